#!/bin/bash
# builds tools/_build/gemm2_probe_<bits> (development probes; the directory is git-ignored and travels with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
PKG=causal-domain-clustering-for-multi-domain-recommendation_amd
for bits in ${@:-0 1 2 4 8 16}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-comment -Iinclude -I$PKG/csrc -DG2_PROBE=$bits \
      tools/gemm2_probe.hip $PKG/csrc/misc.hip -o tools/_build/gemm2_probe_$bits &
done
wait
ls -la tools/_build/

#!/bin/bash
# builds tools/_build/pair_probe_<nstage>_<bits> (development probes; the directory is git-ignored and travels with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
PKG=causal-domain-clustering-for-multi-domain-recommendation_amd
for spec in ${@:-3_0 2_0 3_1 3_2 3_4 3_8 3_16 3_32 3_63}; do
  ns=${spec%%_*}; bits=${spec#*_}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-comment -Iinclude -I$PKG/csrc -DPAIR_PROBE=$bits -DPAIR_NSTAGE=$ns \
      tools/pair_probe.hip $PKG/csrc/misc.hip -o tools/_build/pair_probe_${ns}_$bits &
done
wait
ls tools/_build/ | grep pair

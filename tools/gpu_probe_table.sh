#!/bin/bash
# what each contraction launch of the C2 step waits for: the kernel with parts compiled away (csrc/gemm2.hip G2_PROBE bits:
# 1 no epilogue, 2 no MFMAs, 4 no operand reloads, 8 no global stores, 16 no fragment reads), auto tile configuration
B=tools/_build
O=gpurun_out/$1; mkdir -p $O
for c in fwd1 fwd2 bwdx2 bwdx1; do
  for bits in 0 1 2 4 8 16; do
    echo -n "$c probe=$bits: "; $B/gemm2_probe_$bits $c 1 1 0.2 0 | tail -1
  done
done > $O/gemm2_probes.txt 2>&1
cat $O/gemm2_probes.txt

#!/bin/bash
# tile configurations of cdc_gemm_bf16_nt on one case: tools/gpu_probe_cfgs.sh <outdir> <case> [write_f32 write_bf16]
B=tools/_build
O=gpurun_out/$1; mkdir -p $O
c=$2; wf=${3:-1}; wh=${4:-1}
for cfg in 0 1 2 3 4 5 6 7 8 9 10; do
  echo -n "cfg $cfg: "; $B/gemm2_probe_0 $c $wf $wh 0.2 $cfg | tail -1
done > $O/cfgs_$c.txt 2>&1
cat $O/cfgs_$c.txt

#!/bin/bash
# every development switch of the host side at its non-default value against the training / model parity tests:
#   tools/gpu_switches.sh   -> one line per switch (pass / the failing test)
for kv in CDC_HALF_ONLY=0 CDC_WT_BF16=0 CDC_GATES_LATE=0 CDC_BN_FUSE=0 CDC_FUSE_BCE=0 CDC_FUSED_HEAD=0 CDC_FUSE_ROW_UPDATE=0 CDC_DW_CLASSES=0 \
          CDC_SCALED_REPLAY=0 CDC_OVERLAP=0 CDC_SORT_AHEAD=0 CDC_EARLY_FORK=0 CDC_TABLE_STEP_SIDE=1 CDC_DW_DEFER=0 CDC_PAIR=0 CDC_CGC_MID=0 CDC_FUSE_GATHER=1; do
  env $kv timeout -k 10 400 python -m pytest tests/test_gpu_train.py tests/test_gpu_ple.py tests/test_gpu_models_golden.py -x -q > gpurun_out/switch_last.log 2>&1
  echo "$kv rc=$? $(tail -1 gpurun_out/switch_last.log | cut -c1-90) $(grep -m1 '^FAILED' gpurun_out/switch_last.log | cut -c1-120)"
done

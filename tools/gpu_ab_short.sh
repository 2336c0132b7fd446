#!/bin/bash
# the driver's bench command (--steps 20 --warmup 5) beside the default one, alternating on ONE box: tools/gpu_ab_short.sh <dir>
d=gpurun_out/$1; mkdir -p $d
for i in 1 2 3; do
  for spec in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
    python bench.py --cpu-baseline 0 $spec > $d/b.json 2> $d/err.log || { tail -20 $d/err.log; exit 1; }
    python - "$d/b.json" "$spec" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[2], "| ms/step", round(d["ms_per_step"], 4), "value", round(d["value"]))
PY
  done
done

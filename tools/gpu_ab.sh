#!/bin/bash
# A/B of an environment switch on the default bench: tools/gpu_ab.sh <dir> VAR=a VAR=b ...   (each: bench --steps 300 --warmup 20)
d=gpurun_out/$1; shift
mkdir -p $d
i=0
for kv in "$@"; do
  i=$((i+1))
  env $kv python bench.py --steps 300 --warmup 20 --cpu-baseline 0 > $d/bench_$i.json 2> $d/err_$i.log || { tail -20 $d/err_$i.log; exit 1; }
  python - "$d/bench_$i.json" "$kv" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(sys.argv[2], "ms/step", round(d["ms_per_step"], 4), "value", round(d["value"]), "flush avg ms", r.get("avg_launch_ms"), "kernel sum", r.get("kernel_ms_per_step_sum"))
PY
done

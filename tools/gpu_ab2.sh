#!/bin/bash
# A/B of (environment, bench arguments) pairs: tools/gpu_ab2.sh <dir> "ENV=.. ENV=.. -- bench args" ...
d=gpurun_out/$1; shift
mkdir -p $d
i=0
for spec in "$@"; do
  i=$((i+1))
  envs="${spec%%--*}"; args="${spec#*--}"
  env $envs python bench.py --steps 300 --warmup 20 --cpu-baseline 0 $args > $d/bench_$i.json 2> $d/err_$i.log || { tail -20 $d/err_$i.log; exit 1; }
  python - "$d/bench_$i.json" "$spec" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(sys.argv[2], "| ms/step", round(d["ms_per_step"], 4), "value", round(d["value"]), "kernel sum", round(r.get("kernel_ms_per_step_sum"), 4))
PY
done

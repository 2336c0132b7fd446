#!/bin/bash
# same-box A/B of two builds of the library: tools/gpu_ab_lib.sh <dir> <other.so> [bench args]   (the in-tree build runs first and third)
d=gpurun_out/$1; other=$2; shift 2
mkdir -p $d
L=causal-domain-clustering-for-multi-domain-recommendation_amd/libcdcmdr.so
cp $L $d/new.so
for tag in new old new old; do
  if [ $tag = old ]; then cp $other $L; else cp $d/new.so $L; fi
  python bench.py --steps 300 --warmup 20 --cpu-baseline 0 "$@" > $d/bench_$tag.json 2> $d/err_$tag.log || { tail -20 $d/err_$tag.log; cp $d/new.so $L; exit 1; }
  python - "$d/bench_$tag.json" "$tag" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(sys.argv[2], "| ms/step", round(d["ms_per_step"], 4), "value", round(d["value"]), {k: v for k, v in r.get("back_to_back_us", {}).items() if "tower" in k})
PY
done
cp $d/new.so $L
rm -f $d/new.so

O=gpurun_out/r2g
mkdir -p $O
export CDC_BENCH_BREAKDOWN_ALL=1
CDC_PROFILE_DETAIL=1 CDC_DW_BLOCKS=512 python bench.py --steps 30 --warmup 10 --preroll 200 --cpu-baseline 0 > $O/bench_detail.json 2> $O/err.log
python - <<'PY'
import json
dd=json.loads(open("gpurun_out/r2g/bench_detail.json").read().strip().splitlines()[-1])
print("G2", round(dd["ms_per_step"],4))
items=[(k,v) for k,v in dd["roofline"]["breakdown_all"].items()]
items.sort(key=lambda kv: kv[0].split("#")[1] if "#" in kv[0] else "zzz"+kv[0])
print("   ", "  ".join(f"{k.replace('cdc_','')}={v*1000:.1f}" for k,v in items))
PY
cd /tmp && export TMPDIR=/tmp
CDC_DW_BLOCKS=512 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 --preroll 200 --warmup 10 --steps 200 > $GRAFT_REPO_ROOT/$O/bench.json 2>> $GRAFT_REPO_ROOT/$O/err.log
cd $GRAFT_REPO_ROOT
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
python - <<'PY'
import csv,json
rows=list(csv.DictReader(open("gpurun_out/r2g/kernel_stats.csv")))
d=json.loads(open("gpurun_out/r2g/bench.json").read().strip().splitlines()[-1])
print("ms/step (profiled)", d["ms_per_step"])
steps=None
for r in rows:
    if "k_lazy_flush" in r["Name"]: steps=int(r["Calls"])
for r in rows[:30]:
    print(f'{r["Name"][:60]:60s} per-step {int(r["Calls"])/steps:5.2f} x {float(r["AverageNs"])/1e3:7.2f} us = {float(r["TotalDurationNs"])/steps/1e3:7.1f}')
PY

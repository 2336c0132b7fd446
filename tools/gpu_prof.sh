# usage: bash tools/gpu_prof.sh <outdir> [bench args...]   -> per-kernel per-step table from rocprofv3 --kernel-trace --stats
O=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 --preroll 200 --warmup 10 --steps 200 "$@" > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/err.log
cd $GRAFT_REPO_ROOT
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
python - $O <<'PY'
import csv,json,sys
O=sys.argv[1]
rows=list(csv.DictReader(open(f"{O}/kernel_stats.csv")))
d=json.loads(open(f"{O}/bench.json").read().strip().splitlines()[-1])
print("ms/step (profiled)", round(d["ms_per_step"],4))
steps=None
for r in rows:
    if "k_lazy_flush" in r["Name"] or "k_adam_dense_pass" in r["Name"]: steps=int(r["Calls"])
tot=0
for r in rows[:34]:
    v=float(r["TotalDurationNs"])/steps/1e3; tot+=v
    print(f'{r["Name"][:58]:58s} {int(r["Calls"])/steps:5.2f} x {float(r["AverageNs"])/1e3:7.2f} us = {v:7.1f}')
print("sum", round(tot,1))
PY

O=gpurun_out/r2e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CDC_DW_BLOCKS=512 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 --preroll 200 --warmup 10 --steps 200 > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/err.log
cd $GRAFT_REPO_ROOT
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
python - <<'PY'
import csv,json
rows=list(csv.DictReader(open("gpurun_out/r2e/kernel_stats.csv")))
d=json.loads(open("gpurun_out/r2e/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"])
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  total% {float(r["TotalDurationNs"])/tot*100:5.1f}')
PY

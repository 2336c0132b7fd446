#!/bin/bash
# does the side-stream branch run beside the main one?  kernel trace of a short run, then for each flush launch the kernels whose
# [start, end] intersects it.   usage: tools/gpu_trace_overlap.sh <outdir> [bench args]
O=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 --preroll 70 --warmup 5 --steps 30 "$@" > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/err.log
cd $GRAFT_REPO_ROOT
find $O/kt -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/kt
python - $O <<'PY'
import csv, sys, collections
O = sys.argv[1]
rows = list(csv.DictReader(open(f"{O}/kernel_trace.csv")))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?")) for r in rows))
fl = [e for e in ev if "k_lazy_flush" in e[2]][-10:]
for f in fl[-3:]:
    inter = [(e[2], e[3], round((min(e[1], f[1]) - max(e[0], f[0])) / 1e3, 1)) for e in ev if e is not f and e[0] < f[1] and e[1] > f[0]]
    print("flush", round((f[1] - f[0]) / 1e3, 1), "us on queue", f[3], "overlaps:", inter[:12])
q = collections.Counter(e[3] for e in ev[-2000:])
print("queues of the last 2000 dispatches:", dict(q))
PY

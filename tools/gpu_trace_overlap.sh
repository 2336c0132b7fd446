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
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?") + "/" + r.get("Stream_Id", "?")) for r in rows))
fl = [e for e in ev if "k_lazy_flush" in e[2]][-10:]
for f in fl[-3:]:
    inter = [(e[2], e[3], round((min(e[1], f[1]) - max(e[0], f[0])) / 1e3, 1)) for e in ev if e is not f and e[0] < f[1] and e[1] > f[0]]
    print("flush", round((f[1] - f[0]) / 1e3, 1), "us on queue", f[3], "overlaps:", inter[:12])
q = collections.Counter(e[3] for e in ev[-2000:])
print("queues of the last 2000 dispatches:", dict(q))
# timeline of ONE steady-state step (the third last): from its staging launch to the next one's; offsets in us
st = [i for i, e in enumerate(ev) if "k_stage_batch_next" in e[2]]      # the timed loop's steps (the instrumented eager steps behind it use k_stage_batch)
if len(st) >= 4:
    a, b = st[-4], st[-3]
    t0 = ev[a][0]
    step = ev[a:b]
    side_q = next((e[3] for e in step if "k_lazy_flush" in e[2]), None)
    with open(f"{O}/step_timeline.txt", "w") as f:
        f.write("# one replayed C2 step (rocprofv3 --kernel-trace; the profiler stretches the gaps, the durations are real): queue, start, end, duration (us), kernel\n")
        for e in step:
            f.write(f"{'side' if e[3] == side_q else 'main'} q{e[3]:>3} {(e[0] - t0) / 1e3:8.1f} {(e[1] - t0) / 1e3:8.1f} {(e[1] - e[0]) / 1e3:7.1f}  {e[2]}\n")
        fl = next(e for e in step if "k_lazy_flush" in e[2])
        main = [e for e in step if e[3] != side_q]
        cov = sum(max(0, min(e[1], fl[1]) - max(e[0], fl[0])) for e in main)
        f.write(f"# the slice runs {(fl[1] - fl[0]) / 1e3:.1f} us; main-chain kernels are executing during {100.0 * cov / (fl[1] - fl[0]):.0f} % of that time\n")
        f.write(f"# step length by the trace: {(ev[b][0] - t0) / 1e3:.1f} us\n")
    print(open(f"{O}/step_timeline.txt").read())
PY

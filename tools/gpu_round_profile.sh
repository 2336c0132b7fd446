#!/bin/bash
# Everything profiles/roundN/ holds for the default bench command, in one GPU-box call:
#   tools/gpu_round_profile.sh <outdir>       (outputs under gpurun_out/<outdir>/; copy what is judged into profiles/roundN/)
# 1 default bench line (with cpu_baseline)   2 short run (--steps 20 --warmup 5): steady state independent of the flags
# 3 rocprofv3 --kernel-trace --stats of the same command   4 --pmc FETCH_SIZE and WRITE_SIZE in separate passes
# 5 SQ counters of the replay slice and the MFMA kernels (own passes; never combined with a trace domain other than kernel-trace)
set -o pipefail
O=gpurun_out/$1
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
echo "bench default done"
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 > $O/bench_short.json 2> $O/bench_short.err || exit 1
python bench.py --steps 1000 --warmup 20 --cpu-baseline 0 > $O/bench_long.json 2> $O/bench_long.err || exit 1
python - $O <<'PY'
import json, sys
O = sys.argv[1]
v = {}
for k in ("default", "short", "long"):
    d = json.loads([l for l in open(f"{O}/bench_{k}.json") if l.startswith("{")][-1])
    v[k] = d["ms_per_step"]
    print(k, "steps", d["steps"], "warmup", d["warmup"], "ms/step", round(d["ms_per_step"], 4), "value", round(d["value"]))
print("short vs long: %+.2f%%" % (100 * (v["short"] / v["long"] - 1)))
PY
bash tools/gpu_prof.sh $O > $O/kernel_table.txt 2>&1 || { tail -5 $O/kernel_table.txt; exit 1; }
cat $O/kernel_table.txt | head -30
bash tools/gpu_pmc.sh $O "FETCH_SIZE" > /dev/null 2>&1 || exit 1
bash tools/gpu_pmc.sh $O "WRITE_SIZE" > /dev/null 2>&1 || exit 1
python - $O <<'PY'
import sys, collections
O = sys.argv[1]
acc = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cur = None
    for line in open(f"{O}/pmc_summary_{c}.txt"):
        if not line.startswith(" "):
            cur = line.rstrip("\n")
            acc.setdefault(cur, [])
        elif cur is not None:
            acc[cur].append(line.rstrip("\n"))
with open(f"{O}/pmc_default_fetch_write.txt", "w") as f:
    for k, lines in acc.items():
        f.write(k + "\n")
        for l in lines:
            f.write(l + "\n")
print("pmc fetch/write written:", len(acc), "kernels")
PY
bash tools/gpu_pmc.sh $O "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" k_lazy_flush k_g2_nt k_g2_tn k_cgc_mid k_pair_fwd > /dev/null 2>&1 || exit 1
bash tools/gpu_pmc.sh $O "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" k_g2_nt k_g2_tn k_cgc_mid k_pair_fwd > /dev/null 2>&1 || exit 1
ls $O

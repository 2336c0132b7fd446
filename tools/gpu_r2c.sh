set -x
O=gpurun_out/r2c
mkdir -p $O
export CDC_BENCH_BREAKDOWN_ALL=1 CDC_PROFILE_DETAIL=1
for g in 1 0; do for d in 0.2 0.0; do
CDC_GEMM2=$g python bench.py --steps 30 --warmup 10 --preroll 200 --dropout $d --cpu-baseline 0 > $O/bench_g${g}_d$d.json 2> $O/err.log
done; done
python - <<'PY'
import json
for g in (1,0):
  for d in ("0.2","0.0"):
    dd=json.loads(open(f"gpurun_out/r2c/bench_g{g}_d{d}.json").read().strip().splitlines()[-1])
    print("G2" if g else "OLD", "dropout",d, round(dd["ms_per_step"],4))
    items=[(k,v) for k,v in dd["roofline"]["breakdown_all"].items() if "glinear" in k or "shadow" in k or "transpose" in k]
    items.sort(key=lambda kv: kv[0].split("#")[1])
    print("   ", "  ".join(f"{k.replace('cdc_','')}={v*1000:.1f}" for k,v in items))
PY
cd /tmp && export TMPDIR=/tmp
CDC_PROFILE_DETAIL=0 CDC_GEMM2=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 --graph 0 --preroll 70 --warmup 2 --steps 20 > $GRAFT_REPO_ROOT/$O/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_sq k_g2 k_glinear > $O/pmc_sq_summary.txt 2>&1
rm -rf $O/pmc_sq
cat $O/pmc_sq_summary.txt

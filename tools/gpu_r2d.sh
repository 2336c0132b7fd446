O=gpurun_out/r2d
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -25 $O/pytest.log
export CDC_BENCH_BREAKDOWN_ALL=1 CDC_PROFILE_DETAIL=1
for b in 512 768 1024; do
CDC_DW_BLOCKS=$b CDC_GEMM2=1 python bench.py --steps 30 --warmup 10 --preroll 200 --cpu-baseline 0 > $O/bench_g1_$b.json 2> $O/err.log
done
python - <<'PY'
import json
for b in (512,768,1024):
    dd=json.loads(open(f"gpurun_out/r2d/bench_g1_{b}.json").read().strip().splitlines()[-1])
    print("G2 dw blocks",b, round(dd["ms_per_step"],4), dd["config"]["last_bce_loss"])
    items=[(k,v) for k,v in dd["roofline"]["breakdown_all"].items() if "glinear_bwd_w" in k]
    items.sort(key=lambda kv: kv[0].split("#")[1])
    print("   ", "  ".join(f"{k.replace('cdc_','')}={v*1000:.1f}" for k,v in items))
PY

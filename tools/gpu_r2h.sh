O=gpurun_out/r2h
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -8 $O/pytest.log
export CDC_BENCH_BREAKDOWN_ALL=1
CDC_DW_BLOCKS=512 python bench.py --steps 100 --warmup 10 --preroll 200 --cpu-baseline 0 > $O/bench.json 2> $O/err.log
CDC_SCALED_REPLAY=0 CDC_DW_BLOCKS=512 python bench.py --steps 100 --warmup 10 --preroll 200 --cpu-baseline 0 > $O/bench_noscaled.json 2>> $O/err.log
python - <<'PY'
import json
for f in ("bench","bench_noscaled"):
    dd=json.loads(open(f"gpurun_out/r2h/{f}.json").read().strip().splitlines()[-1])
    print(f, round(dd["ms_per_step"],4), dd["config"]["last_bce_loss"], "gemm tflops", dd["roofline"].get("all_gemm_tflops"))
    print({k:v for k,v in dd["roofline"]["breakdown_all"].items()})
PY

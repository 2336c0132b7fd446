O=gpurun_out/r2f
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -25 $O/pytest.log
export CDC_BENCH_BREAKDOWN_ALL=1
CDC_DW_BLOCKS=512 python bench.py --steps 100 --warmup 10 --preroll 200 --cpu-baseline 0 > $O/bench.json 2> $O/err.log
python - <<'PY'
import json
dd=json.loads(open("gpurun_out/r2f/bench.json").read().strip().splitlines()[-1])
print("G2", round(dd["ms_per_step"],4), dd["config"]["last_bce_loss"], "gemm tflops", dd["roofline"].get("all_gemm_tflops"))
print({k:v for k,v in dd["roofline"]["breakdown_all"].items()})
PY

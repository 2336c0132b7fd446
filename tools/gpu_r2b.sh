set -x
O=gpurun_out/r2b
mkdir -p $O
python -m pytest tests/test_gpu_gemm2.py -x -q > $O/pytest_gemm2.log 2>&1; tail -15 $O/pytest_gemm2.log
python -m pytest tests/test_gpu_ops.py -x -q -k "glinear or batchnorm" > $O/pytest_ops.log 2>&1; tail -5 $O/pytest_ops.log
export CDC_BENCH_BREAKDOWN_ALL=1
CDC_GEMM2=1 python bench.py --steps 50 --warmup 20 --cpu-baseline 0 > $O/bench_g2.json 2> $O/bench_g2.err; tail -3 $O/bench_g2.err
CDC_GEMM2=0 python bench.py --steps 50 --warmup 20 --cpu-baseline 0 > $O/bench_g0.json 2> $O/bench_g0.err
python - <<'PY'
import json
for f in ("bench_g2","bench_g0"):
    try:
        d=json.loads(open(f"gpurun_out/r2b/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],4), d["config"]["last_bce_loss"], d["roofline"].get("all_gemm_tflops"))
        print({k:v for k,v in d["roofline"]["breakdown_all"].items()})
    except Exception as e: print(f,"ERR",e)
PY

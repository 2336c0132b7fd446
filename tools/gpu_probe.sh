B=tools/_build
python -m pytest tests/test_gpu_gemm2.py -x -q 2>&1 | tail -3
for c in bwdx1 bwdx2; do
  for cfg in 1 2 4 5 7 8 9 10; do $B/gemm2_probe_1 $c 1 1 0.2 $cfg; done
  for cfg in 1 2 4 5 7 8 9 10; do $B/gemm2_probe_0 $c 0 1 0.2 $cfg; done
done

set -x
O=gpurun_out/r2a
mkdir -p $O
python -m pytest tests/test_gpu_train.py tests/test_gpu_cdc.py -x -q -m gpu > $O/pytest_new.log 2>&1 || { tail -30 $O/pytest_new.log; exit 1; }
tail -3 $O/pytest_new.log
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 > $O/bench_w5.json 2> $O/bench_w5.err && tail -c 1500 $O/bench_w5.json
python bench.py --steps 200 --warmup 2000 --cpu-baseline 0 > $O/bench_w2000.json 2> $O/bench_w2000.err
CDC_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --preroll 70 --cpu-baseline 0 > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err; echo rehearsal rc=$?
python tools/auc_parity.py --vocab 10000 --sides hip_f32,hip_bf16 --out gpurun_out/auc_v10k > $O/auc_hip.log 2>&1; echo auc rc=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 --graph 0 --preroll 130 --warmup 2 --steps 64 > $GRAFT_REPO_ROOT/$O/pmc_sq.log 2>&1; echo pmc rc=$?
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_sq k_lazy k_gather k_glinear k_bn k_gate k_rowdot > $O/pmc_sq_summary.txt 2>&1
rm -rf $O/pmc_sq
head -60 $O/pmc_sq_summary.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wg
timeout 900 python -m pytest tests/test_gpu_train_fold.py tests/test_gpu_parallel.py -q -x -m gpu > gpurun_out/wg/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/wg/pytest.log | cut -c1-200
for i in 1 2; do python tools/bench_train.py --batch 8 --steps 60 --warmup 20 2>/dev/null | tail -1 | cut -c100-250; done

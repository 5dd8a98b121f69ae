cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wg
timeout 900 python -m pytest tests/test_gpu_train_fold.py tests/test_gpu_model.py -q -x -m gpu > gpurun_out/wg/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/wg/pytest.log | cut -c1-300
for b in 64 8; do for f in 1 0; do echo "pooled $f batch $b: $(DLWP_WGRAD_POOLED=$f python tools/bench_train.py --batch $b --steps 60 --warmup 20 2>/dev/null | tail -1 | cut -c180-215)"; done; done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wg
timeout 900 python -m pytest tests/test_gpu_train_fold.py -q -x -m gpu -k "phase" > gpurun_out/wg/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/wg/pytest.log | cut -c1-200
bash tools/r3_trace.sh t64 64 2>&1 | grep "mse_mae\|span"

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wg
timeout 900 python -m pytest tests/test_gpu_train_fold.py -q -x -m gpu -k "store_phase or unsupported_layers" > gpurun_out/wg/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/wg/pytest.log | cut -c1-250

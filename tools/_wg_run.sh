cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wg
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "channel_block or weight_gradient" > gpurun_out/wg/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/wg/pytest.log | cut -c1-200
bash tools/profile_wgrad.sh r3

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wg
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/wg/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/wg/pytest.log | cut -c1-300
for b in 64 8; do python tools/bench_train.py --batch $b --steps 40 --warmup 20 2>/dev/null | tail -1 | cut -c100-250; done

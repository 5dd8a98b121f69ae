cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wg
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "channel_block or weight_gradient or backward_kernels" > gpurun_out/wg/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/wg/pytest.log | cut -c1-200
timeout 120 tools/microbench/wgrad_cb_phase_timing.bin > gpurun_out/wg/phase.txt 2>&1; grep -A2 "8 waves\|layer 3\|up-sampled" gpurun_out/wg/phase.txt
for b in 64 8; do timeout 600 python tools/tune_wgrad.py --batch $b --layers L2p,L3p,L4,L5r,L6r > gpurun_out/wg/sweep_b$b.txt 2>&1; done
grep -A3 "^L" gpurun_out/wg/sweep_b64.txt
for b in 64 8; do python tools/bench_train.py --batch $b --steps 40 --warmup 20 2>/dev/null | tail -1 | cut -c100-250; done

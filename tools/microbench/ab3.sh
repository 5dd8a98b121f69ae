(cd tools/microbench && hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDLWP_PHASE_TIMING -o /tmp/wgcb.bin wgrad_cb_phase_timing.hip -I../../include 2>/dev/null && /tmp/wgcb.bin | grep -E "ms per launch|per tile" | cut -c1-230)
python -m pytest tests -m gpu -x -q -k "wgrad or weight or pair or fold or train" 2>&1 | grep -E "passed|failed" | tail -2
bash tools/microbench/ab.sh 2>&1 | tail -14

#!/bin/bash
# the weight-gradient kernels of the config-3 training step, one layer at a time: every compiled instance forced and timed
# (tools/tune_wgrad.py, 8 and 64 samples), the channel-block kernel's phase timing (tools/microbench/wgrad_cb_phase_timing.hip)
# and the PMC counters (HBM bytes fetched, MFMA busy) of the sweep.  usage: bash tools/profile_wgrad.sh <tag>  -> gpurun_out/prof/<tag>_wgrad_*
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3}
O=$R/gpurun_out/prof; mkdir -p $O
cd $R
for b in 64 8; do
  timeout 600 python tools/tune_wgrad.py --batch $b --layers L1,L2p,L3p,L4,L5r,L6r 2>&1 | grep -v amdgpu.ids > $O/${TAG}_wgrad_sweep_b$b.txt
done
timeout 120 tools/microbench/wgrad_cb_phase_timing.bin > $O/${TAG}_wgrad_cb_phase_timing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/tune_wgrad.py --batch 64 --layers L2p --iters 3"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/wg_fetch -o p --output-format csv -- $CMD > /dev/null 2> $O/wg_fetch.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $O/wg_mfma -o p --output-format csv -- $CMD > /dev/null 2> $O/wg_mfma.err
python $R/tools/parse_pmc.py $O/${TAG}_wgrad_pmc_layer2_b64.json $O/wg_fetch $O/wg_mfma > $O/${TAG}_wgrad_pmc_layer2_b64.txt 2>&1
rm -rf $O/wg_fetch $O/wg_mfma
grep -A5 "^L" $O/${TAG}_wgrad_sweep_b64.txt | cut -c1-100

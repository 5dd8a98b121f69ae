#!/bin/bash
# Where do the waves of the forward kernels wait?  rocprofv3 --pmc passes of the bench command with the SQ wave-state
# counters (separate passes, --kernel-trace only).  usage: bash tools/profile_stalls.sh r2x  -> gpurun_out/prof/<tag>_stalls.json
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DLWP_ROLLOUT_GROUPS=1   # (as tools/profile_bench.sh: 256-member launches, one at a time)
CMD=${STALL_CMD:-"python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"}
(rocprofv3 --list-avail 2>/dev/null || rocprofv3-avail list 2>/dev/null) | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/${TAG}_sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/st$i -o p --output-format csv -- $CMD > /dev/null 2> $OUT/st$i.err || echo "pass $i failed: $set"
done
python $R/tools/parse_pmc.py $OUT/${TAG}_stalls.json $OUT/st1 $OUT/st2 $OUT/st3 $OUT/st4 $OUT/st5 $OUT/st6 > $OUT/stalls.txt 2>&1
cut -c1-600 $OUT/stalls.txt

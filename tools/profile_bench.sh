#!/bin/bash
# Profiles of the bench command on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats        -> gpurun_out/prof/<tag>_kernel_stats.csv   (per-kernel average duration)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE -> gpurun_out/prof/<tag>_hbm_traffic_b256.json (separate passes; parse_pmc.py
#      applies the gfx950 corrections of MI355X_MICROARCH.md)
#   3. rocprofv3 --pmc MFMA busy / clocks      -> gpurun_out/prof/<tag>_mfma_busy.json
# Copy what should be judged into profiles/.   usage: bash tools/profile_bench.sh r1d
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# One member chain for the profiled rollout: with the default two chains (Executor.member_groups) the graph's kernels run on
# half the members each and two at a time, so their traced durations and per-dispatch counters are not those of the
# 256-member launches bench.py prices in its roofline record.
export DLWP_ROLLOUT_GROUPS=1
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- $CMD > $OUT/${TAG}_stats_bench.json 2> $OUT/stats.err
cp $OUT/stats/s_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
python - <<PY > $OUT/${TAG}_kernel_stats.meta.json
import json, sys
sys.path.insert(0, "$R")
from dlwp_amd import _lib
print(json.dumps({"source_sha": _lib.kernel_source_hash(), "members": 256, "command": "$CMD", "rollout_groups": 1}))
PY
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_write.err
python $R/tools/parse_pmc.py $OUT/${TAG}_hbm_traffic_b256.json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $OUT/pmc_mfma -o p --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_mfma.err
python $R/tools/parse_pmc.py $OUT/${TAG}_mfma_busy.json $OUT/pmc_mfma > $OUT/pmc_mfma.txt 2>&1
head -12 $OUT/${TAG}_kernel_stats.csv
cat $OUT/pmc_traffic.txt | cut -c1-400
cat $OUT/pmc_mfma.txt | cut -c1-400

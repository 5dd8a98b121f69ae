#!/bin/bash
# rocprofv3 HBM evidence for the padding / pooling / up-sampling kernels (north_star: "rocprof HBM GB/s reported for the
# padding kernels").  Run through gpurun from the repo root:  bash tools/profile_pads.sh r2
#   1. un-profiled run of tools/bench_pad.py -> HIP-event times per (kernel, shape)
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE (separate passes) over the same command
#   3. tools/parse_pad_pmc.py joins them -> gpurun_out/prof/<tag>_pad_pool_hbm.json (copy into profiles/)
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
ITERS=5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_pad.py --iters $ITERS --out $OUT/${TAG}_pad_rows.json > $OUT/${TAG}_pad_events.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/pad_stats -o s --output-format csv -- python $R/tools/bench_pad.py --iters $ITERS > /dev/null 2> $OUT/pad_stats.err
cp $(find $OUT/pad_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_pad_pool_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pad_fetch -o p --output-format csv -- python $R/tools/bench_pad.py --iters $ITERS > /dev/null 2> $OUT/pad_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pad_write -o p --output-format csv -- python $R/tools/bench_pad.py --iters $ITERS > /dev/null 2> $OUT/pad_write.err
python $R/tools/parse_pad_pmc.py $OUT/${TAG}_pad_rows.json $ITERS $OUT/${TAG}_pad_pool_hbm.json $OUT/pad_fetch $OUT/pad_write

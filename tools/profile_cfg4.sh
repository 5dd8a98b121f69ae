#!/bin/bash
# Config 4 (1-degree recurrent stack, bf16 storage): rocprofv3 split of the forward into its MFMA-side launches (the
# convolutions, bf16 matrix cores) and its HBM-bound launches (ConvLSTM2D gate update, pooling, copies), with HBM traffic from
# separate --pmc passes.  Run through gpurun from the repo root:  bash tools/profile_cfg4.sh r2 [members]
TAG=${1:-prof}
M=${2:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_cfg4.py --members $M --iters 10"
$CMD > $OUT/${TAG}_cfg4_bf16_m${M}.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/cfg4_stats -o s --output-format csv -- $CMD > /dev/null 2> $OUT/cfg4_stats.err
cp $(find $OUT/cfg4_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_cfg4_bf16_m${M}_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/cfg4_fetch -o p --output-format csv -- $CMD > /dev/null 2> $OUT/cfg4_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/cfg4_write -o p --output-format csv -- $CMD > /dev/null 2> $OUT/cfg4_write.err
python $R/tools/parse_pmc.py $OUT/${TAG}_cfg4_bf16_m${M}_hbm_traffic.json $OUT/cfg4_fetch $OUT/cfg4_write --members $M > $OUT/cfg4_traffic.txt 2>&1
head -14 $OUT/${TAG}_cfg4_bf16_m${M}_kernel_stats.csv | cut -c1-160
cut -c1-220 $OUT/cfg4_traffic.txt | head -14

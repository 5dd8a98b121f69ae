#!/bin/bash
# round-3 GPU call 2: kernel traces at the strong-scaling shares (batch-8 train step, 4-member config-5 rollout, 8-member cfg2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DLWP_ROLLOUT_GROUPS=1
rocprofv3 --kernel-trace --stats -d $O/t8 -o s --output-format csv -- python $R/tools/bench_train.py --batch 8 --steps 40 --warmup 10 > $O/train_b8.json 2> $O/t8.err
cp $O/t8/s_kernel_stats.csv $O/train_b8_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/c5 -o s --output-format csv -- python $R/bench.py --grid 180x360 --channels 12 --members 4 --forwards 40 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/cfg5_m4.json 2> $O/c5.err
cp $O/c5/s_kernel_stats.csv $O/cfg5_m4_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/c2 -o s --output-format csv -- python $R/bench.py --members 8 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/cfg2_m8.json 2> $O/c2.err
cp $O/c2/s_kernel_stats.csv $O/cfg2_m8_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/c4 -o s --output-format csv -- python $R/tools/bench_cfg4.py > $O/cfg4.json 2> $O/c4.err
cp $O/c4/s_kernel_stats.csv $O/cfg4_kernel_stats.csv
# a trace with timestamps for the batch-8 step: gaps between kernels
cp $O/t8/s_kernel_trace.csv $O/train_b8_kernel_trace.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/train_b8_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//2:]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)
span=int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp'])
print('kernels',len(rows),'busy ms',busy/1e6,'span ms',span/1e6,'busy frac',busy/span)
PY
head -30 $O/train_b8_kernel_stats.csv | cut -c1-150
head -12 $O/cfg5_m4_kernel_stats.csv | cut -c1-150
head -12 $O/cfg2_m8_kernel_stats.csv | cut -c1-150
head -20 $O/cfg4_kernel_stats.csv | cut -c1-150
rm -rf $O/t8 $O/c5 $O/c2 $O/c4

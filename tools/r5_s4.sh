#!/bin/bash
# isolated (single-stream 'graph' form) kernel durations of the batch-64 and batch-8 training steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 64 8; do
  DLWP_TRAIN_STEP=graph rocprofv3 --kernel-trace --stats -d $O/tr$b -o s --output-format csv -- python $R/tools/bench_train.py --batch $b --steps 40 --warmup 10 > $O/train_graph_b$b.json 2> $O/tr$b.err
  cp $(find $O/tr$b -name '*kernel_stats.csv' | head -1) $O/train_graph_b${b}_kernel_stats.csv
  rm -rf $O/tr$b
  tail -1 $O/train_graph_b$b.json | cut -c1-300
done
cd $R
for b in 64 8; do for f in graph lanes; do DLWP_TRAIN_STEP=$f python tools/bench_train.py --batch $b --steps 40 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $b form $f', round(d['ms_per_step'],4), 'ms host', round(d.get('host_ms_per_step',0),3))"; done; done

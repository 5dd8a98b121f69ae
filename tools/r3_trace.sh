#!/bin/bash
# kernel timeline of one training step (rocprofv3 --kernel-trace): usage: bash tools/r3_trace.sh <tag> <batch> [env...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; B=$2; shift 2
O=$R/gpurun_out/r3t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace -d $O/$TAG -o s --output-format csv -- python $R/tools/bench_train.py --batch $B --steps 12 --warmup 8 > $O/$TAG.json 2> $O/$TAG.err
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/$TAG/s_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last full step: find the adam kernels
idx=[i for i,r in enumerate(rows) if 'adam' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
step=rows[a+1:b+1]
t0=int(step[0]['Start_Timestamp'])
print('$TAG: kernels in step',len(step),'span us',(int(step[-1]['End_Timestamp'])-t0)/1e3,'busy us',sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)/1e3)
for r in step:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    nm=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]
    print('%8.1f %8.1f %6.1f  q%s  %s'%(s,e,e-s,r.get('Queue_Id','?'),nm))
PY
rm -rf $O/$TAG

#!/bin/bash
# kernel timeline of loader-fed training steps (rocprofv3 --kernel-trace): where do the batch's transfers run relative to the step?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pull in 1 0; do
TAG=fitgen_b64_pull$pull
DLWP_LOADER_PULL=$pull DLWP_TRAIN_STEP=graph rocprofv3 --kernel-trace --memory-copy-trace -d $O/$TAG -o s --output-format csv -- python $R/tools/bench_fit_generator.py --batch 64 --samples 1280 --epochs 1 > $O/$TAG.json 2> $O/$TAG.err
python - <<PY > $O/$TAG.timeline.txt
import csv, glob
rows=list(csv.DictReader(open('$O/$TAG/s_kernel_trace.csv')))
cp=[]
for f in glob.glob('$O/$TAG/*memory_copy_trace.csv'):
    cp+=list(csv.DictReader(open(f)))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),'q'+r.get('Queue_Id','?'),r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]) for r in rows]
ev+=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),'copy',r.get('Direction','')+' '+r.get('Bytes','')) for r in cp]
ev.sort()
# window: the middle of the SECOND fit_generator (timed) -- take events from 55% to 55%+8ms of the span
t0=ev[0][0]; t1=ev[-1][1]
w0=t0+int(0.55*(t1-t0)); w1=w0+6000000
print('$TAG window of 6 ms')
for s,e,q,n in ev:
    if s>=w0 and s<=w1:
        print('%9.1f %9.1f %7.1f  %-5s %s'%((s-w0)/1e3,(e-w0)/1e3,(e-s)/1e3,q,n))
PY
head -150 $O/$TAG.timeline.txt
rm -rf $O/$TAG
done

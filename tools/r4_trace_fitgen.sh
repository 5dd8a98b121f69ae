#!/bin/bash
# kernel timeline of loader-fed training steps (rocprofv3 --kernel-trace): where do the batch's transfers run relative to the step?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pull in ${PULLS:-1 0}; do
B=${BATCH:-64}; TAG=fitgen_b${B}_pull$pull
DLWP_LOADER_PULL=$pull DLWP_TRAIN_STEP=graph rocprofv3 --kernel-trace --memory-copy-trace -d $O/$TAG -o s --output-format csv -- python $R/tools/bench_fit_generator.py --batch $B --samples ${SAMPLES:-1280} --epochs 1 > $O/$TAG.json 2> $O/$TAG.err
python - <<PY > $O/$TAG.timeline.txt
import csv, glob
rows=list(csv.DictReader(open('$O/$TAG/s_kernel_trace.csv')))
print('kernel trace columns', list(rows[0].keys()) if rows else None, len(rows))
cp=[]
for f in glob.glob('$O/$TAG/*memory_copy_trace.csv'):
    cp+=list(csv.DictReader(open(f)))
print('copy trace columns', list(cp[0].keys()) if cp else None, len(cp))
def ts(r,k):
    return int(r[k])
ev=[(ts(r,'Start_Timestamp'),ts(r,'End_Timestamp'),'q'+str(r.get('Queue_Id','?')),r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]) for r in rows]
ev+=[(ts(r,'Start_Timestamp'),ts(r,'End_Timestamp'),'copy',str(r.get('Direction',''))+' '+str(r.get('Bytes',r.get('Size','')))) for r in cp]
ev.sort()
# the timed fit_generator: find the gather / copy_many kernels and print 3 consecutive steps around the 70th percentile
marks=[i for i,e in enumerate(ev) if 'copy_many' in e[3]]
print('events', len(ev), 'copy_many marks', len(marks))
if len(marks) > 12:
    a=marks[int(0.3*len(marks))]; b=marks[int(0.3*len(marks))+3]
    w0=ev[a][0]
    for s,e,q,n in ev[a:b+1]:
        print('%9.1f %9.1f %7.1f  %-5s %s'%((s-w0)/1e3,(e-w0)/1e3,(e-s)/1e3,q,n))
PY
head -200 $O/$TAG.timeline.txt | cut -c1-120
rm -rf $O/$TAG
done

#!/bin/bash
# where the next frame's prefix runs relative to the step: bash tools/exp_timeline.sh VAR   (rocprofv3 kernel trace of the bench, VAR=0 and VAR=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
VAR=${1:-PTTA_PREFIX_FIRST}
for v in 0 1; do
export $VAR=$v
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl$v -o x -- python3 bench.py --steps 20 --warmup 10 --single-block --no-nlspn --no-cpu-baseline > /dev/null 2> gpurun_out/tl$v.log
echo "== $VAR=$v"; python3 - gpurun_out/tl$v/x_kernel_trace.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
adam=[i for i,n in enumerate(names) if n.startswith('adam_multi_kernel')]
for k in (12,13):
    step=rows[adam[k-1]+1:adam[k]+1]
    t0=int(rows[adam[k-1]]['End_Timestamp'])
    qs={}
    for r in step:
        q=r['Queue_Id']; s,e=(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3
        qs.setdefault(q,[]).append((s,e,r['Kernel_Name'][:40]))
    print('step',k,'wall',round((int(step[-1]['End_Timestamp'])-t0)/1e3,1))
    for q,v in sorted(qs.items()):
        print('  queue',q,'kernels',len(v),'first start',round(v[0][0],1),v[0][2],'| last end',round(v[-1][1],1),'| busy us',round(sum(e-s for s,e,_ in v),1))
        if len(v)<40: print('     starts:',' '.join(str(round(s)) for s,e,_ in v))
PY
done

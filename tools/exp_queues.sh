#!/bin/bash
# which HARDWARE QUEUE each launch of one replayed step ran on (rocprofv3 kernel trace): bash tools/exp_queues.sh [step index] [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
K=${1:-22}; shift
for V in "$@"; do export "$V"; done
rm -rf gpurun_out/q
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/q -o x -- python3 bench.py --steps 20 --warmup 10 --single-block --no-nlspn --no-cpu-baseline --no-self-check > gpurun_out/q_bench.json 2> gpurun_out/q.log
python3 - gpurun_out/q/x_kernel_trace.csv $K <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_multi_kernel')]
k = int(sys.argv[2])
t0 = int(rows[adam[k - 1]]['End_Timestamp'])
t1 = int(rows[adam[k]]['End_Timestamp'])
step = [r for r in rows if t0 <= int(r['Start_Timestamp']) <= t1]
qs = sorted({r['Queue_Id'] for r in step})
print('step', k, 'wall us', (t1 - t0) / 1e3, 'queues', qs)
for r in step:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print('%8.1f %7.1f  q%-2d %s' % (s, e - s, qs.index(r['Queue_Id']), r['Kernel_Name'][:90]))
PY
python3 -c "
import json; j=json.load(open('gpurun_out/q_bench.json')); print('ms_per_step under the profiler', j['ms_per_step'])"
rm -rf gpurun_out/q

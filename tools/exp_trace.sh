#!/bin/bash
# one TTA step, kernel by kernel (no graph), under rocprofv3 --kernel-trace; prints the per-kernel summary of one step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_x
PTTA_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -o x -- python3 bench.py --steps 20 --warmup 5 --no-nlspn --no-cpu-baseline > gpurun_out/prof_x.log 2>&1
python3 tools/trace_step.py gpurun_out/prof_x/x_kernel_trace.csv ${1:-30} > gpurun_out/prof_x_summary.txt
python3 bench.py --steps 50 --warmup 10 --no-nlspn --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GRAPH ms_per_step', round(d['ms_per_step'],4))" >> gpurun_out/prof_x_summary.txt

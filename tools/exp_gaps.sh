#!/bin/bash
# graph-replay kernel trace of the bench command -> timeline summary of one step (busy / idle / overlap)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_g
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_g -o x -- python3 bench.py --steps 20 --warmup 5 --no-nlspn --no-cpu-baseline > gpurun_out/prof_g.log 2>&1
python3 tools/trace_gaps.py gpurun_out/prof_g/x_kernel_trace.csv 2 > gpurun_out/prof_g_gaps.txt
python3 tools/trace_gaps.py gpurun_out/prof_g/x_kernel_trace.csv 5 >> gpurun_out/prof_g_gaps.txt
cat gpurun_out/prof_g_gaps.txt | cut -c1-200

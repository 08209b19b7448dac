#!/bin/bash
# the ordered launch sequence of ONE replayed step (graph replay, frame pipelining off so that the step is one stream) under rocprofv3 --kernel-trace:
#   bash tools/exp_seq.sh [ENV=VAL ...]   -> gpurun_out/seq.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s
for V in "$@"; do export "$V"; done
PTTA_PIPELINE=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_s -o s -- python3 bench.py --steps 10 --warmup 3 --single-block --no-nlspn --no-cpu-baseline > gpurun_out/prof_s.log 2>&1
python3 tools/trace_sequence.py gpurun_out/prof_s/s_kernel_trace.csv ${SEQ_STEP:-} > gpurun_out/seq.txt
rm -rf gpurun_out/prof_s
grep -E "gemm|bn_|head_|loss" gpurun_out/seq.txt; tail -1 gpurun_out/seq.txt

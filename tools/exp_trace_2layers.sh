#!/bin/bash
# kernel trace of the MSG_CHN 2layers step (kernel by kernel): per-kernel summary of one step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_2l
cat > /tmp/run_2l.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
print(bench.msgchn_2layers_workload())
PY
PTTA_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_2l -o x -- python3 /tmp/run_2l.py > gpurun_out/prof_2l.log 2>&1
python3 tools/trace_step.py gpurun_out/prof_2l/x_kernel_trace.csv ${1:-40} > gpurun_out/prof_2l_summary.txt
cut -c1-150 gpurun_out/prof_2l_summary.txt

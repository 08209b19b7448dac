#!/bin/bash
# NLSPN step: kernel trace grouped by (kernel, grid)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_n
cat > /tmp/run_nl.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
print(bench.nlspn_workload(1, 3))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n -o x -- python3 /tmp/run_nl.py > gpurun_out/prof_n.log 2>&1
python3 tools/prof_by_grid.py gpurun_out/prof_n "${1:-}" ${2:-45} > gpurun_out/prof_n_grid.txt 2>&1
tail -2 gpurun_out/prof_n.log | cut -c1-400

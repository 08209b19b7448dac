#!/bin/bash
# in-kernel s_memtime stamps of the generic stride-1 conv (TIMING variant, selected for the 3344-block launches = the full-resolution 64->64 layers)
cd $GRAFT_REPO_ROOT
cat > /tmp/run_nl3.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
d = bench.nlspn_workload(1, 1)
PY
PTTA_S1_STAMPS=3344 python3 /tmp/run_nl3.py 2>&1 | grep "^blk" > gpurun_out/s1_stamps.txt
wc -l gpurun_out/s1_stamps.txt; tail -48 gpurun_out/s1_stamps.txt

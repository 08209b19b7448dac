#!/bin/bash
# kernel trace of the CostDCNet (480x640) and NLSPN (352x1216) steps: per-kernel totals
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_o
cat > /tmp/run_other.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
which = sys.argv[1]
print(bench.costdcnet_workload(4) if which == 'costdcnet' else bench.nlspn_workload(2, 3))
PY
for W in costdcnet nlspn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_o/$W -o x -- python3 /tmp/run_other.py $W > gpurun_out/prof_o_$W.log 2>&1
  python3 tools/prof_top.py gpurun_out/prof_o/$W 25 > gpurun_out/prof_o_$W.txt 2>&1
done

#!/bin/bash
# A/B of environment switches for the NLSPN step on ONE box: bash tools/exp_nlspn_ab.sh "VAR=a" "VAR=b"
cd $GRAFT_REPO_ROOT
cat > /tmp/run_nl2.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
d = bench.nlspn_workload(3, 3)
print('ms_per_step %.3f eval %.3f finite %s' % (d['ms_per_step'], d['eval_forward_ms'], d['finite']))
PY
rm -f gpurun_out/ab_nl.txt
for rep in 1 2; do
  for V in "$@"; do
    echo "$V rep $rep $(env $V python3 /tmp/run_nl2.py 2>/dev/null | tail -1)" >> gpurun_out/ab_nl.txt
  done
done
cat gpurun_out/ab_nl.txt

#!/bin/bash
# NLSPN / CostDCNet step times of the tree's build (A) against proxytta/libptta_hip.alt.so (B) on one box, 3 interleaved repeats
cd $GRAFT_REPO_ROOT
L=tta-depth-completion_amd/proxytta
cp $L/libptta_hip.so /tmp/libA.so; cp $L/libptta_hip.alt.so /tmp/libB.so
for rep in 1 2 3; do
  for V in A B; do
    cp /tmp/lib$V.so $L/libptta_hip.so
    timeout 300 python3 - <<PY
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
n = bench.nlspn_workload(4, 3); c = bench.costdcnet_workload(12)
print('lib $V rep $rep nlspn ms_per_step %.3f eval %.3f | costdcnet ms_per_step %.3f eval %.3f' % (n['ms_per_step'], n['eval_forward_ms'], c['ms_per_step'], c['eval_forward_ms']))
PY
  done
done
cp /tmp/libA.so $L/libptta_hip.so

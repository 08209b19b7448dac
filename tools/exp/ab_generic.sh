#!/bin/bash
# A/B of two library builds on ONE box for the generic engine's workloads (NLSPN config 3, CostDCNet): A = libptta_hip.so, B = libptta_hip.alt.so
cd $GRAFT_REPO_ROOT
L=tta-depth-completion_amd/proxytta
cp $L/libptta_hip.so /tmp/libA.so; cp $L/libptta_hip.alt.so /tmp/libB.so
cat > /tmp/run_other.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
n = bench.nlspn_workload(2, 3, dtype='fp32', with_mixed=True); c = bench.costdcnet_workload(4)
print('nlspn step %.3f ms eval %.3f | mixed %.3f | costdcnet step %.3f eval %.3f' % (n['ms_per_step'], n['eval_forward_ms'], n['mixed_mode']['ms_per_step'], c['ms_per_step'], c['eval_forward_ms']))
PY
for rep in 1 2 3; do for V in A B; do cp /tmp/lib$V.so $L/libptta_hip.so; echo -n "lib $V rep $rep  "; python3 /tmp/run_other.py 2>/dev/null | tail -1; done; done
cp /tmp/libA.so $L/libptta_hip.so

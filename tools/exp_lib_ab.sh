#!/bin/bash
# A/B of two builds on ONE box: proxytta/libptta_hip.so (A, the tree's build) against proxytta/libptta_hip.alt.so (B, built from another
# revision beforehand); 3 interleaved repeats each
cd $GRAFT_REPO_ROOT
L=tta-depth-completion_amd/proxytta
cp $L/libptta_hip.so /tmp/libA.so; cp $L/libptta_hip.alt.so /tmp/libB.so
for rep in 1 2 3; do
  for V in A B; do
    cp /tmp/lib$V.so $L/libptta_hip.so
    python3 bench.py --steps 50 --warmup 10 --no-nlspn --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib $V rep $rep ms_per_step', round(d['ms_per_step'],4), 'plain', round(d['config']['ms_per_step_without_frame_pipelining'] or 0,4))"
  done
done
cp /tmp/libA.so $L/libptta_hip.so

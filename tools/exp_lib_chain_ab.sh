#!/bin/bash
# A/B/C of library builds on ONE box: chain microbenchmark (full / half resolution) and the headline step, interleaved repeats
cd $GRAFT_REPO_ROOT
L=tta-depth-completion_amd/proxytta
cp $L/libptta_hip.so /tmp/libA.so; VS="A"
if [ -f $L/libptta_hip.alt.so ]; then cp $L/libptta_hip.alt.so /tmp/libB.so; VS="A B"; fi
if [ -f $L/libptta_hip.alt2.so ]; then cp $L/libptta_hip.alt2.so /tmp/libC.so; VS="$VS C"; fi
for rep in $(seq 1 ${REPS:-2}); do
  for V in $VS; do
    cp /tmp/lib$V.so $L/libptta_hip.so
    python3 tools/bench_chain.py 2>/dev/null | grep "1/[12] x" | cut -c1-70 | sed "s/^/lib $V rep $rep  /"
    [ -n "$NOBENCH" ] || python3 bench.py --steps 50 --warmup 10 --no-self-check --no-nlspn --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib $V rep $rep ms_per_step', round(d['ms_per_step'],4), ' '.join('%s %.0f' % (k, v['us_per_step']) for k,v in d['roofline_by_class'].items() if isinstance(v,dict)))"
  done
done
cp /tmp/libA.so $L/libptta_hip.so

#!/bin/bash
# A/B of environment switches on ONE box: bash tools/exp_ab.sh "VAR=a" "VAR=b" ...   (3 interleaved repeats each, graph replay)
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab.txt
for rep in 1 2 3; do
  for V in "$@"; do
    env $V python3 bench.py --steps 100 --warmup 10 --no-nlspn --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V rep $rep ms_per_step', round(d['ms_per_step'],4))" >> gpurun_out/ab.txt
  done
done

#!/bin/bash
# A/B of bench settings on ONE box: bash tools/exp_ab.sh "PTTA_BENCH_OPTIONS=thru=0" "PTTA_BENCH_DTYPE=fp32" ...   (3 interleaved repeats each, graph replay)
# prints the pipelined (headline) and the call-by-call step time of every variant
cd $GRAFT_REPO_ROOT
REPS=${REPS:-3}
rm -f gpurun_out/ab.txt
for rep in $(seq 1 $REPS); do
  for V in "$@"; do
    env $V python3 bench.py --steps 50 --warmup 10 --no-nlspn --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V rep $rep ms_per_step', round(d['ms_per_step'],4), 'plain', round(d['config']['ms_per_step_without_frame_pipelining'] or 0,4))" >> gpurun_out/ab.txt
  done
done
cat gpurun_out/ab.txt

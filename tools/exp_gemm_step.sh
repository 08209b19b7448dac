#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in ws t128 duo wide; do
  for AUX in 1 0; do
    PTTA_GEMM=$V PTTA_AUX_STREAM=$AUX python3 bench.py --steps 60 --warmup 10 --no-nlspn --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gemm $V aux $AUX ms_per_step', round(d['ms_per_step'],4))" >> gpurun_out/gemm_step.txt
  done
done

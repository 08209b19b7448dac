#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for V in ws t128 duo wide; do
  rm -rf gpurun_out/prof_a
  PTTA_GEMM=$V PTTA_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -o x -- python3 bench.py --steps 6 --warmup 2 --no-nlspn --no-cpu-baseline > /dev/null 2>&1
  echo "== $V" >> gpurun_out/gemm_var.txt
  python3 tools/trace_step.py gpurun_out/prof_a/x_kernel_trace.csv 40 | grep gemm >> gpurun_out/gemm_var.txt
done

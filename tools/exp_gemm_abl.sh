#!/bin/bash
# ablations of gemm_x3_duo_kernel: rebuild heads.o with a switch, time the K=512 GEMMs of one step
cd $GRAFT_REPO_ROOT/tta-depth-completion_amd/csrc
for ABL in NONE WS_ABL_NOSTORE "WS_ABL_NOSTORE -DWS_ABL_NOLOAD" "WS_ABL_NOSTORE -DWS_ABL_NOLOAD -DWS_ABL_NOMFMA"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -D$ABL -c heads.hip -o heads.o && make -s ARCH=gfx950 >/dev/null 2>&1
  cd $GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
  rm -rf gpurun_out/prof_a
  PTTA_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -o x -- python3 bench.py --steps 6 --warmup 2 --no-nlspn --no-cpu-baseline > /dev/null 2>&1
  echo "== $ABL" >> gpurun_out/gemm_abl.txt
  python3 tools/trace_step.py gpurun_out/prof_a/x_kernel_trace.csv 40 | grep gemm >> gpurun_out/gemm_abl.txt
  cd $GRAFT_REPO_ROOT/tta-depth-completion_amd/csrc
done

#!/bin/bash
# CostDCNet: parity report of every golden + step time, bf16x6 forward (default) against bf16x3 (PTTA_X6=0)
mkdir -p gpurun_out
L=gpurun_out/cd_diag.log; : > $L
for x in 1 0; do
  echo "== PTTA_X6=$x" >> $L
  PTTA_X6=$x python tools/costdc_report.py 2>&1 | grep -v amdgpu.ids >> $L
  PTTA_X6=$x python - >> $L 2>&1 <<'PY'
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tta-depth-completion_amd')
import bench, json
print(json.dumps(bench.costdcnet_workload(frames=12)))
PY
done
cat $L

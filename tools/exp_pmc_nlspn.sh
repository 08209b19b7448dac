#!/bin/bash
# SQ counters per kernel for the NLSPN step: bash tools/exp_pmc_nlspn.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_n
cat > /tmp/run_nl.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
print(bench.nlspn_workload(1, 1))
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d gpurun_out/pmc_n -o x -- python3 /tmp/run_nl.py > gpurun_out/pmc_n.log 2>&1
python3 tools/pmc_summary.py "gpurun_out/pmc_n/*counter_collection.csv" > gpurun_out/pmc_n_summary.txt 2>&1

#!/bin/bash
# SQ counters per kernel for one TTA step stream (kernel by kernel): bash tools/exp_pmc.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_x
PTTA_GRAPH=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d gpurun_out/pmc_x -o x -- python3 bench.py --steps 4 --warmup 2 --no-nlspn --no-cpu-baseline > gpurun_out/pmc_x.log 2>&1
python3 tools/pmc_summary.py "gpurun_out/pmc_x/*counter_collection.csv" > gpurun_out/pmc_x_summary.txt 2>&1

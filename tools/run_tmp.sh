#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py tests/test_gpu_head_trainer.py -x -q 2>&1 | tail -5
bash tools/exp_ab.sh "PTTA_S1_SMALL=0" "PTTA_S1_SMALL=1"
cat gpurun_out/ab.txt

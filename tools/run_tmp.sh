#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_costdcnet.py tests/test_gpu_nlspn.py tests/test_gpu_syncbn.py tests/test_gpu_parity.py -x -q > gpurun_out/t.txt 2>&1
grep -a "passed\|failed\|Error" gpurun_out/t.txt | tail -5

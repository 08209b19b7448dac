#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_syncbn.py -x -q 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_costdcnet.py -x -q 2>&1 | tail -8

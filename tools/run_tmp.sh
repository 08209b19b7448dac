python -m pytest tests/test_gpu_parity.py -x -q -k merged 2>&1 | tail -3

python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py tests/test_gpu_head_trainer.py -x -q 2>&1 | tail -3

python -m pytest tests/test_gpu_syncbn.py tests/test_gpu_costdcnet_syncbn.py -x -q -s  2>&1 | grep -v "Gloo\|socket\|amdgpu.ids" | tail -30

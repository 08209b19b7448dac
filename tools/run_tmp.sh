python -m pytest tests/test_gpu_costdcnet_syncbn.py tests/test_gpu_syncbn.py -x -q 2>&1 | grep -v "Gloo\|socket\|amdgpu.ids" | tail -30
python bench.py --workload costdcnet-shared --steps 30 --warmup 5 2>&1 | tail -3

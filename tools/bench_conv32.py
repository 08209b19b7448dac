#!/usr/bin/env python3
"""One 32->32 3x3 stride-1 convolution of the hot path on a KITTI-size map, repeated: run under `rocprofv3 --kernel-trace --stats` to read the
kernel's duration (the test hook allocates and synchronises around every call, so host-side timing means nothing), or under `--pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` as the CALIBRATION of those counters on this kernel's own access pattern: the bytes are known exactly (input and output map
= pixels x 32 channels x element size), tools/traffic_from_pmc.py --calibrate turns the counters into factors.
  python3 tools/bench_conv32.py [batch] [fp32|narrow]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tta-depth-completion_amd'))
import torch
from proxytta.engine import op_conv32

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
h, w = 352, 1216
x = torch.randn(b, h, w, 32, device='cuda')
wt = torch.randn(32, 32, 3, 3, device='cuda') * 0.05
bias = torch.randn(32, device='cuda')
for _ in range(12):
    y = op_conv32(x, wt, bias, 0, relu_in=True, x3=True, dtype=dtype)
torch.cuda.synchronize()
print('ok', dtype, 'map bytes', b * h * w * 32 * (4 if dtype == 'fp32' else 2), float(y.abs().mean()))

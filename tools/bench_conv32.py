#!/usr/bin/env python3
"""One 32->32 3x3 stride-1 convolution of the hot path on a KITTI-size map, repeated: run under `rocprofv3 --kernel-trace --stats` to read the
kernel's duration (the test hook allocates and synchronises around every call, so host-side timing means nothing).  PTTA_S1_ABL selects a
resource ablation of the kernel (csrc/conv32.hip): 1 no stores, 2 no global loads, 4 no MFMAs, 8 no LDS reads."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tta-depth-completion_amd'))
import torch
from proxytta.engine import op_conv32

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h, w = 352, 1216
x = torch.randn(b, h, w, 32, device='cuda')
wt = torch.randn(32, 32, 3, 3, device='cuda') * 0.05
bias = torch.randn(32, device='cuda')
for _ in range(12):
    y = op_conv32(x, wt, bias, 0, relu_in=True, x3=True)
torch.cuda.synchronize()
print('ok', float(y.abs().mean()))

#!/usr/bin/env python3
"""In-kernel phase stamps of conv32_s1_x3_kernel<float, RELU, plain> (library built with `make DIAG=1`): a short chain of dependent launches on a
KITTI-size map; the stamped instantiation prints per-tile cycle counts from two blocks.  python tools/diag_x3.py [batch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tta-depth-completion_amd'))
import torch
from proxytta import _lib
from proxytta._lib import ptr
lib = _lib.load()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h, w = 352, 1216
x = torch.randn(b, h, w, 32, device='cuda')
wt = torch.randn(32, 32, 3, 3, device='cuda') * 0.05
bias = torch.randn(32, device='cuda')
a, c, aux = torch.empty_like(x), torch.empty_like(x), torch.randn_like(x)
us = ctypes.c_float(0)
rc = lib.ptta_op_conv32_chain(ptr(x), ptr(wt), ptr(bias), ptr(a), ptr(c), ptr(aux), b, h, w, 1, 8, 3, 1, ctypes.byref(us), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print('rc', rc, 'us per launch (with stamps + printf)', us.value)

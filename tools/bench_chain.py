#!/usr/bin/env python3
"""What ONE 32->32 stride-1 convolution launch costs inside a replayed hipGraph, by map size and epilogue (ptta_op_conv32_chain: a chain of
dependent launches, one graph, hipEvents around the replays).  The step's ~45 low-resolution launches are priced here without the profiler's
per-kernel overhead."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tta-depth-completion_amd'))
import torch
from proxytta import _lib
from proxytta._lib import ptr

lib = _lib.load()
H, W = 352, 1216
wt = torch.randn(32, 32, 3, 3, device='cuda') * 0.05
bias = torch.randn(32, device='cuda')
print('%-28s %8s %8s %8s %8s %8s %8s %8s %8s %8s   (us per LAYER: a replayed graph of 40 dependent launches -- plain, ReLU-mask and skip-add epilogues; '
      'direct = plain, launched directly; loop40 / loop6 = the layer loop, ONE launch per 40 / 6 plain layers with a device-wide barrier '
      'between layers, launched directly, bitwise equal to the direct chain; r = relaxed polls and one fence pair instead of acquire polls; nofence = loop40 without any fence: timing ablation, values unchecked)' % ('map', 'plain', 'mask', 'add', 'direct', 'loop40', 'loop6', 'loop40r', 'loop6r', 'nofence'))
for name, b, s in (('1/16 x2', 2, 16), ('1/8 x2', 2, 8), ('1/4 x1', 1, 4), ('1/4 x2', 2, 4), ('1/2 x1', 1, 2), ('1/2 x2', 2, 2), ('1/1 x1', 1, 1), ('1/1 x2', 2, 1)):
    h, w = H // s, W // s
    x = torch.randn(b, h, w, 32, device='cuda')
    a, c, aux = torch.empty_like(x), torch.empty_like(x), torch.randn_like(x)
    row = []
    for flags in (0, 2, 4, 8):
        us = ctypes.c_float(0)
        rc = lib.ptta_op_conv32_chain(ptr(x), ptr(wt), ptr(bias), ptr(a), ptr(c), ptr(aux), b, h, w, 1, flags, 40, 20, ctypes.byref(us),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        row.append(us.value)
    direct_out = {40: c.clone()}                     # 40 layers: the last one (r = 39) wrote buf_b
    if s >= 4:                                       # the layer loop: small maps only (one co-resident block per CU)
        for reps, var in ((40, 0), (6, 0), (40, 32), (6, 32), (40, 64)):
            if reps not in direct_out:
                rc = lib.ptta_op_conv32_chain(ptr(x), ptr(wt), ptr(bias), ptr(a), ptr(c), ptr(aux), b, h, w, 1, 8, reps, 2, ctypes.byref(us),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
                direct_out[reps] = c.clone()
            a.zero_(); c.zero_()
            rc = lib.ptta_op_conv32_chain(ptr(x), ptr(wt), ptr(bias), ptr(a), ptr(c), ptr(aux), b, h, w, 1, 16 | var, reps, 20, ctypes.byref(us),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, 'layer loop: rc %d (-62: a barrier spin ran out)' % rc
            assert var == 64 or torch.equal(c, direct_out[reps]) and bool(torch.isfinite(c).all()) and float(c.abs().max()) > 0, 'layer loop differs from the direct chain'
            row.append(us.value)
    else:
        row += [float('nan')] * 5
    mb = b * h * w * 32 * 4 / 1e6
    print('%-28s %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f   tensor %.1f MB' % (('%s (%dx%d)' % (name, h, w),) + tuple(row[:9]) + (mb,)))

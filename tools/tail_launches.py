#!/usr/bin/env python3
"""Every bracketed launch of ONE profiled step (ptta_profile: kernel by kernel on one stream, hipEvents around each launch) in launch order:
class and microseconds -- what the step's launches cost alone, without the profiler's per-kernel overhead (debug tensor "prof_seq").
usage: python tools/tail_launches.py [mixed|fp32] [N frames per call]"""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
import numpy as np
import torch
import bench
from proxytta import synth
from proxytta.engine import ADAPTED, Engine

dtype = sys.argv[1] if len(sys.argv) > 1 else 'mixed'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, W = bench.H, bench.W
eng = Engine(N, H, W, dtype=dtype, **bench.HP)
sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(bench.MODE).items()}
eng.load_state_dict(sd)
for name in ADAPTED:
    eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(i * 16, H, W, N)] for i in range(4)]
for i in range(10):
    eng.step(*frames[i % 4])
CLASSES = ['s1_relu_large', 's1_relu_small', 's1_plain_large', 's1_plain_small', 'strided_large', 'strided_small', 'heads', 'in_out_convs', 'rest']
runs = []
for rep in range(9):
    eng.profile(True)
    eng.step(*frames[rep % 4])
    torch.cuda.synchronize()
    runs.append(eng.debug_tensor('prof_seq').cpu().numpy().reshape(-1, 2))
    eng.profile(False)
n = min(len(r) for r in runs)
us = np.median(np.stack([r[:n, 1] for r in runs]), axis=0)
cum = 0.0
print('# %s: %d bracketed launches of one step, median of 9 profiled steps (one stream, events around each launch)' % (dtype, n))
for k in range(n):
    cum += us[k]
    print('%3d  %-16s %7.1f us   cum %7.1f' % (k, CLASSES[int(runs[0][k, 0])], us[k], cum))

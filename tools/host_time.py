#!/usr/bin/env python3
"""Host-side cost of one pipelined step call (no sync inside the loop): per-call perf_counter times of the first calls after a synchronize
(queue empty: the true host cost) and the steady state (GPU-bound: the host waits on a full queue).  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from proxytta import synth
from proxytta.engine import ADAPTED, Engine
H, W = bench.H, bench.W
dtype = sys.argv[1] if len(sys.argv) > 1 and '=' not in sys.argv[1] else 'mixed'
opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in sys.argv[1:] if '=' in kv}
eng = Engine(1, H, W, dtype=dtype, options=opts, **bench.HP)
sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(bench.MODE).items()}
eng.load_state_dict(sd)
for name in ADAPTED:
    eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(i, H, W, 1)] for i in range(4)]
for i in range(10):
    eng.step(*frames[i % 4], next_frame=frames[(i + 1) % 4])
torch.cuda.synchronize()
ts = []
t_all = time.perf_counter()
for i in range(10, 50):
    t0 = time.perf_counter()
    eng.step(*frames[i % 4], next_frame=frames[(i + 1) % 4])
    ts.append((time.perf_counter() - t0) * 1e6)
t_host = (time.perf_counter() - t_all) * 1e6
torch.cuda.synchronize()
t_tot = (time.perf_counter() - t_all) * 1e6
print(dtype, opts)
print('host us per call:', ' '.join('%d' % t for t in ts))
print('host loop %.0f us for 40 calls (%.0f per call); with final sync %.0f us (%.0f per step)' % (t_host, t_host / 40, t_tot, t_tot / 40))

#!/usr/bin/env python3
"""Frames/s of the fused step as a function of the per-GPU batch (the reference's scripts use n_batch 16)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np, torch
from proxytta import synth
from proxytta.engine import Engine, adapted_names
MODE = 'meta_selfsup_seq_1layer_ema'
for n in [int(x) for x in (sys.argv[1:] or ['1', '2', '4', '8'])]:
    eng = Engine(n, 352, 1216, dtype='fp32', lr=1e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(MODE).items()}
    eng.load_state_dict(sd)
    for name in adapted_names():
        eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
    img, sp = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(0, 352, 1216, n)]
    for _ in range(5):
        eng.step(img, sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        eng.step(img, sp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print('batch %d: %.3f ms/step  %.1f frames/s' % (n, dt * 1e3, n / dt), flush=True)
    eng.close()

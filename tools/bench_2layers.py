#!/usr/bin/env python3
"""Step time of the MSG_CHN `2layers` meta layer (Res_Conv(32,128), the recipe of bash/adapt/adapt_msgchn_vkitti.sh) beside the
`1layer` headline configuration, 352x1216, batch 1.   python tools/bench_2layers.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
import numpy as np
import torch

from proxytta import synth
from proxytta.engine import Engine

H, W = 352, 1216
for meta, mode in (('1layer', 'meta_selfsup_seq_1layer_ema'), ('2layers', 'meta_selfsup_seq_2layers_ema')):
    eng = Engine(1, H, W, meta=meta, lr=1e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(mode).items()}
    eng.load_state_dict(sd)
    for name in eng.adapted:
        eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(i, H, W, 1)] for i in range(4)]
    for i in range(10):
        eng.step(*frames[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        eng.step(*frames[i % 4])
    torch.cuda.synchronize()
    print('%s: %.3f ms/step (%d adapted tensors)' % (meta, (time.perf_counter() - t0) * 10, len(eng.adapted)))
    eng.close()

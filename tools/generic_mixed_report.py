#!/usr/bin/env python3
"""NLSPN / CostDCNet in the mixed mode of the generic engine (fp32 storage; single-MFMA products for the proxy frames and / or the data
gradients) against the REAL reference's full-size fixtures, with the step time of each variant.  python tools/generic_mixed_report.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from proxytta import synth  # noqa: E402
from proxytta.engine import Engine  # noqa: E402
from tests.util import rel_mae  # noqa: E402
from tests.test_gpu_nlspn import nlspn_frame  # noqa: E402
from tests.test_gpu_costdcnet import costdc_frame, MAX_DEPTH  # noqa: E402

GD = os.path.join(ROOT, 'tests', 'golden')
VARIANTS = [('fp32', 'fp32', ()), ('mixed proxy-only', 'mixed', ('backward',)), ('mixed backward-only', 'mixed', ('proxy',)), ('mixed (both)', 'mixed', ()),
            ('mixed backward-only, rounded w (r5)', 'mixed', ('proxy', 'bwd_rounded_w')), ('mixed (both), rounded w (r5)', 'mixed', ('bwd_rounded_w',))]


def pix(t, g):
    return t.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']]


def build(backbone, n, h, w, hp, dtype, keep):
    kw = dict(legacy_offset=True) if backbone == 'nlspn' else dict(max_predict_depth=MAX_DEPTH)
    eng = Engine(n, h, w, backbone=backbone, dtype=dtype, keep=keep, **kw, **hp)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in (synth.formula_state_dict_nlspn() if backbone == 'nlspn' else synth.formula_state_dict_costdcnet()).items()}
    if backbone == 'costdcnet':
        for k in list(sd):
            if k.startswith('enc2d.') and '.downsample.1.' in k:
                sd[k] = sd[k.replace('.downsample.1.', '.norm3.')]
        eng.load_state_dict(sd)
    else:
        eng.load_state_dict({k: v for k, v in sd.items() if v.dtype == torch.float32})
    adapted = {}
    for k in eng.adapted:
        p = sd[k].clone().contiguous()
        adapted[k] = (p, torch.zeros_like(p), torch.zeros_like(p))
        eng.bind_adapted(k, *adapted[k])
    return eng, adapted


def run(backbone, name):
    g = np.load(os.path.join(GD, name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    v = [float(x) for x in g['hp']]
    hp = dict(lr=v[0], betas=(v[1], v[2]), eps=v[3], weight_decay=v[4], w_sparse_depth=v[5], w_smoothness=v[6], w_cos=v[7],
              max_input_depth=v[8] if backbone == 'nlspn' else None)
    same = int(g['same_frame']) == 1 if 'same_frame' in g.files else True
    for label, dtype, keep in VARIANTS:
        try:
            eng, adapted = build(backbone, n, h, w, hp, dtype, keep)
        except RuntimeError as e:                      # CostDCNet: ptta_create refuses the mode (see profiles/r06_nlspn_costdcnet_mixed.txt)
            print('%-28s %-36s refused: %s' % (name, label, str(e)[:60]), flush=True)
            continue
        fr = nlspn_frame(0, h, w, n) if backbone == 'nlspn' else costdc_frame(0, h, w, n, float(g['density']))
        raw, image1, sparse = [torch.from_numpy(x).cuda() for x in fr]
        out = []
        for s in range(steps if same else 1):
            p = 's%d/' % s
            info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
            d_eval = eng.forward_eval(image1, sparse)
            li = np.abs(info.cpu().numpy() - g[p + 'loss_info']) / np.maximum(np.abs(g[p + 'loss_info']), 1e-12)
            gerr = [rel_mae(eng.grad(k[len(p) + 5:], adapted[k[len(p) + 5:]][0]), g[k]) for k in g.files if k.startswith(p + 'grad/')] if s == 0 else []
            out.append('s%d train %.2e eval %.2e loss %.1e%s' % (s, rel_mae(pix(depth, g), g[p + 'depth_train_pix']), rel_mae(pix(d_eval, g), g[p + 'depth_eval_pix']),
                                                                  li.max(), (' grad max %.2e' % max(gerr)) if gerr else ''))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.step(image1, sparse, loss_image=raw)
        torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            eng.forward_eval(image1, sparse)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / 5
        print('%-28s %-36s step %.2f ms eval %.2f ms | %s' % (name, label, 1e3 * ts, 1e3 * te, ' | '.join(out)), flush=True)
        eng.close()


for nm in ('nlspn_352x1216_legacy_inner3', 'nlspn_228x304_legacy'):
    run('nlspn', nm)
for nm in ('costdcnet_480x640', 'costdcnet_320x400'):
    run('costdcnet', nm)

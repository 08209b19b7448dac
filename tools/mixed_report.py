#!/usr/bin/env python3
"""What tests/test_gpu_mixed.py bounds, measured: the mixed mode against every reference fixture (relative MAE per quantity and step).
Bounds in the tests = 2x the worst figure printed here.  python tools/mixed_report.py [dtype]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from proxytta import synth  # noqa: E402
from tests.util import golden_hp, make_engine, rel_mae  # noqa: E402

DT = sys.argv[1] if len(sys.argv) > 1 else 'mixed'
GD = os.path.join(ROOT, 'tests', 'golden')


def run(name, meta='1layer', full=False):
    g = np.load(os.path.join(GD, name + '.npz'))
    m = [int(x) for x in g['meta']]
    h, w, n, steps = m[:4]
    frame0 = m[4] if len(m) > 4 else 0
    hp, gain = golden_hp(g)
    hb = float(g['head_bias']) if 'head_bias' in g.files else 0.0
    eng, sd, adapted = make_engine(n, h, w, DT, hp, gain, None, meta=meta, head_bias=hb)
    for s in range(steps):
        image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(frame0 + s, h, w, n)]
        p = 's%d/' % s
        info, depth = eng.step(image, sparse, want_depth=True)
        d_eval = eng.forward_eval(image, sparse)
        li = np.abs(info.cpu().numpy() - g[p + 'loss_info']) / np.maximum(np.abs(g[p + 'loss_info']), 1e-12)
        if full:
            flat = lambda t: t.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']]
            dt, de = rel_mae(flat(depth), g[p + 'depth_train_pix']), rel_mae(flat(d_eval), g[p + 'depth_eval_pix'])
        else:
            dt, de = rel_mae(depth, g[p + 'depth_train']), rel_mae(d_eval, g[p + 'depth_eval'])
        out = '%-32s s%d depth_train %.2e depth_eval %.2e loss_info %.1e' % (name, s, dt, de, li.max())
        if p + 'row_idx' in g.files:
            idx = g[p + 'row_idx']
            out += ' emb %.1e ref %.1e' % (rel_mae(eng.debug_tensor('emb').view(-1, 512).cpu()[idx], g[p + 'emb_rows']),
                                           rel_mae(eng.debug_tensor('ref').view(-1, 512).cpu()[idx], g[p + 'ref_rows']))
        gs, ps, ms, vs = [], [], [], []
        for k, (prm, mm, vv) in adapted.items():
            if p + 'grad/' + k not in g.files:
                continue
            gref = g[p + 'grad/' + k]
            if np.abs(gref).max() < 1e-6:
                continue
            gs.append(rel_mae(eng.grad(k, prm), gref)); ps.append(rel_mae(prm, g[p + 'param/' + k]))
            if p + 'exp_avg/' + k in g.files:
                ms.append(rel_mae(mm, g[p + 'exp_avg/' + k])); vs.append(rel_mae(vv, g[p + 'exp_avg_sq/' + k]))
        if gs:
            out += ' | grad max %.2e param max %.2e' % (max(gs), max(ps))
        if ms:
            out += ' exp_avg %.2e exp_avg_sq %.2e' % (max(ms), max(vs))
        bufs = [rel_mae(sd[k[len(p) + 4:]], g[k]) for k in g.files if k.startswith(p + 'buf/') and not k[len(p) + 4:].startswith('proj_t')]
        if bufs:
            out += ' buf %.1e' % max(bufs)
        print(out, flush=True)
    eng.close()


for nm, meta in (('msgchn_1layer_352x1216', '1layer'), ('msgchn_1layer_256x320', '1layer'), ('msgchn_2layers_352x1216', '2layers'),
                 ('msgchn_2layers_256x320', '2layers'), ('msgchn_1layer_64x96_seq10', '1layer')):
    run(nm, meta, full=True)
for nm in ('msgchn_1layer_32x48', 'msgchn_1layer_64x96', 'msgchn_1layer_36x52_pad', 'msgchn_1layer_32x48_n2', 'msgchn_1layer_32x48_wcos1',
           'msgchn_2layers_32x48', 'msgchn_1layer_64x96_gate_below', 'msgchn_1layer_64x96_gate_above'):
    run(nm, '2layers' if '2layers' in nm else '1layer')

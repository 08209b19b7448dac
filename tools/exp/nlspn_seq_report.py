"""NLSPN over the 24-frame REAL-reference sequence (tests/golden/nlspn_96x320_legacy_seq24.npz): scored depth per step, both modes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np, torch
from tests.util import rel_mae
from tests.test_gpu_nlspn import make_nlspn, nlspn_frame
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'nlspn_96x320_legacy_seq24.npz'))
h, w, n, steps = [int(x) for x in g['meta']]
lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid = [float(x) for x in g['hp']]
for dtype, keep in (('fp32', ()), ('mixed', ()), ('mixed', ('bwd_rounded_w',))):
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=mid, dtype=dtype, keep=keep)
    eng, sd, adapted = make_nlspn(n, h, w, hp, legacy=True)
    tr, ev, li = [], [], []
    for s in range(steps):
        raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(s, h, w, n)]
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        d_eval = eng.forward_eval(image1, sparse)
        pix = lambda t: t.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']]
        tr.append(rel_mae(pix(depth), g['s%d/depth_train_pix' % s])); ev.append(rel_mae(pix(d_eval), g['s%d/depth_eval_pix' % s]))
        li.append(float(np.max(np.abs(info.cpu().numpy() - g['s%d/loss_info' % s]) / np.maximum(np.abs(g['s%d/loss_info' % s]), 1e-12))))
    print('%-6s %-18s eval max %.2e last %.2e | train max %.2e | loss max %.1e' % (dtype, ','.join(keep), max(ev), ev[-1], max(tr), max(li)))
    print('       eval per step: ' + ' '.join('%.1e' % x for x in ev))
    eng.close()

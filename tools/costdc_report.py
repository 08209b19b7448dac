#!/usr/bin/env python3
"""CostDCNet HIP path vs the reference goldens: prints every error figure (no asserts).  python tools/costdc_report.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
from tests.test_gpu_costdcnet import costdc_frame, make_costdc  # noqa: E402
from tests.util import rel_mae  # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')


def pix(t, g):
    return t.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']]


for name in sys.argv[1:] or ['costdcnet_64x96', 'costdcnet_64x64_n2', 'costdcnet_72x100_pad', 'costdcnet_320x400', 'costdcnet_480x640']:
    g = np.load(os.path.join(G, name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, md = [float(x) for x in g['hp']]
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=None)
    sampled = 'pix_idx' in g.files
    for impl in (os.environ['IMPLS'].split(',') if os.environ.get('IMPLS') else (['default'] if sampled else ['naive', 'default'])):
        eng, sd, ad = make_costdc(n, h, w, hp, impl=impl)
        for s in range(steps):
            raw, im, sp = [torch.from_numpy(x).cuda() for x in costdc_frame(s, h, w, n, float(g['density']))]
            info, depth = eng.step(im, sp, loss_image=raw, want_depth=True)
            p = 's%d/' % s
            dt = rel_mae(pix(depth, g), g[p + 'depth_train_pix']) if sampled else rel_mae(depth, g[p + 'depth_train'])
            gerr = {k: rel_mae(eng.grad(k, ad[k][0]), g[p + 'grad/' + k]) for k in eng.adapted}
            perr = {k: rel_mae(ad[k][0], g[p + 'param/' + k]) for k in eng.adapted}
            berr = {k[len(p) + 4:]: rel_mae(sd[k[len(p) + 4:]], g[k]) for k in g.files if k.startswith(p + 'buf/')}
            de = eng.forward_eval(im, sp)
            dev = rel_mae(pix(de, g), g[p + 'depth_eval_pix']) if sampled else rel_mae(de, g[p + 'depth_eval'])
            # eval forward from the REFERENCE's post-step parameters (isolates the eval path from Adam's +-lr sign noise)
            keep = {k: ad[k][0].clone() for k in eng.adapted}
            for k in eng.adapted:
                ad[k][0].copy_(torch.from_numpy(g[p + 'param/' + k]))
            de2 = eng.forward_eval(im, sp)
            dev2 = rel_mae(pix(de2, g), g[p + 'depth_eval_pix']) if sampled else rel_mae(de2, g[p + 'depth_eval'])
            for k in eng.adapted:
                ad[k][0].copy_(keep[k])
            wg = max(gerr, key=gerr.get); wb = max(berr, key=berr.get)
            print('%-22s %-7s s%d depth %.2e loss %s grad max %.2e (%s) param max %.2e buf max %.2e (%s) eval %.2e eval@refparams %.2e' % (
                name, impl, s, dt, np.array2string(np.abs(info.cpu().numpy() / g[p + 'loss_info'] - 1), precision=1), gerr[wg], wg,
                max(perr.values()), berr[wb], wb, dev, dev2), flush=True)
        eng.close()

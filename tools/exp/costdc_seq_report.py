"""CostDCNet over the 16-frame REAL-reference sequence (tests/golden/costdcnet_96x128_seq16.npz): scored depth per step beside the reference's
own separation after a one-ulp change of one weight"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np, torch
from tests.util import rel_mae
name = sys.argv[1] if len(sys.argv) > 1 else 'costdcnet_96x128_seq16'
src = open(os.path.join(ROOT, 'tools', 'generic_mixed_report.py')).read().split("\nfor nm in ('nlspn_352x1216_legacy_inner3'")[0]
G = {'__file__': os.path.join(ROOT, 'tools', 'generic_mixed_report.py'), '__name__': 'gmr'}
exec(compile(src, 'gmr', 'exec'), G)
g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
h, w, n, steps = [int(x) for x in g['meta']]
v = [float(x) for x in g['hp']]
hp = dict(lr=v[0], betas=(v[1], v[2]), eps=v[3], weight_decay=v[4], w_sparse_depth=v[5], w_smoothness=v[6], w_cos=v[7], max_input_depth=None)
pix = lambda t: t.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']]
floor = [rel_mae(g['alt/s%d/depth_eval_pix' % s], g['s%d/depth_eval_pix' % s]) for s in range(steps)]
print('reference vs itself (1 ulp): ' + ' '.join('%.1e' % x for x in floor))
for impl in ('default', 'naive'):
    if impl == 'naive': os.environ['PTTA_CONV_IMPL'] = 'naive'
    eng, adapted = G['build']('costdcnet', n, h, w, hp, 'fp32', ())
    os.environ.pop('PTTA_CONV_IMPL', None)
    ev, tr, li = [], [], []
    for s in range(steps):
        raw, image1, sparse = [torch.from_numpy(x).cuda() for x in G['costdc_frame'](s, h, w, n, float(g['density']))]
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        ev.append(rel_mae(pix(eng.forward_eval(image1, sparse)), g['s%d/depth_eval_pix' % s])); tr.append(rel_mae(pix(depth), g['s%d/depth_train_pix' % s]))
        li.append(float(np.max(np.abs(info.cpu().numpy() - g['s%d/loss_info' % s]) / np.maximum(np.abs(g['s%d/loss_info' % s]), 1e-12))))
    print('%-8s eval: ' % impl + ' '.join('%.1e' % x for x in ev))
    print('         train: ' + ' '.join('%.1e' % x for x in tr) + ' | loss max %.1e' % max(li))
    eng.close()

"""Measured deviations of the NLSPN inner_iter = 3 run at 352x1216 from the reference fixture (bounds of
tests/test_gpu_nlspn.py::test_three_steps_on_one_full_size_frame_match_the_reference)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')]
from tests.test_gpu_nlspn import make_nlspn, nlspn_frame
from tests.util import rel_mae
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'nlspn_352x1216_legacy_inner3.npz'))
h, w, n, steps = [int(x) for x in g['meta']]
lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid = [float(x) for x in g['hp']]
hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=mid)
eng, sd, adapted = make_nlspn(n, h, w, hp, legacy=True)
raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
for s in range(steps):
    p = 's%d/' % s
    info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
    ev = eng.forward_eval(image1, sparse)
    f = lambda t, key: rel_mae(t.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']], g[key + '_pix'])
    print('step', s, 'train depth %.2e eval depth %.2e' % (f(depth, p + 'depth_train'), f(ev, p + 'depth_eval')), 'loss_info', info.cpu().numpy(), g[p + 'loss_info'])
eng.close()

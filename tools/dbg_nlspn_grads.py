"""Debug aid: is the step-2 gradient gap a stale-state bug or Adam amplifying fp32 sign noise?  Fresh engine loaded
with the ORACLE's post-step-1 parameters and Adam state, then one step on frame 1, compared with the oracle's step 2."""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tta-depth-completion_amd')
import numpy as np, torch
from tests.test_gpu_nlspn import make_nlspn, _oracle, nlspn_frame, HP
from tests.util import rel_mae
n, h, w = 1, 32, 64
N, o = _oracle()
raw, image1, sparse = [torch.from_numpy(x) for x in nlspn_frame(0, h, w, n)]
o.step(image1, sparse, loss_image=raw)
eng, sd, adapted = make_nlspn(n, h, w)
for i, k in enumerate(o.names):
    adapted[k][0].copy_(o.P[k].detach()); adapted[k][1].copy_(o.opt.m[i]); adapted[k][2].copy_(o.opt.v[i])
eng.set_adam_step(1)
raw, image1, sparse = [torch.from_numpy(x) for x in nlspn_frame(1, h, w, n)]
r = o.step(image1, sparse, loss_image=raw)
info, depth = eng.step(image1.cuda(), sparse.cuda(), loss_image=raw.cuda(), want_depth=True)
print('depth', rel_mae(depth, r['depth']))
worst = sorted(((rel_mae(eng.grad(k, adapted[k][0]), r['grads'][k]), k) for k in eng.adapted), reverse=True)
print('worst', [(round(e, 5), k) for e, k in worst[:6]], 'median', worst[len(worst) // 2][0])
print('param', max(rel_mae(adapted[k][0], o.P[k].detach()) for k in eng.adapted))

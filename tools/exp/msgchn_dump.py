"""MSG_CHN steps on the loaded library (both modes, both meta layers, shapes whose stride-1 launches end in half tiles and not): gradients,
parameters after three steps, loss terms and depths to an .npz -- tools/exp/lib_bitwise.sh compares two builds bit for bit"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np, torch
from proxytta import synth
from tests.util import make_engine
res = {}
HP = dict(lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=0.1, w_cos=1.0, max_input_depth=100.0)
for (n, h, w) in ((1, 352, 1216), (1, 360, 1216), (2, 352, 1216), (1, 256, 320), (3, 200, 616), (1, 488, 1600)):
    for dtype in ('mixed', 'fp32'):
        for meta in (('1layer', '2layers') if (n, h, w) == (1, 352, 1216) else ('1layer',)):
            eng, sd, adapted = make_engine(n, h, w, dtype=dtype, hp=HP, meta=meta)
            tag = '%dx%dx%d/%s/%s' % (n, h, w, dtype, meta)
            frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(50 + 7 * i, h, w, n)] for i in range(3)]
            for s in range(3):
                if s == 1:
                    info, depth = eng.step(*frames[s], next_frame=frames[2], want_depth=True)     # one pipelined call in the middle
                else:
                    info, depth = eng.step(*frames[s], want_depth=True)
                res['%s/s%d/info' % (tag, s)] = info.cpu().numpy(); res['%s/s%d/depth' % (tag, s)] = depth.cpu().numpy()
                if s == 0:
                    for k in adapted: res['%s/grad/%s' % (tag, k)] = eng.grad(k, adapted[k][0]).cpu().numpy()
            res['%s/eval' % tag] = eng.forward_eval(*frames[0]).cpu().numpy()
            for k in adapted: res['%s/param/%s' % (tag, k)] = adapted[k][0].cpu().numpy()
            eng.close()
np.savez(sys.argv[1], **res)
print('wrote', sys.argv[1], len(res), 'arrays')

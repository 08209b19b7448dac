#!/usr/bin/env python3
"""Measured gradient errors behind the bounds of tests/test_gpu_head_trainer.py, test_gpu_nlspn.py, test_gpu_parity.py and
test_gpu_fullsize.py, default arithmetic (bf16x3) against exact arithmetic, first step: python tools/grad_report.py   (GPU)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
from proxytta import synth  # noqa: E402
from proxytta.engine import HEAD_PARAMS  # noqa: E402
from tests.util import rel_mae  # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')


def head():
    from tests.test_gpu_head_trainer import make_head_engine
    for name in ('head_reverse_32x48_n2', 'head_forward_32x48_n2', 'head_reverse_64x96'):
        z = np.load(os.path.join(G, name + '.npz'))
        h, w, n, steps = (int(v) for v in z['meta'])
        lr, b1, b2, eps, wd, tau = (float(v) for v in z['hp'])
        reverse = 'reverse' in str(z['loss_type'])
        for mode in ('default', 'exact'):
            os.environ.pop('PTTA_ARITH', None)
            if mode == 'exact':
                os.environ['PTTA_ARITH'] = 'exact'
            eng, sd, _, _ = make_head_engine(n, h, w, dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd), tau)
            os.environ.pop('PTTA_ARITH', None)
            image, sparse = (torch.from_numpy(a).cuda() for a in synth.synthetic_frame(0, h, w, n))
            eng.head_forward(image, sparse, reverse); eng.head_backward()
            worst_max = worst_mean = 0.0; wk = ''
            for k in HEAD_PARAMS:
                g = eng.head_grad(k, sd[k])
                if g is None:
                    continue
                g = g.cpu().numpy().astype(np.float64)
                key = 's0/grad/' + k
                if key in z.files:
                    ref = z[key].astype(np.float64)
                else:
                    idx = np.linspace(0, g.shape[0] - 1, 24).astype(np.int64); ref = z[key + '#rows'].astype(np.float64); g = g[idx]
                if np.abs(ref).max() < 1e-8:
                    continue
                d = np.abs(g - ref)
                a, b = d.max() / np.abs(ref).max(), d.mean() / np.abs(ref).mean()
                if a > worst_max:
                    worst_max, wk = a, k
                worst_mean = max(worst_mean, b)
            print('head %-24s %-7s rows %5d  worst entry / largest entry %.2e (%s)  worst mean error / mean magnitude %.2e' % (name, mode, n * (h // 4) * (w // 4), worst_max, wk, worst_mean), flush=True)
            eng.close()


def head_other():
    """tests/test_gpu_head_trainer.py::test_head_trainer_against_oracle_other_shape, both arithmetic modes, both steps"""
    from oracle import head_oracle as HO
    from tests.test_gpu_head_trainer import make_head_engine
    n, h, w = 3, 48, 80
    hp = dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    for loss_type in ('head_selfsup_seq_ema_reverse', 'head_selfsup_seq_ema'):
        for mode in ('default', 'exact'):
            os.environ.pop('PTTA_ARITH', None)
            if mode == 'exact':
                os.environ['PTTA_ARITH'] = 'exact'
            eng, sd, sd_np, _ = make_head_engine(n, h, w, hp, 0.99)
            os.environ.pop('PTTA_ARITH', None)
            o = HO.HeadTrainerOracle(sd_np, loss_type, max_input_depth=80.0, tau=0.99, **hp)
            for s in range(2):
                image, sparse = synth.synthetic_frame(10 + s, h, w, n)
                r = o.step(torch.from_numpy(image), torch.from_numpy(sparse))
                eng.head_forward(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda(), 'reverse' in loss_type)
                loss = eng.head_backward()
                wm = wa = 0.0; wk = ''
                for k, g in r['grads'].items():
                    want = g.numpy().astype(np.float64)
                    if np.abs(want).max() < 1e-8:
                        continue
                    d = np.abs(eng.head_grad(k, sd[k]).cpu().numpy().astype(np.float64) - want)
                    a, b = d.max() / np.abs(want).max(), d.mean() / np.abs(want).mean()
                    if a > wm:
                        wm, wk = a, k
                    wa = max(wa, b)
                print('head-other %-30s %-7s step %d rows %d loss diff %.1e worst entry / largest %.2e (%s) worst mean / mean %.2e' % (loss_type, mode, s, n * (h // 4) * (w // 4), abs(float(loss) - r['loss']), wm, wk, wa), flush=True)
                eng.head_adam_step()
            eng.close()


def nlspn():
    from tests.test_gpu_nlspn import make_nlspn, nlspn_frame
    for name in ('nlspn_32x64', 'nlspn_48x80_n2', 'nlspn_32x64_legacy', 'nlspn_40x56_n2_legacy', 'nlspn_96x320_legacy', 'nlspn_228x304_legacy', 'nlspn_352x1216_legacy'):
        g = np.load(os.path.join(G, name + '.npz'))
        h, w, n, steps = [int(x) for x in g['meta']]
        lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid = [float(x) for x in g['hp']]
        hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=mid)
        names = [str(x) for x in g['adapted_names']]
        for impl in (('default',) if h * w > 100000 else ('naive', 'default')):
            eng, sd, ad = make_nlspn(n, h, w, hp, impl=impl, legacy=bool(int(g['legacy'])))
            raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
            info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
            gn = np.array([float(eng.grad(k, ad[k][0]).double().norm()) for k in names])
            en = np.abs(gn - g['s0/grad_norms']) / np.maximum(np.abs(g['s0/grad_norms']), 1e-6)
            worst = 0.0; wk = ''
            for key in g.files:
                if key.startswith('s0/grad/'):
                    k = key[len('s0/grad/'):]
                    e = rel_mae(eng.grad(k, ad[k][0]), g[key])
                    if e > worst:
                        worst, wk = e, k
            dk = 's0/depth_train' if 's0/depth_train' in g.files else None
            de = rel_mae(depth, g[dk]) if dk else rel_mae(depth.cpu().numpy().reshape(-1)[g['pix_idx']], g['s0/depth_train_pix'])
            print('nlspn %-24s %-7s depth %.2e  worst gradient norm error %.2e (%s)  worst full-gradient rel MAE %.2e (%s)' % (
                name, impl, de, en.max(), names[int(en.argmax())], worst, wk), flush=True)
            eng.close()


def msgchn():
    from tests.util import golden_hp, make_engine
    for name, meta in (('msgchn_1layer_32x48', '1layer'), ('msgchn_1layer_64x96', '1layer'), ('msgchn_1layer_36x52_pad', '1layer'), ('msgchn_1layer_32x48_n2', '1layer'),
                       ('msgchn_1layer_32x48_wcos1', '1layer'), ('msgchn_2layers_32x48', '2layers'), ('msgchn_1layer_256x320', '1layer'), ('msgchn_2layers_256x320', '2layers'),
                       ('msgchn_1layer_352x1216', '1layer'), ('msgchn_2layers_352x1216', '2layers')):
        g = np.load(os.path.join(G, name + '.npz'))
        h, w, n, steps = [int(x) for x in g['meta'][:4]]
        frame0 = int(g['meta'][4]) if len(g['meta']) > 4 else 0
        hp, gain = golden_hp(g)
        for impl in ((None,) if h * w > 50000 else ('exact', None)):
            eng, sd, ad = make_engine(n, h, w, 'fp32', hp, gain, impl, meta=meta)
            first = later = pfirst = plater = 0.0
            for s in range(steps):
                image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(frame0 + s, h, w, n)]
                eng.step(image, sparse)
                for k in ad:
                    key = 's%d/grad/%s' % (s, k)
                    if key in g.files and np.abs(g[key]).max() >= 1e-6:
                        e = rel_mae(eng.grad(k, ad[k][0]), g[key]); pe = rel_mae(ad[k][0], g['s%d/param/%s' % (s, k)])
                        if s == 0:
                            first = max(first, e); pfirst = max(pfirst, pe)
                        else:
                            later = max(later, e); plater = max(plater, pe)
            print('msg_chn %-26s %-7s steps %d worst gradient rel MAE: first step %.2e later %.2e | parameters: first %.2e later %.2e' % (name, impl or 'default', steps, first, later, pfirst, plater), flush=True)
            eng.close()


if __name__ == '__main__':
    which = sys.argv[1:] or ['head', 'head_other', 'nlspn', 'msgchn']
    for wname in which:
        globals()[wname]()

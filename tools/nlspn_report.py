#!/usr/bin/env python3
"""NLSPN at 352x1216: (i) depth / loss agreement of the bf16x3 matrix-core path with the exact direct-kernel path,
(ii) step and eval time per frame at batch 1, 2, 4.   python tools/nlspn_report.py"""
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

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
H, W = 352, 1216


def make(n, naive=False):
    if naive:
        os.environ['PTTA_CONV_IMPL'] = 'naive'
    else:
        os.environ.pop('PTTA_CONV_IMPL', None)
    eng = Engine(n, H, W, backbone='nlspn', legacy_offset=True, lr=1e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
    os.environ.pop('PTTA_CONV_IMPL', None)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
    eng.load_state_dict({k: v for k, v in sd.items() if v.dtype == torch.float32})
    keep = {k: (sd[k].clone().contiguous(), torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in eng.adapted}
    for k in eng.adapted:
        eng.bind_adapted(k, *keep[k])
    return eng, keep


def frame(n):
    image01, sparse = synth.synthetic_frame(0, H, W, n)
    raw = np.floor(image01 * 255.0).astype(np.float32)
    return torch.from_numpy(raw).cuda(), torch.from_numpy(((raw / np.float32(255.0) - MEAN) / STD).astype(np.float32)).cuda(), torch.from_numpy(sparse).cuda()


def rel(a, b):
    return float((a - b).abs().mean() / b.abs().mean())


def main():
    raw, image, sparse = frame(1)
    res = {}
    for mode in ('naive', 'default'):
        eng, keep = make(1, naive=(mode == 'naive'))
        d_eval0 = eng.forward_eval(image, sparse)
        info, d_train = eng.step(image, sparse, loss_image=raw, want_depth=True)
        d_eval1 = eng.forward_eval(image, sparse)
        res[mode] = (d_eval0.clone(), d_train.clone(), d_eval1.clone(), info.cpu().numpy(), {k: keep[k][0].clone() for k in keep})
        eng.close()
    a, b = res['default'], res['naive']
    print('depth rel MAE (bf16x3 vs exact): eval before step %.2e, train %.2e, eval after one step %.2e' % (rel(a[0], b[0]), rel(a[1], b[1]), rel(a[2], b[2])))
    print('loss_info default', a[3], 'exact', b[3])
    print('max |param difference| after one Adam step (lr 1e-3): %.2e' % max(float((a[4][k] - b[4][k]).abs().max()) for k in a[4]))
    for n in (1, 2, 4):
        raw, image, sparse = frame(n)
        eng, keep = make(n)
        eng.step(image, sparse, loss_image=raw)
        eng.forward_eval(image, sparse)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.step(image, sparse, loss_image=raw)
        torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            eng.forward_eval(image, sparse)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / 5
        print('batch %d: step %.2f ms (%.2f ms/frame), eval %.2f ms (%.2f ms/frame)' % (n, ts * 1e3, ts * 1e3 / n, te * 1e3, te * 1e3 / n))
        eng.close()


if __name__ == '__main__':
    main()

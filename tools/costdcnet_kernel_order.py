#!/usr/bin/env python3
"""Which enumeration of the 27 kernel offsets do the reference's PRETRAINED sparse-encoder weights expect?  (build container only)

MinkowskiEngine is absent from the reference tree; oracle/minkowski_lite.py (and csrc/costdc_kernels.hip after it) index a 3x3x3 kernel as
k = (d0+1) + 3 (d1+1) + 9 (d2+1) -- first spatial axis fastest -- which with formula weights no test can tell from the opposite order.  The
published weights (external_src/costdcnet/weights/{enc2d,enc3d,unet3d}.pth) can: the network was trained with ONE of the two, so on a synthetic
indoor scene (planes, a sphere, shading correlated with depth; 1500 sampled points) the prediction error of the real reference network is
computed under both enumerations (and, as a control, with the sparse kernels' offsets shuffled at random).
Prints MAE / RMSE of the eval forward against the scene's true depth for each."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from oracle import minkowski_lite as ML  # noqa: E402
ML.install(sys.modules)
import make_golden as MG  # noqa: E402

W = '/root/reference/external_src/costdcnet/weights'
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)


def scene(h, w, seed):
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    u, v = (x - w / 2) / w, (y - h / 2) / h
    depth = 3.0 + 1.5 * u + 0.8 * v                                     # back wall, slanted
    floor = 1.2 / np.maximum(v + 0.15, 1e-3) * 0.35                     # floor plane in perspective
    depth = np.minimum(depth, np.where(v > 0.05, floor, 1e9))
    cx, cy, r = 0.15, -0.05, 0.18
    d2 = (u - cx) ** 2 + (v - cy) ** 2
    sph = 2.0 - np.sqrt(np.maximum(r * r - d2, 0.0)) * 2.0
    depth = np.where(d2 < r * r, np.minimum(depth, sph), depth).astype(np.float32)
    depth = np.clip(depth, 0.4, 7.5)
    shade = 1.0 / (0.4 + 0.25 * depth)
    tex = 0.08 * np.sin(x / 7.0) * np.sin(y / 5.0)
    img = np.stack([np.clip(shade + tex, 0, 1), np.clip(0.9 * shade + 0.5 * tex, 0, 1), np.clip(0.8 * shade - tex, 0, 1)], 0).astype(np.float32)
    sparse = np.zeros((h, w), np.float32)
    idx = rng.choice(h * w, 1500, replace=False)
    sparse.reshape(-1)[idx] = depth.reshape(-1)[idx]
    return img[None], sparse[None, None], depth[None, None]


def main():
    h, w = 240, 320
    ema, _ = MG.import_reference()
    model = ema.ExternalModel_Adapt('costdcnet', 0.1, 10.0, max_input_depth=None, device=torch.device('cpu'))
    net = model.model.model
    for part in ('enc2d', 'enc3d', 'unet3d'):
        sd = torch.load(os.path.join(W, part + '.pth'), map_location='cpu')
        getattr(net, part).load_state_dict(sd)
    net.eval()
    img, sparse, depth = scene(h, w, 0)
    image = torch.from_numpy(((img - MEAN) / STD).astype(np.float32))
    first = ML.kernel_offsets
    r = (-1, 0, 1)
    orders = {
        'first axis fastest (shipped)': lambda ks: first(ks),
        'last axis fastest': lambda ks: [(0, 0, 0)] if ks == 1 else [(d0, d1, d2) for d0 in r for d1 in r for d2 in r],
    }
    rng = np.random.RandomState(1)
    perm = rng.permutation(27)
    orders['random permutation (control)'] = lambda ks: [(0, 0, 0)] if ks == 1 else [first(3)[j] for j in perm]
    for name, fn in orders.items():
        ML.kernel_offsets = fn
        with torch.no_grad():
            out = model.forward(image=image, sparse_depth=torch.from_numpy(sparse), loss_type='pretrain')
        out = out[0] if isinstance(out, (list, tuple)) else out
        e = (out.numpy() - depth)
        print('%-34s MAE %.4f m  RMSE %.4f m   (at the 1500 given points: MAE %.4f)' % (
            name, np.abs(e).mean(), np.sqrt((e ** 2).mean()), np.abs(e[sparse > 0]).mean()), flush=True)
    ML.kernel_offsets = first


if __name__ == '__main__':
    main()

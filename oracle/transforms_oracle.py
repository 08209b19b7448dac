"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's crop / flip augmentation, numpy index arithmetic.
Only tests/ may import this.  Pinned by tests/golden/transforms_geometric.npz (outputs of the real class on CPU).

Follows src/transforms.py: decisions :230 (`do_random_transform`), :337-350 (crop coin, shape, start offsets), :391-403
(flip coins); data movement :955-988 (crop), :990-1011 (horizontal flip), :1013-1034 (vertical flip); intrinsics :380-383,
:1330-1378.
"""
import numpy as np
import torch


def draw(n, H, W, crop, flips, prob):
    """Decisions in the reference's draw order from torch's global CPU generator / numpy's global state."""
    do_random_transform = torch.rand(n) <= prob
    d = {'crop': None, 'hflip': np.zeros(n, bool), 'vflip': np.zeros(n, bool)}
    enabled = -1 not in crop
    is_range = enabled and len(crop) == 4
    if (enabled and bool(torch.rand(1) <= 0.50)) or is_range:
        if is_range:
            ch = np.random.randint(low=crop[0], high=crop[2] + 1)
            cw = np.random.randint(low=crop[1], high=crop[3] + 1)
        else:
            ch, cw = crop
        sy = torch.randint(low=0, high=H - ch + 1, size=(n,)).numpy()
        sx = torch.randint(low=0, high=W - cw + 1, size=(n,)).numpy()
        d['crop'] = (int(ch), int(cw), sy, sx)
    if 'horizontal' in flips:
        d['hflip'] = torch.logical_and(do_random_transform, torch.rand(n) <= 0.50).numpy()
    if 'vertical' in flips:
        d['vflip'] = torch.logical_and(do_random_transform, torch.rand(n) <= 0.50).numpy()
    return d


def apply(x, d):
    """x: N x C x H x W float32 -> cropped then flipped copy."""
    n, c, H, W = x.shape
    ch, cw, sy, sx = d['crop'] if d['crop'] is not None else (H, W, np.zeros(n, int), np.zeros(n, int))
    out = np.empty((n, c, ch, cw), np.float32)
    for b in range(n):
        ys = sy[b] + (np.arange(ch)[::-1] if d['vflip'][b] else np.arange(ch))
        xs = sx[b] + (np.arange(cw)[::-1] if d['hflip'][b] else np.arange(cw))
        out[b] = x[b][:, ys][:, :, xs]
    return out


def adjust_intrinsics(K, d, H, W):
    K = K.copy()
    if d['crop'] is not None:
        K[:, 0, 2] -= np.float32(W - d['crop'][1])
        K[:, 1, 2] -= np.float32(H - d['crop'][0])
    return K

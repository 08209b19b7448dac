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


# ---- rotation / resize-and-crop / photometric jitter: PARITY UNPINNED ------------------------------------------------------------
# The reference calls torchvision.transforms.functional (src/transforms.py:730-838 adjust_*, :1060-1066 rotate, :1263-1267 resize);
# torchvision (pinned to 0.10.1+cu111 by the reference's README.md:74) is absent from this image and from the reference tree, so the
# functions below restate its tensor code path (torchvision/transforms/functional.py rotate / resize / adjust_*,
# functional_tensor.py _gen_affine_grid / _apply_grid_transform / _blend / rgb_to_grayscale) with the torch primitives it calls.
import math  # noqa: E402

import torch.nn.functional as F  # noqa: E402


def tv_rotate(img, angle, bilinear):
    """functional.rotate(img (C,H,W) float tensor, angle degrees, interpolation, expand=False, center=None, fill=None)."""
    rot = math.radians(-angle)                                  # rotate() passes -angle to _get_inverse_affine_matrix
    # _get_inverse_affine_matrix(center (0,0), angle, translate (0,0), scale 1, shear (0,0)): [d, -b, 0, -c, a, 0] with
    # a = cos, b = -sin, c = sin, d = cos of `rot`
    a, b_, c_, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    matrix = [d, -b_, 0.0, -c_, a, 0.0]
    h, w = img.shape[-2], img.shape[-1]
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3)                              # _gen_affine_grid(theta, w, h, ow = w, oh = h)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    grid = base.view(1, h * w, 3).bmm(theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h])).view(1, h, w, 2)
    return F.grid_sample(img.unsqueeze(0), grid, mode='bilinear' if bilinear else 'nearest', padding_mode='zeros', align_corners=False)[0]


def rotate(x, do, angles, bilinear):
    out = x.clone()
    for b in range(x.shape[0]):
        if do[b]:
            out[b] = tv_rotate(x[b], float(angles[b]), bilinear)
    return out


def resize_and_crop(x, do, rh, rw, sy, sx, bilinear, depth_div=False):
    """src/transforms.py:1250-1281: functional.resize (= F.interpolate, align_corners=False for bilinear) then the crop."""
    n, c, H, W = x.shape
    out = x.clone()
    for b in range(n):
        if do[b]:
            r = F.interpolate(x[b:b + 1], size=[int(rh[b]), int(rw[b])], mode='bilinear' if bilinear else 'nearest', align_corners=False if bilinear else None)[0]
            r = r[..., int(sy[b]):int(sy[b]) + H, int(sx[b]):int(sx[b]) + W]
            if depth_div:
                r = r / (float(rw[b]) / W)
            out[b] = r
    return out


def _gray(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)


def _blend(img1, img2, ratio):
    ratio = float(ratio)
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, 255.0).to(img1.dtype)


def photometric(x, brightness=None, contrast=None, saturation=None):
    """src/transforms.py:236-311: float images -> uint8, then per sample adjust_brightness / adjust_contrast / adjust_saturation
    (each (do[b], factor[b]) pair or None), then .float()."""
    u = x.to(torch.uint8)
    for b in range(x.shape[0]):
        if brightness is not None and brightness[0][b]:
            u[b] = _blend(u[b], torch.zeros_like(u[b]), brightness[1][b])
        if contrast is not None and contrast[0][b]:
            mean = torch.mean(_gray(u[b]).to(torch.float32), dim=(-3, -2, -1), keepdim=True)
            u[b] = _blend(u[b], mean, contrast[1][b])
        if saturation is not None and saturation[0][b]:
            u[b] = _blend(u[b], _gray(u[b]), saturation[1][b])
    return u.float()

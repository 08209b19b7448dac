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


# ---- gamma, hue, noise, patch removal, crop-and-pad, resize-and-pad (enabled by no adapt script) -----------------------------------
# gamma / hue / pad / resize are torchvision calls in the reference (src/transforms.py:783, :808, :1128-1132, :1196-1207): restated from
# torchvision 0.10.1's tensor code path (functional_tensor.py adjust_gamma / adjust_hue / _rgb2hsv / _hsv2rgb / pad / resize,
# functional.py convert_image_dtype) -- PARITY UNPINNED for those four; add_noise (:839-876), remove_random_patches / random_nonzero
# (:878-953) and the index arithmetic of crop_and_pad (:1072-1135) / resize_and_pad (:1137-1220) are torch-only code of the reference and
# are pinned by tests/golden/transforms_extra.npz (outputs of the REAL class, tests/golden/make_golden_transforms_extra.py).
def tv_convert_to_float(img_u8):
    return img_u8.to(torch.float32) / 255.0                       # convert_image_dtype(uint8 -> float32)


def tv_convert_to_u8(img_f):
    return img_f.mul(255.0 + 1.0 - 1e-3).to(torch.uint8)          # convert_image_dtype(float32 -> uint8): eps = 1e-3


def tv_adjust_gamma(img, gamma, gain=1):
    """functional_tensor.adjust_gamma: uint8 images go through [0, 1] floats and back; FLOAT images are clamped to [0, 1] as they are
    (the reference reaches this branch when gamma is the only photometric option: do_photometric_transforms (:102-106) leaves gamma
    out, so the images are not cast to uint8 first)."""
    result = img
    if not torch.is_floating_point(img):
        result = tv_convert_to_float(result)
    result = (gain * result ** float(gamma)).clamp(0, 1)
    if not torch.is_floating_point(img):
        result = tv_convert_to_u8(result)
    return result


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc = (maxc - r) / cr_divisor
    gc = (maxc - g) / cr_divisor
    bc = (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = (hr + hg + hb)
    h = torch.fmod((h / 6.0 + 1.0), 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp((v * (1.0 - s)), 0.0, 1.0)
    q = torch.clamp((v * (1.0 - s * f)), 0.0, 1.0)
    t = torch.clamp((v * (1.0 - (s * (1.0 - f)))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def tv_adjust_hue(img, hue_factor):
    """functional_tensor.adjust_hue on a (3, H, W) tensor: uint8 -> [0, 1] floats -> HSV, h = (h + factor) % 1, -> RGB -> (x * 255).to(uint8)."""
    hue_factor = float(hue_factor)
    if not (-0.5 <= hue_factor <= 0.5):
        raise ValueError('hue_factor ({}) is not in [-0.5, 0.5].'.format(hue_factor))
    orig_dtype = img.dtype
    if img.dtype == torch.uint8:
        img = img.to(dtype=torch.float32) / 255.0
    img = _rgb2hsv(img)
    h, s, v = img.unbind(dim=-3)
    h = (h + hue_factor) % 1.0
    img = torch.stack((h, s, v), dim=-3)
    out = _hsv2rgb(img)
    if orig_dtype == torch.uint8:
        out = (out * 255.0).to(dtype=orig_dtype)
    return out


def tv_pad(img, padding, fill=0, padding_mode='constant'):
    """functional_tensor.pad(img (C, H, W), [left, top, right, bottom]): constant / edge (= replicate) / reflect / symmetric."""
    left, top, right, bottom = [int(p) for p in padding]
    if padding_mode == 'constant':
        return F.pad(img, [left, right, top, bottom], mode='constant', value=float(fill))
    if padding_mode == 'symmetric':
        H, W = img.shape[-2:]
        xi = [i for i in range(left)][::-1] + list(range(W)) + [W - 1 - i for i in range(right)]
        yi = [i for i in range(top)][::-1] + list(range(H)) + [H - 1 - i for i in range(bottom)]
        return img[..., torch.tensor(yi, dtype=torch.long), :][..., torch.tensor(xi, dtype=torch.long)]
    mode = {'edge': 'replicate', 'reflect': 'reflect'}[padding_mode]
    return F.pad(img.unsqueeze(0).float(), [left, right, top, bottom], mode=mode)[0].to(img.dtype)


def tv_resize(img, size, bilinear):
    """functional_tensor.resize(img (C, H, W), [h, w]): F.interpolate, align_corners=False for bilinear, no antialiasing (0.10.1)."""
    return F.interpolate(img.unsqueeze(0), size=[int(size[0]), int(size[1])], mode='bilinear' if bilinear else 'nearest',
                         align_corners=False if bilinear else None)[0]


def tv_adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def tv_adjust_contrast(img, f):
    mean = torch.mean(_gray(img).to(torch.float32), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, f)


def tv_adjust_saturation(img, f):
    return _blend(img, _gray(img), f)


def photometric_full(x, brightness=None, contrast=None, gamma=None, hue=None, saturation=None):
    """src/transforms.py:236-311 with every option: the images are cast to uint8 only when brightness / contrast / hue / saturation is
    CONFIGURED (pass (zeros, factors) for a configured option whose coin came up 'no'); the reference's order is brightness, contrast,
    gamma, hue, saturation, then .float()."""
    as_u8 = any(p is not None for p in (brightness, contrast, hue, saturation))
    u = x.to(torch.uint8) if as_u8 else x.clone()
    for b in range(x.shape[0]):
        if brightness is not None and brightness[0][b]:
            u[b] = _blend(u[b], torch.zeros_like(u[b]), brightness[1][b])
        if contrast is not None and contrast[0][b]:
            mean = torch.mean(_gray(u[b]).to(torch.float32), dim=(-3, -2, -1), keepdim=True)
            u[b] = _blend(u[b], mean, contrast[1][b])
        if gamma is not None and gamma[0][b]:
            u[b] = tv_adjust_gamma(u[b], gamma[1][b])
        if hue is not None and hue[0][b]:
            u[b] = tv_adjust_hue(u[b], hue[1][b])
        if saturation is not None and saturation[0][b]:
            u[b] = _blend(u[b], _gray(u[b]), saturation[1][b])
    return u.float()


def add_noise(x, do, noise, noise_type, spread):
    """:839-876 with the noise field given (`noise`: what torch.randn / torch.rand returned for the sample)."""
    out = x.clone()
    for b in range(x.shape[0]):
        if do[b]:
            out[b] = x[b] + spread * (noise[b] if noise_type == 'gaussian' else (noise[b] - 0.5))
    return out


def remove_patches(x, do, selected, patch_sizes):
    """:878-924 with the selection given: selected[b] = (ys, xs) index tensors of the chosen nonzero pixels (random_nonzero's result)."""
    out = x.clone()
    for b in range(x.shape[0]):
        if not do[b]:
            continue
        image = x[b]
        mask = torch.sum(torch.abs(image), dim=0, keepdim=True)
        mask = torch.where(mask > 0, torch.ones_like(mask), torch.zeros_like(mask))
        ys, xs = selected[b]
        mask[0, ys.long(), xs.long()] = float('inf')
        k = [int(v) for v in patch_sizes[b]]
        mask = F.max_pool2d(mask.unsqueeze(0), kernel_size=k, stride=1, padding=[int(v // 2) for v in k])[0]
        mask[mask == float('inf')] = 0.0
        out[b] = mask * image
    return out


def crop_and_pad(x, do, sy, sx, ey, ex, pad_top, pad_bottom, pad_left, pad_right, padding_mode='constant', fill=0):
    """:1072-1135."""
    out = x.clone()
    for b in range(x.shape[0]):
        if do[b]:
            im = x[b][..., int(sy[b]):int(ey[b]), int(sx[b]):int(ex[b])]
            out[b] = tv_pad(im, (int(pad_left[b]), int(pad_top[b]), int(pad_right[b]), int(pad_bottom[b])), fill, padding_mode)
    return out


def resize_and_pad(x, do, rh, rw, pad_top, pad_bottom, pad_left, pad_right, bilinear, padding_mode='constant', fill=0):
    """:1137-1220 (max_shape = the tensor's own H x W; a larger result is cropped from its bottom-right corner as the reference does)."""
    n, c, H, W = x.shape
    out = x.clone()
    for b in range(n):
        if do[b]:
            im = tv_resize(x[b], (int(rh[b]), int(rw[b])), bilinear)
            im = tv_pad(im, (int(pad_left[b]), int(pad_top[b]), int(pad_right[b]), int(pad_bottom[b])), fill, padding_mode)
            h, w = im.shape[-2:]
            if H < h or W < w:
                im = im[..., h - H:h, w - W:w]
            out[b] = im
    return out


def apply_draw(tf, d, arrs, padding_modes=('constant',), interpolation_modes=('nearest',), generator=None):
    """Every tensor of `arrs` (N x C x H x W float32, CPU) through the decisions `d` of proxytta.Transforms.draw in the reference's order
    (src/transforms.py:236-655) with the restatements above; `tf` = the Transforms object (noise type / spread).  The patch-removal
    selection is drawn here, last, from torch's generator -- as random_nonzero does (:926-953)."""
    n = arrs[0].shape[0]
    pm = list(padding_modes) + [list(padding_modes)[-1]] * (len(arrs) - len(padding_modes))
    im = list(interpolation_modes) + [list(interpolation_modes)[-1]] * (len(arrs) - len(interpolation_modes))
    bil = [m in ('bilinear', 2) for m in im]
    out = [a.clone() for a in arrs]
    if any(d.get(k) is not None for k in ('brightness', 'contrast', 'gamma', 'hue', 'saturation')):
        out = [photometric_full(a, d.get('brightness'), d.get('contrast'), d.get('gamma'), d.get('hue'), d.get('saturation')) for a in out]
    if d.get('noise') is not None:
        do, fields = d['noise']
        out = [add_noise(a, do, f, tf.random_noise_type, tf.random_noise_spread) for a, f in zip(out, fields)]
    if d.get('crop') is not None or d.get('hflip') is not None or d.get('vflip') is not None:
        H, W = out[0].shape[-2:]
        ch, cw, sy, sx = d['crop'] if d.get('crop') is not None else (H, W, torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32))
        dd = {'crop': (ch, cw, np.asarray(sy), np.asarray(sx)),
              'hflip': np.asarray(d['hflip']).astype(bool) if d.get('hflip') is not None else np.zeros(n, bool),
              'vflip': np.asarray(d['vflip']).astype(bool) if d.get('vflip') is not None else np.zeros(n, bool)}
        out = [torch.from_numpy(apply(a.numpy(), dd)) for a in out]
    if d.get('rotate') is not None:
        out = [rotate(a, d['rotate'][0], d['rotate'][1], b_) for a, b_ in zip(out, bil)]
    if d.get('resize') is not None:
        do, rh, rw, sy, sx = d['resize'][:5]
        out = [resize_and_crop(a, do, rh, rw, sy, sx, b_, depth_div=bool(tf.resize_scaling_depth) and i != 0) if i < len(interpolation_modes) else a
               for i, (a, b_) in enumerate(zip(out, bil))]
    if d.get('crop_pad') is not None:
        do, sy, sx, ey, ex, pt, pb, pl, pr = d['crop_pad']
        out = [crop_and_pad(a, do, sy, sx, ey, ex, pt, pb, pl, pr, m) for a, m in zip(out, pm)]
    if d.get('resize_pad') is not None:
        do, rh, rw, pt, pb, pl, pr = d['resize_pad']
        out = [resize_and_pad(a, do, rh, rw, pt, pb, pl, pr, b_, m) for a, b_, m in zip(out, bil, pm)]
    if d.get('remove') is not None:
        do, densities, sizes = d['remove']
        res = []
        for a in out:
            selected = [None] * n
            for b in range(n):
                if do[b]:
                    nz = (a[b].abs().sum(dim=0) > 0).nonzero(as_tuple=True)
                    count = int(nz[0].shape[0])
                    perm = torch.randperm(count, generator=generator)[0:int(densities[b].float() * count)]   # fp32 product, src/transforms.py:944
                    selected[b] = (nz[0][perm], nz[1][perm])
            res.append(remove_patches(a, do, selected, sizes))
        out = res
    return out

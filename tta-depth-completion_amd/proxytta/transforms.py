"""Host-side mirror of the reference's `Transforms` (src/transforms.py:7-191 constructor, :192-664 `transform`): the
augmentations the adapt scripts enable (bash/adapt/*.sh) -- random crop to shape, horizontal / vertical flip, rotation,
resize-and-crop with the intrinsics adjustment, brightness / contrast / saturation jitter.  The random DRAWS are made here with
torch's CPU generator and numpy's global state in the reference's order (so a CPU run of the reference with the same
`torch.manual_seed` / `np.random.seed` takes the same decisions); the data movement is hand-written HIP (csrc/augment.hip):
one launch per tensor and transform.

Parity: crop / flip are bit-exact against the real class (tests/golden/transforms_geometric.npz).  Rotation, resizing and the
photometric jitter are `torchvision.transforms.functional` calls in the reference (torchvision is absent from this image and
from the reference tree): they restate torchvision 0.10.1's tensor algorithms and are PARITY UNPINNED, held to
oracle/transforms_oracle.py (the same algorithms on torch's own grid_sample / interpolate) and to property tests.
Gamma, hue, noise, patch removal, crop-and-pad and resize-and-pad (enabled by no adapt script) are built too: gamma / hue / pad /
resize restate torchvision (parity unpinned); noise, patch removal, every draw and the crop / pad index arithmetic are pinned by
tests/golden/transforms_extra.npz (outputs of the real class).  Image normalisation is fused into the engine's first convolution
(`Engine.set_image_norm`) and is refused here.
"""
import random

import numpy as np
import torch

from . import _lib
from ._lib import ptr


def _unset(v):
    return -1 in list(v)


class Transforms(object):
    def __init__(self,
                 normalized_image_range=None,
                 random_brightness=[-1, -1],
                 random_contrast=[-1, -1],
                 random_gamma=[-1, -1],
                 random_hue=[-1, -1],
                 random_saturation=[-1, -1],
                 random_noise_type='none',
                 random_noise_spread=-1,
                 random_remove_patch_percent_range=[-1, -1],
                 random_remove_patch_size=[1, 1],
                 random_crop_to_shape=[-1, -1],
                 random_flip_type=['none'],
                 random_rotate_max=0,
                 random_crop_and_pad=[-1, -1],
                 random_resize_and_crop=[-1, -1],
                 random_resize_and_pad=[-1, -1],
                 resize_scaling_depth=False):
        if normalized_image_range is not None:
            raise NotImplementedError(
                'proxytta.Transforms: image normalisation is fused into the engine: Engine.set_image_norm(normalized_image_range)')
        # src/transforms.py:84-96
        self.do_random_brightness = not _unset(random_brightness); self.random_brightness = random_brightness
        self.do_random_contrast = not _unset(random_contrast); self.random_contrast = random_contrast
        self.do_random_saturation = not _unset(random_saturation); self.random_saturation = random_saturation
        self.do_random_gamma = not _unset(random_gamma); self.random_gamma = random_gamma
        self.do_random_hue = not _unset(random_hue); self.random_hue = random_hue
        # :102-106: gamma does NOT count -- with gamma alone the images are not cast to uint8 and torchvision's float branch applies
        self.do_photometric_transforms = self.do_random_brightness or self.do_random_contrast or self.do_random_hue or self.do_random_saturation
        # :108-125
        self.do_random_noise = random_noise_type != 'none' and random_noise_spread > -1
        self.random_noise_type, self.random_noise_spread = random_noise_type, random_noise_spread
        if self.do_random_noise and random_noise_type not in ('gaussian', 'uniform'):
            raise ValueError('Unsupported noise type: {}'.format(random_noise_type))
        self.do_random_remove_patch = not _unset(random_remove_patch_percent_range)
        self.random_remove_patch_percent_range = random_remove_patch_percent_range
        if len(random_remove_patch_size) == 4:
            self.random_remove_patch_size_height = list(range(random_remove_patch_size[0], random_remove_patch_size[2] + 2, 2))
            self.random_remove_patch_size_width = list(range(random_remove_patch_size[1], random_remove_patch_size[3] + 2, 2))
        else:
            self.random_remove_patch_size_height = [random_remove_patch_size[0]]
            self.random_remove_patch_size_width = [random_remove_patch_size[1]]
        # :154-161, :174-182
        self.do_random_crop_and_pad = not _unset(random_crop_and_pad)
        self.random_crop_and_pad_min, self.random_crop_and_pad_max = random_crop_and_pad[0], random_crop_and_pad[1]
        if self.do_random_crop_and_pad:
            assert self.random_crop_and_pad_min < self.random_crop_and_pad_max
            assert self.random_crop_and_pad_max <= 1
        self.do_random_resize_and_pad = not _unset(random_resize_and_pad)
        self.random_resize_and_pad_min, self.random_resize_and_pad_max = random_resize_and_pad[0], random_resize_and_pad[1]
        if self.do_random_resize_and_pad:
            assert self.random_resize_and_pad_min < self.random_resize_and_pad_max
            assert self.random_resize_and_pad_min > 0
            assert self.random_resize_and_pad_max <= 1.0
        # :151-152, :165-172, :184
        self.do_random_rotate = random_rotate_max > 0
        self.random_rotate_max = random_rotate_max
        self.do_random_resize_and_crop = not _unset(random_resize_and_crop)
        self.random_resize_and_crop_min, self.random_resize_and_crop_max = random_resize_and_crop[0], random_resize_and_crop[1]
        if self.do_random_resize_and_crop:
            assert self.random_resize_and_crop_min < self.random_resize_and_crop_max
            assert self.random_resize_and_crop_min >= 1.0
        self.resize_scaling_depth = resize_scaling_depth
        # src/transforms.py:126-148
        self.do_random_crop_to_shape = not _unset(random_crop_to_shape)
        self.do_random_crop_to_shape_exact = False
        self.do_random_crop_to_shape_range = False
        if self.do_random_crop_to_shape:
            if len(random_crop_to_shape) == 2:
                self.do_random_crop_to_shape_exact = True
                self.random_crop_to_shape_height, self.random_crop_to_shape_width = random_crop_to_shape
            elif len(random_crop_to_shape) == 4:
                self.do_random_crop_to_shape_range = True
                (self.random_crop_to_shape_height_min, self.random_crop_to_shape_width_min,
                 self.random_crop_to_shape_height_max, self.random_crop_to_shape_width_max) = random_crop_to_shape
            else:
                raise ValueError('Unsupported input for random crop to shape: {}'.format(random_crop_to_shape))
        self.do_random_horizontal_flip = 'horizontal' in random_flip_type
        self.do_random_vertical_flip = 'vertical' in random_flip_type
        self.last_draw = None

    # ---- decisions (host, reference draw order) ------------------------------------------------------
    def draw(self, n_batch, n_height, n_width, random_transform_probability=0.0, generator=None, channels=(3,)):
        """The random decisions of one `transform` call, drawn in the reference's order (:230, :244-311, :337-350, :391-497) from
        torch's CPU generator (`generator=None`: the global one `torch.manual_seed` seeds) and numpy's global state for
        the range crop (:346-352).  `channels`: the channel count of every tensor of images_arr -- add_noise draws its field per tensor and
        sample at this point of the stream (:862-866).  The patch-removal SELECTION (randperm over the nonzero pixels, :946-947) depends on the
        data and is drawn by `apply`, last, as in the reference."""
        def rand(k):
            return torch.rand(k, generator=generator)
        d = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': None, 'resize': None, 'brightness': None, 'contrast': None, 'saturation': None,
             'gamma': None, 'hue': None, 'noise': None, 'crop_pad': None, 'resize_pad': None, 'remove': None}
        do_random_transform = rand(n_batch) <= random_transform_probability
        # photometric (:244-311): brightness draws `>= 0.50`, the others `<= 0.50`; factor = (max - min) * rand + min
        if self.do_random_brightness:
            do = torch.logical_and(do_random_transform, rand(n_batch) >= 0.50)
            lo, hi = self.random_brightness
            d['brightness'] = (do.to(torch.uint8), ((hi - lo) * rand(n_batch) + lo).float())
        if self.do_random_contrast:
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            lo, hi = self.random_contrast
            d['contrast'] = (do.to(torch.uint8), ((hi - lo) * rand(n_batch) + lo).float())
        if self.do_random_gamma:                                       # :279-290
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            lo, hi = self.random_gamma
            d['gamma'] = (do.to(torch.uint8), ((hi - lo) * rand(n_batch) + lo).float())
        if self.do_random_hue:                                         # :292-303
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            lo, hi = self.random_hue
            d['hue'] = (do.to(torch.uint8), ((hi - lo) * rand(n_batch) + lo).float())
        if self.do_random_saturation:
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            lo, hi = self.random_saturation
            d['saturation'] = (do.to(torch.uint8), ((hi - lo) * rand(n_batch) + lo).float())
        if self.do_random_noise:                                       # :322-332; the fields: add_noise's loops, tensor by tensor, sample by sample
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            fields = []
            for c in channels:
                f = torch.zeros(n_batch, int(c), n_height, n_width)
                for b in range(n_batch):
                    if do[b]:
                        f[b] = (torch.randn(int(c), n_height, n_width, generator=generator) if self.random_noise_type == 'gaussian'
                                else torch.rand(int(c), n_height, n_width, generator=generator))
                fields.append(f)
            d['noise'] = (do.to(torch.uint8), fields)
        do_crop = (self.do_random_crop_to_shape and bool(rand(1) <= 0.50)) or self.do_random_crop_to_shape_range
        if do_crop:
            if self.do_random_crop_to_shape_exact:
                ch, cw = self.random_crop_to_shape_height, self.random_crop_to_shape_width
            if self.do_random_crop_to_shape_range:
                ch = np.random.randint(low=self.random_crop_to_shape_height_min, high=self.random_crop_to_shape_height_max + 1)
                cw = np.random.randint(low=self.random_crop_to_shape_width_min, high=self.random_crop_to_shape_width_max + 1)
            start_y = torch.randint(low=0, high=n_height - ch + 1, size=(n_batch,), generator=generator)
            start_x = torch.randint(low=0, high=n_width - cw + 1, size=(n_batch,), generator=generator)
            d['crop'] = (int(ch), int(cw), start_y.to(torch.int32), start_x.to(torch.int32))
            n_height, n_width = int(ch), int(cw)                       # :385-386
        if self.do_random_horizontal_flip:
            d['hflip'] = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50).to(torch.uint8)
        if self.do_random_vertical_flip:
            d['vflip'] = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50).to(torch.uint8)
        if self.do_random_rotate:                                      # :409-428: the angles come from NUMPY's global state
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            values = np.random.rand(n_batch)
            angles = (self.random_rotate_max - (-self.random_rotate_max)) * values + (-self.random_rotate_max)
            d['rotate'] = (do.to(torch.uint8), torch.from_numpy(np.asarray(angles, dtype=np.float64)))
        if self.do_random_resize_and_crop:                             # :430-497
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            r_h = torch.randint(low=int(self.random_resize_and_crop_min * n_height), high=int(self.random_resize_and_crop_max * n_height),
                                size=(n_batch,), generator=generator)
            r_w = torch.randint(low=int(self.random_resize_and_crop_min * n_width), high=int(self.random_resize_and_crop_max * n_width),
                                size=(n_batch,), generator=generator)
            sy, sx = [], []
            for b in range(n_batch):                                   # y then x, sample by sample
                sy.append(torch.randint(low=0, high=int(r_h[b]) - n_height + 1, size=(1,), generator=generator))
                sx.append(torch.randint(low=0, high=int(r_w[b]) - n_width + 1, size=(1,), generator=generator))
            d['resize'] = (do.to(torch.uint8), r_h.to(torch.int32), r_w.to(torch.int32), torch.cat(sy).to(torch.int32), torch.cat(sx).to(torch.int32),
                           n_height, n_width)
        if self.do_random_crop_and_pad:                                # :508-571
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            max_h, min_h = int(self.random_crop_and_pad_max * n_height), int(self.random_crop_and_pad_min * n_height)
            max_w, min_w = int(self.random_crop_and_pad_max * n_width), int(self.random_crop_and_pad_min * n_width)
            r_h = torch.randint(low=min_h, high=max_h, size=(n_batch,), generator=generator)
            r_w = torch.randint(low=min_w, high=max_w, size=(n_batch,), generator=generator)
            sy = torch.cat([torch.randint(low=0, high=max_h - int(v), size=(1,), generator=generator) for v in r_h])
            sx = torch.cat([torch.randint(low=0, high=max_w - int(v), size=(1,), generator=generator) for v in r_w])
            ey = torch.minimum(sy + r_h, torch.full_like(sy, n_height))
            ex = torch.minimum(sx + r_w, torch.full_like(sx, n_width))
            d_h = (n_height - (ey - sy)).int()
            pad_top = (d_h * rand(n_batch)).int()
            d_w = (n_width - (ex - sx)).int()
            pad_left = (d_w * rand(n_batch)).int()
            d['crop_pad'] = (do.to(torch.uint8), sy.to(torch.int32), sx.to(torch.int32), ey.to(torch.int32), ex.to(torch.int32),
                             pad_top.to(torch.int32), (d_h - pad_top).to(torch.int32), pad_left.to(torch.int32), (d_w - pad_left).to(torch.int32))
        if self.do_random_resize_and_pad:                              # :573-620
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            r_h = torch.randint(low=int(self.random_resize_and_pad_min * n_height), high=int(self.random_resize_and_pad_max * n_height),
                                size=(n_batch,), generator=generator)
            r_w = torch.randint(low=int(self.random_resize_and_pad_min * n_width), high=int(self.random_resize_and_pad_max * n_width),
                                size=(n_batch,), generator=generator)
            d_h = (n_height - r_h).int()
            pad_top = (d_h * rand(n_batch)).int()
            pad_bottom = d_h - pad_top
            d_w = (n_width - r_w).int()
            pad_left = (d_w * rand(n_batch)).int()
            pad_right = d_w - pad_left
            z = torch.zeros_like(pad_top)
            d['resize_pad'] = (do.to(torch.uint8), r_h.to(torch.int32), r_w.to(torch.int32), torch.maximum(pad_top, z).to(torch.int32),
                               torch.maximum(pad_bottom, z).to(torch.int32), torch.maximum(pad_left, z).to(torch.int32), torch.maximum(pad_right, z).to(torch.int32))
        if self.do_random_remove_patch:                                # :630-655: patch sizes from PYTHON's random module
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            lo, hi = self.random_remove_patch_percent_range
            densities = (hi - lo) * rand(n_batch) + lo
            sizes = [[random.choice(self.random_remove_patch_size_height), random.choice(self.random_remove_patch_size_width)] for _ in range(n_batch)]
            d['remove'] = (do.to(torch.uint8), densities, sizes)
        return d

    # ---- data movement (device) ----------------------------------------------------------------------
    def apply(self, images_arr, draw, interpolation_modes=('nearest',), padding_modes=('constant',), generator=None):
        """Every N x C x H x W cuda tensor of images_arr through the decisions `draw`, in the reference's order: photometric jitter
        (three-channel tensors only make sense there), crop + flip, rotation, resize-and-crop.  interpolation_modes: one of
        'nearest' / 'bilinear' (or the reference's PIL enums 0 / 2) per tensor; rotation repeats the last one for further tensors (:1058-1060),
        resize-and-crop leaves further tensors untouched (:1252), both as the reference."""
        lib = _lib.load()
        dev = images_arr[0].device
        n, _, H, W = images_arr[0].shape
        stream = torch.cuda.current_stream().cuda_stream
        up = lambda t: None if t is None else t.to(dev, non_blocking=True)
        modes = list(interpolation_modes) + [list(interpolation_modes)[-1]] * (len(images_arr) - len(interpolation_modes))
        bil = [m in ('bilinear', 2) for m in modes]
        for m in modes:
            if m not in ('nearest', 'bilinear', 0, 2):
                raise NotImplementedError('interpolation mode %r (nearest and bilinear are built)' % (m,))
        out = []
        for t in images_arr:
            if not t.is_cuda:
                raise RuntimeError('proxytta.Transforms moves data on the GPU only (no CPU fallback)')
            assert t.shape[0] == n and tuple(t.shape[-2:]) == (H, W), 'all tensors of images_arr share N, H, W'
            out.append(t.float().contiguous())
        # ---- photometric (:236-311): uint8-valued images; every tensor of images_arr (the reference passes [image] only) ----
        pmodes = list(padding_modes) + [list(padding_modes)[-1]] * (len(images_arr) - len(padding_modes))
        for m in pmodes:
            if m not in self._PAD_MODES:
                raise NotImplementedError('padding mode %r (torchvision functional.pad: constant, edge, reflect, symmetric)' % (m,))
        if self.do_photometric_transforms or draw.get('gamma') is not None:
            bb, cc, gg, hh, ss = [draw.get(k) for k in ('brightness', 'contrast', 'gamma', 'hue', 'saturation')]
            if hh is not None and bool(((hh[1] < -0.5) | (hh[1] > 0.5)).any()):
                raise ValueError('hue_factor is not in [-0.5, 0.5].')                 # torchvision's adjust_hue
            args = []
            for pr in (bb, cc, gg, hh, ss):
                args += [None, None] if pr is None else [up(pr[0]), up(pr[1])]
            scratch = torch.empty(256 * n, device=dev, dtype=torch.float64)
            res = []
            for t in out:
                if t.shape[1] != 3:
                    raise ValueError('photometric transforms take N x 3 x H x W images')
                o = torch.empty_like(t)
                rc = lib.ptta_photometric_full(ptr(t), ptr(o), n, H, W, *[ptr(a) for a in args], ptr(scratch), stream)
                if rc != 0:
                    raise RuntimeError('ptta_photometric_full failed (%d)' % rc)
                res.append(o)
            out = res
        # ---- noise (:322-332, :839-876) ----
        if draw.get('noise') is not None:
            do, fields = draw['noise']
            do = up(do)
            res = []
            for t, f in zip(out, fields):
                assert tuple(f.shape) == tuple(t.shape), 'draw(channels=...) must list the channel count of every tensor'
                o = torch.empty_like(t)
                f_d = up(f.float().contiguous())
                rc = lib.ptta_add_noise(ptr(t), ptr(f_d), ptr(o), n, t.shape[1], H, W, ptr(do), float(self.random_noise_spread),
                                        int(self.random_noise_type == 'uniform'), stream)
                if rc != 0:
                    raise RuntimeError('ptta_add_noise failed (%d)' % rc)
                res.append(o)
            out = res
        # ---- crop + flip ----
        ch, cw, sy, sx = draw['crop'] if draw['crop'] is not None else (H, W, None, None)
        sy, sx, hf, vf = up(sy), up(sx), up(draw['hflip']), up(draw['vflip'])
        if not (sy is None and hf is None and vf is None):
            res = []
            for t in out:
                o = torch.empty((n, t.shape[1], ch, cw), device=dev, dtype=torch.float32)
                rc = lib.ptta_crop_flip(ptr(t), ptr(o), n, t.shape[1], H, W, ch, cw, ptr(sy), ptr(sx), ptr(hf), ptr(vf), stream)
                if rc != 0:
                    raise RuntimeError('ptta_crop_flip failed (%d)' % rc)
                res.append(o)
            out = res
        # ---- rotation (:409-428) ----
        if draw.get('rotate') is not None:
            do, ang = up(draw['rotate'][0]), up(draw['rotate'][1].float())
            res = []
            for t, b_ in zip(out, bil):
                o = torch.empty_like(t)
                rc = lib.ptta_rotate(ptr(t), ptr(o), n, t.shape[1], ch, cw, ptr(do), ptr(ang), int(b_), stream)
                if rc != 0:
                    raise RuntimeError('ptta_rotate failed (%d)' % rc)
                res.append(o)
            out = res
        # ---- resize and crop (:430-497) ----
        if draw.get('resize') is not None:
            do, rh, rw, ry, rx = [up(a) for a in draw['resize'][:5]]
            res = []
            for i, (t, b_) in enumerate(zip(out, bil)):
                if i >= len(interpolation_modes):
                    # the reference's resize_and_crop zips images_arr with the list AS GIVEN (src/transforms.py:1252; only rotate extends
                    # it, :1058-1060): tensors beyond it are left unresized there, so they are here (tta_main passes one mode per tensor)
                    res.append(t)
                    continue
                o = torch.empty_like(t)
                rc = lib.ptta_resize_crop(ptr(t), ptr(o), n, t.shape[1], ch, cw, ptr(do), ptr(rh), ptr(rw), ptr(ry), ptr(rx), int(b_),
                                          int(bool(self.resize_scaling_depth) and i != 0), stream)
                if rc != 0:
                    raise RuntimeError('ptta_resize_crop failed (%d)' % rc)
                res.append(o)
            out = res
        # ---- crop and pad (:508-571, :1072-1135) ----
        if draw.get('crop_pad') is not None:
            do, sy_, sx_, ey_, ex_, pt, pb, pl, pr_ = [up(a) for a in draw['crop_pad']]
            res = []
            for t, m in zip(out, pmodes):
                o = torch.empty_like(t)
                rc = lib.ptta_crop_pad(ptr(t), ptr(o), n, t.shape[1], ch, cw, ptr(do), ptr(sy_), ptr(sx_), ptr(ey_), ptr(ex_), ptr(pt), ptr(pl),
                                       self._PAD_MODES[m], 0.0, stream)
                if rc != 0:
                    raise RuntimeError('ptta_crop_pad failed (%d)' % rc)
                res.append(o)
            out = res
        # ---- resize and pad (:573-620, :1137-1220) ----
        if draw.get('resize_pad') is not None:
            do, rh, rw, pt, pb, pl, pr_ = [up(a) for a in draw['resize_pad']]
            res = []
            for t, b_, m in zip(out, bil, pmodes):
                o = torch.empty_like(t)
                rc = lib.ptta_resize_pad(ptr(t), ptr(o), n, t.shape[1], ch, cw, ptr(do), ptr(rh), ptr(rw), ptr(pt), ptr(pl), int(b_),
                                         self._PAD_MODES[m], 0.0, stream)
                if rc != 0:
                    raise RuntimeError('ptta_resize_pad failed (%d)' % rc)
                res.append(o)
            out = res
        # ---- patch removal (:630-655, :878-953): the selection is random_nonzero's -- nonzero pixels in row-major order, a randperm over them
        # (drawn here, last, per tensor and sample as the reference does), the first int(density * count) of it ----
        if draw.get('remove') is not None:
            do, densities, sizes = draw['remove']
            if any(bool(do[b]) and (int(k[0]) % 2 == 0 or int(k[1]) % 2 == 0) for b, k in enumerate(sizes)):
                # max_pool2d(kernel k, stride 1, padding k // 2) returns H + 1 rows for an even k: the reference's `mask * image` then fails to broadcast
                raise ValueError('remove_random_patches: patch sizes must be odd (got %r)' % (sizes,))
            # (device copies held in locals until the launches are enqueued: a temporary's block is handed to the next allocation at once)
            ph = up(torch.tensor([int(k[0]) for k in sizes], dtype=torch.int32))
            pw = up(torch.tensor([int(k[1]) for k in sizes], dtype=torch.int32))
            do_d = up(do)
            res = []
            for t in out:
                sel = torch.zeros((n, ch, cw), device=dev, dtype=torch.uint8)
                for b in range(n):
                    if not bool(do[b]):
                        continue
                    nz = (t[b].abs().sum(dim=0) > 0).nonzero(as_tuple=True)           # (device reduction + index list: host logic of the draw)
                    count = int(nz[0].shape[0])
                    # int(density * n) with a float32 0-dim density, as src/transforms.py:946-947: the product is rounded to fp32 before truncation
                    perm = torch.randperm(count, generator=generator)[0:int(densities[b].float() * count)].to(dev)
                    sel[b, nz[0][perm], nz[1][perm]] = 1
                o = torch.empty_like(t)
                rc = lib.ptta_remove_patches(ptr(t), ptr(o), n, t.shape[1], ch, cw, ptr(do_d), ptr(sel), ptr(ph), ptr(pw), stream)
                if rc != 0:
                    raise RuntimeError('ptta_remove_patches failed (%d)' % rc)
                res.append(o)
            out = res
        return out

    _PAD_MODES = {'constant': 0, 'edge': 1, 'reflect': 2, 'symmetric': 3}

    def transform(self, images_arr, intrinsics_arr=[], padding_modes=['constant'], interpolation_modes=['nearest'],
                  random_transform_probability=0.00, generator=None):
        if images_arr[0].ndim != 4:
            raise ValueError('Unsupported number of dimensions: {}'.format(images_arr[0].ndim))
        n, _, H, W = images_arr[0].shape
        d = self.draw(n, H, W, random_transform_probability, generator, channels=[t.shape[1] for t in images_arr])
        self.last_draw = d
        images_arr = self.apply(list(images_arr), d, interpolation_modes, padding_modes, generator)
        intrinsics_arr = list(intrinsics_arr)
        if d['crop'] is not None:
            # the reference subtracts (n_width - crop_width, n_height - crop_height) from every sample's optical centre,
            # whatever the start offsets were (:380-383)
            intrinsics_arr = self.adjust_intrinsics(intrinsics_arr, x_offsets=float(W - d['crop'][1]), y_offsets=float(H - d['crop'][0]))
        if d['resize'] is not None:
            # :447-451 scales, :493-497 offsets -- for EVERY sample, whether or not its do_resize_and_crop coin came up (as the reference)
            do, rh, rw, sy, sx, nh, nw = d['resize']
            intrinsics_arr = self.adjust_intrinsics(intrinsics_arr, x_scales=rw.float() / nw, y_scales=rh.float() / nh)
            intrinsics_arr = self.adjust_intrinsics(intrinsics_arr, x_offsets=(rw - nw).float(), y_offsets=(rh - nh).float())
        outputs = []
        if len(images_arr) > 0:
            outputs.append(images_arr)
        if len(intrinsics_arr) > 0:
            outputs.append(list(intrinsics_arr))
        return outputs[0] if len(outputs) == 1 else outputs

    def adjust_intrinsics(self, intrinsics_arr, x_scales=1.0, y_scales=1.0, x_offsets=0.0, y_offsets=0.0):
        """src/transforms.py:1330-1378: fx, cx scaled by x_scales, cx -= x_offsets (same for y); scalars or one value per sample."""
        out = []
        for K in intrinsics_arr:
            K = K.clone()
            cast = lambda v: v.to(K.device, K.dtype) if torch.is_tensor(v) else v
            x_scales, y_scales, x_offsets, y_offsets = cast(x_scales), cast(y_scales), cast(x_offsets), cast(y_offsets)
            K[:, 0, 0] = K[:, 0, 0] * x_scales
            K[:, 0, 2] = K[:, 0, 2] * x_scales - x_offsets
            K[:, 1, 1] = K[:, 1, 1] * y_scales
            K[:, 1, 2] = K[:, 1, 2] * y_scales - y_offsets
            out.append(K)
        return out

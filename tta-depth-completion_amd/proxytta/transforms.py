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
Gamma, hue, noise, crop-and-pad, resize-and-pad and patch removal (enabled by no adapt script) raise NotImplementedError; image
normalisation is fused into the engine's first convolution (`Engine.set_image_norm`) and is refused here.
"""
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
        unsupported = {
            'normalized_image_range': normalized_image_range is not None,
            'random_gamma': not _unset(random_gamma), 'random_hue': not _unset(random_hue),
            'random_noise': random_noise_type != 'none' and random_noise_spread > -1,
            'random_remove_patch_percent_range': not _unset(random_remove_patch_percent_range),
            'random_crop_and_pad': not _unset(random_crop_and_pad),
            'random_resize_and_pad': not _unset(random_resize_and_pad)}
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(
                'proxytta.Transforms builds what the adapt scripts enable (crop, flip, rotate, resize_and_crop, brightness, contrast, '
                'saturation); not built: %s.  Image normalisation is fused into the engine: Engine.set_image_norm(normalized_image_range)'
                % ', '.join(bad))
        # src/transforms.py:84-96
        self.do_random_brightness = not _unset(random_brightness); self.random_brightness = random_brightness
        self.do_random_contrast = not _unset(random_contrast); self.random_contrast = random_contrast
        self.do_random_saturation = not _unset(random_saturation); self.random_saturation = random_saturation
        self.do_photometric_transforms = self.do_random_brightness or self.do_random_contrast or self.do_random_saturation
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
    def draw(self, n_batch, n_height, n_width, random_transform_probability=0.0, generator=None):
        """The random decisions of one `transform` call, drawn in the reference's order (:230, :244-311, :337-350, :391-497) from
        torch's CPU generator (`generator=None`: the global one `torch.manual_seed` seeds) and numpy's global state for
        the range crop (:346-352)."""
        def rand(k):
            return torch.rand(k, generator=generator)
        d = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': None, 'resize': None, 'brightness': None, 'contrast': None, 'saturation': None}
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
        if self.do_random_saturation:
            do = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50)
            lo, hi = self.random_saturation
            d['saturation'] = (do.to(torch.uint8), ((hi - lo) * rand(n_batch) + lo).float())
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
        return d

    # ---- data movement (device) ----------------------------------------------------------------------
    def apply(self, images_arr, draw, interpolation_modes=('nearest',)):
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
        if self.do_photometric_transforms:
            bb, cc, ss = [draw[k] for k in ('brightness', 'contrast', 'saturation')]
            args = []
            for pr in (bb, cc, ss):
                args += [None, None] if pr is None else [up(pr[0]), up(pr[1])]
            scratch = torch.empty(256 * n, device=dev, dtype=torch.float64)
            res = []
            for t in out:
                if t.shape[1] != 3:
                    raise ValueError('photometric transforms take N x 3 x H x W images')
                o = torch.empty_like(t)
                rc = lib.ptta_photometric(ptr(t), ptr(o), n, H, W, *[ptr(a) for a in args], ptr(scratch), stream)
                if rc != 0:
                    raise RuntimeError('ptta_photometric failed (%d)' % rc)
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
        return out

    def transform(self, images_arr, intrinsics_arr=[], padding_modes=['constant'], interpolation_modes=['nearest'],
                  random_transform_probability=0.00, generator=None):
        if images_arr[0].ndim != 4:
            raise ValueError('Unsupported number of dimensions: {}'.format(images_arr[0].ndim))
        n, _, H, W = images_arr[0].shape
        d = self.draw(n, H, W, random_transform_probability, generator)
        self.last_draw = d
        images_arr = self.apply(list(images_arr), d, interpolation_modes)
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

"""Host-side mirror of the geometric part of the reference's `Transforms` (src/transforms.py:7-191 constructor,
:192-664 `transform`): random crop to shape, horizontal and vertical flip, intrinsics adjustment.  The random DRAWS are
made here with torch's generator in the reference's order (so a CPU run of the reference with the same
`torch.manual_seed` / `np.random.seed` takes the same decisions); the data movement is one `ptta_crop_flip` launch per
tensor (csrc/augment.hip).  Photometric augmentation, rotation, resize / pad and patch removal are stage-1/2 features
outside the hot path and raise NotImplementedError; image normalisation is fused into the engine's
first convolution (`Engine.set_image_norm`) and is refused here for the same reason.
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
            'random_brightness': not _unset(random_brightness), 'random_contrast': not _unset(random_contrast),
            'random_gamma': not _unset(random_gamma), 'random_hue': not _unset(random_hue),
            'random_saturation': not _unset(random_saturation),
            'random_noise': random_noise_type != 'none' and random_noise_spread > -1,
            'random_remove_patch_percent_range': not _unset(random_remove_patch_percent_range),
            'random_rotate_max': random_rotate_max > 0, 'random_crop_and_pad': not _unset(random_crop_and_pad),
            'random_resize_and_crop': not _unset(random_resize_and_crop),
            'random_resize_and_pad': not _unset(random_resize_and_pad)}
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(
                'proxytta.Transforms builds the crop / flip path only (src/transforms.py:337-407); not built: %s.  '
                'Image normalisation is fused into the engine: Engine.set_image_norm(normalized_image_range)' % ', '.join(bad))
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
        """The random decisions of one `transform` call, drawn in the reference's order (:230, :337-350, :391-403) from
        torch's CPU generator (`generator=None`: the global one `torch.manual_seed` seeds) and numpy's global state for
        the range crop (:346-352)."""
        def rand(k):
            return torch.rand(k, generator=generator)
        d = {'crop': None, 'hflip': None, 'vflip': None}
        do_random_transform = rand(n_batch) <= random_transform_probability
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
        if self.do_random_horizontal_flip:
            d['hflip'] = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50).to(torch.uint8)
        if self.do_random_vertical_flip:
            d['vflip'] = torch.logical_and(do_random_transform, rand(n_batch) <= 0.50).to(torch.uint8)
        return d

    # ---- data movement (device) ----------------------------------------------------------------------
    def apply(self, images_arr, draw):
        """Crop + flip every N x C x H x W cuda tensor of images_arr with the decisions `draw`."""
        lib = _lib.load()
        dev = images_arr[0].device
        n, _, H, W = images_arr[0].shape
        ch, cw, sy, sx = draw['crop'] if draw['crop'] is not None else (H, W, None, None)
        up = lambda t: None if t is None else t.to(dev, non_blocking=True)
        sy, sx, hf, vf = up(sy), up(sx), up(draw['hflip']), up(draw['vflip'])
        if sy is None and hf is None and vf is None:
            return [t.float() for t in images_arr]
        stream = torch.cuda.current_stream().cuda_stream
        out = []
        for t in images_arr:
            if not t.is_cuda:
                raise RuntimeError('proxytta.Transforms moves data on the GPU only (no CPU fallback)')
            t = t.float().contiguous()
            assert t.shape[0] == n and tuple(t.shape[-2:]) == (H, W), 'all tensors of images_arr share N, H, W'
            o = torch.empty((n, t.shape[1], ch, cw), device=dev, dtype=torch.float32)
            rc = lib.ptta_crop_flip(ptr(t), ptr(o), n, t.shape[1], H, W, ch, cw, ptr(sy), ptr(sx), ptr(hf), ptr(vf), stream)
            if rc != 0:
                raise RuntimeError('ptta_crop_flip failed (%d)' % rc)
            out.append(o)
        return out

    def transform(self, images_arr, intrinsics_arr=[], padding_modes=['constant'], interpolation_modes=['nearest'],
                  random_transform_probability=0.00, generator=None):
        if images_arr[0].ndim != 4:
            raise ValueError('Unsupported number of dimensions: {}'.format(images_arr[0].ndim))
        n, _, H, W = images_arr[0].shape
        d = self.draw(n, H, W, random_transform_probability, generator)
        self.last_draw = d
        images_arr = self.apply(list(images_arr), d)
        if d['crop'] is not None:
            # the reference subtracts (n_width - crop_width, n_height - crop_height) from every sample's optical centre,
            # whatever the start offsets were (:380-383)
            intrinsics_arr = self.adjust_intrinsics(list(intrinsics_arr), x_offsets=float(W - d['crop'][1]),
                                                    y_offsets=float(H - d['crop'][0]))
        outputs = []
        if len(images_arr) > 0:
            outputs.append(images_arr)
        if len(intrinsics_arr) > 0:
            outputs.append(list(intrinsics_arr))
        return outputs[0] if len(outputs) == 1 else outputs

    def adjust_intrinsics(self, intrinsics_arr, x_scales=1.0, y_scales=1.0, x_offsets=0.0, y_offsets=0.0):
        """src/transforms.py:1330-1378 for scalar factors: fx, cx scaled by x_scales, cx -= x_offsets (same for y)."""
        out = []
        for K in intrinsics_arr:
            K = K.clone()
            K[:, 0, 0] = K[:, 0, 0] * x_scales
            K[:, 0, 2] = K[:, 0, 2] * x_scales - x_offsets
            K[:, 1, 1] = K[:, 1, 1] * y_scales
            K[:, 1, 2] = K[:, 1, 2] * y_scales - y_offsets
            out.append(K)
        return out

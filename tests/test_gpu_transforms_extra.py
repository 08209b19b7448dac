"""GPU: the six augmentations of Transforms that no adapt script enables (gamma, hue, noise, crop-and-pad, resize-and-pad, patch removal), HIP
kernels behind proxytta.Transforms against the outputs of the REAL reference class (tests/golden/transforms_extra.npz) and the oracle.
Noise, patch removal and crop-and-pad are bit-exact (copies, one rounded multiply-add); gamma / hue go through powf and a float -> uint8
truncation (one level on a few pixels); bilinear resizing is held to the tolerance of the resize-and-crop test."""
import random

import numpy as np
import pytest
import torch

from oracle import transforms_oracle as TO
from proxytta.transforms import Transforms
from tests.test_oracle_transforms_extra import cases, load_case, seed_all

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', list(cases()))
def test_transform_reproduces_the_reference(name):
    cfg, arrs, want = load_case(name)
    t = Transforms(**cfg['kw'])
    seed_all(cfg['seed'])
    got = t.transform([a.cuda() for a in arrs], padding_modes=cfg['pmodes'], interpolation_modes=cfg['imodes'],
                      random_transform_probability=cfg['prob'])
    kw = cfg['kw']
    photometric = any(k in kw for k in ('random_brightness', 'random_contrast', 'random_gamma', 'random_hue', 'random_saturation'))
    resized = 'random_resize_and_pad' in kw
    for i, (g, w) in enumerate(zip(got, want)):
        g = g.cpu().numpy()
        assert g.shape == w.shape
        if photometric and ('random_noise_type' in kw or resized):
            # a level of difference in the uint8 stage, carried through noise / resizing
            assert np.abs(g - w).max() <= 1.01 and (np.abs(g - w) > 1e-3).mean() < 5e-3
        elif photometric:
            assert np.abs(g - w).max() <= 1.0 and (g != w).mean() < 3e-3
        elif resized and cfg['imodes'][min(i, len(cfg['imodes']) - 1)] == 'bilinear':
            assert np.abs(g - w).max() < 0.05 and np.abs(g - w).mean() < 1e-3
        elif resized:
            assert (g != w).mean() < 1e-3
        else:
            np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize('shape', [(2, 37, 53), (2, 352, 1216)])
def test_gamma_and_hue_match_the_torchvision_algorithm(shape):
    n, H, W = shape
    rng = np.random.default_rng(21)
    x = rng.random((n, 3, H, W), dtype=np.float32) * 255.0
    on = torch.ones(n, dtype=torch.uint8)
    fg, fh = torch.tensor([0.61, 1.57][:n]), torch.tensor([0.37, -0.21][:n])
    t = Transforms(random_gamma=[0.5, 1.6], random_hue=[-0.4, 0.4])
    base = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': None, 'resize': None, 'brightness': None, 'contrast': None, 'saturation': None}
    for gg, hh in (((on, fg), (on, fh)), ((on, fg), (1 - on, fh)), ((1 - on, fg), (on, fh))):
        d = dict(base, gamma=gg, hue=hh)
        [y] = t.apply([torch.from_numpy(x).cuda()], d)
        ref = TO.photometric_full(torch.from_numpy(x), gamma=gg, hue=hh).numpy()
        got = y.cpu().numpy()
        assert np.abs(got - ref).max() <= 1.0 and (got != ref).mean() < 3e-3
        assert np.array_equal(got, np.floor(got)) and got.min() >= 0 and got.max() <= 255
    # gamma as the only option: float images, torchvision's float branch (x ** gamma clamped to [0, 1])
    tg = Transforms(random_gamma=[0.5, 1.6])
    x01 = rng.random((n, 3, H, W), dtype=np.float32) * 1.2
    [y] = tg.apply([torch.from_numpy(x01).cuda()], dict(base, gamma=(on, fg)))
    ref = TO.photometric_full(torch.from_numpy(x01), gamma=(on, fg)).numpy()
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-6, atol=1e-7)
    with pytest.raises(ValueError):
        t.apply([torch.from_numpy(x).cuda()], dict(base, hue=(on, torch.tensor([0.7, 0.0][:n]))))


@pytest.mark.parametrize('mode', ['constant', 'edge', 'reflect', 'symmetric'])
def test_crop_pad_and_resize_pad_full_size(mode):
    n, H, W = 2, 352, 1216
    rng = np.random.default_rng(22)
    x = np.floor(rng.random((n, 3, H, W), dtype=np.float32) * 255.0)
    sd = (rng.random((n, 1, H, W), dtype=np.float32) * 80 * (rng.random((n, 1, H, W)) < 0.05)).astype(np.float32)
    t = Transforms(random_crop_and_pad=[0.6, 0.9], random_resize_and_pad=[0.6, 0.9])
    base = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': None, 'resize': None}
    i32 = lambda *v: torch.tensor(v, dtype=torch.int32)
    do = torch.tensor([1, 0], dtype=torch.uint8)
    sy, sx, ey, ex = i32(17, 0), i32(100, 0), i32(300, 10), i32(1100, 10)
    pt, pl = i32(40, 0), i32(7, 0)
    pb, pr = i32(H - 283 - 40, 0), i32(W - 1000 - 7, 0)
    d = dict(base, crop_pad=(do, sy, sx, ey, ex, pt, pb, pl, pr))
    im, dep = t.apply([torch.from_numpy(x).cuda(), torch.from_numpy(sd).cuda()], d, ['bilinear', 'nearest'], [mode])
    np.testing.assert_array_equal(im.cpu().numpy(), TO.crop_and_pad(torch.from_numpy(x), do, sy, sx, ey, ex, pt, pb, pl, pr, mode).numpy())
    np.testing.assert_array_equal(dep.cpu().numpy(), TO.crop_and_pad(torch.from_numpy(sd), do, sy, sx, ey, ex, pt, pb, pl, pr, mode).numpy())
    rh, rw = i32(250, H), i32(900, W)
    pt, pl = i32(60, 0), i32(200, 0)
    pb, pr = i32(H - 250 - 60, 0), i32(W - 900 - 200, 0)
    d = dict(base, resize_pad=(do, rh, rw, pt, pb, pl, pr))
    im, dep = t.apply([torch.from_numpy(x).cuda(), torch.from_numpy(sd).cuda()], d, ['bilinear', 'nearest'], [mode])
    ref_im = TO.resize_and_pad(torch.from_numpy(x), do, rh, rw, pt, pb, pl, pr, True, mode).numpy()
    ref_dep = TO.resize_and_pad(torch.from_numpy(sd), do, rh, rw, pt, pb, pl, pr, False, mode).numpy()
    assert np.abs(im.cpu().numpy() - ref_im).max() < 0.05 and np.abs(im.cpu().numpy() - ref_im).mean() < 1e-3
    assert (dep.cpu().numpy() != ref_dep).mean() < 1e-3
    np.testing.assert_array_equal(im.cpu().numpy()[1], x[1])               # the sample whose coin said no is copied


def test_noise_and_patch_removal_full_size():
    n, H, W = 2, 352, 1216
    rng = np.random.default_rng(23)
    x = np.floor(rng.random((n, 3, H, W), dtype=np.float32) * 255.0)
    sd = (rng.random((n, 1, H, W), dtype=np.float32) * 80 * (rng.random((n, 1, H, W)) < 0.05)).astype(np.float32)
    for kind in ('gaussian', 'uniform'):
        t = Transforms(random_noise_type=kind, random_noise_spread=2.5)
        seed_all(5)
        d = t.draw(n, H, W, 1.0, channels=[3, 1])
        got = t.apply([torch.from_numpy(x).cuda(), torch.from_numpy(sd).cuda()], d)
        ref = TO.apply_draw(t, d, [torch.from_numpy(x), torch.from_numpy(sd)])
        for g, r in zip(got, ref):
            np.testing.assert_array_equal(g.cpu().numpy(), r.numpy())
    t = Transforms(random_remove_patch_percent_range=[0.2, 0.4], random_remove_patch_size=[5, 7, 9, 11])
    seed_all(6)
    d = t.draw(n, H, W, 1.0, channels=[1])
    d['remove'] = (torch.ones(n, dtype=torch.uint8), d['remove'][1], d['remove'][2])
    g1 = torch.Generator().manual_seed(9)
    [got] = t.apply([torch.from_numpy(sd).cuda()], d, generator=g1)
    g2 = torch.Generator().manual_seed(9)
    [ref] = TO.apply_draw(t, d, [torch.from_numpy(sd)], generator=g2)
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())
    assert 0 < float((got.cpu() != 0).sum()) < float((torch.from_numpy(sd) != 0).sum())       # points were removed, not all of them


def test_constructor_refuses_only_what_the_engine_fuses():
    with pytest.raises(NotImplementedError):
        Transforms(normalized_image_range=[0, 1])
    with pytest.raises(ValueError):
        Transforms(random_noise_type='salt', random_noise_spread=1.0)
    te = Transforms(random_remove_patch_percent_range=[0.1, 0.2], random_remove_patch_size=[4, 3])
    with pytest.raises(ValueError):                      # an even patch size fails in the reference too (max_pool2d returns H + 1 rows)
        d_ = te.draw(1, 8, 8, 1.0, channels=[1])
        d_['remove'] = (torch.ones(1, dtype=torch.uint8), d_['remove'][1], d_['remove'][2])
        te.apply([torch.ones(1, 1, 8, 8).cuda()], d_)
    t = Transforms(random_crop_and_pad=[0.5, 1.0])
    with pytest.raises(NotImplementedError):
        t.apply([torch.zeros(1, 1, 8, 8).cuda()], t.draw(1, 8, 8, 1.0, channels=[1]), padding_modes=['circular'])

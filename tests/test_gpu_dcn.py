"""Modulated deformable convolution (the reference's native `DCN` extension) on libptta_hip.
Re-expresses the checks of external_src/NLSPN/src/model/deformconv/test.py: zero offset + unit mask == nn.Conv2d
(:69-110), identity weights (:142-181), im2col_step invariance (:219-260) and gradients (:405-435), the latter
against autograd of the oracle's differentiable restatement."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import proxytta_oracle as O
from tests.util import rel_mae

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize('cfg', [dict(B=2, C=4, Co=6, H=9, W=11, k=3, s=1, p=1, d=1, group=2, dg=2),
                                 dict(B=1, C=1, Co=1, H=16, W=20, k=3, s=1, p=1, d=1, group=1, dg=1),     # NLSPN propagation
                                 dict(B=1, C=1, Co=1, H=12, W=10, k=1, s=1, p=0, d=1, group=1, dg=1),     # NLSPN confidence gather
                                 dict(B=2, C=4, Co=4, H=10, W=12, k=3, s=2, p=1, d=2, group=1, dg=1)])
def test_zero_offset_is_plain_conv(cfg):
    from proxytta import dcn
    B, C, Co, H, W, k = cfg['B'], cfg['C'], cfg['Co'], cfg['H'], cfg['W'], cfg['k']
    x, w, b = _rand(B, C, H, W, seed=1), _rand(Co, C // cfg['group'], k, k, seed=2, scale=0.3), _rand(Co, seed=3)
    ref = F.conv2d(x, w, b, stride=cfg['s'], padding=cfg['p'], dilation=cfg['d'], groups=cfg['group'])
    Ho, Wo = ref.shape[-2:]
    off = torch.zeros(B, 2 * k * k * cfg['dg'], Ho, Wo)
    msk = torch.ones(B, k * k * cfg['dg'], Ho, Wo)
    for step in (1, 64):                                   # im2col_step has no effect
        out = dcn.modulated_deform_conv_forward(x.cuda(), w.cuda(), b.cuda(), off.cuda(), msk.cuda(), k, k, cfg['s'], cfg['s'],
                                                cfg['p'], cfg['p'], cfg['d'], cfg['d'], cfg['group'], cfg['dg'], step).cpu()
        assert rel_mae(out, ref) < 1e-5


@pytest.mark.parametrize('cfg', [dict(B=2, C=4, Co=6, H=9, W=11, k=3, s=1, p=1, d=1, group=2, dg=2),
                                 dict(B=1, C=1, Co=1, H=24, W=32, k=3, s=1, p=1, d=1, group=1, dg=1),
                                 dict(B=1, C=1, Co=1, H=12, W=10, k=1, s=1, p=0, d=1, group=1, dg=1)])
def test_forward_and_gradients_against_oracle(cfg):
    from proxytta import dcn
    B, C, Co, H, W, k = cfg['B'], cfg['C'], cfg['Co'], cfg['H'], cfg['W'], cfg['k']
    K = k * k
    x = _rand(B, C, H, W, seed=11).requires_grad_(True)
    w = _rand(Co, C // cfg['group'], k, k, seed=12, scale=0.4).requires_grad_(True)
    b = _rand(Co, seed=13).requires_grad_(True)
    Ho = (H + 2 * cfg['p'] - (cfg['d'] * (k - 1) + 1)) // cfg['s'] + 1
    Wo = (W + 2 * cfg['p'] - (cfg['d'] * (k - 1) + 1)) // cfg['s'] + 1
    off = (_rand(B, 2 * K * cfg['dg'], Ho, Wo, seed=14) * 1.7).requires_grad_(True)      # samples leave the image at the borders
    msk = torch.sigmoid(_rand(B, K * cfg['dg'], Ho, Wo, seed=15)).requires_grad_(True)
    ref = O.mdconv_forward(x, w, b, off, msk, cfg['s'], cfg['p'], cfg['d'], cfg['group'], cfg['dg'])
    gy = _rand(*ref.shape, seed=16)
    gx, gw, gb, goff, gm = torch.autograd.grad(ref, (x, w, b, off, msk), gy)
    out = dcn.modulated_deform_conv_forward(x.detach().cuda(), w.detach().cuda(), b.detach().cuda(), off.detach().cuda(),
                                            msk.detach().cuda(), k, k, cfg['s'], cfg['s'], cfg['p'], cfg['p'], cfg['d'], cfg['d'],
                                            cfg['group'], cfg['dg'], 64)
    assert rel_mae(out, ref.detach()) < 1e-5
    got = dcn.modulated_deform_conv_backward(x.detach().cuda(), w.detach().cuda(), b.detach().cuda(), off.detach().cuda(),
                                             msk.detach().cuda(), gy.cuda(), k, k, cfg['s'], cfg['s'], cfg['p'], cfg['p'],
                                             cfg['d'], cfg['d'], cfg['group'], cfg['dg'], 64)
    for name, a, r in zip(('input', 'offset', 'mask', 'weight', 'bias'), got, (gx, goff, gm, gw, gb)):
        assert rel_mae(a, r) < 2e-5, name


def test_autograd_function_and_nlspn_propagation_shape():
    """One NLSPN propagation step (nlspnmodel_adapt.py:239-253,332-338): C=1, 3x3, weight=ones, bias=0."""
    from proxytta.dcn import ModulatedDeformConvFunction
    B, H, W = 1, 352, 1216
    feat = _rand(B, 1, H, W, seed=21).abs().cuda().requires_grad_(True)
    offset = (_rand(B, 18, H, W, seed=22) * 2.0).cuda().requires_grad_(True)
    aff = torch.softmax(_rand(B, 9, H, W, seed=23), 1).cuda().requires_grad_(True)
    w = torch.ones(1, 1, 3, 3).cuda()
    b = torch.zeros(1).cuda()
    out = ModulatedDeformConvFunction.apply(feat, offset, aff, w, b, 1, 1, 1, 1, 1, 64)
    assert out.shape == feat.shape and torch.isfinite(out).all()
    out.sum().backward()
    assert torch.isfinite(feat.grad).all() and torch.isfinite(offset.grad).all() and torch.isfinite(aff.grad).all()
    # d(sum out)/d aff[k] = the sampled feature; sum over taps of aff * sample == out
    assert rel_mae((aff.detach() * aff.grad).sum(1, keepdim=True), out.detach()) < 1e-5

"""GPU parity tests: the HIP path (through the C-ABI) against the golden vectors produced by the
real reference and against the oracle on the same seeded inputs.  Tolerances are the ones of
BASELINE.json's north_star: float32, relative MAE <= 1e-3 on depth maps (we hold fp32 mode to
1e-4 or better); the mixed-precision mode has its own file, tests/test_gpu_mixed.py."""
import os

import numpy as np
import pytest
import torch

from proxytta import synth
from tests.util import ONE, TWO, golden_hp, make_engine, rel_mae

pytestmark = pytest.mark.gpu

CASES = ['msgchn_1layer_32x48', 'msgchn_1layer_64x96', 'msgchn_1layer_36x52_pad', 'msgchn_1layer_32x48_n2',
         'msgchn_1layer_32x48_wcos1']


def _torch_conv(x_nhwc, w, b, mode, relu):
    import torch.nn.functional as F
    x = x_nhwc.permute(0, 3, 1, 2)
    if relu:
        x = F.relu(x)
    if mode == 0:
        y = F.conv2d(x, w, b, padding=1)
    elif mode == 1:
        y = F.conv2d(x, w, b, stride=2, padding=1)
    else:
        y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('naive', [True, False])
@pytest.mark.parametrize('mode', [0, 1, 2])
@pytest.mark.parametrize('relu', [False, True])
def test_op_conv32_forward(mode, relu, naive):
    from proxytta.engine import op_conv32
    g = torch.Generator().manual_seed(mode * 7 + relu)
    x = torch.randn(2, 12, 44, 32, generator=g)            # W=44: one full and one ragged 32-pixel tile
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.1        # asymmetric in (out, in, ky, kx)
    b = torch.randn(32, generator=g)
    ref = _torch_conv(x, w, b, mode, relu)
    got = op_conv32(x.cuda(), w.cuda(), b.cuda(), mode, relu_in=relu, in_major=(mode == 2), naive=naive).cpu()
    assert got.shape == ref.shape
    assert rel_mae(got, ref) < 2e-6


@pytest.mark.parametrize('naive', [True, False])
@pytest.mark.parametrize('mode', [0, 1, 2])
def test_op_conv32_input_gradient(mode, naive):
    """The backward re-packings (in_major / flip) reproduce autograd's conv backward-data."""
    from proxytta.engine import op_conv32
    g = torch.Generator().manual_seed(100 + mode)
    x = torch.randn(1, 8, 36, 32, generator=g, requires_grad=True)
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
    y = _torch_conv(x, w, None, mode, False)
    gy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x, gy)
    # S1 -> S1 (transposed+flipped); S2 -> T2 (in-major); T2 -> S2 (as stored)
    bmode, in_major, flip = {0: (0, True, True), 1: (2, True, False), 2: (1, False, False)}[mode]
    got = op_conv32(gy.cuda(), w.cuda(), None, bmode, in_major=in_major, flip=flip, naive=naive).cpu()
    assert rel_mae(got, gx) < 2e-6


@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('shape', [(2, 12, 44), (1, 8, 32), (1, 19, 70)])
def test_op_conv32_bf16x3(shape, relu):
    """fp32 storage, three bf16 MFMAs per product (LDS-staged stride-1 kernel): fp32-faithful to ~1e-5."""
    from proxytta.engine import op_conv32
    b, h, w = shape
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(b, h, w, 32, generator=g) * 3.0
    wt = torch.randn(32, 32, 3, 3, generator=g) * 0.1
    bias = torch.randn(32, generator=g)
    ref = _torch_conv(x, wt, bias, 0, relu)
    got = op_conv32(x.cuda(), wt.cuda(), bias.cuda(), 0, relu_in=relu, x3=True).cpu()
    assert rel_mae(got, ref) < 2e-5
    # and the backward re-packing through the same kernel
    xr = x.clone().requires_grad_(True)
    y = _torch_conv(xr, wt, None, 0, False)
    gy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad(y, xr, gy)
    got = op_conv32(gy.cuda(), wt.cuda(), None, 0, in_major=True, flip=True, x3=True).cpu()
    assert rel_mae(got, gx) < 2e-5
    # stride-2 and transposed geometries (direct-load bf16x3 kernel)
    if h % 2 == 0 and w % 2 == 0:
        for mode in (1, 2):
            ref = _torch_conv(x, wt, bias, mode, relu)
            got = op_conv32(x.cuda(), wt.cuda(), bias.cuda(), mode, relu_in=relu, in_major=(mode == 2), x3=True).cpu()
            assert rel_mae(got, ref) < 2e-5, mode


@pytest.mark.parametrize('shape', [(2, 22, 76), (1, 19, 70), (2, 88, 304)])
@pytest.mark.parametrize('relu', [False, True])
def test_layer_loop_trial_equals_the_chain_of_launches(shape, relu):
    """The layer-loop trial (conv32_s1_small_loop_kernel: dependent stride-1 layers in ONE launch, a device-wide barrier between layers)
    against the same chain as separate launches: bit for bit, incl. a ragged map and one of more tiles than blocks; the first layer of
    the chain against torch.  A barrier spin that runs out reports -62 instead of hanging."""
    import ctypes
    from proxytta import _lib
    from proxytta._lib import ptr
    lib = _lib.load()
    b, h, w = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(b, h, w, 32, generator=g).cuda()
    wt = (torch.randn(32, 32, 3, 3, generator=g) * 0.05).cuda()
    bias = torch.randn(32, generator=g).cuda()
    us = ctypes.c_float(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for reps in (1, 2, 7):
        a1, c1 = torch.zeros_like(x), torch.zeros_like(x)
        assert lib.ptta_op_conv32_chain(ptr(x), ptr(wt), ptr(bias), ptr(a1), ptr(c1), None, b, h, w, int(relu), 8, reps, 1, ctypes.byref(us), st) == 0
        a2, c2 = torch.zeros_like(x), torch.zeros_like(x)
        assert lib.ptta_op_conv32_chain(ptr(x), ptr(wt), ptr(bias), ptr(a2), ptr(c2), None, b, h, w, int(relu), 16, reps, 1, ctypes.byref(us), st) == 0
        assert torch.equal(a1, a2) and torch.equal(c1, c2), reps
        assert float(a1.abs().max()) > 0 and bool(torch.isfinite(a1).all())
        if reps == 1:
            ref = _torch_conv(x.cpu(), wt.cpu(), bias.cpu(), 0, relu)
            assert rel_mae(a1.cpu(), ref) < 2e-5


@pytest.mark.parametrize('shape', [(1, 8, 64), (2, 20, 48), (1, 88, 1216)])
def test_op_conv32_narrow(shape):
    """The NARROW instantiations of the tuned kernels (bf16 maps, one bf16 MFMA per product: the mixed mode's proxy-pass and gradient
    maps) against torch on bf16-rounded operands: small maps (one-tile kernel), ragged tiles, and a map of > 256 tiles (the persistent
    LDS-staged stride-1 / stride-2 kernels and the direct transposed kernel)."""
    from proxytta.engine import op_conv32
    g = torch.Generator().manual_seed(5)
    b_, h_, w_ = shape
    x = torch.randn(b_, h_, w_, 32, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
    b = torch.randn(32, generator=g)
    for mode in (0, 1, 2):
        for relu in (True, False):
            xr = x.bfloat16().float()
            wr = w.bfloat16().float()
            ref = _torch_conv(xr, wr, b, mode, relu)
            got = op_conv32(x.cuda(), w.cuda(), b.cuda(), mode, relu_in=relu, in_major=(mode == 2), dtype='narrow').cpu()
            assert rel_mae(got, ref) < 4e-3, (mode, relu)        # output rounded to bf16 (2^-9 per element)
            assert rel_mae(got, ref.bfloat16().float()) < 4e-4, (mode, relu)     # ... and beyond that rounding only fp32 summation order


def _run_golden(name, impl, golden_dir):
    # gradients of the L1 / total-variation terms are sums of sign() functions: a 1e-5 perturbation of the depth map (bf16x3
    # arithmetic) flips a few signs, so the default-mode gradient tolerance is looser than the depth tolerance.  Bounds = 2x the worst
    # figure measured on MI355X over these fixtures (tools/grad_report.py, round 3): gradients, first step: exact 2.1e-6, default
    # 5.2e-3; later steps (they also carry Adam's sign-like first update): exact 6.8e-4 (the w_cos = 1 fixture; 1.2e-6 elsewhere),
    # default 9.3e-3; post-step parameters: exact 2.3e-5, default 3.6e-4
    gtol1, gtol, ptol = (5e-6, 1.4e-3, 5e-5) if impl in ('naive', 'exact') else (1.1e-2, 1.9e-2, 7.5e-4)
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    hp, gain = golden_hp(g)
    eng, sd, adapted = make_engine(n, h, w, 'fp32', hp, gain, impl)
    for s in range(steps):
        image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(s, h, w, n)]
        p = 's%d/' % s
        bufs = {k: v.clone() for k, v in sd.items() if k.endswith(('running_mean', 'running_var'))}
        # split call first, fused step second
        depth, emb, ref = eng.forward_train(image, sparse)
        assert rel_mae(depth, g[p + 'depth_train']) < 1e-4, (name, s)
        idx = g[p + 'row_idx']
        assert tuple(emb.shape) == tuple(g[p + 'emb_shape'])
        assert rel_mae(emb.cpu()[idx], g[p + 'emb_rows']) < 2e-4
        assert rel_mae(ref.cpu()[idx], g[p + 'ref_rows']) < 2e-4
        # forward_train just updated the BatchNorm1d running stats: put them back so that the fused
        # step below starts from the state the reference saw
        for k, v in bufs.items():
            sd[k].copy_(v)
        info, depth2 = eng.step(image, sparse, want_depth=True)
        torch.cuda.synchronize()
        assert rel_mae(depth2, g[p + 'depth_train']) < 1e-4
        np.testing.assert_allclose(info.cpu().numpy(), g[p + 'loss_info'], rtol=1e-4)
        gw, gb = eng.debug_tensor('gW').view(32, 32, 3, 3), eng.debug_tensor('gB')
        assert rel_mae(gw, g[p + 'grad/conv1_rgb_meta.weight']) < (gtol1 if s == 0 else gtol), (name, s)
        assert rel_mae(gb, g[p + 'grad/conv1_rgb_meta.bias']) < (gtol1 if s == 0 else gtol)
        for k, (prm, m, v) in adapted.items():
            assert rel_mae(prm, g[p + 'param/' + k]) < ptol, k
            assert rel_mae(m, g[p + 'exp_avg/' + k]) < gtol
            assert rel_mae(v, g[p + 'exp_avg_sq/' + k]) < 2 * gtol
        for k in g.files:
            if k.startswith(p + 'buf/'):
                key = k[len(p) + 4:]
                if key.startswith('proj_t'):
                    continue
                assert rel_mae(sd[key], g[k]) < 1e-4, k
        d_eval = eng.forward_eval(image, sparse)
        assert rel_mae(d_eval, g[p + 'depth_eval']) < 1e-4
    assert eng.adam_step_count() == steps
    eng.close()


@pytest.mark.parametrize('impl', ['naive', 'exact', None])
@pytest.mark.parametrize('name', CASES)
def test_step_matches_reference_golden(golden_dir, name, impl):
    """impl: naive = direct kernels, exact = fp32 MFMA everywhere, None = shipped default (bf16x3
    arithmetic on fp32 storage for the stride-1 convs)."""
    _run_golden(name, impl, golden_dir)


def test_loss_gate(golden_dir):
    g = np.load(os.path.join(golden_dir, 'adapt_loss_gate.npz'))
    h, w, n, rows, dim = [int(x) for x in g['meta']]
    eng, _, _ = make_engine(n, h, w, 'fp32', dict(max_input_depth=80.0))
    image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(11, h, w, n, density=0.1)]
    validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
    u = lambda tag, *shape: torch.from_numpy(
        (synth.hash_uniform(tag, int(np.prod(shape))) * 2 - 1).reshape(shape).astype(np.float32)).cuda()
    for tag in ('near', 'far'):
        depth = 20 + 10 * u('gate/depth', n, 1, h, w)
        emb = u('gate/emb', rows, dim)
        ref = emb + float(g[tag + '/noise']) * u('gate/noise' + tag, rows, dim)
        info = eng.loss_forward(image, depth, sparse, validity, emb, ref, 1.0, 2.0, 0.1)
        gd, gr = eng.loss_backward(image, depth, sparse, validity, emb, ref)
        np.testing.assert_allclose(info.cpu().numpy(), g[tag + '/loss_info'], rtol=2e-5)
        np.testing.assert_allclose(gd.cpu().numpy(), g[tag + '/grad_depth'], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(gr.cpu().numpy(), g[tag + '/grad_ref'], rtol=2e-4, atol=1e-9)
    eng.close()


def test_against_oracle_midsize():
    """128x256 (not in the golden set): fused step and eval forward against the oracle."""
    from oracle import proxytta_oracle as O
    n, h, w = 1, 128, 256
    hp = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, w_sparse_depth=1.0, w_smoothness=2.0,
              w_cos=0.1, max_input_depth=80.0)
    eng, sd, adapted = make_engine(n, h, w, 'fp32', hp)
    o = O.MsgChnOracle(synth.formula_state_dict(ONE), ONE, max_input_depth=80.0, lr=1e-3, weight_decay=1e-4,
                       w_sd=1.0, w_sm=2.0, w_cos=0.1)
    for s in range(2):
        image, sparse = synth.synthetic_frame(40 + s, h, w, n)
        r = o.step(torch.from_numpy(image), torch.from_numpy(sparse))
        info, depth = eng.step(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda(), want_depth=True)
        assert rel_mae(depth, r['depth']) < 1e-4
        li = r['loss_info']
        np.testing.assert_allclose(info.cpu().numpy(), [li['loss'], li['loss_smooth'], li['loss_sparse_depth'], li['loss_cos']], rtol=1e-4)
        for k, (prm, m, v) in adapted.items():
            assert rel_mae(prm, o.P[k].detach()) < 1e-3
        d_eval = eng.forward_eval(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda())
        assert rel_mae(d_eval, o.forward_eval(torch.from_numpy(image), torch.from_numpy(sparse))) < 1e-4
    eng.close()


def test_merged_head_linears_equal_the_two_layer_form():
    """emb = pred(proj(feat_zero)): proj.3 and pred.0 have nothing between them and run as ONE GEMM with W' = W_pred0 W_proj3 (derived in
    double when the weights are loaded).  Same embedding as the layer-by-layer form to fp32 rounding; the BatchNorm buffers likewise."""
    import os
    n, h, w = 1, 128, 256
    image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(7, h, w, n)]
    out = {}
    for fuse in ('1', '0'):
        eng, sd, adapted = make_engine(n, h, w, 'fp32', options={'fuse_heads': int(fuse)})
        depth, emb, ref = eng.forward_train(image, sparse)
        out[fuse] = (emb.clone(), ref.clone(), sd['pred.1.running_mean'].clone(), sd['pred.1.running_var'].clone(), sd['proj.1.running_var'].clone())
        eng.close()
    # measured 8.6e-6 = the size of either form's own bf16x3 deviation from fp32 arithmetic (dropped lo*lo terms, 2^-16 per product)
    assert rel_mae(out['1'][0], out['0'][0]) < 2e-5
    # ref: bit-identical while proj's hidden was materialised in both forms; heads v2 (default, with the merged form) recomputes it inside the
    # GEMM from analytic BatchNorm statistics (measured 5e-6)
    assert rel_mae(out['1'][1], out['0'][1]) < 2e-5
    for k in (2, 3, 4):
        assert rel_mae(out['1'][k], out['0'][k]) < 2e-5


@pytest.mark.parametrize('mode,fixture', [(ONE, 'msgchn_1layer_32x48'), (TWO, 'msgchn_2layers_32x48')])
def test_facade_reference_style_driver(golden_dir, mode, fixture):
    """tta_main-style loop (forward / compute_loss / zero_grad / backward / optimizer.step) through the
    ExternalModel_Adapt mirror gives the same numbers as the golden run (both meta layers)."""
    from proxytta.model import CANONICAL_LOSS_TYPE, ExternalModel_Adapt
    g = np.load(os.path.join(golden_dir, fixture + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    hp, gain = golden_hp(g)
    model = ExternalModel_Adapt('msg_chn', 0.0, 80.0, max_input_depth=hp['max_input_depth'], device=torch.device('cuda'))
    model._prepare_head(mode)
    model.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict(mode, gain).items()})
    params = model.adapt_parameters(mode='meta')
    names = [k for k, _ in model.model.model.named_parameters() if 'meta' in k]
    opt = torch.optim.Adam(params, lr=hp['lr'], betas=hp['betas'], eps=hp['eps'], weight_decay=hp['weight_decay'])
    for s in range(steps):
        image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(s, h, w, n)]
        validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
        model.train()
        depth, emb, ref = model.forward(image=image, sparse_depth=sparse, loss_type=CANONICAL_LOSS_TYPE)
        loss, info = model.compute_loss(input_rgb=image, output_depth=depth, sparse_depth=sparse, validity_map=validity,
                                        embedding=emb, reference=ref, w_loss_sparse_depth=hp['w_sparse_depth'],
                                        w_loss_smoothness=hp['w_smoothness'], w_loss_cos=hp['w_cos'], loss_type='adapt')
        opt.zero_grad()
        loss.backward()
        opt.step()
        p = 's%d/' % s
        assert abs(float(loss.detach()) - g[p + 'loss_info'][0]) < 2e-4 * abs(g[p + 'loss_info'][0])
        for k, prm in zip(names, params):
            if np.abs(g[p + 'grad/' + k]).max() < 1e-6:
                continue                          # analytically-zero gradient, see test_2layers_...
            assert rel_mae(prm.grad, g[p + 'grad/' + k]) < 3.2e-2, k        # measured <= 1.56e-2 (2layers, second step)
            assert rel_mae(prm, g[p + "param/" + k]) < 2e-3, k
        model.eval()
        with torch.no_grad():
            d_eval = model.forward(image=image, sparse_depth=sparse, loss_type=CANONICAL_LOSS_TYPE)
        assert rel_mae(d_eval, g[p + 'depth_eval']) < 3e-4


def test_full_size_properties():
    """352x1216 (BASELINE config): finite, deterministic bit-for-bit across two handles, Adam moves
    the parameters, and the eval depth after the step differs from before."""
    n, h, w = 1, 352, 1216
    hp = dict(w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
    image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(0, h, w, n)]
    outs = []
    for use_graph in (True, False):         # hipGraph replay and kernel-by-kernel launches: same bits
        eng, sd, adapted = make_engine(n, h, w, 'fp32', hp)
        eng.set_graph(use_graph)
        before = eng.forward_eval(image, sparse).clone()
        info, depth = eng.step(image, sparse, want_depth=True)
        info, depth = eng.step(image, sparse, want_depth=True)      # second step = first graph REPLAY
        after = eng.forward_eval(image, sparse)
        torch.cuda.synchronize()
        assert torch.isfinite(info).all() and torch.isfinite(depth).all() and torch.isfinite(after).all()
        assert (after - before).abs().max() > 0
        outs.append((info.cpu(), depth.cpu(), after.cpu(), adapted['conv1_rgb_meta.weight'][0].cpu().clone()))
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('impl', ['exact', None])
def test_2layers_meta_layer_matches_reference_golden(golden_dir, impl):
    """prepare_mode meta_selfsup_seq_2layers_ema (Res_Conv(32,128) with train-mode BatchNorm2d): the
    recipe of bash/adapt/adapt_msgchn_vkitti.sh.  Seven adapted tensors, BN running statistics."""
    g = np.load(os.path.join(golden_dir, 'msgchn_2layers_32x48.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    hp, gain = golden_hp(g)
    eng, sd, adapted = make_engine(n, h, w, 'fp32', hp, gain, impl, meta='2layers')
    # measured (tools/grad_report.py): exact 2.0e-6 both steps; default 2.5e-3 first step, 1.56e-2 second step
    gtol = 1e-5 if impl == 'exact' else 3.2e-2
    for s in range(steps):
        image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(s, h, w, n)]
        p = 's%d/' % s
        info, depth = eng.step(image, sparse, want_depth=True)
        torch.cuda.synchronize()
        assert rel_mae(depth, g[p + 'depth_train']) < 1e-4
        np.testing.assert_allclose(info.cpu().numpy(), g[p + 'loss_info'], rtol=2e-4)
        for k, (prm, m, v) in adapted.items():
            gref = g[p + 'grad/' + k]
            got = eng.grad(k, prm)
            if np.abs(gref).max() < 1e-6:           # conv bias in front of a BatchNorm: analytically zero
                assert float(got.abs().max()) < 1e-5
                assert np.abs(prm.cpu().numpy() - g[p + 'param/' + k]).max() <= 2.5 * hp['lr'] * (s + 1)
                continue
            assert rel_mae(got, gref) < gtol, (k, s)
            assert rel_mae(prm, g[p + 'param/' + k]) < (1e-4 if impl == 'exact' else 2e-3), k
        for k in g.files:
            if k.startswith(p + 'buf/'):
                key = k[len(p) + 4:]
                if key.startswith('proj_t'):
                    continue
                assert rel_mae(sd[key], g[k]) < 2e-3, k
        d_eval = eng.forward_eval(image, sparse)
        assert rel_mae(d_eval, g[p + 'depth_eval']) < 3e-4
    eng.close()


def test_outlier_removal_matches_reference_golden(golden_dir):
    """OutlierRemoval(7, 1.5) fused HIP stencil vs the reference's output (bit-exact: selection only)."""
    from proxytta.model import OutlierRemoval
    g = np.load(os.path.join(golden_dir, 'outlier_removal.npz'))
    _, sparse = synth.synthetic_frame(7, 40, 56, 2, density=0.2, dmin=1.0, dmax=20.0)
    sparse = torch.from_numpy(sparse).cuda()
    validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
    sd, vm = OutlierRemoval(7, 1.5).remove_outliers(sparse, validity)
    np.testing.assert_array_equal(sd.cpu().numpy(), g['sparse_out'])
    np.testing.assert_array_equal(vm.cpu().numpy(), g['validity_out'])
    # full size (352x1216) against the oracle, bit-exact
    from oracle import proxytta_oracle as O
    _, big = synth.synthetic_frame(3, 352, 1216, 1, density=0.05, dmin=1.0, dmax=80.0)
    big = torch.from_numpy(big).cuda()
    vbig = (big > 0).float()
    sd2, vm2 = OutlierRemoval(7, 1.5).remove_outliers(big, vbig)
    rs, rv = O.remove_outliers(big.cpu(), vbig.cpu(), 7, 1.5)
    assert torch.equal(sd2.cpu(), rs) and torch.equal(vm2.cpu(), rv)
    assert 0 < float(vm2.sum()) < float(vbig.sum())          # some points were removed, not all


def test_eval_metrics_on_device():
    from oracle import proxytta_oracle as O
    from proxytta.model import eval_metrics
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(2, 1, 352, 1216, generator=g) * 90.0
    gt[torch.rand(gt.shape, generator=g) < 0.7] = 0.0            # semi-dense ground truth
    out = (gt + torch.randn(gt.shape, generator=g)).clamp(min=0.1) + (gt == 0) * 5.0
    ref = O.eval_metrics(out, gt, 0.5, 80.0)
    got = eval_metrics(out.cuda(), gt.cuda(), 0.5, 80.0).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5)


GATE_CASES = ['msgchn_1layer_64x96_gate_below', 'msgchn_1layer_64x96_gate_above']


@pytest.mark.parametrize('path', ['graph', 'pipelined', 'pipelined_graph', 'eager'])
@pytest.mark.parametrize('impl', ['exact', None])
@pytest.mark.parametrize('name', GATE_CASES)
def test_fused_step_on_both_sides_of_the_cosine_gate(golden_dir, name, impl, path):
    """The `loss_cos < 0.3 => w_loss_cos = 0` gate (src/external_model_adapt.py:424-425) INSIDE the fused step -- loss_finalize_block
    run by the two gradient kernels -- through ptta_step and the frame-pipelined call (ptta_step_pipelined), each with direct launches
    (the default: 'eager', 'pipelined') and with hipGraph replay ('graph', 'pipelined_graph'), against whole steps of the REAL reference on either side of the gate (L_cos = 0.20: the branch trained heads
    take, the cosine term drops out of loss and gradients; L_cos = 0.37: it stays and, with w_cos = 300, is ~8 % of the adapted gradient,
    so a gate taken the wrong way fails every bound below)."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    hp, gain = golden_hp(g)
    eng, sd, adapted = make_engine(n, h, w, 'fp32', hp, gain, impl, head_bias=float(g['head_bias']), options={'graph': 1 if 'graph' in path else 0})
    # bounds = 2x the worst figure measured on MI355X on these two fixtures (tools/gate_report.py, round 4): first-step gradients exact 9.1e-7 /
    # default 5.6e-3; later steps exact 2.0e-3 (one ReLU decision after Adam's sign-like first update, w_cos = 300) / default 1.2e-2;
    # post-step parameters exact 2.7e-5 / default 6.0e-4.  A gate taken the wrong way moves the first-step gradient by ~8e-2.
    gtol1, gtol, ptol = (5e-6, 4.1e-3, 6e-5) if impl == 'exact' else (1.2e-2, 2.5e-2, 1.2e-3)
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(s, h, w, n)] for s in range(steps + 1)]
    below = 'below' in name
    for s in range(steps):
        image, sparse = frames[s]
        p = 's%d/' % s
        info, depth = eng.step(image, sparse, want_depth=True, next_frame=frames[s + 1] if 'pipelined' in path else None)
        torch.cuda.synchronize()
        li = info.cpu().numpy()
        np.testing.assert_allclose(li, g[p + 'loss_info'], rtol=1e-4)
        assert (li[3] < 0.3) == below
        dense = hp['w_sparse_depth'] * li[2] + hp['w_smoothness'] * li[1]
        assert abs(li[0] - dense) < 1e-4 * li[0] if below else li[0] > dense + 100.0
        assert rel_mae(depth, g[p + 'depth_train']) < 1e-4
        gw, gb = eng.debug_tensor('gW').view(32, 32, 3, 3), eng.debug_tensor('gB')
        assert rel_mae(gw, g[p + 'grad/conv1_rgb_meta.weight']) < (gtol1 if s == 0 else gtol), (name, s)
        assert rel_mae(gb, g[p + 'grad/conv1_rgb_meta.bias']) < (gtol1 if s == 0 else gtol)
        for k, (prm, m, v) in adapted.items():
            assert rel_mae(prm, g[p + 'param/' + k]) < ptol, k
            assert rel_mae(m, g[p + 'exp_avg/' + k]) < gtol
            assert rel_mae(v, g[p + 'exp_avg_sq/' + k]) < 2 * gtol
        d_eval = eng.forward_eval_last() if 'pipelined' in path else eng.forward_eval(image, sparse)
        assert rel_mae(d_eval, g[p + 'depth_eval']) < 1e-4
    eng.close()

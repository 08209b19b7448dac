"""Edge cases and the other BASELINE.json shapes, HIP path vs the oracle (through the C-ABI)."""

import numpy as np
import pytest
import torch

from oracle import proxytta_oracle as O
from proxytta import synth
from tests.util import ONE, make_engine, rel_mae

pytestmark = pytest.mark.gpu
HP = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1,
          max_input_depth=80.0)


def _oracle():
    return O.MsgChnOracle(synth.formula_state_dict(ONE), ONE, max_input_depth=80.0, lr=1e-3, w_sd=1.0, w_sm=2.0, w_cos=0.1)


# post-update eval depth after this path's OWN Adam step, against the oracle's: the north_star tolerance (measured on MI355X, round 3:
# 2.9e-4 at 256x320, 5.6e-4 at 480x640, 9.7e-5 at 3 x 48x80)
EVAL_TOL = 1e-3


@pytest.mark.parametrize('shape', [(1, 256, 320), (1, 480, 640), (3, 48, 80)])
def test_other_config_shapes_against_oracle(shape):
    """BASELINE.json configs[0] (320x256 VOID frame), the 640x480 VOID shape and an odd batch."""
    n, h, w = shape
    eng, sd, adapted = make_engine(n, h, w, 'fp32', HP)
    o = _oracle()
    image, sparse = synth.synthetic_frame(5, h, w, n, density=1500.0 / (h * w) if h >= 256 else 0.05, dmin=0.2, dmax=8.0)
    r = o.step(torch.from_numpy(image), torch.from_numpy(sparse))
    info, depth = eng.step(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda(), want_depth=True)
    assert rel_mae(depth, r['depth']) < 1e-4
    li = r['loss_info']
    np.testing.assert_allclose(info.cpu().numpy(), [li['loss'], li['loss_smooth'], li['loss_sparse_depth'], li['loss_cos']], rtol=2e-4)
    # the post-update depth also carries the (sign-flip) gradient noise of the step through Adam (entries with a near-zero
    # gradient move by +-lr either way); with 1500-point indoor frames the depth itself is O(1): a bound for this path's own
    # update, and the tight check from the ORACLE's post-step parameters
    ref_eval = o.forward_eval(torch.from_numpy(image), torch.from_numpy(sparse))
    d_eval = eng.forward_eval(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda())
    assert rel_mae(d_eval, ref_eval) < EVAL_TOL, rel_mae(d_eval, ref_eval)
    # this path's OWN update in parameter space: wherever the gradient is clear of the sign-flip noise floor (|g| > 1e-3 of the
    # tensor's largest entry) Adam's first step is -lr * sign(g) on both sides: identical parameters to 1e-3 of one step
    for k, (prm, m, v) in adapted.items():
        gref = r['grads'][k]
        clear = gref.abs() > 1e-3 * gref.abs().max()
        assert float(clear.float().mean()) > 0.9, k
        dp = (prm.cpu() - o.P[k].detach()).abs()
        assert float(dp[clear].max()) < 1e-3 * HP['lr'], (k, float(dp[clear].max()))
    for k, (prm, m, v) in adapted.items():
        prm.copy_(o.P[k].detach())
    d_eval = eng.forward_eval(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda())
    assert rel_mae(d_eval, ref_eval) < 1e-4
    eng.close()


def test_frame_without_valid_points_gives_nan_like_the_reference():
    """sparse_depth_consistency_loss_func divides by sum(w) with no eps (src/loss_utils.py:116-137): a frame
    with no valid point yields NaN in the reference; the HIP path must not hide it."""
    n, h, w = 1, 32, 48
    eng, sd, adapted = make_engine(n, h, w, 'fp32', HP)
    image, _ = synth.synthetic_frame(1, h, w, n)
    sparse = torch.zeros(n, 1, h, w)
    o = _oracle()
    r = o.step(torch.from_numpy(image), sparse)
    info, _ = eng.step(torch.from_numpy(image).cuda(), sparse.cuda())
    assert np.isnan(r['loss_info']['loss_sparse_depth']) and bool(torch.isnan(info[2]))
    assert bool(torch.isnan(info[0]))
    eng.close()


def test_max_input_depth_clamp_and_none():
    """max_input_depth clamps the sparse input in forward AND in the loss (external_model_adapt.py:108,:192-193);
    None disables both."""
    n, h, w = 1, 32, 48
    image, sparse = synth.synthetic_frame(2, h, w, n, dmin=1.0, dmax=200.0)
    for mid in (80.0, None):
        hp = dict(HP); hp['max_input_depth'] = mid
        eng, sd, adapted = make_engine(n, h, w, 'fp32', hp)
        o = O.MsgChnOracle(synth.formula_state_dict(ONE), ONE, max_input_depth=mid, lr=1e-3, w_sd=1.0, w_sm=2.0, w_cos=0.1)
        r = o.step(torch.from_numpy(image), torch.from_numpy(sparse))
        info, depth = eng.step(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda(), want_depth=True)
        assert rel_mae(depth, r['depth']) < 1e-4
        assert abs(float(info[2]) - r['loss_info']['loss_sparse_depth']) < 2e-4 * abs(r['loss_info']['loss_sparse_depth'])
        eng.close()


def test_explicit_validity_and_separate_loss_image():
    """step(image1, sparse, validity_map=filtered validity, loss_image=image): tta_main.py:610 feeds the
    normalised image to the network and the un-normalised one to the smoothness weights (:620)."""
    n, h, w = 2, 32, 48
    eng, sd, adapted = make_engine(n, h, w, 'fp32', HP)
    o = _oracle()
    image, sparse = synth.synthetic_frame(3, h, w, n, density=0.2)
    loss_image = 255.0 * image
    sp_t = torch.from_numpy(sparse)
    val = torch.where(sp_t > 0, torch.ones_like(sp_t), sp_t)
    sd_f, val_f = O.remove_outliers(sp_t, val, 7, 1.5)
    r = o.step(torch.from_numpy(image), sd_f, validity_map=val_f, loss_image=torch.from_numpy(loss_image))
    info, depth = eng.step(torch.from_numpy(image).cuda(), sd_f.cuda(), validity=val_f.cuda(),
                           loss_image=torch.from_numpy(loss_image).cuda(), want_depth=True)
    assert rel_mae(depth, r['depth']) < 1e-4
    li = r['loss_info']
    np.testing.assert_allclose(info.cpu().numpy(), [li['loss'], li['loss_smooth'], li['loss_sparse_depth'], li['loss_cos']], rtol=2e-4)
    eng.close()


def test_shared_parameter_step_single_rank_equals_fused_step():
    """proxytta.distributed.shared_parameter_step (split calls + gradient all-reduce + Adam) with one rank is the
    fused step."""
    from proxytta.distributed import shared_parameter_step
    n, h, w = 1, 32, 48
    image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(4, h, w, n)]
    e1, sd1, ad1 = make_engine(n, h, w, 'fp32', HP)
    e2, sd2, ad2 = make_engine(n, h, w, 'fp32', HP)
    info1, _ = e1.step(image, sparse)
    info2, _ = shared_parameter_step(e2, image, sparse, w=(HP['w_sparse_depth'], HP['w_smoothness'], HP['w_cos']))
    torch.cuda.synchronize()
    assert torch.allclose(info1, info2, rtol=1e-6)
    # round 4: the fused step forms d loss_cos / d ref inside the backward GEMM's operand staging, the split calls write it as a tensor first.
    # The same expression compiled twice may differ in the last bit, and the GEMM's bf16 hi / lo split of its A operand turns a last-bit change
    # into a 2^-17 step of the represented value: the adapted gradients agree to 2.9e-6 / 2.0e-6 of their mean magnitude (measured, 32x48;
    # 64x96: 2.3e-6 / 1.3e-6) -- the size of bf16x3's own error -- no longer bit for bit (option cos_in_gemm = 0: bit-identical again).  Adam's first
    # step is lr * sign(g) wherever |g| >> eps, so the parameters are identical except at entries whose gradient is within that noise of zero.
    for name, k in (('gW', 'conv1_rgb_meta.weight'), ('gB', 'conv1_rgb_meta.bias')):
        g1, g2 = e1.debug_tensor(name), e2.debug_tensor(name)
        assert rel_mae(g1, g2) < 6e-6, name
        clear = (g1.abs() > 1e-3 * g1.abs().max()).view(-1)
        assert torch.allclose(ad1[k][0].reshape(-1)[clear], ad2[k][0].reshape(-1)[clear], rtol=0, atol=1e-7), k
        assert clear.float().mean() > 0.97
    e1.close(); e2.close()


@pytest.mark.parametrize('n,h,w', [(1, 64, 96), (2, 176, 608), (1, 352, 1216)])
def test_sign_bit_masks_equal_float_masks_bit_for_bit(n, h, w):
    """Round 4: the backward reads ONE word of sign bits per pixel (written by the forward epilogues) instead of the fp32 pre-activation
    pixel, the prediction heads' backward runs as one launch from those bits and the fused init block writes only the bits of its first
    map.  Same predicate (> 0) on the same stored values: three steps with the option mask_bits = 0 (float masks, the unfused launches) and with the
    default must agree bit for bit -- depth, loss terms, adapted parameters and Adam moments.  352x1216 is where the fused large-map kernels
    run; 64x96 takes the small-map kernels; two frames of 176x608 index the bit planes with a batch (b % nb) and mix both kernel families."""
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(20 + i, h, w, n)] for i in range(3)]
    runs = []
    for bits in (0, 1):
        eng, sd, ad = make_engine(n, h, w, 'fp32', HP, options={'mask_bits': bits})
        out = []
        for im, sp in frames:
            info, depth = eng.step(im, sp, want_depth=True)
            out.append((info.clone(), depth.clone()))
        torch.cuda.synchronize()
        runs.append((out, {k: [t.clone() for t in v] for k, v in ad.items()}))
        eng.close()
    (o0, a0), (o1, a1) = runs
    for (i0, d0), (i1, d1) in zip(o0, o1):
        assert torch.equal(i0, i1) and torch.equal(d0, d1)
    for k in a0:
        for t0, t1 in zip(a0[k], a1[k]):
            assert torch.equal(t0, t1), k


def test_shared_parameter_step_2layers_meta():
    """The general path of shared_parameter_step (gradients read from / written back to the library, on-device Adam)
    with the 7-tensor 2layers meta layer, one rank: equals the fused step."""
    from proxytta.distributed import shared_parameter_step
    n, h, w = 1, 32, 48
    image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(4, h, w, n)]
    # (this test is about the plumbing of the general path: with the tensor form of d loss_cos / d ref in the fused step the two are bit-identical;
    # the in-GEMM form is compared with the split calls in test_shared_parameter_step_single_rank_equals_fused_step)
    e1, sd1, ad1 = make_engine(n, h, w, 'fp32', HP, meta='2layers', options={'cos_in_gemm': 0})
    e2, sd2, ad2 = make_engine(n, h, w, 'fp32', HP, meta='2layers', options={'cos_in_gemm': 0})
    assert len(e2.adapted) == 7
    info1, _ = e1.step(image, sparse)
    info2, _ = shared_parameter_step(e2, image, sparse, w=(HP['w_sparse_depth'], HP['w_smoothness'], HP['w_cos']))
    torch.cuda.synchronize()
    assert torch.allclose(info1, info2, rtol=1e-6)
    for k in ad1:
        assert torch.allclose(ad1[k][0], ad2[k][0], rtol=0, atol=1e-7), k
    e1.close(); e2.close()


@pytest.mark.parametrize('rng,shape', [([0, 1], (2, 32, 48)), ([-1, 1], (1, 32, 48)),
                                       ([[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]], (1, 32, 48)),
                                       ([[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]], (1, 36, 52))])
def test_fused_image_normalisation(rng, shape):
    """set_image_norm(normalized_image_range): the engine takes the RAW 0..255 image, normalises it inside the
    first convolution (or inside the dual-corner padding for sizes that are not multiples of 16) and the loss uses
    the raw image -- equal to the reference flow normalize_images -> forward(image1) / compute_loss(image)
    (transforms.py:668-710, tta_main.py:610,620)."""
    n, h, w = shape
    eng, sd, adapted = make_engine(n, h, w, 'fp32', HP)
    o = _oracle()
    image01, sparse = synth.synthetic_frame(5, h, w, n, density=0.2)
    raw = torch.from_numpy(np.floor(image01 * 255.0).astype(np.float32))
    if rng == [0, 1]:
        img1 = raw / 255.0
    elif rng == [-1, 1]:
        img1 = 2.0 * (raw / 255.0) - 1.0
    else:
        mean = torch.tensor(rng[0]).view(1, 3, 1, 1); std = torch.tensor(rng[1]).view(1, 3, 1, 1)
        img1 = (raw / 255.0 - mean) / std
    sp_t = torch.from_numpy(sparse)
    r = o.step(img1, sp_t, loss_image=raw)
    eng.set_image_norm(rng)
    info, depth = eng.step(raw.cuda(), sp_t.cuda(), want_depth=True)
    assert rel_mae(depth, r['depth']) < 1e-4
    li = r['loss_info']
    np.testing.assert_allclose(info.cpu().numpy(), [li['loss'], li['loss_smooth'], li['loss_sparse_depth'], li['loss_cos']], rtol=2e-4)
    d_eval = eng.forward_eval(raw.cuda(), sp_t.cuda())
    assert rel_mae(d_eval, o.forward_eval(img1, sp_t)) < 1e-4
    with pytest.raises(ValueError):
        eng.set_image_norm([0, 2])
    eng.close()


def test_adam_step_count_continues_across_frame_shapes(tmp_path):
    """torch.optim.Adam keeps ONE step count per parameter; the façade builds one engine per (N, H, W).  A shape change
    (last short batch of a loader with drop_last=False, src/tta_main.py:281) must continue the bias correction, and a
    checkpoint restored through restore_model(optimizer=...) must rebind the replaced Adam state."""
    from proxytta.model import ExternalModel_Adapt
    model = ExternalModel_Adapt('msg_chn', 0.0, 80.0, max_input_depth=80.0, device=torch.device('cuda'))
    model._prepare_head(ONE)
    model.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict(ONE).items()})
    params = model.adapt_parameters(mode='meta')
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    model.model.set_hparams(w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1)
    model.model.bind_optimizer(opt)
    o = _oracle()
    shapes = [(1, 32, 48), (1, 32, 48), (2, 32, 64), (1, 32, 48)]
    for i, (n, h, w) in enumerate(shapes):
        image, sparse = synth.synthetic_frame(60 + i, h, w, n)
        o.step(torch.from_numpy(image), torch.from_numpy(sparse))
        model.step(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda())
        if i == 1:                      # save + restore in the middle of the sequence
            path = str(tmp_path / 'ckpt.pth')
            model.save_model(path, i, opt)
            model.restore_model(path, optimizer=opt)
    assert model.model._adam_t == len(shapes)
    assert float(opt.state[params[0]]['step']) == len(shapes)
    for k, prm in zip(('conv1_rgb_meta.weight', 'conv1_rgb_meta.bias'), params):
        assert rel_mae(prm, o.P[k].detach()) < 2e-3, k
    assert rel_mae(opt.state[params[0]]['exp_avg'], o.opt.m[0]) < 3e-2


def test_get_time_hook():
    """forward(loss_type containing 'time') accumulates wall-clock like the reference's networks do
    (network_exp_msg_chn_adapt.py:338-340,407-414); forward(loss_type='get_time') reads [total, train, eval]."""
    from proxytta.model import CANONICAL_LOSS_TYPE, ExternalModel_Adapt
    model = ExternalModel_Adapt('msg_chn', 0.0, 80.0, max_input_depth=80.0, device=torch.device('cuda'))
    model._prepare_head(ONE)
    model.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict(ONE).items()})
    image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(0, 32, 48, 1)]
    assert model.forward(image, sparse, loss_type='get_time') == [0.0, 0.0, 0.0]
    model.train()
    model.forward(image, sparse, loss_type=CANONICAL_LOSS_TYPE + '_time')
    model.eval()
    model.forward(image, sparse, loss_type=CANONICAL_LOSS_TYPE + '_time')
    model.forward(image, sparse, loss_type=CANONICAL_LOSS_TYPE)           # not timed
    total, train, evalt = model.forward(image, sparse, loss_type='get_time')
    assert train > 0 and evalt > 0 and abs(total - train - evalt) < 1e-9


def test_adapt_loop_with_look_ahead_equals_the_plain_loop():
    """ExternalModel_Adapt.adapt (one TTA step + the scored eval forward per frame, src/tta_main.py:579-636, :729-736) with the data loader's
    look-ahead frame announced: same depths and parameters as without it, also across a shape change (new engine: no pipelining there)."""
    from proxytta.model import ExternalModel_Adapt
    shapes = [(1, 32, 48), (1, 32, 48), (1, 32, 48), (1, 48, 64), (1, 48, 64)]
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(70 + i, h, w, n)] for i, (n, h, w) in enumerate(shapes)]
    out = {}
    for ahead in (False, True):
        model = ExternalModel_Adapt('msg_chn', 0.0, 80.0, max_input_depth=80.0, device=torch.device('cuda'))
        model._prepare_head(ONE)
        model.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict(ONE).items()})
        params = model.adapt_parameters(mode='meta')
        opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        model.model.set_hparams(w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1)
        model.model.bind_optimizer(opt)
        rec = []
        for i, (image, sparse) in enumerate(frames):
            nxt = frames[i + 1] if (ahead and i + 1 < len(frames)) else None
            depth, info = model.adapt(image, sparse, inner_iter=1, next_frame=nxt)
            rec.append((depth.clone(), info.clone(), [p.detach().clone() for p in params]))
        torch.cuda.synchronize()
        out[ahead] = rec
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for pa, pb in zip(a[2], b[2]):
            assert torch.equal(pa, pb)


# sizes chosen to land tile edges everywhere: widths around the 32-pixel tile and the 64-pixel strided tile, heights around the 8- / 4-row tiles,
# not divisible by 16 (dual-corner padded path), odd batches; one full tile row more / less than a multiple
SWEEP = [(1, 16, 16), (1, 17, 33), (2, 24, 100), (1, 40, 129), (3, 33, 65), (1, 72, 31), (2, 47, 94), (1, 129, 67), (1, 64, 257), (2, 81, 49)]


@pytest.mark.parametrize('dtype', ['fp32', 'mixed'])
@pytest.mark.parametrize('shape', SWEEP)
def test_shape_sweep_against_oracle(shape, dtype):
    """Every kernel's tile-edge handling at once: one full step + the scored eval forward from the ORACLE's post-step parameters, at ten
    shapes around the tile sizes, both precision modes.  Bounds as test_other_config_shapes_against_oracle (fp32) / tests/test_gpu_mixed.py
    (mixed: the real frames' forward is the fp32 one)."""
    n, h, w = shape
    eng, sd, adapted = make_engine(n, h, w, dtype, HP)
    o = _oracle()
    image, sparse = synth.synthetic_frame(11 + h + w, h, w, n, density=0.08, dmin=0.2, dmax=8.0)
    r = o.step(torch.from_numpy(image), torch.from_numpy(sparse))
    info, depth = eng.step(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda(), want_depth=True)
    assert rel_mae(depth, r['depth']) < 1e-4, rel_mae(depth, r['depth'])
    li = r['loss_info']
    np.testing.assert_allclose(info.cpu().numpy(), [li['loss'], li['loss_smooth'], li['loss_sparse_depth'], li['loss_cos']],
                               rtol=2e-4 if dtype == 'fp32' else 3e-3)
    for k, (prm, m, v) in adapted.items():
        g = eng.grad(k, prm)
        assert rel_mae(g, r['grads'][k]) < (3e-2 if dtype == 'fp32' else 8e-2), (k, rel_mae(g, r['grads'][k]))
        prm.copy_(o.P[k].detach())
    d_eval = eng.forward_eval(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda())
    ref_eval = o.forward_eval(torch.from_numpy(image), torch.from_numpy(sparse))
    assert rel_mae(d_eval, ref_eval) < 1e-4, rel_mae(d_eval, ref_eval)
    eng.close()


"""CostDCNet backbone (BASELINE config 5) on libptta_hip against the oracle
(oracle/costdcnet_oracle.py) and the golden vectors generated from the real reference (tests/golden/costdcnet_*.npz;
MinkowskiEngine provided by oracle/minkowski_lite.py on both sides: the sparse encoder's arithmetic is parity-unpinned)."""
import os

import numpy as np
import pytest
import torch

from proxytta import synth
from proxytta.engine import Engine
from tests.util import rel_mae

pytestmark = pytest.mark.gpu

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
MAX_DEPTH = 8.0
HP = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=None)
# 'naive': direct fp32 kernels (exact arithmetic); 'default': matrix-core convolutions, bf16x6 forward for the real frames
# (three-way operand split, fp32-grade products), bf16x3 for the proxy frames and the data gradients.
# Bounds = 2x the worst figure measured on MI355X (tools/costdc_report.py, round 3): depth 1.5e-6 / 2.4e-6, gradients 5.0e-3 /
# 5.1e-3 (rel. MAE of the worst of the 32 tensors; post-step parameters 9.3e-5; the reference's own fp32 gradients sit 1.8e-3 from an fp64 evaluation,
# tests/golden/costdcnet_fp64.npz), post-update eval depth 2.5e-4 / 6.0e-4.
TOL = {'naive': dict(depth=1e-5, emb=1e-3, grad=1e-2, param=2e-4, eval=1e-3), 'default': dict(depth=1e-5, emb=5e-3, grad=1e-2, param=2e-4, eval=1e-3)}
MODES = ['naive', 'default']


def costdc_frame(idx, h, w, n, density=0.05):
    image01, sparse = synth.synthetic_frame(idx, h, w, n, density=density, dmin=0.3, dmax=7.5)
    raw = np.floor(image01 * 255.0).astype(np.float32)
    return raw, ((raw / np.float32(255.0) - MEAN) / STD).astype(np.float32), sparse


def make_costdc(n, h, w, hp=HP, impl='default'):
    old = os.environ.get('PTTA_CONV_IMPL')
    if impl == 'naive':
        os.environ['PTTA_CONV_IMPL'] = 'naive'
    else:
        os.environ.pop('PTTA_CONV_IMPL', None)
    try:
        eng = Engine(n, h, w, backbone='costdcnet', max_predict_depth=MAX_DEPTH, **hp)
    finally:
        if old is None:
            os.environ.pop('PTTA_CONV_IMPL', None)
        else:
            os.environ['PTTA_CONV_IMPL'] = old
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_costdcnet().items()}
    for k in list(sd):            # one BatchNorm behind two names
        if k.startswith('enc2d.') and '.downsample.1.' in k:
            sd[k] = sd[k.replace('.downsample.1.', '.norm3.')]
    eng.load_state_dict(sd)
    adapted = {}
    for k in eng.adapted:
        p = sd[k].clone().contiguous()
        adapted[k] = (p, torch.zeros_like(p), torch.zeros_like(p))
        eng.bind_adapted(k, *adapted[k])
    return eng, sd, adapted


def _oracle(hp=HP):
    from oracle import costdcnet_oracle as CO
    return CO, CO.CostDcnOracle(synth.formula_state_dict_costdcnet(), max_depth=MAX_DEPTH, max_input_depth=hp['max_input_depth'], lr=hp['lr'],
                                betas=hp['betas'], eps=hp['eps'], weight_decay=hp['weight_decay'], w_sd=hp['w_sparse_depth'],
                                w_sm=hp['w_smoothness'], w_cos=hp['w_cos'])


def test_adapted_set_is_the_reference_list():
    eng, sd, adapted = make_costdc(1, 64, 96)
    CO, o = _oracle()
    assert eng.adapted == o.names and len(eng.adapted) == 32 and sum(eng.adapted_numel.values()) == 5200
    assert eng.rows == 1 * 2 * 3
    eng.close()


@pytest.mark.parametrize('impl', MODES)
def test_intermediates_match_oracle(impl):
    """Stage by stage: 2-D features, fused volume, cost volume, bottleneck, depth, embeddings (training forward)."""
    n, h, w = 1, 64, 96
    tol = TOL[impl]
    eng, sd, adapted = make_costdc(n, h, w, impl=impl)
    CO, o = _oracle()
    raw, image1, sparse = [torch.from_numpy(x) for x in costdc_frame(0, h, w, n)]
    pred, emb, ref, inter = CO.network_forward(o.P, image1, sparse, True, MAX_DEPTH, want_intermediates=True)
    depth, e_, r_ = eng.forward_train(image1.cuda(), sparse.cuda())
    h4, w4 = h // 4, w // 4
    f2 = eng.debug_tensor('feat2d').view(2 * n, h4, w4, 16)[:n].permute(0, 3, 1, 2)
    assert rel_mae(f2, inter['feat2d'].detach()) < tol['depth'] * 2
    vol = eng.debug_tensor('vol').view(2 * n, 16, h4, w4, 32)[:n].permute(0, 4, 1, 2, 3)
    assert rel_mae(vol[:, 16:], inter['vol'][:, 16:].detach()) < 1e-4, 'sparse encoder / densify'
    assert rel_mae(vol[:, :16], inter['vol'][:, :16].detach()) < tol['depth'] * 2, 'fusion mask'
    cost = eng.debug_tensor('cost').view(n, 16, h4, w4, 16).permute(0, 4, 1, 2, 3)
    assert rel_mae(cost, inter['cost'].detach()) < tol['depth'] * 5
    assert rel_mae(depth, pred.detach()) < tol['depth']
    assert rel_mae(e_, emb) < tol['emb'] and rel_mae(r_, ref.detach()) < tol['emb']
    eng.close()


@pytest.mark.parametrize('impl', MODES)
@pytest.mark.parametrize('name', ['costdcnet_64x96', 'costdcnet_64x64_n2', 'costdcnet_72x100_pad'])
def test_step_matches_golden(golden_dir, name, impl):
    tol = TOL[impl]
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, max_depth = [float(x) for x in g['hp']]
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=None)
    eng, sd, adapted = make_costdc(n, h, w, hp, impl=impl)
    names = [str(x) for x in g['adapted_names']]
    assert eng.adapted == names
    for s in range(steps):
        raw, image1, sparse = [torch.from_numpy(x).cuda() for x in costdc_frame(s, h, w, n, float(g['density']))]
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        p = 's%d/' % s
        if s > 0:
            # Adam's FIRST update moves every entry by lr * g / (|g| + eps) (lr = 3e-3 in the CostDCNet scripts): entries whose
            # gradient is near zero move by a sign- and size-dependent amount (exact arithmetic: 3.5e-2 of the mean parameter
            # at 64x96), so the SECOND step is held to the north-star tolerance only (measured 6.1e-4, exact mode 6.8e-4)
            assert rel_mae(depth, g[p + 'depth_train']) < 1.5e-3, (name, s)
            continue
        assert rel_mae(depth, g[p + 'depth_train']) < tol['depth'], (name, s)
        np.testing.assert_allclose(info.cpu().numpy(), g[p + 'loss_info'], rtol=2e-3, atol=1e-7)
        for k in names:
            assert rel_mae(eng.grad(k, adapted[k][0]), g[p + 'grad/' + k]) < tol['grad'], k
            assert np.abs(adapted[k][0].cpu().numpy() - g[p + 'param/' + k]).max() <= 2.0 * lr * 1.01, k     # at worst a flipped +-lr step
            assert rel_mae(adapted[k][0], g[p + 'param/' + k]) < 5 * tol['param'], k
        for k in g.files:                       # tracked BatchNorm buffers (BatchNorm3d, heads, sparse encoder)
            if k.startswith(p + 'buf/'):
                assert rel_mae(sd[k[len(p) + 4:]], g[k]) < 2e-3, k
        assert rel_mae(eng.forward_eval(image1, sparse), g[p + 'depth_eval']) < tol['eval']        # the scored tensor, after this path's OWN update
        # the eval path itself, from the REFERENCE's post-step parameters (the tracked running statistics are this
        # engine's own, updated by the training forward above): tight
        keep = {k: adapted[k][0].clone() for k in names}
        for k in names:
            adapted[k][0].copy_(torch.from_numpy(g[p + 'param/' + k]))
        assert rel_mae(eng.forward_eval(image1, sparse), g[p + 'depth_eval']) < 1e-5            # measured 1.2e-6 (both modes)
        for k in names:
            adapted[k][0].copy_(keep[k])
    assert eng.adam_step_count() == steps
    eng.close()


@pytest.mark.parametrize('name', ['costdcnet_320x400', 'costdcnet_480x640'])
def test_full_size_step_matches_reference(golden_dir, name):
    """The ScanNet script's frame (bash/adapt/adapt_costdc_scannet.sh: 320x400) and BASELINE config 5's 480x640 VOID frame
    (1500 points), default arithmetic, against sampled pixels / block means / checksums produced by the reference."""
    from tests.test_gpu_fullsize import _check_map
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, max_depth = [float(x) for x in g['hp']]
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=None)
    eng, sd, adapted = make_costdc(n, h, w, hp)
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in costdc_frame(0, h, w, n, float(g['density']))]
    info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
    _check_map(depth, g, 's0/depth_train', 1e-5)                                            # measured 2.4e-6
    np.testing.assert_allclose(info.cpu().numpy(), g['s0/loss_info'], rtol=1e-4, atol=1e-7)
    for k in eng.adapted:
        assert rel_mae(eng.grad(k, adapted[k][0]), g['s0/grad/' + k]) < TOL['default']['grad'], k
        assert np.abs(adapted[k][0].cpu().numpy() - g['s0/param/' + k]).max() <= 2.0 * lr * 1.01, k
    # the scored tensor (src/tta_main.py:729-736): eval depth after this path's OWN Adam step, north_star tolerance 1e-3
    # (measured 1.1e-6 at 320x400, 6.0e-4 at 480x640; exact-arithmetic mode 1.0e-6 / 2.6e-4)
    d_eval = eng.forward_eval(image1, sparse)
    _check_map(d_eval, g, 's0/depth_eval', 1e-3)
    # ... and against the fp64 evaluation of the same step (tests/golden/costdcnet_fp64.npz): the reference's own fp32 result is
    # 9.9e-5 / 2.9e-4 away from it (one near-zero gradient entry takes the opposite first step); this path may be at most 1e-3
    g64 = np.load(os.path.join(golden_dir, 'costdcnet_fp64.npz'))
    e64 = g64[name + '/depth_eval_pix']
    mine = d_eval.detach().cpu().numpy().reshape(-1)[g['pix_idx']].astype(np.float64)
    ref_noise = float(g64[name + '/reference_vs_fp64'][1])
    err64 = float(np.abs(mine - e64).mean() / np.abs(e64).mean())
    assert err64 < 1e-3, (err64, ref_noise)
    for k in eng.adapted:                                                                  # the eval path from the reference's parameters
        adapted[k][0].copy_(torch.from_numpy(g['s0/param/' + k]))
    _check_map(eng.forward_eval(image1, sparse), g, 's0/depth_eval', 1e-5)                  # measured 1.1e-6
    eng.close()


def test_split_calls_equal_fused_step():
    """forward_train / loss_forward / loss_backward / backward / adam_step (the reference's call sequence,
    src/tta_main.py:610-633) give the same parameters as the fused ptta_step."""
    n, h, w = 1, 64, 96
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in costdc_frame(3, h, w, n)]
    validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
    eng, sd, ad1 = make_costdc(n, h, w)
    eng.step(image1, sparse, loss_image=raw)
    eng.close()
    eng, sd, ad2 = make_costdc(n, h, w)
    depth, emb, ref = eng.forward_train(image1, sparse)
    eng.loss_forward(raw, depth, sparse, validity, emb, ref, 1.0, 2.0, 0.1)
    gd, gr = eng.loss_backward(raw, depth, sparse, validity, emb, ref)
    eng.backward_all(gd, gr, {k: v[0] for k, v in ad2.items()})
    eng.adam_step()
    torch.cuda.synchronize()
    for k in ad1:
        assert torch.equal(ad1[k][0], ad2[k][0]), k
    eng.close()


def test_external_model_adapt_facade_costdcnet(golden_dir):
    """The reference's driver calls (src/tta_main.py:309-354, 610-633, 729-736) against the façade:
    ExternalModel_Adapt('costdcnet') -> _prepare_head -> load weights -> adapt_parameters('meta_bn') -> Adam -> forward /
    compute_loss / backward / optimizer.step -> eval forward; golden step 0."""
    from proxytta.model import CANONICAL_LOSS_TYPE, ExternalModel_Adapt
    g = np.load(os.path.join(golden_dir, 'costdcnet_64x96.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, max_depth = [float(x) for x in g['hp']]
    model = ExternalModel_Adapt('costdcnet', 0.1, max_depth, max_input_depth=None, device=torch.device('cuda'))
    model._prepare_head('meta_selfsup_seq_1layer_ema')
    model.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict_costdcnet().items()})
    params = model.adapt_parameters(mode='meta_bn')
    names = [str(x) for x in g['adapted_names']]
    assert len(params) == 32 and model.model.adapted == names
    opt = torch.optim.Adam(params, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in costdc_frame(0, h, w, n, float(g['density']))]
    validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
    model.train()
    depth, emb, ref = model.forward(image=image1, sparse_depth=sparse, loss_type=CANONICAL_LOSS_TYPE)
    loss, info = model.compute_loss(input_rgb=raw, output_depth=depth, sparse_depth=sparse, validity_map=validity, embedding=emb,
                                    reference=ref, w_loss_sparse_depth=w_sd, w_loss_smoothness=w_sm, w_loss_cos=w_cos, loss_type='adapt')
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert rel_mae(depth, g['s0/depth_train']) < 1e-5
    assert abs(float(loss.detach()) - g['s0/loss_info'][0]) < 2e-3 * abs(g['s0/loss_info'][0])
    for k, prm in zip(names, params):
        assert rel_mae(prm.grad, g['s0/grad/' + k]) < TOL['default']['grad'], k
    model.eval()
    with torch.no_grad():
        d_eval = model.forward(image=image1, sparse_depth=sparse, loss_type=CANONICAL_LOSS_TYPE)
    assert rel_mae(d_eval, g['s0/depth_eval']) < 1e-3            # measured 2.3e-4
    # the checkpoint carries the reference's key set (ResBlock.norm3 listed twice) and the updated running statistics
    sd = model.model.state_dict()
    assert list(sd.keys()) == [k for k, _ in synth.costdcnet_keys()]
    assert rel_mae(sd['unet3d.inc.double_conv.0.bn1.running_mean'], g['s0/buf/unet3d.inc.double_conv.0.bn1.running_mean']) < 2e-3


def test_graph_replay_equals_kernel_by_kernel_launches():
    """ptta_step / ptta_forward_eval replay captured hipGraphs from their second call on (the op-list engine): three steps and
    two eval forwards with replay give bit-identical parameters, statistics and depth maps to PTTA_GRAPH=0."""
    n, h, w = 1, 64, 96
    res = []
    for graph in (0, 1):
        eng, sd, ad = make_costdc(n, h, w)
        eng._chk(eng.lib.ptta_set_graph(eng.handle, graph), 'ptta_set_graph')
        for s in range(3):
            raw, image1, sparse = [torch.from_numpy(x).cuda() for x in costdc_frame(s, h, w, n)]
            info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
            d_eval = eng.forward_eval(image1, sparse)
        torch.cuda.synchronize()
        res.append((info.clone(), depth.clone(), d_eval.clone(), {k: v[0].clone() for k, v in ad.items()},
                    sd['unet3d.inc.double_conv.0.bn1.running_mean'].clone()))
        eng.close()
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[4], b[4])
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


@pytest.mark.parametrize('shape', [(64, 96), (72, 100)])
def test_fused_image_normalisation(shape):
    """Engine.set_image_norm([mean, std]) fuses Transforms.normalize_images (src/transforms.py:668-710) into the input staging -- or,
    for sizes that are not multiples of 16, into the dual-corner padding (the reference normalises before it pads: the padding stays
    zero): a step on the RAW image equals a step on the pre-normalised image bit for bit."""
    n = 1
    h, w = shape
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in costdc_frame(0, h, w, n)]
    res = []
    for fused in (False, True):
        eng, sd, ad = make_costdc(n, h, w)
        if fused:
            eng.set_image_norm([tuple(float(v) for v in MEAN.reshape(-1)), tuple(float(v) for v in STD.reshape(-1))])
        info, depth = eng.step(raw if fused else image1, sparse, loss_image=raw, want_depth=True)
        d_eval = eng.forward_eval(raw if fused else image1, sparse)
        torch.cuda.synchronize()
        res.append((info.clone(), depth.clone(), d_eval.clone()))
        eng.close()
    # (v / 255 - mean) / std on the host (numpy) and on the device are the same three float32 operations
    assert rel_mae(res[1][1], res[0][1]) < 1e-6 and rel_mae(res[1][2], res[0][2]) < 1e-6
    np.testing.assert_allclose(res[1][0].cpu().numpy(), res[0][0].cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize('impl', ['default', 'naive'])
def test_sequence_of_frames_stays_as_close_to_the_reference_as_the_reference_to_itself(golden_dir, impl):
    """ONE CostDCNet parameter set adapted over 16 different 96x128 frames by the REAL reference (tests/golden/make_golden_costdcnet.py
    costdcnet_96x128_seq16), the scored eval forward after every step -- and, in the same fixture, the reference run again with one adapted
    weight ONE ULP off (`alt/`): its own trajectory is 2.0e-4 away after three frames, 1.2e-3 after four and 7e-2 after sixteen (the arg-max
    over the cost volume turns every rounding difference of the update into a different depth plane for some pixels).  What can be asserted:
    the north_star's 1e-3 while the reference itself is inside it (first three frames: measured 9.7e-5 default, 2.2e-4 exact arithmetic),
    and afterwards the same ORDER as that floor (measured worst ratio 1.6; bound 3x + 5e-4)."""
    g = np.load(os.path.join(golden_dir, 'costdcnet_96x128_seq16.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, max_depth = [float(x) for x in g['hp']]
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=None)
    eng, sd, adapted = make_costdc(n, h, w, hp, impl)
    pix = lambda t: t.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']]
    floor = [rel_mae(g['alt/s%d/depth_eval_pix' % s], g['s%d/depth_eval_pix' % s]) for s in range(steps)]
    assert floor[2] < 1e-3 < floor[3] and floor[-1] > 2e-2            # the fixture's own statement
    for s in range(steps):
        raw, image1, sparse = [torch.from_numpy(x).cuda() for x in costdc_frame(s, h, w, n, float(g['density']))]
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        err = rel_mae(pix(eng.forward_eval(image1, sparse)), g['s%d/depth_eval_pix' % s])
        if s < 3:
            assert err < 5e-4, (s, err)
            np.testing.assert_allclose(info.cpu().numpy(), g['s%d/loss_info' % s], rtol=2e-3, atol=1e-7)
        assert err < 3.0 * floor[s] + 5e-4, (s, err, floor[s])
    eng.close()


# quarter-resolution maps of 4..33 pixels per side, the dual-corner padded path (sizes not divisible by 8), odd batches
COSTDC_SWEEP = [(1, 64, 64), (1, 40, 104), (2, 56, 72), (1, 68, 132), (3, 48, 40), (1, 100, 60)]


@pytest.mark.parametrize('shape', COSTDC_SWEEP)
def test_shape_sweep_against_oracle(shape):
    """One full step (default arithmetic) + the eval forward from the ORACLE's post-step parameters and buffers at shapes around the tile
    sizes of the 2-D and the P3D kernels."""
    n, h, w = shape
    eng, sd, adapted = make_costdc(n, h, w)
    CO, o = _oracle()
    raw, image1, sparse = [torch.from_numpy(x) for x in costdc_frame(2 + w, h, w, n)]
    r = o.step(image1, sparse, loss_image=raw)
    info, depth = eng.step(image1.cuda(), sparse.cuda(), loss_image=raw.cuda(), want_depth=True)
    assert rel_mae(depth, r['depth']) < TOL['default']['depth'], rel_mae(depth, r['depth'])
    li = r['loss_info']
    np.testing.assert_allclose(info.cpu().numpy(), [li['loss'], li['loss_smooth'], li['loss_sparse_depth'], li['loss_cos']], rtol=1e-3, atol=1e-7)
    for k in eng.adapted:
        assert rel_mae(eng.grad(k, adapted[k][0]), r['grads'][k]) < 2 * TOL['default']['grad'], (k, rel_mae(eng.grad(k, adapted[k][0]), r['grads'][k]))
    eng.close()


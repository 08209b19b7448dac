"""NLSPN backbone on libptta_hip against the oracle (oracle/nlspn_oracle.py) and the golden
vectors generated from the real reference (tests/golden/nlspn_*.npz)."""
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
HP = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1,
          max_input_depth=80.0)


def nlspn_frame(idx, h, w, n):
    image01, sparse = synth.synthetic_frame(idx, h, w, n, density=0.1)
    raw = np.floor(image01 * 255.0).astype(np.float32)
    return raw, ((raw / np.float32(255.0) - MEAN) / STD).astype(np.float32), sparse


# 'naive': direct fp32 kernels everywhere (exact arithmetic, tight bounds); 'default': bf16x3 matrix-core convolutions
# (each product carries ~2^-17 relative error, through ~40 normalised layers).
# Gradient bounds = 2x the worst figure measured on MI355X over every fixture incl. 352x1216 (tools/grad_report.py, round 3):
#   worst full-gradient rel. MAE of an adapted tensor: naive 6.8e-3, default 2.1e-2 (conv3.0.downsample.1.*: the deep BatchNorm
#   affine gradients are ill-conditioned -- the fp32 CPU reference itself sits 1.3e-2 from an fp64 evaluation of the same graph);
#   worst gradient-NORM error over the 88 tensors: naive 2.1e-3, default 4.4e-3;  depth: naive 3.9e-7, default 2.6e-6.
TOL = {'naive': dict(feat=1e-4, depth=1e-5, emb=1e-3, grad=1.4e-2, gnorm=5e-3, param=2e-3),
       'default': dict(feat=1e-3, depth=1e-4, emb=5e-3, grad=4.2e-2, gnorm=9e-3, param=4e-3)}
MODES = ['naive', 'default']


def make_nlspn(n, h, w, hp=HP, impl='default', legacy=False):
    old = os.environ.get('PTTA_CONV_IMPL')
    if impl == 'naive':
        os.environ['PTTA_CONV_IMPL'] = 'naive'
    else:
        os.environ.pop('PTTA_CONV_IMPL', None)
    try:
        eng = Engine(n, h, w, backbone='nlspn', legacy_offset=legacy, **hp)
    finally:
        if old is None:
            os.environ.pop('PTTA_CONV_IMPL', None)
        else:
            os.environ['PTTA_CONV_IMPL'] = old
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
    eng.load_state_dict({k: v for k, v in sd.items() if v.dtype == torch.float32})
    adapted = {}
    for k in eng.adapted:
        p = sd[k].clone().contiguous()
        adapted[k] = (p, torch.zeros_like(p), torch.zeros_like(p))
        eng.bind_adapted(k, *adapted[k])
    return eng, sd, adapted


def _oracle(hp=HP):
    from oracle import nlspn_oracle as N
    return N, N.NlspnOracle(synth.formula_state_dict_nlspn(), max_input_depth=hp['max_input_depth'], lr=hp['lr'], betas=hp['betas'],
                            eps=hp['eps'], weight_decay=hp['weight_decay'], w_sd=hp['w_sparse_depth'], w_sm=hp['w_smoothness'],
                            w_cos=hp['w_cos'])


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def test_adapted_set_is_the_reference_list():
    eng, sd, adapted = make_nlspn(1, 32, 64)
    N, o = _oracle()
    assert eng.adapted == o.names and len(eng.adapted) == 88 and sum(eng.adapted_numel.values()) == 40048
    eng.close()


@pytest.mark.parametrize('impl', MODES)
@pytest.mark.parametrize('shape', [(1, 32, 64), (2, 48, 80)])
def test_forward_train_and_eval_match_oracle(shape, impl):
    n, h, w = shape
    tol = TOL[impl]
    eng, sd, adapted = make_nlspn(n, h, w, impl=impl)
    N, o = _oracle()
    raw, image1, sparse = [torch.from_numpy(x) for x in nlspn_frame(0, h, w, n)]
    with torch.no_grad():
        d_ref, e_ref, r_ref, inter = N.network_forward(o.P, image1, torch.clamp(sparse, 0, 80.0), True, want_intermediates=True)
    depth, emb, ref = eng.forward_train(image1.cuda(), sparse.cuda())
    fe6 = eng.debug_tensor('fe6').view(2 * n, h // 16, w // 16, 512)[:n]
    assert rel_mae(fe6, nhwc(inter['fe6'])) < tol['feat']
    assert rel_mae(eng.debug_tensor('pred_init').view(n, 1, h, w), inter['pred_init']) < tol['feat']
    assert rel_mae(eng.debug_tensor('confidence').view(n, 1, h, w), inter['confidence']) < tol['feat']
    off9 = eng.debug_tensor('off9').view(n, 18, h, w)
    aff9 = eng.debug_tensor('aff9').view(n, 9, h, w)
    assert rel_mae(off9, inter['offset']) < tol['feat']
    assert rel_mae(aff9, inter['aff']) < tol['feat']
    assert rel_mae(depth, d_ref) < tol['depth']
    assert rel_mae(emb, e_ref) < tol['emb'] and rel_mae(ref, r_ref) < tol['emb']
    d_eval = eng.forward_eval(image1.cuda(), sparse.cuda())
    assert rel_mae(d_eval, o.forward_eval(image1, sparse)) < tol['depth']
    eng.close()


@pytest.mark.parametrize('impl', MODES)
@pytest.mark.parametrize('name', ['nlspn_32x64', 'nlspn_48x80_n2', 'nlspn_32x64_canonical', 'nlspn_32x64_legacy', 'nlspn_40x56_n2_legacy'])
def test_step_matches_golden(golden_dir, name, impl):
    tol = TOL[impl]
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid = [float(x) for x in g['hp']]
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=mid)
    eng, sd, adapted = make_nlspn(n, h, w, hp, impl=impl, legacy=bool(int(g['legacy'])))
    names = [str(x) for x in g['adapted_names']]
    assert eng.adapted == names
    for s in range(steps):
        raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(s, h, w, n)]
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        p = 's%d/' % s
        assert rel_mae(depth, g[p + 'depth_train']) < (tol['depth'] if s == 0 else 1e-3)     # first step: 2.6e-6 measured; north_star 1e-3
        np.testing.assert_allclose(info.cpu().numpy(), g[p + 'loss_info'], rtol=2e-3)
        if s == 0:
            # Gradients of the deep BatchNorm affine parameters are ill-conditioned: the fp32 CPU reference itself sits
            # ~1e-2 from an fp64 evaluation (a few ReLU-mask / sign flips), so 3e-2 is the meaningful bound; after the
            # first update Adam turns that noise into +-lr parameter moves, so later steps are pinned on depth and loss
            # here and on gradients by test_second_step_from_oracle_state.
            gn = np.array([float(eng.grad(k, adapted[k][0]).double().norm()) for k in names])
            np.testing.assert_allclose(gn, g[p + 'grad_norms'], rtol=tol['gnorm'], atol=1e-6)
            for key in g.files:
                if key.startswith(p + 'grad/'):
                    k = key[len(p + 'grad/'):]
                    assert rel_mae(eng.grad(k, adapted[k][0]), g[key]) < tol['grad'], k
                    assert rel_mae(adapted[k][0], g[p + 'param/' + k]) < tol['param'], k
        d_eval = eng.forward_eval(image1, sparse)
        assert rel_mae(d_eval, g[p + 'depth_eval']) < 1e-3
    eng.close()


@pytest.mark.parametrize('impl', MODES)
def test_second_step_from_oracle_state(impl):
    """Step 2 of a sequence, started from the oracle's exact post-step-1 parameters and Adam moments: gradients, the
    Adam update and the depth map against the oracle's second step."""
    n, h, w = 1, 32, 64
    N, o = _oracle()
    raw, image1, sparse = [torch.from_numpy(x) for x in nlspn_frame(0, h, w, n)]
    o.step(image1, sparse, loss_image=raw)
    tol = TOL[impl]
    eng, sd, adapted = make_nlspn(n, h, w, impl=impl)
    for i, k in enumerate(o.names):
        adapted[k][0].copy_(o.P[k].detach())
        adapted[k][1].copy_(o.opt.m[i])
        adapted[k][2].copy_(o.opt.v[i])
    eng.set_adam_step(1)
    raw, image1, sparse = [torch.from_numpy(x) for x in nlspn_frame(1, h, w, n)]
    r = o.step(image1, sparse, loss_image=raw)
    info, depth = eng.step(image1.cuda(), sparse.cuda(), loss_image=raw.cuda(), want_depth=True)
    assert eng.adam_step_count() == 2
    assert rel_mae(depth, r['depth']) < tol['depth']
    for i, k in enumerate(eng.adapted):
        assert rel_mae(eng.grad(k, adapted[k][0]), r['grads'][k]) < tol['grad'], k
        assert rel_mae(adapted[k][0], o.P[k].detach()) < tol['param'], k
        assert rel_mae(adapted[k][1], o.opt.m[i]) < tol['grad'], k
    eng.close()


def test_fused_image_normalisation_nlspn():
    n, h, w = 1, 32, 64
    eng, sd, adapted = make_nlspn(n, h, w)
    N, o = _oracle()
    raw, image1, sparse = [torch.from_numpy(x) for x in nlspn_frame(1, h, w, n)]
    eng.set_image_norm([[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]])
    d = eng.forward_eval(raw.cuda(), sparse.cuda())
    assert rel_mae(d, o.forward_eval(image1, sparse)) < 1e-4
    eng.close()


def test_sizes_not_divisible_by_16_have_the_reference_shapes():
    """Odd encoder maps (nlspnmodel_adapt.py:474-490): 36 x 52 -> fe6 is 3 x 4, the embedding has N * 12 rows."""
    eng = Engine(2, 36, 52, backbone='nlspn', **HP)
    assert eng.rows == 2 * 3 * 4
    eng.close()
    with pytest.raises(RuntimeError):
        Engine(1, 8, 52, backbone='nlspn', **HP)


def test_external_model_adapt_facade_nlspn(golden_dir):
    """The reference's driver calls (src/tta_main.py:309-354, 610-633, 729-736) against the façade:
    ExternalModel_Adapt('nlspn') -> _prepare_head -> load weights -> adapt_parameters('meta_bn') -> Adam -> step -> eval."""
    from proxytta.model import ExternalModel_Adapt
    g = np.load(os.path.join(golden_dir, 'nlspn_32x64.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    model = ExternalModel_Adapt('nlspn', 0.0, 80.0, max_input_depth=80.0, device=torch.device('cuda'))
    model._prepare_head('meta_selfsup_seq_1layer_ema')
    model.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict_nlspn().items()})
    params = model.adapt_parameters(mode='meta_bn')
    assert len(params) == 88 and sum(p.numel() for p in params) == 40048
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    model.model.bind_optimizer(opt)
    model.model.set_hparams(w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1)
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
    model.train()
    info, depth = model.step(image1, sparse, loss_image=raw, want_depth=True)
    assert rel_mae(depth, g['s0/depth_train']) < 1e-3
    np.testing.assert_allclose(info.cpu().numpy(), g['s0/loss_info'], rtol=2e-3)
    assert float(opt.state[params[0]]['step']) == 1.0 and float(opt.state[params[0]]['exp_avg'].abs().sum()) > 0
    model.eval()
    d_eval = model.forward(image1, sparse, loss_type='adapt_meta_selfsup_seq_ema_reverse')
    assert rel_mae(d_eval, g['s0/depth_eval']) < 1e-3
    # fused normalisation through the façade
    model.set_image_norm([[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]])
    model.eval()
    d2 = model.forward(raw, sparse, loss_type='adapt_meta_selfsup_seq_ema_reverse')
    assert rel_mae(d2, d_eval) < 1e-4


def test_error_paths_fail_loudly():
    """Unknown keys, wrong shapes, unbound adapted tensors and MSG_CHN-only entry points are errors with a message,
    never a silent fallback."""
    eng = Engine(1, 32, 64, backbone='nlspn', **HP)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
    with pytest.raises(RuntimeError, match='unknown state_dict key'):
        eng.load_state_dict({'conv99.0.weight': sd['conv6.0.weight']})
    with pytest.raises(RuntimeError, match='shape mismatch'):
        eng.load_state_dict({'conv6.0.weight': sd['dec5.0.weight']})
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, 32, 64, 1)]
    with pytest.raises(RuntimeError, match='not bound'):
        eng.forward_eval(image1, sparse)
    for k in eng.adapted:
        eng.bind_adapted(k, sd[k].clone(), torch.zeros_like(sd[k]), torch.zeros_like(sd[k]))
    with pytest.raises(RuntimeError, match='not loaded'):
        eng.forward_eval(image1, sparse)
    with pytest.raises(RuntimeError, match='not an adapted parameter'):
        eng.bind_adapted('conv6.0.weight', sd['conv6.0.weight'], sd['conv6.0.weight'], sd['conv6.0.weight'])
    with pytest.raises(RuntimeError, match='without a preceding'):
        eng.backward_all(torch.zeros(1, 1, 32, 64, device='cuda'), None, {})
    eng.close()


def test_reference_style_driver_nlspn(golden_dir):
    """src/tta_main.py:610-633 verbatim against the façade: forward -> compute_loss -> zero_grad -> loss.backward() ->
    torch.optim.Adam.step() on the 88 adapted tensors, then the eval forward; golden step 0."""
    from proxytta.model import ExternalModel_Adapt
    g = np.load(os.path.join(golden_dir, 'nlspn_32x64.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    model = ExternalModel_Adapt('nlspn', 0.0, 80.0, max_input_depth=80.0, device=torch.device('cuda'))
    model._prepare_head('meta_selfsup_seq_1layer_ema')
    model.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict_nlspn().items()})
    params = model.adapt_parameters(mode='meta_bn')
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
    validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
    model.train()
    depth, emb, ref = model.forward(image=image1, sparse_depth=sparse, loss_type='adapt_meta_selfsup_seq_ema_reverse')
    loss, info = model.compute_loss(input_rgb=raw, output_depth=depth, sparse_depth=sparse, validity_map=validity, embedding=emb,
                                    reference=ref, w_loss_sparse_depth=1.0, w_loss_smoothness=2.0, w_loss_cos=0.1, loss_type='adapt')
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert rel_mae(depth.detach(), g['s0/depth_train']) < 1e-3
    np.testing.assert_allclose([float(torch.as_tensor(info[k]).detach()) for k in ('loss', 'loss_smooth', 'loss_sparse_depth', 'loss_cos')], g['s0/loss_info'], rtol=2e-3)
    named = dict(model.model.model.named_parameters())
    names = [str(x) for x in g['adapted_names']]
    gn = np.array([float(named[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(gn, g['s0/grad_norms'], rtol=8e-2, atol=1e-6)
    for key in g.files:
        if key.startswith('s0/param/'):
            k = key[len('s0/param/'):]
            assert rel_mae(named[k].detach(), g[key]) < 4e-3, k
    model.eval()
    d_eval = model.forward(image1, sparse, loss_type='adapt_meta_selfsup_seq_ema_reverse')
    assert rel_mae(d_eval, g['s0/depth_eval']) < 1e-3


@pytest.mark.parametrize('shape', [(1, 352, 1216), (3, 240, 1216), (4, 224, 320)])
def test_full_size_properties(shape):
    """BASELINE config 3 size (352x1216) and the per-GPU batches of the reference's two NLSPN scripts (n_batch 12 at
    240x1216 and 16 at 224x320 over 4 GPUs, bash/adapt/adapt_nlspn_{vkitti,nyuv2}.sh), where the CPU oracle is too slow: (i) the bf16x3 matrix-core path agrees
    with the exact direct-kernel path (validated against oracle and golden vectors above) on depth and loss,
    (ii) the training forward's depth equals the eval forward's depth for the same parameters (both normalise
    with batch statistics; the proxy half of the training batch must not leak into the real half's statistics),
    (iii) a step moves every one of the 88 adapted tensors by at most lr per entry (Adam's first step) and by a
    non-zero amount."""
    n, h, w = shape
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
    out = {}
    for impl in MODES:
        eng, sd, adapted = make_nlspn(n, h, w, impl=impl)
        d_train, emb, ref = eng.forward_train(image1, sparse)
        d_eval = eng.forward_eval(image1, sparse)
        assert torch.isfinite(d_train).all() and torch.isfinite(emb).all() and torch.isfinite(ref).all()
        assert rel_mae(d_train, d_eval) < 1e-6
        before = {k: adapted[k][0].clone() for k in eng.adapted}
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        assert torch.isfinite(info).all()
        for k in eng.adapted:
            delta = (adapted[k][0] - before[k]).abs()
            assert float(delta.max()) <= HP['lr'] * 1.001 and float(delta.max()) > 0, k
        out[impl] = (depth.clone(), info.cpu().numpy().copy())
        eng.close()
    assert rel_mae(out['default'][0], out['naive'][0]) < 1e-3
    np.testing.assert_allclose(out['default'][1], out['naive'][1], rtol=2e-3)


def test_shared_parameter_step_nlspn():
    """proxytta.distributed.shared_parameter_step on the NLSPN engine (88 gradients as one flat message, ptta_set_grad,
    on-device Adam) with one rank equals the fused step up to the float atomics of the propagation gradient."""
    from proxytta.distributed import shared_parameter_step
    n, h, w = 1, 32, 64
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
    e1, sd1, ad1 = make_nlspn(n, h, w)
    e2, sd2, ad2 = make_nlspn(n, h, w)
    info1, _ = e1.step(image1, sparse, loss_image=raw)
    info2, _ = shared_parameter_step(e2, image1, sparse, loss_image=raw, w=(HP['w_sparse_depth'], HP['w_smoothness'], HP['w_cos']))
    torch.cuda.synchronize()
    assert torch.allclose(info1, info2, rtol=1e-5)
    assert e2.adam_step_count() == 1
    for k in ad1:
        # two FUSED steps on identical engines already differ by 0.4e-5 .. 2.6e-5 on conv5.0.bn1.bias (measured in round 2:
        # the propagation gradient's float atomics + Adam's sign-like first step), so 1e-5 was inside the run-to-run noise
        assert rel_mae(ad2[k][0], ad1[k][0]) < 1e-4, k
    e1.close(); e2.close()


@pytest.mark.parametrize('name', ['nlspn_96x320_legacy', 'nlspn_352x1216_legacy', 'nlspn_228x304_legacy'])      # 228 x 304: NYUv2, not divisible by 16
def test_full_size_step_matches_reference(golden_dir, name):
    """Default arithmetic at 96x320 and at the BASELINE size 352x1216 against vectors produced by the REAL reference
    (tests/golden/make_golden_nlspn.py <size>): 4096 sampled pixels, 8x8 block means and checksums of the training and
    the post-update eval depth, loss terms, gradient norms of all 88 adapted tensors, full gradients of ten of them."""
    from tests.test_gpu_fullsize import _check_map
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid = [float(x) for x in g['hp']]
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=mid)
    eng, sd, adapted = make_nlspn(n, h, w, hp, legacy=True)
    names = [str(x) for x in g['adapted_names']]
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
    info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
    _check_map(depth, g, 's0/depth_train', 1e-4)                       # measured 2.6e-6 (north_star: 1e-3 relative MAE)
    np.testing.assert_allclose(info.cpu().numpy(), g['s0/loss_info'], rtol=2e-3)
    gn = np.array([float(eng.grad(k, adapted[k][0]).double().norm()) for k in names])
    np.testing.assert_allclose(gn, g['s0/grad_norms'], rtol=TOL['default']['gnorm'], atol=1e-6)
    for key in g.files:
        if key.startswith('s0/grad/'):
            k = key[len('s0/grad/'):]
            assert rel_mae(eng.grad(k, adapted[k][0]), g[key]) < TOL['default']['grad'], k
    # the reference's eval output BEFORE its skimage hole filling (exact zeros stay zeros on both sides)
    _check_map(eng.forward_eval(image1, sparse), g, 's0/depth_eval', 1e-3)
    eng.close()


def test_syncbn_adapted_list_has_94_tensors():
    """tta_main.py:326 converts to SyncBatchNorm BEFORE adapt_parameters('meta_bn'): the heads' BatchNorm1d join the adapted
    list (94 tensors, module order proj.1, proj_t.1, pred.1).  proj_t carries the cosine-term gradient; proj / pred feed the
    detached embedding and keep zero gradients.  Gradients and the update against the oracle with the same list."""
    from oracle import nlspn_oracle as N
    n, h, w = 1, 32, 64
    eng = Engine(n, h, w, backbone='nlspn', syncbn_adapted=True, **HP)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
    eng.load_state_dict({k: v for k, v in sd.items() if v.dtype == torch.float32})
    adapted = {}
    for k in eng.adapted:
        p = sd[k].clone().contiguous()
        adapted[k] = (p, torch.zeros_like(p), torch.zeros_like(p))
        eng.bind_adapted(k, *adapted[k])
    o = N.NlspnOracle(synth.formula_state_dict_nlspn(), max_input_depth=HP['max_input_depth'], lr=HP['lr'], w_sd=1.0, w_sm=2.0, w_cos=0.1,
                      syncbn_adapted=True)
    assert len(eng.adapted) == 94 and eng.adapted == o.names
    assert eng.adapted[-6:] == ['proj.1.weight', 'proj.1.bias', 'proj_t.1.weight', 'proj_t.1.bias', 'pred.1.weight', 'pred.1.bias']
    raw, image1, sparse = [torch.from_numpy(x) for x in nlspn_frame(0, h, w, n)]
    r = o.step(image1, sparse, loss_image=raw)
    eng.step(image1.cuda(), sparse.cuda(), loss_image=raw.cuda())
    for k in ('proj_t.1.weight', 'proj_t.1.bias'):
        assert rel_mae(eng.grad(k, adapted[k][0]), r['grads'][k]) < TOL['default']['grad'], k
        assert rel_mae(adapted[k][0], o.P[k].detach()) < TOL['default']['param'], k
    for k in ('proj.1.weight', 'pred.1.bias'):
        assert float(eng.grad(k, adapted[k][0]).abs().max()) == 0.0 and torch.equal(adapted[k][0], sd[k])
    eng.close()


def test_heads_batchnorm_buffers_are_updated():
    """meta_bn drops the running statistics of BatchNorm2d only: the heads' BatchNorm1d stay in train mode WITH tracking, so
    every TTA step moves proj / proj_t / pred running_mean, running_var (momentum 0.1, unbiased) and num_batches_tracked --
    what save_model writes after adaptation."""
    n, h, w = 1, 32, 64
    eng = Engine(n, h, w, backbone='nlspn', **HP)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
    eng.load_state_dict(sd)
    for k in eng.adapted:
        p = sd[k].clone().contiguous()
        eng.bind_adapted(k, p, torch.zeros_like(p), torch.zeros_like(p))
    before = {k: sd[k].clone() for k in sd if k.startswith(('proj', 'pred')) and 'running' in k}
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
    eng.forward_train(image1, sparse)
    hpre = eng.debug_tensor('proj_t.h').view(-1, 1024)          # pre-BatchNorm activations of proj_t
    torch.cuda.synchronize()
    exp_mean = 0.9 * before['proj_t.1.running_mean'] + 0.1 * hpre.mean(0)
    exp_var = 0.9 * before['proj_t.1.running_var'] + 0.1 * hpre.var(0, unbiased=True)
    assert rel_mae(sd['proj_t.1.running_mean'], exp_mean) < 1e-4 and rel_mae(sd['proj_t.1.running_var'], exp_var) < 1e-4
    for k in ('proj.1', 'proj_t.1', 'pred.1'):
        assert int(sd[k + '.num_batches_tracked']) == 1
        assert not torch.equal(sd[k + '.running_mean'], before[k + '.running_mean'])
    eng.close()


def test_graph_replay_equals_kernel_by_kernel_launches():
    """The NLSPN step (about 500 launches) and eval forward replayed from hipGraphs (second call on) against PTTA_GRAPH=0: bit-identical
    except for the propagation gradient's LDS float atomics (order-dependent in both modes): parameters within 1e-6."""
    n, h, w = 1, 32, 64
    res = []
    for graph in (0, 1):
        eng, sd, ad = make_nlspn(n, h, w)
        eng._chk(eng.lib.ptta_set_graph(eng.handle, graph), 'ptta_set_graph')
        for s in range(3):
            raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(s, h, w, n)]
            info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
            d_eval = eng.forward_eval(image1, sparse)
        torch.cuda.synchronize()
        res.append((info.clone(), depth.clone(), d_eval.clone(), {k: v[0].clone() for k, v in ad.items()}))
        eng.close()
    a, b = res
    assert rel_mae(b[1], a[1]) < 1e-4 and rel_mae(b[2], a[2]) < 1e-4          # measured 1.6e-5 after three steps
    np.testing.assert_allclose(b[0].cpu().numpy(), a[0].cpu().numpy(), rtol=1e-3)   # two kernel-by-kernel runs differ by 5e-5 (float atomics)
    tot = same = 0
    for k in a[3]:
        d = (a[3][k] - b[3][k]).abs()
        tot += d.numel(); same += int((d < 1e-4).sum())          # lr = 1e-3: an entry that took the opposite Adam step would differ by 2e-3
    # measured 0.995 typically (two kernel-by-kernel runs: 0.998), 0.988 once in ~6 runs: the atomics' order decides the sign of gradients
    # that are zero up to rounding, and three Adam steps at lr = 1e-3 turn a flipped sign into 2e-3.  0.97 = the observed spread doubled.
    assert same >= 0.97 * tot, (same, tot)


def test_eval_forward_fills_holes_like_the_reference(golden_dir):
    """src/nlspn_model_adapt.py:124-127: the eval forward's exact zeros (the clamp at nlspnmodel_adapt.py:371) are filled by
    biharmonic inpainting on the host.  At 228x304 the reference's eval output holds one hole (tests/golden/nlspn_228x304_legacy.npz:
    n_zero_eval, taken before its skimage call): the façade returns a map without zeros that equals the raw network output everywhere
    else; `fill_holes = False` gives the raw output (what the fixtures pin)."""
    from proxytta.model import ExternalModel_Adapt
    g = np.load(os.path.join(golden_dir, 'nlspn_228x304_legacy.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid = [float(x) for x in g['hp']]
    model = ExternalModel_Adapt('nlspn', 0.0, 80.0, max_input_depth=mid, offset=True, device=torch.device('cuda'))
    model._prepare_head('meta_selfsup_seq_1layer_ema')
    model.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict_nlspn().items()})
    params = model.adapt_parameters(mode='meta_bn')
    opt = torch.optim.Adam(params, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    model.model.bind_optimizer(opt)
    model.model.set_hparams(w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos)
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
    model.train()
    model.step(image1, sparse, loss_image=raw)
    model.eval()
    model.model.fill_holes = False
    d_raw = model.forward(image1, sparse, loss_type='adapt_meta_selfsup_seq_ema_reverse')
    model.model.fill_holes = True
    d_fill = model.forward(image1, sparse, loss_type='adapt_meta_selfsup_seq_ema_reverse')
    holes = d_raw == 0
    assert int(holes.sum()) == int(g['s0/n_zero_eval']) >= 1
    assert not bool((d_fill == 0).any())
    assert torch.equal(d_fill[~holes], d_raw[~holes])
    ys, xs = torch.where(holes[0, 0])
    for y, x in zip(ys.tolist(), xs.tolist()):
        nb = d_raw[0, 0, max(y - 2, 0):y + 3, max(x - 2, 0):x + 3]
        nb = nb[nb > 0]
        assert float(nb.min()) * 0.5 <= float(d_fill[0, 0, y, x]) <= float(nb.max()) * 1.5


def test_three_steps_on_one_full_size_frame_match_the_reference(golden_dir):
    """BASELINE config 3 AS STATED: inner_iter = 3 TTA steps on ONE 352x1216 frame (src/tta_main.py:579-636), the scored eval forward after
    each, against the REAL reference run the same way (tests/golden/make_golden_nlspn.py 352x1216 inner3).  Depth maps and loss terms are held
    along the whole sequence (the adapted gradients only at the first step: from the second step on they carry Adam's sign-like first update
    of near-zero gradient entries, see test_second_step_from_oracle_state)."""
    from tests.test_gpu_fullsize import _check_map
    g = np.load(os.path.join(golden_dir, 'nlspn_352x1216_legacy_inner3.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    assert steps == 3 and int(g['same_frame']) == 1
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid = [float(x) for x in g['hp']]
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=mid)
    eng, sd, adapted = make_nlspn(n, h, w, hp, legacy=True)
    names = [str(x) for x in g['adapted_names']]
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
    for s in range(steps):
        p = 's%d/' % s
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        # measured on MI355X (round 4, tools/nlspn_inner3_report.py): training depth 2.6e-6 / 6.3e-6 / 7.7e-6, eval depth 6.3e-6 / 7.7e-6 / 8.8e-6
        # over the three steps, loss terms to 1.5e-6
        _check_map(depth, g, p + 'depth_train', 3e-5)
        np.testing.assert_allclose(info.cpu().numpy(), g[p + 'loss_info'], rtol=2e-5)
        if s == 0:
            gn = np.array([float(eng.grad(k, adapted[k][0]).double().norm()) for k in names])
            np.testing.assert_allclose(gn, g[p + 'grad_norms'], rtol=TOL['default']['gnorm'], atol=1e-6)
        _check_map(eng.forward_eval(image1, sparse), g, p + 'depth_eval', 3e-5)          # (north_star bound on the scored tensor: 1e-3)
    eng.close()


@pytest.mark.parametrize('dtype', ['fp32', 'mixed'])
def test_one_parameter_set_over_a_sequence_of_frames_follows_the_reference(golden_dir, dtype):
    """The reference adapts ONE parameter set over a whole dataset (src/tta_main.py:504-804).  24 different 96x320 frames through the REAL
    reference (tests/golden/make_golden_nlspn.py 96x320 seq24), the scored eval forward after every step: measured on MI355X
    (tools/exp/nlspn_seq_report.py) the scored depth stays at <= 8.3e-5 in the fp32 mode and <= 1.34e-4 in the mixed mode (both level off after
    ~10 frames; round 5's mixed mode with bf16-rounded weights in the data gradients: 4.1e-4 and still rising) -- asserted at 2x at EVERY
    step, far inside the north_star's 1e-3."""
    g = np.load(os.path.join(golden_dir, 'nlspn_96x320_legacy_seq24.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    assert steps == 24 and int(g['same_frame']) == 0
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid = [float(x) for x in g['hp']]
    hp = dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm, w_cos=w_cos, max_input_depth=mid, dtype=dtype)
    eng, sd, adapted = make_nlspn(n, h, w, hp, legacy=True)
    bound, lbound = (1.7e-4, 4.4e-4) if dtype == 'fp32' else (2.7e-4, 1.6e-3)
    pix = lambda t: t.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']]
    # the fixture's own floor: the reference with one adapted weight one ulp off is 5.2e-5 from itself after the 24 frames
    floor = rel_mae(g['alt/s%d/depth_eval_pix' % (steps - 1)], g['s%d/depth_eval_pix' % (steps - 1)])
    assert 2e-5 < floor < 1.2e-4
    for s in range(steps):
        raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(s, h, w, n)]
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        assert rel_mae(pix(depth), g['s%d/depth_train_pix' % s]) < bound, s
        np.testing.assert_allclose(info.cpu().numpy(), g['s%d/loss_info' % s], rtol=lbound)
        assert rel_mae(pix(eng.forward_eval(image1, sparse)), g['s%d/depth_eval_pix' % s]) < bound, s
    eng.close()


# tile edges of the generic engine's kernels (8x32-pixel tiles, 16-channel sub-chunks), the decoder crops of sizes not divisible by 16,
# a one-tile map, odd batches
NLSPN_SWEEP = [(1, 32, 48), (1, 24, 72), (2, 40, 56), (1, 33, 100), (3, 48, 36), (1, 70, 130)]


@pytest.mark.parametrize('shape', NLSPN_SWEEP)
def test_shape_sweep_against_oracle(shape):
    """One full step (default arithmetic, legacy offsets) + the scored eval forward from the ORACLE's post-step parameters at shapes around every
    tile size; gradients by norm (the full-tensor bound needs the exact mode, test_step_matches_golden)."""
    n, h, w = shape
    eng, sd, adapted = make_nlspn(n, h, w, legacy=True)
    from oracle import nlspn_oracle as N
    o = N.NlspnOracle(synth.formula_state_dict_nlspn(), max_input_depth=HP['max_input_depth'], lr=HP['lr'], betas=HP['betas'], eps=HP['eps'],
                      weight_decay=HP['weight_decay'], w_sd=HP['w_sparse_depth'], w_sm=HP['w_smoothness'], w_cos=HP['w_cos'], legacy=True)
    raw, image1, sparse = [torch.from_numpy(x) for x in nlspn_frame(3 + h, h, w, n)]
    r = o.step(image1, sparse, loss_image=raw)
    info, depth = eng.step(image1.cuda(), sparse.cuda(), loss_image=raw.cuda(), want_depth=True)
    assert rel_mae(depth, r['depth']) < TOL['default']['depth'], rel_mae(depth, r['depth'])
    li = r['loss_info']
    np.testing.assert_allclose(info.cpu().numpy(), [li['loss'], li['loss_smooth'], li['loss_sparse_depth'], li['loss_cos']], rtol=2e-3)
    gn = np.array([float(eng.grad(k, adapted[k][0]).double().norm()) for k in o.names])
    gr = np.array([float(r['grads'][k].double().norm()) for k in o.names])
    np.testing.assert_allclose(gn, gr, rtol=3 * TOL['default']['gnorm'], atol=1e-6)
    for k in o.names:
        adapted[k][0].copy_(o.P[k].detach())
    assert rel_mae(eng.forward_eval(image1.cuda(), sparse.cuda()), o.forward_eval(image1, sparse)) < TOL['default']['depth']
    eng.close()


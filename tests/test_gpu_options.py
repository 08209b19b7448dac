"""ptta_set_option (include/ptta.h): every non-default value of a per-handle switch is a slower form of the SAME step.

The defaults are what bench.py times; the other values are what the library falls back to by itself where the default form does not
apply (small maps, N > 16, a statistics exchange) -- so each one is run here, at the size where the default kernels are the fused /
large-map ones (352x1216) and at a size that takes the small-map kernels (64x96), through the plain and the pipelined entry point,
and compared with the default handle bit for bit."""

import pytest
import torch

from proxytta import synth
from tests.util import make_engine, rel_mae

pytestmark = pytest.mark.gpu
HP = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1,
          max_input_depth=80.0)

# options whose other value must not change one bit of the step
BITWISE = [{'stamps': 1}, {'thru': 0}, {'adam_in_wgrad': 0}, {'fuse_first': 0}, {'fuse_first': 2}, {'fuse_head_bwd': 0}, {'mask_bits': 0}, {'aux_stream': 0}, {'graph': 1},
           {'thru': 0, 'fuse_first': 0, 'fuse_head_bwd': 0, 'mask_bits': 0}]


def _run(options, n, h, w, frames, pipelined, dtype='fp32', meta='1layer'):
    eng, sd, ad = make_engine(n, h, w, dtype, HP, meta=meta, options=options)
    for k, v in (options or {}).items():
        assert eng.get_option(k) == v
    out = []
    for i, (im, sp) in enumerate(frames):
        if pipelined:
            info, depth = eng.step(im, sp, want_depth=True, next_frame=frames[min(i + 1, len(frames) - 1)])
            ev = eng.forward_eval_last()
        else:
            info, depth = eng.step(im, sp, want_depth=True)
            ev = eng.forward_eval(im, sp)
        out.append((info.clone(), depth.clone(), ev.clone()))
    torch.cuda.synchronize()
    params = {k: [t.clone() for t in v] for k, v in ad.items()}
    eng.close()
    return out, params


def _frames(n, h, w, k=3):
    return [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(40 + i, h, w, n)] for i in range(k)]


@pytest.mark.parametrize('size', [(1, 352, 1216), (1, 64, 96), (2, 176, 608)])
@pytest.mark.parametrize('pipelined', [False, True])
def test_bitwise_options_equal_the_default_step(size, pipelined):
    n, h, w = size
    frames = _frames(n, h, w)
    base, pbase = _run(None, n, h, w, frames, pipelined)
    for opts in BITWISE:
        got, pgot = _run(opts, n, h, w, frames, pipelined)
        for s, ((i0, d0, e0), (i1, d1, e1)) in enumerate(zip(base, got)):
            assert torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(e0, e1), (opts, s)
        for k in pbase:
            for t0, t1 in zip(pbase[k], pgot[k]):
                assert torch.equal(t0, t1), (opts, k)


@pytest.mark.parametrize('opts', [{'fuse_heads': 0}, {'heads_v2': 0}, {'cos_in_gemm': 0}])
def test_arithmetic_reordering_options_stay_within_bf16x3_error(opts):
    """These three change the ORDER of fp32 operations in the heads (a folded weight, analytic BatchNorm statistics, the cosine gradient
    formed from rounded operands): same depth_train bit for bit (the heads do not feed it), loss terms to 1e-6, the post-update depth to
    the size of one flipped Adam sign (measured <= 4e-5 at 352x1216 over three steps)."""
    n, h, w = 1, 352, 1216
    frames = _frames(n, h, w)
    base, pbase = _run(None, n, h, w, frames, False)
    got, pgot = _run(opts, n, h, w, frames, False)
    assert torch.equal(base[0][1], got[0][1])
    for (i0, d0, e0), (i1, d1, e1) in zip(base, got):
        assert torch.allclose(i0, i1, rtol=2e-5)
        assert rel_mae(d1, d0) < 2e-4 and rel_mae(e1, e0) < 2e-4


def test_options_of_the_2layers_meta_block_are_bitwise():
    n, h, w = 1, 128, 256
    frames = _frames(n, h, w)
    base, pbase = _run(None, n, h, w, frames, False, meta='2layers')
    for opts in ({'thru': 0}, {'mask_bits': 0, 'fuse_first': 0}, {'graph': 1, 'aux_stream': 0}):
        got, pgot = _run(opts, n, h, w, frames, False, meta='2layers')
        for (i0, d0, e0), (i1, d1, e1) in zip(base, got):
            assert torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(e0, e1), opts
        for k in pbase:
            assert torch.equal(pbase[k][0], pgot[k][0]), (opts, k)


def test_option_errors_and_mixed_mode_keys():
    eng, sd, ad = make_engine(1, 64, 96, 'fp32', HP)
    with pytest.raises(RuntimeError, match='unknown key'):
        eng.set_option('no_such_switch', 1)
    with pytest.raises(RuntimeError, match='out of range'):
        eng.set_option('thru', 3)
    assert eng.get_option('graph') == 0 and eng.get_option('fuse_first') == 1
    eng.close()
    # the mixed mode is defined on the default kernels: only the stream / graph switches move
    eng, sd, ad = make_engine(1, 64, 96, 'mixed', HP)
    for k in ('mask_bits', 'fuse_first', 'fuse_heads', 'heads_v2', 'cos_in_gemm', 'fuse_head_bwd'):
        with pytest.raises(RuntimeError, match='mixed mode'):
            eng.set_option(k, 0)
    eng.set_option('thru', 0); eng.set_option('aux_stream', 0); eng.set_option('graph', 1); eng.set_option('stamps', 1)
    im, sp = _frames(1, 64, 96, 1)[0]
    eng.step(im, sp)
    torch.cuda.synchronize()
    eng.close()


def test_option_change_on_a_live_handle_recaptures():
    """Setting an option after steps have been replayed drops the captured graphs; the following steps equal those of a handle created
    with the option."""
    n, h, w = 1, 352, 1216
    frames = _frames(n, h, w, 4)
    ref, pref = _run(None, n, h, w, frames, True)
    eng, sd, ad = make_engine(n, h, w, 'fp32', HP)
    out = []
    for i, (im, sp) in enumerate(frames):
        if i == 2:
            eng.set_option('thru', 0); eng.set_option('mask_bits', 0)
        info, depth = eng.step(im, sp, want_depth=True, next_frame=frames[min(i + 1, 3)])
        out.append((info.clone(), depth.clone()))
    torch.cuda.synchronize()
    for (i0, d0, _), (i1, d1) in zip(ref, out):
        assert torch.equal(i0, i1) and torch.equal(d0, d1)
    eng.close()

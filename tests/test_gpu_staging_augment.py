"""SURVEY.md 8f-3 on the GPU: crop / flip augmentation (ptta_crop_flip behind proxytta.Transforms) against the REAL
reference's outputs and the index-arithmetic oracle; pinned double-buffered host staging against plain copies."""
import os

import numpy as np
import pytest
import torch

from oracle import proxytta_oracle as O
from oracle import transforms_oracle as TO
from proxytta import synth
from proxytta.staging import FrameStager
from proxytta.transforms import Transforms
from tests.util import ONE, make_engine

pytestmark = pytest.mark.gpu
HP = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1,
          max_input_depth=80.0)


def _cases(golden_dir):
    z = np.load(os.path.join(golden_dir, 'transforms_geometric.npz'))
    for name in z['names']:
        p = str(name) + '/'
        cfg = z[p + 'cfg']
        seed, n, H, W = (int(v) for v in cfg[:4])
        crop = [int(v) for v in cfg[4:4 + int(cfg[8])]]
        flips = (['horizontal'] if cfg[9] else []) + (['vertical'] if cfg[10] else []) or ['none']
        yield str(name), z, p, seed, n, H, W, crop, flips, float(z[p + 'prob'])


def test_transforms_reproduce_reference_outputs_bit_exact(golden_dir):
    """Same seed -> same decisions -> the device crop/flip equals the reference class's output bit for bit."""
    for name, z, p, seed, n, H, W, crop, flips, prob in _cases(golden_dir):
        t = Transforms(random_crop_to_shape=crop, random_flip_type=flips)
        torch.manual_seed(seed)
        np.random.seed(seed)
        image = torch.from_numpy(z[p + 'image'].astype(np.float32)).cuda()
        sparse = torch.from_numpy(z[p + 'sparse']).cuda()
        [im, sd], [K] = t.transform(images_arr=[image, sparse], intrinsics_arr=[torch.from_numpy(z[p + 'K']).cuda()],
                                    random_transform_probability=prob)
        np.testing.assert_array_equal(im.cpu().numpy(), z[p + 'image_out'], err_msg=name)
        np.testing.assert_array_equal(sd.cpu().numpy(), z[p + 'sparse_out'], err_msg=name)
        np.testing.assert_array_equal(K.cpu().numpy(), z[p + 'K_out'], err_msg=name)


@pytest.mark.parametrize('shape', [(2, 352, 1216, 320, 1216), (3, 480, 640, 416, 512), (1, 37, 53, 37, 53), (2, 40, 64, 1, 1)])
def test_crop_flip_full_size_against_index_arithmetic(shape):
    """KITTI / VOID sizes, a ragged size with no crop, and the 1x1 crop: bit-exact against the oracle; flipping twice
    and cropping to the full frame are the identity."""
    n, H, W, ch, cw = shape
    rng = np.random.default_rng(H * W)
    x = rng.random((n, 4, H, W), dtype=np.float32)
    d = {'crop': (ch, cw, rng.integers(0, H - ch + 1, n), rng.integers(0, W - cw + 1, n)),
         'hflip': rng.random(n) < 0.5, 'vflip': rng.random(n) < 0.5}
    d['hflip'][0] = True
    t = Transforms(random_crop_to_shape=[ch, cw], random_flip_type=['horizontal', 'vertical'])
    dd = {'crop': (ch, cw, torch.from_numpy(d['crop'][2]).int(), torch.from_numpy(d['crop'][3]).int()),
          'hflip': torch.from_numpy(d['hflip'].astype(np.uint8)), 'vflip': torch.from_numpy(d['vflip'].astype(np.uint8))}
    xg = torch.from_numpy(x).cuda()
    [y] = t.apply([xg], dd)
    np.testing.assert_array_equal(y.cpu().numpy(), TO.apply(x, d))
    # size-independent properties
    flip = {'crop': None, 'hflip': torch.ones(n, dtype=torch.uint8), 'vflip': torch.ones(n, dtype=torch.uint8)}
    [once] = t.apply([xg], flip)
    [twice] = t.apply([once], flip)
    assert torch.equal(twice, xg) and not torch.equal(once, xg)
    ident = {'crop': (H, W, torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32)), 'hflip': None, 'vflip': None}
    assert torch.equal(t.apply([xg], ident)[0], xg)


def test_transforms_refuse_what_is_not_built():
    # (gamma, hue, noise, patch removal, crop-and-pad and resize-and-pad are built: tests/test_gpu_transforms_extra.py)
    with pytest.raises(NotImplementedError):
        Transforms(normalized_image_range=[0, 1])
    with pytest.raises(ValueError):
        Transforms(random_crop_to_shape=[1, 2, 3])


# ---- rotation / resize-and-crop / photometric jitter (torchvision calls in the reference: parity unpinned; oracle = torchvision's
# ---- tensor algorithms on torch's own grid_sample / interpolate) ------------------------------------------------------------------
def _img(n, c, H, W, seed, integer=False):
    rng = np.random.default_rng(seed)
    x = rng.random((n, c, H, W), dtype=np.float32) * 255.0
    return np.floor(x) if integer else x


@pytest.mark.parametrize('shape', [(2, 37, 53), (2, 352, 1216), (1, 480, 640)])
def test_rotate_matches_torchvision_algorithm(shape):
    n, H, W = shape
    x = _img(n, 3, H, W, 11)
    do = torch.tensor([1] + [0] * (n - 1) if n > 1 else [1], dtype=torch.uint8)
    ang = torch.tensor([4.3, -2.0][:n], dtype=torch.float64)
    t = Transforms(random_rotate_max=5)
    d = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': (do, ang), 'resize': None}
    xg = torch.from_numpy(x).cuda()
    for mode in ('nearest', 'bilinear'):
        [y] = t.apply([xg], d, [mode])
        ref = TO.rotate(torch.from_numpy(x), do.tolist(), ang.tolist(), mode == 'bilinear').numpy()
        got = y.cpu().numpy()
        if mode == 'nearest':
            # a pixel whose source coordinate sits within float rounding of a half-integer may pick the neighbour: < 0.1 % of them
            assert (got != ref).mean() < 1e-3
        else:
            assert np.abs(got - ref).max() < 0.05 and np.abs(got - ref).mean() < 1e-3          # 0-255 scale
        if n > 1:
            np.testing.assert_array_equal(got[1], x[1])            # coin said no: untouched
    # angle 0 = identity (nearest: exactly)
    d0 = dict(d, rotate=(torch.ones(n, dtype=torch.uint8), torch.zeros(n, dtype=torch.float64)))
    assert torch.equal(t.apply([xg], d0, ['nearest'])[0], xg)
    assert float((t.apply([xg], d0, ['bilinear'])[0] - xg).abs().max()) < 0.1          # float32 grid coordinates (as torchvision): 1e-4 px x 255
    # 4 degrees forth and back: the interior comes back (bilinear smoothing aside), the corners were rotated out to zero
    ds = dict(d, rotate=(torch.ones(n, dtype=torch.uint8), torch.full((n,), 4.0, dtype=torch.float64)))
    [r1] = t.apply([xg], ds, ['nearest'])
    assert float(r1[:, :, 0, 0].abs().max()) == 0.0


@pytest.mark.parametrize('shape', [(2, 37, 53), (2, 352, 1216), (1, 480, 640)])
def test_resize_and_crop_matches_torchvision_algorithm(shape):
    n, H, W = shape
    x = _img(n, 3, H, W, 12)
    sd = _img(n, 1, H, W, 13)
    do = torch.tensor(([1, 0] * n)[:n], dtype=torch.uint8)
    rh = torch.tensor([int(1.27 * H), int(1.1 * H)][:n], dtype=torch.int32)
    rw = torch.tensor([int(1.41 * W), int(1.2 * W)][:n], dtype=torch.int32)
    sy = torch.tensor([(int(1.27 * H) - H) // 2, 1][:n], dtype=torch.int32)
    sx = torch.tensor([int(1.41 * W) - W, 0][:n], dtype=torch.int32)
    t = Transforms(random_resize_and_crop=[1.0, 1.5])
    d = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': None, 'resize': (do, rh, rw, sy, sx, H, W)}
    im, dep = t.apply([torch.from_numpy(x).cuda(), torch.from_numpy(sd).cuda()], d, ['bilinear', 'nearest'])
    ref_im = TO.resize_and_crop(torch.from_numpy(x), do.tolist(), rh, rw, sy, sx, True).numpy()
    ref_dep = TO.resize_and_crop(torch.from_numpy(sd), do.tolist(), rh, rw, sy, sx, False).numpy()
    assert np.abs(im.cpu().numpy() - ref_im).max() < 0.05 and np.abs(im.cpu().numpy() - ref_im).mean() < 1e-3
    assert (dep.cpu().numpy() != ref_dep).mean() < 1e-3
    # scale 1 with a zero offset = identity, both modes
    one = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': None,
           'resize': (torch.ones(n, dtype=torch.uint8), torch.full((n,), H, dtype=torch.int32), torch.full((n,), W, dtype=torch.int32),
                      torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32), H, W)}
    a, b = t.apply([torch.from_numpy(x).cuda(), torch.from_numpy(sd).cuda()], one, ['bilinear', 'nearest'])
    assert torch.equal(b.cpu(), torch.from_numpy(sd)) and float((a.cpu() - torch.from_numpy(x)).abs().max()) < 1e-3
    # resize_scaling_depth divides the non-image tensors by rw / W
    ts = Transforms(random_resize_and_crop=[1.0, 1.5], resize_scaling_depth=True)
    _, dep2 = ts.apply([torch.from_numpy(x).cuda(), torch.from_numpy(sd).cuda()], d, ['bilinear', 'nearest'])
    np.testing.assert_allclose(dep2.cpu().numpy()[0], dep.cpu().numpy()[0] / (float(rw[0]) / W), rtol=1e-6)


@pytest.mark.parametrize('shape', [(2, 37, 53), (2, 352, 1216)])
def test_photometric_matches_torchvision_algorithm(shape):
    n, H, W = shape
    x = _img(n, 3, H, W, 14)                                           # fractional values: the uint8 cast truncates
    on = torch.ones(n, dtype=torch.uint8)
    off = torch.zeros(n, dtype=torch.uint8)
    t = Transforms(random_brightness=[0.6, 1.4], random_contrast=[0.6, 1.4], random_saturation=[0.6, 1.4])
    fb, fc, fs = torch.tensor([1.31, 0.7][:n]), torch.tensor([0.64, 1.38][:n]), torch.tensor([1.22, 0.61][:n])
    for bb, cc, ss in ((on, on, on), (on, off, off), (off, on, off), (off, off, on), (off, off, off)):
        d = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': None, 'resize': None, 'brightness': (bb, fb), 'contrast': (cc, fc), 'saturation': (ss, fs)}
        [y] = t.apply([torch.from_numpy(x).cuda()], d)
        ref = TO.photometric(torch.from_numpy(x), (bb, fb), (cc, fc), (ss, fs)).numpy()
        got = y.cpu().numpy()
        assert np.abs(got - ref).max() <= 1.0 and (got != ref).mean() < 2e-3       # a product within float rounding of an integer may truncate one level lower
        assert np.array_equal(got, np.floor(got)) and got.min() >= 0 and got.max() <= 255
    # factor 1 everywhere = the uint8 cast alone
    one = torch.ones(n)
    d = {'crop': None, 'hflip': None, 'vflip': None, 'rotate': None, 'resize': None, 'brightness': (on, one), 'contrast': (on, one), 'saturation': (on, one)}
    np.testing.assert_array_equal(t.apply([torch.from_numpy(x).cuda()], d)[0].cpu().numpy(), np.floor(x))


def test_adapt_script_flags_construct_and_run():
    """Transforms(**flags of bash/adapt/adapt_msgchn_vkitti.sh:34-41) as src/tta_main.py:446-462 builds its two objects, one call
    each as :595-605: same seed -> the oracle's draw-by-draw restatement gives the same tensors; intrinsics follow :447-451, :493-497."""
    n, H, W = 2, 96, 160
    geo = Transforms(random_crop_to_shape=[-1, -1], random_flip_type=['horizontal'], random_rotate_max=5,
                     random_crop_and_pad=[-1, -1], random_resize_and_pad=[-1, -1], random_resize_and_crop=[1.0, 1.5])
    pho = Transforms(normalized_image_range=None, random_brightness=[0.6, 1.4], random_contrast=[0.6, 1.4], random_gamma=[-1, -1],
                     random_hue=[-1, -1], random_saturation=[0.6, 1.4], random_noise_type='none', random_noise_spread=-1)
    image = torch.from_numpy(_img(n, 3, H, W, 21, integer=True)).cuda()
    sparse = torch.from_numpy((_img(n, 1, H, W, 22) * (np.random.default_rng(5).random((n, 1, H, W)) < 0.05)).astype(np.float32)).cuda()
    K = torch.tensor([[[100., 0., 80.], [0., 110., 48.], [0., 0., 1.]]] * n).cuda()
    seen = 0
    for seed in range(6):
        torch.manual_seed(seed); np.random.seed(seed)
        [im, sd], [K2] = geo.transform(images_arr=[image, sparse], intrinsics_arr=[K], interpolation_modes=[2, 0], random_transform_probability=1.0)
        d = geo.last_draw
        [im1] = pho.transform(images_arr=[im], random_transform_probability=1.0)
        dp = pho.last_draw
        assert im.shape == image.shape and sd.shape == sparse.shape and im1.shape == image.shape
        # the oracle, fed the same decisions
        x, s_ = image.cpu(), sparse.cpu()
        hf = d['hflip'].bool().numpy()
        dd = {'crop': None, 'hflip': hf, 'vflip': np.zeros(n, bool)}
        x, s_ = torch.from_numpy(TO.apply(x.numpy(), dd)), torch.from_numpy(TO.apply(s_.numpy(), dd))
        x, s_ = TO.rotate(x, d['rotate'][0].tolist(), d['rotate'][1].tolist(), True), TO.rotate(s_, d['rotate'][0].tolist(), d['rotate'][1].tolist(), False)
        do, rh, rw, sy, sx, nh, nw = d['resize']
        x, s_ = TO.resize_and_crop(x, do.tolist(), rh, rw, sy, sx, True), TO.resize_and_crop(s_, do.tolist(), rh, rw, sy, sx, False)
        assert np.abs(im.cpu().numpy() - x.numpy()).mean() < 1e-3 and (sd.cpu().numpy() != s_.numpy()).mean() < 2e-3
        ref1 = TO.photometric(im.cpu(), dp['brightness'], dp['contrast'], dp['saturation'])
        assert (im1.cpu() - ref1).abs().max() <= 1.0
        Kr = K.cpu().clone()
        for b in range(n):                                              # every sample, whatever its coin (as the reference)
            Kr[b, 0, 0] *= float(rw[b]) / nw; Kr[b, 0, 2] = Kr[b, 0, 2] * (float(rw[b]) / nw) - float(rw[b] - nw)
            Kr[b, 1, 1] *= float(rh[b]) / nh; Kr[b, 1, 2] = Kr[b, 1, 2] * (float(rh[b]) / nh) - float(rh[b] - nh)
        np.testing.assert_allclose(K2.cpu().numpy(), Kr.numpy(), rtol=1e-6)
        seen += int(d['rotate'][0].sum()) + int(do.sum())
    assert seen > 0


def test_flipped_frame_gives_flipped_free_step():
    """Augmented tensors feed the engine like any other frame: a step on the flipped+cropped frame equals the oracle's step
    on the same (oracle-augmented) frame."""
    n, H, W, ch, cw = 1, 48, 96, 32, 64
    image, sparse = synth.synthetic_frame(3, H, W, n)
    d = {'crop': (ch, cw, np.array([9]), np.array([17])), 'hflip': np.array([True]), 'vflip': np.array([False])}
    dd = {'crop': (ch, cw, torch.tensor([9], dtype=torch.int32), torch.tensor([17], dtype=torch.int32)),
          'hflip': torch.tensor([1], dtype=torch.uint8), 'vflip': None}
    t = Transforms(random_crop_to_shape=[ch, cw], random_flip_type=['horizontal'])
    im, sd = t.apply([torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda()], dd)
    eng, _, _ = make_engine(n, ch, cw, 'fp32', HP)
    info, depth = eng.step(im, sd, want_depth=True)
    o = O.MsgChnOracle(synth.formula_state_dict(ONE), ONE, max_input_depth=80.0, lr=1e-3, w_sd=1.0, w_sm=2.0, w_cos=0.1)
    r = o.step(torch.from_numpy(TO.apply(image, d)), torch.from_numpy(TO.apply(sparse, d)))
    assert float((depth.cpu() - r['depth']).abs().mean() / r['depth'].abs().mean()) < 1e-4


# ---- host staging ---------------------------------------------------------------------------------------
def test_stager_delivers_frames_in_order_and_reuses_slots():
    n, h, w = 2, 40, 64
    st = FrameStager(n, h, w)
    frames = [synth.synthetic_frame(k, h, w, n) for k in range(7)]
    st.submit(*frames[0])
    for k in range(7):
        if k + 1 < 7:
            st.submit(*frames[k + 1])                      # frame k+1 travels while frame k is consumed
        im, sd = st.acquire()
        got = (im.clone(), sd.clone())
        st.release()
        np.testing.assert_array_equal(got[0].cpu().numpy(), frames[k][0])
        np.testing.assert_array_equal(got[1].cpu().numpy(), frames[k][1])
    assert st.in_flight() == 0
    with pytest.raises(RuntimeError):
        st.acquire()
    st.submit(*frames[0]); st.submit(*frames[1]); st.submit(*frames[2])
    with pytest.raises(RuntimeError):
        st.submit(*frames[3])                              # all slots in flight


def test_staged_sequence_equals_resident_sequence():
    """Five steps fed through the stager (copy stream, events) leave exactly the parameters the same five steps leave
    when the frames are resident -- the overlap changes no result."""
    n, h, w = 1, 64, 96
    frames = [synth.synthetic_frame(k, h, w, n) for k in range(5)]

    def run(staged):
        eng, _, adapted = make_engine(n, h, w, 'fp32', HP)
        infos = []
        if staged:
            st = FrameStager(n, h, w)
            st.submit(*frames[0])
            for k in range(5):
                if k + 1 < 5:
                    st.submit(*frames[k + 1])
                im, sd = st.acquire()
                # 'pipelined': the next slot (its copy still in flight) is announced, ordered on the library's prefix stream
                nxt = st.peek_next(eng.prefix_stream()) if staged == 'pipelined' else None
                infos.append(eng.step(im, sd, next_frame=nxt)[0])
                st.release()
        else:
            for k in range(5):
                infos.append(eng.step(torch.from_numpy(frames[k][0]).cuda(), torch.from_numpy(frames[k][1]).cuda())[0])
        torch.cuda.synchronize()
        return torch.stack(infos).cpu(), {k: v[0].detach().cpu().clone() for k, v in adapted.items()}
    ia, pa = run(False)
    for mode in (True, 'pipelined'):
        ib, pb = run(mode)
        assert torch.equal(ia, ib), mode
        for k in pa:
            assert torch.equal(pa[k], pb[k]), (mode, k)


@pytest.mark.parametrize('graph', [0, 1])
@pytest.mark.parametrize('meta,n,size', [('1layer', 1, (64, 128)), ('2layers', 1, (64, 128)), ('1layer', 2, (64, 128)),
                                         ('1layer', 1, (352, 1216)), ('2layers', 1, (352, 1216))])
def test_pipelined_steps_equal_plain_steps(meta, n, size, graph):
    """ptta_step_pipelined: the parameter-independent prefix of frame k+1 (sparse-depth pooling, frozen RGB encoder, depth-only head of the
    stage-1 encoder) runs on its own stream beside the step of frame k.  Same parameters, losses and depths as ptta_step, call by call --
    also with an eval forward between two calls, an unannounced frame (prefix recomputed in line) and a plain step in the middle."""
    # (352x1216: the size bench.py times -- persistent 480-block launches, the fused first-two-layer and head-backward kernels and
    # the prefix stream beside full-chip kernels only exist together there)
    from tests.util import make_engine
    h, w = size
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(20 + i, h, w, n)] for i in range(7)]
    out = {}
    for mode in ('plain', 'pipelined'):
        # graph = 0 (default): both forms enqueue their kernels directly; 1: both replay captured hipGraphs
        eng, sd, adapted = make_engine(n, h, w, 'fp32', dict(lr=1e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0), meta=meta,
                                       options={'graph': graph})
        rec = []
        for i in range(6):
            nxt = frames[i + 1] if mode == 'pipelined' else None
            if i == 3:
                nxt = None if mode == 'plain' else frames[6]               # announce the WRONG frame once: step 4 recomputes its prefix
            if i == 5 and mode == 'pipelined':
                nxt = None                                                   # a plain ptta_step in the middle of the stream
            if i == 0 and mode == 'pipelined':                               # two steps on the same frame (inner_iter 2): the prefix is reused
                eng.step(frames[0][0], frames[0][1], next_frame=frames[0])
            elif i == 0:
                eng.step(frames[0][0], frames[0][1])
            # steps 2 and 4 with an explicit validity map and a separate loss image (other graph keys, staged per buffer set)
            extra = dict(validity=(frames[i][1] > 0).float(), loss_image=frames[i][0] * 0.5 + 0.1) if i in (2, 4) else {}
            info, depth = eng.step(frames[i][0], frames[i][1], want_depth=True, next_frame=nxt, **extra)
            evl = eng.forward_eval_last() if (i == 2 and mode == 'pipelined') else None   # the scored forward from the adapted frame's own prefix
            ev = eng.forward_eval(frames[i][0], frames[i][1]) if i in (1, 2) else None     # the scored forward between two steps
            assert evl is None or torch.equal(evl, ev)
            rec.append((info.clone(), depth.clone(), None if ev is None else ev.clone(), {k: v[0].clone() for k, v in adapted.items()}))
        torch.cuda.synchronize()
        out[mode] = rec
        eng.close()
    for a, b in zip(out['plain'], out['pipelined']):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert (a[2] is None) == (b[2] is None) and (a[2] is None or torch.equal(a[2], b[2]))
        for k in a[3]:
            assert torch.equal(a[3][k], b[3][k]), k


def test_refilled_buffer_gets_a_fresh_prefix():
    """The prepared prefix is recognised by a frame TOKEN, not by pointer identity (include/ptta.h ptta_step_pipelined): a caller that
    refills the SAME device buffers with another frame between the announcing and the consuming call -- what a fixed staging slot or a
    caching allocator does -- gets that frame's own prefix, not the stale one.  Also at the C-ABI with explicit tokens."""
    from tests.util import make_engine
    n, h, w = 1, 64, 128
    hp = dict(lr=1e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
    A, B, C = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(40 + i, h, w, n)] for i in range(3)]
    out = {}
    for mode in ('plain', 'refill', 'refill_tokens', 'stale_token'):
        eng, sd, adapted = make_engine(n, h, w, 'fp32', hp)
        slot = [B[0].clone(), B[1].clone()]
        if mode == 'plain':
            eng.step(*A)
            info, depth = eng.step(*C, want_depth=True)
        else:
            tok = dict(frame_token=11, next_token=12) if mode != 'refill' else {}
            eng.step(*A, next_frame=slot, **tok)                   # announces B (in `slot`)
            torch.cuda.synchronize()                               # B's prefix has been computed from the handle's copy of it
            slot[0].copy_(C[0]); slot[1].copy_(C[1])               # the same buffers now hold C
            tok = dict(refill={}, refill_tokens=dict(frame_token=13, next_token=14), stale_token=dict(frame_token=12, next_token=14))[mode]
            info, depth = eng.step(*slot, want_depth=True, next_frame=A, **tok)
        torch.cuda.synchronize()
        out[mode] = (info.clone(), depth.clone(), {k: v[0].clone() for k, v in adapted.items()})
        eng.close()
    for mode in ('refill', 'refill_tokens'):
        assert torch.equal(out['plain'][0], out[mode][0]) and torch.equal(out['plain'][1], out[mode][1]), mode
        for k in out['plain'][2]:
            assert torch.equal(out['plain'][2][k], out[mode][2][k]), (mode, k)
    # the contract, seen from the other side: a caller that passes the OLD token for new content asked for the prefix of B and gets it
    assert not torch.equal(out['plain'][1], out['stale_token'][1])


@pytest.mark.parametrize('why', ['naive', 'profiling'])
def test_eval_last_after_a_step_that_fell_back_to_the_plain_path(why):
    """ptta_step_pipelined runs as a plain ptta_step with the validation kernels (PTTA_CONV_IMPL=naive), under profiling, with SyncBatchNorm / gradient
    exchange or padded sizes; ptta_forward_eval_last must then still return the scored forward of the frame just adapted
    (round-3 advisor finding: it raised), and ExternalModel_Adapt.adapt(..., next_frame=...) must run."""
    import os
    from tests.util import make_engine
    n, h, w = 1, 64, 96
    hp = dict(lr=1e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(60 + i, h, w, n)] for i in range(3)]
    eng, sd, adapted = make_engine(n, h, w, 'fp32', hp, impl='naive' if why == 'naive' else None)
    if why == 'profiling':
        eng.profile(True)
    for i in range(2):
        eng.step(*frames[i], next_frame=frames[i + 1])
        a = eng.forward_eval_last()
        b = eng.forward_eval(*frames[i])
        assert torch.equal(a, b)
    eng.close()


def test_eval_forward_between_two_steps_on_one_frame_invalidates_the_kept_prefix():
    """inner_iter > 1: the prefix of a frame is kept for its next step (next_token == frame_token).  A full eval forward of ANOTHER frame in
    between overwrites that buffer set (round-3 advisor finding: the following step silently reused the wrong prefix)."""
    from tests.util import make_engine
    n, h, w = 1, 64, 96
    hp = dict(lr=1e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
    f0, f1 = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(80 + i, h, w, n)] for i in range(2)]
    out = {}
    for mode in ('plain', 'pipelined'):
        eng, sd, adapted = make_engine(n, h, w, 'fp32', hp)
        nxt = dict(next_frame=f0) if mode == 'pipelined' else {}
        eng.step(*f0, **nxt)
        ev = eng.forward_eval(*f1)
        info, depth = eng.step(*f0, want_depth=True, **nxt)
        torch.cuda.synchronize()
        out[mode] = (ev.clone(), info.clone(), depth.clone())
        eng.close()
    for a, b in zip(out['plain'], out['pipelined']):
        assert torch.equal(a, b)

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
    with pytest.raises(NotImplementedError):
        Transforms(random_brightness=[0.5, 1.5])
    with pytest.raises(NotImplementedError):
        Transforms(normalized_image_range=[0, 1])
    with pytest.raises(NotImplementedError):
        Transforms(random_rotate_max=10)
    with pytest.raises(ValueError):
        Transforms(random_crop_to_shape=[1, 2, 3])


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
                infos.append(eng.step(im, sd)[0])
                st.release()
        else:
            for k in range(5):
                infos.append(eng.step(torch.from_numpy(frames[k][0]).cuda(), torch.from_numpy(frames[k][1]).cuda())[0])
        torch.cuda.synchronize()
        return torch.stack(infos).cpu(), {k: v[0].detach().cpu().clone() for k, v in adapted.items()}
    ia, pa = run(False)
    ib, pb = run(True)
    assert torch.equal(ia, ib)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k

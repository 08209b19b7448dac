"""CPU: the host-side draws of proxytta.Transforms (gamma, hue, noise, crop-and-pad, resize-and-pad, patch removal) + the oracle's restatements
against tests/golden/transforms_extra.npz -- outputs of the REAL reference class (tests/golden/make_golden_transforms_extra.py).  Pins every draw
(order and arithmetic), add_noise, remove_random_patches and the crop / pad index arithmetic; the torchvision functionals inside the fixture are
the oracle's own restatement (parity unpinned for those)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import transforms_oracle as TO
from proxytta.transforms import Transforms

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'transforms_extra.npz')


def cases():
    z = np.load(GOLD)
    for name in z['names']:
        yield str(name)


def load_case(name):
    z = np.load(GOLD)
    p = name + '/'
    cfg = json.loads(str(z[p + 'cfg']))
    arrs = [torch.from_numpy(z[p + 'image'].astype(np.float32))]
    want = [z[p + 'image_out']]
    if p + 'sparse_out' in z.files:
        arrs.append(torch.from_numpy(z[p + 'sparse']))
        want.append(z[p + 'sparse_out'])
    return cfg, arrs, want


def seed_all(seed):
    torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)


@pytest.mark.parametrize('name', list(cases()))
def test_draws_and_oracle_reproduce_the_reference(name):
    cfg, arrs, want = load_case(name)
    t = Transforms(**cfg['kw'])
    seed_all(cfg['seed'])
    d = t.draw(cfg['n'], cfg['H'], cfg['W'], cfg['prob'], channels=[a.shape[1] for a in arrs])
    got = TO.apply_draw(t, d, arrs, cfg['pmodes'], cfg['imodes'])
    for g, w in zip(got, want):
        assert tuple(g.shape) == tuple(w.shape)
        np.testing.assert_array_equal(g.numpy(), w)


def test_padding_modes_of_the_restated_pad():
    x = torch.arange(2 * 4 * 5, dtype=torch.float32).reshape(2, 4, 5)
    assert TO.tv_pad(x, (2, 1, 1, 2), 0, 'reflect')[0, 0].tolist() == [7.0, 6.0, 5.0, 6.0, 7.0, 8.0, 9.0, 8.0]
    assert TO.tv_pad(x, (2, 1, 1, 2), 0, 'symmetric')[0, 0].tolist() == [1.0, 0.0, 0.0, 1.0, 2.0, 3.0, 4.0, 4.0]
    assert TO.tv_pad(x, (2, 1, 1, 2), 0, 'edge')[0, -1].tolist() == [15.0, 15.0, 15.0, 16.0, 17.0, 18.0, 19.0, 19.0]
    assert TO.tv_pad(x, (2, 1, 1, 2), 7, 'constant')[0, 0].tolist() == [7.0] * 8

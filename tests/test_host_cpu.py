"""CPU-only checks: the C-ABI library loads and exports every declared symbol (no compute calls),
include/ptta.h and the ctypes table agree, the host mirror keeps the reference's surface."""
import numpy as np
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from proxytta import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libptta_hip.so not built (run __graft_entry__.build())')
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'ptta.h')).read()
    declared = set(re.findall(r'\b(ptta_[a-z0-9_]+)\s*\(', header))
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name
    abi = int(re.search(r'#define PTTA_ABI_VERSION (\d+)', header).group(1))
    assert lib.ptta_version() == abi == _lib.PTTA_ABI_VERSION


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from proxytta.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(1, 32, 48)


def test_mirror_surface_and_state_dict_keys():
    from proxytta import synth
    from proxytta.model import ExternalModel_Adapt, MsgChnModel_Adapt
    for name in ('forward', 'compute_loss', '_prepare_head', 'adapt_parameters', 'parameters', 'train', 'eval', 'to',
                 'data_parallel', 'distributed_data_parallel', 'restore_model', 'save_model', 'convert_syncbn',
                 'step', 'adapt'):
        assert hasattr(ExternalModel_Adapt, name), name
    m = MsgChnModel_Adapt(device=torch.device('cpu'))
    m._prepare_head('meta_selfsup_seq_1layer_ema')
    assert sorted(m.model.state_dict().keys()) == sorted(k for k, _ in synth.msg_chn_keys())
    params = m.adapt_parameters('meta')
    assert [tuple(p.shape) for p in params] == [(32, 32, 3, 3), (32,)]
    with pytest.raises(ValueError):
        ExternalModel_Adapt('unknown', 0, 1, device=torch.device('cpu'))


def test_checkpoint_roundtrip(tmp_path):
    from proxytta.model import MsgChnModel_Adapt
    m = MsgChnModel_Adapt(device=torch.device('cpu'))
    m._prepare_head('meta_selfsup_seq_1layer_ema')
    opt = torch.optim.Adam(m.adapt_parameters('meta'), lr=1e-3)
    path = str(tmp_path / 'ckpt.pth')
    m.save_model(path, 7, opt)
    ck = torch.load(path)
    assert set(ck) == {'net', 'optimizer', 'train_step'} and ck['train_step'] == 7    # msg_chn_model_adapt.py:513-545
    m2 = MsgChnModel_Adapt(device=torch.device('cpu'))
    m2._prepare_head('meta_selfsup_seq_1layer_ema')
    _, step = m2.restore_model(path)
    assert step == 7
    for k, v in m.model.state_dict().items():
        assert torch.equal(v, m2.model.state_dict()[k])


def test_nlspn_key_table_and_adapted_list():
    """NLSPN state_dict key table (303 keys / 31,533,196 values, checked against the reference when the golden
    vectors were generated) and the 'meta_bn' adapted list derived from it (88 tensors / 40,048 values)."""
    from proxytta import synth
    from proxytta.nlspn import nlspn_adapted_names
    keys = synth.nlspn_keys()
    assert len(keys) == 303
    import numpy as np
    assert sum(int(np.prod(s)) if len(s) else 1 for _, s in keys) == 31533196
    names = nlspn_adapted_names([k for k, _ in keys])
    shapes = dict(keys)
    assert len(names) == 88 and sum(int(np.prod(shapes[k])) for k in names) == 40048
    assert names[:2] == ['conv1_rgb_meta.weight', 'conv1_rgb_meta.bias'] and names[2] == 'conv2.0.bn1.weight'


def test_costdcnet_key_table_and_adapted_list():
    """371 state_dict keys (the table is checked against the real reference in tests/golden/make_golden_costdcnet.py) and the
    32-tensor / 5,200-value adapt_parameters('meta_bn') list in the reference's order."""
    from proxytta import synth
    from proxytta.costdcnet import costdcnet_adapted_names
    keys = synth.costdcnet_keys()
    assert len(keys) == 371
    names = costdcnet_adapted_names([k for k, _ in keys])
    shapes = dict(keys)
    assert len(names) == 32 and names[:2] == ['conv1_rgb_meta.weight', 'conv1_rgb_meta.bias'] and names[2] == 'enc2d.norm1.weight'
    assert sum(int(np.prod(shapes[k])) for k in names) == 5200
    sd = synth.formula_state_dict_costdcnet()
    assert np.array_equal(sd['enc2d.layer2.0.downsample.1.weight'], sd['enc2d.layer2.0.norm3.weight'])      # one module, two names


def test_biharmonic_hole_filling_properties():
    """proxytta/inpaint.py restates skimage.restoration.inpaint_biharmonic (absent here: parity unpinned) for the NLSPN adapter's
    eval-time hole filling (src/nlspn_model_adapt.py:124-127, src/data_utils.py:327-354).  Size-independent properties: the interior
    stencil annihilates cubic surfaces (holes in one are restored to rounding), known pixels are never touched, a map without holes
    comes back as the same object's values, results stay inside the known range, independent regions do not interact."""
    from proxytta.inpaint import inpaint_biharmonic, inpainting
    H, W = 48, 64
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    f = (2 + 0.03 * x + 0.02 * y + 0.001 * x * y + 1e-4 * x ** 2 * y - 2e-5 * y ** 3).astype(np.float32)
    d = np.stack([f, f])[:, None].copy()
    d[0, 0, 10:13, 20:24] = 0; d[0, 0, 30, 40] = 0; d[1, 0, 5, 5] = 0; d[1, 0, 0, 0] = 0; d[1, 0, 47, 5:8] = 0
    holes = d == 0
    out = inpainting(d.copy())
    assert not (out == 0).any()
    np.testing.assert_array_equal(out[~holes], d[~holes])
    np.testing.assert_allclose(out[0, 0, 10:13, 20:24], f[10:13, 20:24], rtol=1e-6)       # interior: exact for a cubic
    np.testing.assert_allclose(out[0, 0, 30, 40], f[30, 40], rtol=1e-6)
    np.testing.assert_allclose(out[1, 0, 5, 5], f[5, 5], rtol=1e-6)
    assert abs(out[1, 0, 0, 0] - f[0, 0]) < 0.05 and np.abs(out[1, 0, 47, 5:8] - f[47, 5:8]).max() < 0.05      # borders: reflect stencil
    assert out.min() >= f.min() and out.max() <= f.max()
    clean = np.stack([f])[:, None].copy()
    assert inpainting(clean) is clean
    # two regions solved independently = both solved together when they are far apart
    m = np.zeros((H, W), bool); m[10:12, 10:12] = True; m[30:33, 50] = True
    a = inpaint_biharmonic(np.where(m, 0, f), m)
    m1 = np.zeros_like(m); m1[10:12, 10:12] = True
    b = inpaint_biharmonic(np.where(m1, 0, f), m1)
    np.testing.assert_array_equal(a[10:12, 10:12], b[10:12, 10:12])
    with pytest.raises(ValueError):
        inpaint_biharmonic(f, m[:-1])

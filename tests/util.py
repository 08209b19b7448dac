"""Shared helpers for the GPU parity tests."""
import os

import numpy as np
import torch

from proxytta import synth

ONE = 'meta_selfsup_seq_1layer_ema'


def rel_mae(a, b, floor=1e-4):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.abs(a - b).mean() / max(np.abs(b).mean(), floor))


TWO = 'meta_selfsup_seq_2layers_ema'


def make_engine(n, h, w, dtype='fp32', hp=None, gain=1.0, impl=None, meta='1layer', head_bias=0.0, options=None, keep=()):
    """Engine with formula weights; returns (engine, state dict on device, adapted dict)."""
    from proxytta.engine import Engine, adapted_names
    os.environ.pop('PTTA_CONV_IMPL', None)
    os.environ.pop('PTTA_ARITH', None)
    if impl == 'naive':
        os.environ['PTTA_CONV_IMPL'] = 'naive'
        os.environ['PTTA_ARITH'] = 'exact'
    elif impl == 'exact':
        os.environ['PTTA_ARITH'] = 'exact'
    hp = dict(hp or {})
    eng = Engine(n, h, w, dtype=dtype, meta=meta, options=options, keep=keep, **hp)
    os.environ.pop('PTTA_CONV_IMPL', None)
    os.environ.pop('PTTA_ARITH', None)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(TWO if meta == '2layers' else ONE, gain, head_bias).items()}
    eng.load_state_dict(sd)
    adapted = {}
    for name in adapted_names(meta):
        p = sd[name]
        adapted[name] = (p, torch.zeros_like(p), torch.zeros_like(p))
        eng.bind_adapted(name, *adapted[name])
    return eng, sd, adapted


def golden_hp(g):
    lr, b1, b2, eps, wd, w_sd, w_sm, w_cos, mid, gain = [float(x) for x in g['hp']]
    return dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, w_sparse_depth=w_sd, w_smoothness=w_sm,
                w_cos=w_cos, max_input_depth=mid), gain

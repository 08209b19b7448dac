"""Golden vectors of the reference's STAGE-2 head trainer, from the REAL reference on CPU.

Drives exactly the calls of src/head_main.py:259-276 (`_prepare_head(prepare_mode)`, `prepare_parameters('head_selfsup_ema')`,
Adam over the returned head parameters) and :464-480 (`model.train(prepare=True)`, `model.forward(loss_type=...)`,
`compute_loss(loss_type='prepare')`, zero_grad / backward / step) with the two loss types that reach `_update_head`
(network_exp_msg_chn_adapt.py:678-699, EMA :701-703):
    head_selfsup_seq_ema_reverse : emb = pred(proj(feat_zero).detach()), ref = proj(feat).detach()   -> only `pred` trains
    head_selfsup_seq_ema         : emb = pred(proj(feat)),               ref = proj(feat_zero).detach() -> `proj` and `pred` train
Same import shims as make_golden.py.  Weights come from proxytta.synth formulas (loaded AFTER prepare_parameters, which
re-creates the heads, head_main.py:268); `proj_t` starts as a perturbed copy of `proj` so the EMA is visible.
Usage:  python tests/golden/make_golden_head.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from make_golden import synth  # noqa: E402

PREPARE_MODE = 'meta_selfsup_seq_1layer_ema'
HEAD_HP = dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
CASES = [('head_reverse_32x48_n2', 'head_selfsup_seq_ema_reverse', 32, 48, 2, 3),
         ('head_forward_32x48_n2', 'head_selfsup_seq_ema', 32, 48, 2, 3),
         ('head_reverse_64x96', 'head_selfsup_seq_ema_reverse', 64, 96, 1, 2)]


def perturbed_target(sd):
    """proj_t = proj * (1 + 0.05 * sin(index)) for the six parameter tensors (buffers stay copies)."""
    out = {}
    for k, v in sd.items():
        if k.startswith('proj_t.') and not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
            a = np.asarray(v, np.float32)
            out[k] = (a * (1.0 + 0.05 * np.sin(np.arange(a.size, dtype=np.float64)).reshape(a.shape))).astype(np.float32)
    return out


def put(out, key, a):
    """Small tensors whole; 512 x 512 (and 512 x 32) matrices as 24 sampled rows + float64 checksums."""
    a = np.asarray(a)
    if a.size <= 4096:
        out[key] = a.copy()
        return
    idx = np.linspace(0, a.shape[0] - 1, 24).astype(np.int64)
    out[key + '#rows'] = a[idx].copy()
    out[key + '#sum'] = np.array([a.sum(dtype=np.float64), np.abs(a).sum(dtype=np.float64)])


def run_case(ema, name, loss_type, h, w, n, steps):
    model = ema.ExternalModel_Adapt('msg_chn', 0.0, 80.0, max_input_depth=80.0, device=torch.device('cpu'))
    model._prepare_head(PREPARE_MODE)
    net = model.model.model
    head_params = model.prepare_parameters('head_selfsup_ema')
    sd = synth.formula_state_dict(PREPARE_MODE, 1.0)
    sd.update(perturbed_target(sd))
    assert list(sd.keys()) == list(net.state_dict().keys())
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    names = [k for k, p in net.named_parameters() if any(p is q for q in head_params)]
    assert len(names) == 12 and all(('proj' in k or 'pred' in k) and '_t' not in k for k in names), names
    opt = torch.optim.Adam(head_params, **HEAD_HP)
    out = {'meta': np.array([h, w, n, steps], dtype=np.int64), 'loss_type': np.array(loss_type),
           'hp': np.array([HEAD_HP['lr'], HEAD_HP['betas'][0], HEAD_HP['betas'][1], HEAD_HP['eps'], HEAD_HP['weight_decay'], 0.999]),
           'head_names': np.array(names)}
    for s in range(steps):
        image_np, sparse_np = synth.synthetic_frame(s, h, w, n)
        image, sparse = torch.from_numpy(image_np), torch.from_numpy(sparse_np)
        model.train(prepare=True)
        output_depth, embedding, reference = model.forward(image=image, sparse_depth=sparse, loss_type=loss_type)
        assert output_depth is None
        loss, info = model.compute_loss(input_rgb=image, output_depth=output_depth, validity_map=None, ground_truth=None,
                                        embedding=embedding, reference=reference, loss_type='prepare')
        opt.zero_grad()
        loss.backward()
        p = 's%d/' % s
        out[p + 'loss'] = np.array(float(loss))
        idx, out[p + 'emb_rows'] = MG.sample_rows(embedding.detach().numpy())
        _, out[p + 'ref_rows'] = MG.sample_rows(reference.detach().numpy())
        out[p + 'row_idx'] = idx
        named = dict(net.named_parameters())
        for k in names:
            g = named[k].grad
            out[p + 'has_grad/' + k] = np.array(g is not None)
            if g is not None:
                put(out, p + 'grad/' + k, g.detach().numpy())
        opt.step()
        state = net.state_dict()
        for k in state:
            if k.startswith(('proj', 'pred')):
                put(out, p + 'after/' + k, state[k].detach().numpy())
        print(name, s, float(loss), [k for k in names if named[k].grad is None])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    ema, _ = MG.import_reference()
    for c in CASES:
        if len(sys.argv) > 1 and c[0] not in sys.argv[1:]:
            continue
        run_case(ema, *c)

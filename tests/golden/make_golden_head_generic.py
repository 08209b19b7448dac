"""Golden vectors of the reference's STAGE-2 head trainer for the NLSPN and CostDCNet backbones, from the REAL reference on CPU.

Drives the calls of src/head_main.py:259-276 (`_prepare_head(prepare_mode)`, `prepare_parameters('head_selfsup_ema')`, Adam over the returned
head parameters) and :464-480 (`model.train(prepare=True)`, `model.forward(loss_type=...)`, `compute_loss(loss_type='prepare')`, zero_grad /
backward / step) with both loss types that reach `_update_head`:
  NLSPN     (external_src/NLSPN/src/model/nlspnmodel_adapt.py:1014-1060, EMA :1314-1316)
  CostDCNet (external_src/costdcnet/CostDCNet_adapt.py:258-303, EMA :426-428)
      head_selfsup_seq_ema         : emb = pred(proj(fe(real).detach())),  ref = proj_t(fe(zero image)).detach()
      head_selfsup_seq_ema_reverse : emb = pred(proj(fe(zero image).detach())), ref = proj_t(fe(real)).detach()
  -- unlike MSG_CHN the reference branch goes through the EMA TARGET `proj_t`, and `proj` trains in both directions.
`train(prepare=True)` (src/nlspn_model_adapt.py:360-368, src/costdcnet_model_adapt.py) puts every BatchNorm2d / 3d that is not a head's into eval
mode: the backbone normalises with its LOADED running statistics; the heads' BatchNorm1d -- proj_t's too: the isinstance test names BatchNorm2d
and SyncBatchNorm only, and convert_syncbn() cannot run on CPU -- stay in train mode (batch statistics, running statistics updated).
Same import shims as make_golden_nlspn.py / make_golden_costdcnet.py (imported from them, nothing copied).
Usage:  python tests/golden/make_golden_head_generic.py [nlspn|costdcnet]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HEAD_HP = dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
PREPARE = 'meta_selfsup_seq_1layer_ema'


def perturbed_target(sd):
    """proj_t = proj * (1 + 0.05 * sin(index)) for its parameter tensors (buffers stay copies): the EMA becomes visible."""
    out = {}
    for k, v in sd.items():
        if k.startswith('proj_t.') and not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
            a = np.asarray(v, np.float32)
            out[k] = (a * (1.0 + 0.05 * np.sin(np.arange(a.size, dtype=np.float64)).reshape(a.shape))).astype(np.float32)
    return out


def put(out, key, a):
    """Small tensors whole; large matrices as 8 sampled rows + float64 checksums."""
    a = np.asarray(a)
    if a.size <= 4096:
        out[key] = a.copy()
        return
    idx = np.linspace(0, a.shape[0] - 1, 8).astype(np.int64)
    out[key + '#rows'] = a[idx].copy()
    out[key + '#sum'] = np.array([a.sum(dtype=np.float64), np.abs(a).sum(dtype=np.float64)])


def sample_rows(x, k=8):
    idx = np.linspace(0, x.shape[0] - 1, min(k, x.shape[0])).astype(np.int64)
    return idx, x[idx]


def run_case(model, net, sd, frames, name, loss_type, steps, meta):
    head_params = model.prepare_parameters('head_selfsup_ema')        # re-creates the heads (head_main.py:268): weights are loaded AFTER it
    sd = dict(sd)
    sd.update(perturbed_target(sd))
    assert list(sd.keys()) == list(net.state_dict().keys()), 'key table drifted from reference'
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    names = [k for k, p in net.named_parameters() if any(p is q for q in head_params)]
    assert len(names) == 12 and all(('proj' in k or 'pred' in k) and '_t' not in k for k in names), names
    opt = torch.optim.Adam(head_params, **HEAD_HP)
    out = {'meta': np.array(meta, dtype=np.int64), 'loss_type': np.array(loss_type),
           'hp': np.array([HEAD_HP['lr'], HEAD_HP['betas'][0], HEAD_HP['betas'][1], HEAD_HP['eps'], HEAD_HP['weight_decay'], 0.999]),
           'head_names': np.array(names)}
    for s in range(steps):
        image, sparse = frames(s)
        model.train(prepare=True)
        output_depth, embedding, reference = model.forward(image=image, sparse_depth=sparse, loss_type=loss_type)
        assert output_depth is None
        loss, info = model.compute_loss(input_rgb=image, output_depth=output_depth, validity_map=None, ground_truth=None,
                                        embedding=embedding, reference=reference, loss_type='prepare')
        opt.zero_grad()
        loss.backward()
        p = 's%d/' % s
        out[p + 'loss'] = np.array(float(loss))
        idx, out[p + 'emb_rows'] = sample_rows(embedding.detach().numpy())
        _, out[p + 'ref_rows'] = sample_rows(reference.detach().numpy())
        out[p + 'row_idx'] = idx
        out[p + 'emb_shape'] = np.array(embedding.shape)
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
        print(name, s, float(loss), 'no grad:', [k for k in names if named[k].grad is None], 'emb', tuple(embedding.shape),
              'emb |.| %.3g ref |.| %.3g' % (float(embedding.abs().mean()), float(reference.abs().mean())), flush=True)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def nlspn_cases():
    import make_golden_nlspn as MN
    ema = MN.import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from proxytta import synth
    for name, loss_type, h, w, n, steps in (('head_nlspn_forward_48x80_n2', 'head_selfsup_seq_ema', 48, 80, 2, 2),
                                            ('head_nlspn_reverse_48x80_n2', 'head_selfsup_seq_ema_reverse', 48, 80, 2, 2),
                                            ('head_nlspn_reverse_96x320', 'head_selfsup_seq_ema_reverse', 96, 320, 1, 2)):
        model = ema.ExternalModel_Adapt('nlspn', 0.0, 80.0, max_input_depth=80.0, offset=True, device=torch.device('cpu'))
        model._prepare_head(PREPARE)
        net = model.model.model
        sd = synth.formula_state_dict_nlspn(PREPARE)

        def frames(s, h=h, w=w, n=n):
            raw, image1, sparse = MN.nlspn_frame(s, h, w, n)
            return torch.from_numpy(image1), torch.from_numpy(sparse)
        run_case(model, net, sd, frames, name, loss_type, steps, [h, w, n, steps])


def costdcnet_cases():
    import make_golden_costdcnet as MC          # installs the MinkowskiEngine stand-in (oracle/minkowski_lite.py: parity unpinned for the sparse branch)
    ema, _ = MC.MG.import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from proxytta import synth
    # sizes divisible by 32: the heads' rows are the H/32 x W/32 bottleneck of the cost-volume UNet (CostDCNet_adapt.py:287-290)
    for name, loss_type, h, w, n, steps in (('head_costdcnet_forward_64x96_n2', 'head_selfsup_seq_ema', 64, 96, 2, 2),
                                            ('head_costdcnet_reverse_64x96_n2', 'head_selfsup_seq_ema_reverse', 64, 96, 2, 2),
                                            ('head_costdcnet_reverse_160x224', 'head_selfsup_seq_ema_reverse', 160, 224, 1, 2)):
        model = ema.ExternalModel_Adapt('costdcnet', 0.1, MC.MAX_DEPTH, max_input_depth=None, device=torch.device('cpu'))
        model._prepare_head(PREPARE)
        net = model.model.model
        sd = synth.formula_state_dict_costdcnet(PREPARE)

        def frames(s, h=h, w=w, n=n):
            raw, image1, sparse = MC.costdc_frame(s, h, w, n, 0.05)
            return torch.from_numpy(image1), torch.from_numpy(sparse)
        run_case(model, net, sd, frames, name, loss_type, steps, [h, w, n, steps])


if __name__ == '__main__':
    which = sys.argv[1:] or ['nlspn', 'costdcnet']
    if 'nlspn' in which:
        nlspn_cases()
    if 'costdcnet' in which:
        import subprocess
        if len(which) > 1:      # each backbone's shims want their own interpreter
            raise SystemExit(subprocess.call([sys.executable, os.path.abspath(__file__), 'costdcnet']))
        costdcnet_cases()

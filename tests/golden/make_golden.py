"""Generate golden vectors by running the REAL reference (read-only at /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  No reference
file is modified or copied: it is imported in place with four harness-side shims
(SURVEY.md §8c): (1) ``.cuda()``/``.to('cuda')`` become no-ops, (2) ``torchvision`` is an empty
stub, (3) ``src/msg_chn_model_adapt.py`` is exec'd with the stray 26-character prefix on its
first line dropped in memory, (4) cwd = reference root (the adapters use relative sys.path).

Driven surface: ``ExternalModel_Adapt`` (src/external_model_adapt.py:29) -> ``_prepare_head``,
``adapt_parameters('meta')``, ``forward``, ``compute_loss(loss_type='adapt')``, ``backward``,
``torch.optim.Adam.step`` and the eval forward — exactly the calls of src/tta_main.py:583-633
and :729-736.  Weights and frames come from ``proxytta.synth`` formulas so only outputs are
stored.  Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
from proxytta import synth  # noqa: E402

LOSS_TYPE = 'adapt_meta_selfsup_seq_ema_reverse'
HP = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
          w_sd=1.0, w_sm=2.0, w_cos=0.1, max_input_depth=80.0)


def import_reference():
    os.chdir(REF)
    sys.path.insert(0, os.path.join(REF, 'src'))
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    orig_to = nn.Module.to

    def to(self, *a, **k):
        a = [x for x in a if 'cuda' not in str(x)]
        if 'cuda' in str(k.get('device', '')):
            k.pop('device')
        return orig_to(self, *a, **k) if (a or k) else self
    nn.Module.to = to
    sys.modules['torchvision'] = types.ModuleType('torchvision')
    path = os.path.join(REF, 'src', 'msg_chn_model_adapt.py')
    src = open(path).read()
    prefix = 'src/msg_chn_model_adapt.py'
    assert src.startswith(prefix)
    mod = types.ModuleType('msg_chn_model_adapt')
    mod.__file__ = path
    sys.modules['msg_chn_model_adapt'] = mod
    exec(compile(src[len(prefix):], path, 'exec'), mod.__dict__)
    import external_model_adapt
    import net_utils
    return external_model_adapt, net_utils


def sample_rows(x, k=24):
    idx = np.linspace(0, x.shape[0] - 1, min(k, x.shape[0])).astype(np.int64)
    return idx, x[idx]


def run_case(ema, name, prepare_mode, h, w, n, steps, w_cos=None, gain=1.0, head_bias=0.0):
    hp = dict(HP)
    if w_cos is not None:
        hp['w_cos'] = w_cos
    model = ema.ExternalModel_Adapt('msg_chn', 0.0, 80.0, max_input_depth=hp['max_input_depth'],
                                    device=torch.device('cpu'))
    model._prepare_head(prepare_mode)
    net = model.model.model
    sd = synth.formula_state_dict(prepare_mode, gain, head_bias)
    assert list(sd.keys()) == list(net.state_dict().keys()), 'key table drifted from reference'
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    params = model.adapt_parameters(mode='meta')
    opt = torch.optim.Adam(params, lr=hp['lr'], betas=hp['betas'], eps=hp['eps'],
                           weight_decay=hp['weight_decay'])
    out = {'meta': np.array([h, w, n, steps], dtype=np.int64),
           'hp': np.array([hp['lr'], hp['betas'][0], hp['betas'][1], hp['eps'], hp['weight_decay'],
                           hp['w_sd'], hp['w_sm'], hp['w_cos'], hp['max_input_depth'], gain],
                          dtype=np.float64),
           'head_bias': np.array(head_bias, dtype=np.float64)}
    names = [k for k, _ in net.named_parameters() if 'meta' in k]
    for s in range(steps):
        image_np, sparse_np = synth.synthetic_frame(s, h, w, n)
        image, sparse = torch.from_numpy(image_np), torch.from_numpy(sparse_np)
        validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)   # tta_main.py:583-586
        model.train()
        depth, emb, ref = model.forward(image=image, sparse_depth=sparse, loss_type=LOSS_TYPE)
        loss, info = model.compute_loss(
            input_rgb=image.detach(), output_depth=depth, sparse_depth=sparse.detach(),
            validity_map=validity.detach(), embedding=emb, reference=ref,
            w_loss_sparse_depth=hp['w_sd'], w_loss_smoothness=hp['w_sm'], w_loss_cos=hp['w_cos'],
            loss_type='adapt')
        opt.zero_grad()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if 'meta' in k}
        opt.step()
        model.eval()
        with torch.no_grad():
            depth_eval = model.forward(image=image, sparse_depth=sparse, loss_type=LOSS_TYPE)
        p = 's%d/' % s
        out[p + 'depth_train'] = depth.detach().numpy()
        out[p + 'depth_eval'] = depth_eval.numpy()
        e, r = emb.detach().numpy(), ref.detach().numpy()
        idx, out[p + 'emb_rows'] = sample_rows(e)
        _, out[p + 'ref_rows'] = sample_rows(r)
        out[p + 'row_idx'] = idx
        out[p + 'emb_shape'] = np.array(e.shape)
        out[p + 'emb_abs_mean'] = np.array(np.abs(e).mean(dtype=np.float64))
        out[p + 'ref_abs_mean'] = np.array(np.abs(r).mean(dtype=np.float64))
        out[p + 'loss_info'] = np.array([float(info[k].detach()) for k in
                                         ('loss', 'loss_smooth', 'loss_sparse_depth', 'loss_cos')])
        state = opt.state_dict()['state']
        for i, k in enumerate(names):
            out[p + 'grad/' + k] = grads[k].numpy()
            out[p + 'param/' + k] = dict(net.named_parameters())[k].detach().numpy().copy()
            out[p + 'exp_avg/' + k] = state[i]['exp_avg'].numpy().copy()
            out[p + 'exp_avg_sq/' + k] = state[i]['exp_avg_sq'].numpy().copy()
        for k, v in net.state_dict().items():
            if 'running_' in k and ('proj.' in k or 'pred.' in k or 'meta' in k):
                out[p + 'buf/' + k] = v.numpy().copy()
        print(name, 'step', s, 'loss_info', out[p + 'loss_info'], 'depth mean',
              float(depth.mean()), 'grad|w|', float(sum(g.abs().sum() for g in grads.values())))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def run_loss_gate(ema):
    """compute_loss(loss_type='adapt') on crafted embeddings: one pair on each side of the
    ``loss_cos < 0.3`` gate (src/external_model_adapt.py:424-425), with input gradients."""
    model = ema.ExternalModel_Adapt('msg_chn', 0.0, 80.0, max_input_depth=80.0,
                                    device=torch.device('cpu'))
    h, w, n, rows, dim = 24, 40, 2, 60, 512
    image_np, sparse_np = synth.synthetic_frame(11, h, w, n, density=0.1)
    image, sparse = torch.from_numpy(image_np), torch.from_numpy(sparse_np)
    validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
    u = lambda tag, *shape: torch.from_numpy(
        (synth.hash_uniform(tag, int(np.prod(shape))) * 2 - 1).reshape(shape).astype(np.float32))
    out = {'meta': np.array([h, w, n, rows, dim]), 'w': np.array([1.0, 2.0, 0.1])}
    for tag, noise in (('near', 0.2), ('far', 3.0)):
        depth = (20 + 10 * u('gate/depth', n, 1, h, w)).requires_grad_(True)
        emb = u('gate/emb', rows, dim)
        ref = (emb + noise * u('gate/noise' + tag, rows, dim)).requires_grad_(True)
        loss, info = model.compute_loss(
            input_rgb=image, output_depth=depth, sparse_depth=sparse, validity_map=validity,
            embedding=emb, reference=ref, w_loss_sparse_depth=1.0, w_loss_smoothness=2.0,
            w_loss_cos=0.1, loss_type='adapt')
        loss.backward()
        out[tag + '/loss_info'] = np.array([float(info['loss'].detach()), float(info['loss_smooth'].detach()),
                                            float(info['loss_sparse_depth'].detach()), float(info['loss_cos'].detach())])
        out[tag + '/grad_depth'] = depth.grad.numpy()
        out[tag + '/grad_ref'] = ref.grad.numpy() if ref.grad is not None else np.zeros((rows, dim), np.float32)
        out[tag + '/noise'] = np.array(noise)
        print('gate case', tag, out[tag + '/loss_info'])
    np.savez_compressed(os.path.join(HERE, 'adapt_loss_gate.npz'), **out)


def run_outlier(net_utils):
    _, sparse = synth.synthetic_frame(7, 40, 56, 2, density=0.2, dmin=1.0, dmax=20.0)
    sparse = torch.from_numpy(sparse)
    validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
    sd, vm = net_utils.OutlierRemoval(7, 1.5).remove_outliers(sparse, validity)
    np.savez_compressed(os.path.join(HERE, 'outlier_removal.npz'),
                        sparse_out=sd.numpy(), validity_out=vm.numpy(),
                        meta=np.array([7, 40, 56, 2]), params=np.array([0.2, 1.0, 20.0, 7, 1.5]))
    print('outlier removal: kept', int(vm.sum()), 'of', int(validity.sum()))


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ema, net_utils = import_reference()
    one, two = 'meta_selfsup_seq_1layer_ema', 'meta_selfsup_seq_2layers_ema'
    if len(sys.argv) > 1 and sys.argv[1] == 'gate':
        # whole steps on both sides of the `loss_cos < 0.3` gate (src/external_model_adapt.py:424-425): the heads' output biases are one
        # large common vector (synth.formula_state_dict head_bias), L_cos = 0.2 (gate fires: the cosine term drops out of loss and
        # gradients -- the branch trained heads take) and 0.37 (just above: it stays); w_cos = 300 so that the cosine term is ~8 % of the adapted gradient when it is on
        run_case(ema, 'msgchn_1layer_64x96_gate_below', one, 64, 96, 1, 3, w_cos=300.0, head_bias=5.0)
        run_case(ema, 'msgchn_1layer_64x96_gate_above', one, 64, 96, 1, 3, w_cos=300.0, head_bias=3.5)
        raise SystemExit(0)
    run_case(ema, 'msgchn_1layer_32x48', one, 32, 48, 1, 3)
    run_case(ema, 'msgchn_1layer_64x96', one, 64, 96, 1, 3)
    run_case(ema, 'msgchn_1layer_36x52_pad', one, 36, 52, 1, 2)        # dual-corner padding path
    run_case(ema, 'msgchn_1layer_32x48_n2', one, 32, 48, 2, 2)         # batch 2
    run_case(ema, 'msgchn_1layer_32x48_wcos1', one, 32, 48, 1, 2, w_cos=1.0)
    run_case(ema, 'msgchn_2layers_32x48', two, 32, 48, 1, 2)
    run_loss_gate(ema)
    run_outlier(net_utils)

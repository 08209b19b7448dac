#!/usr/bin/env python3
"""Algorithmic work of one CostDCNet TTA step, counted with forward hooks on the REAL reference (build container only:
python tools/costdcnet_count.py [H W]).  Same counting rule as SURVEY.md section 8d for MSG_CHN:

  * per conv / linear / sparse-conv layer executed in the TRAINING forward (grad pass + proxy pass + heads): input + output + weight
    ELEMENTS once, MACs = output positions x kernel taps x Cin x Cout (sparse convolutions: the kernel-map pairs actually visited);
  * the MINIMAL backward: a data gradient for every grad-pass layer downstream of the first adapted tensor (Encoder2D's first
    BatchNorm: everything except enc2d.conv1; the frozen sparse encoder has no backward) + the weight gradient of conv1_rgb_meta +
    the data gradient through proj_t: input + output + weight elements and the forward layer's MACs once more per data gradient;
  * BatchNorm / ELU / pooling / upsampling / fusion / softmax regression / loss / Adam traffic counts as fused (0), as in section 8d.

The reference is imported in place with make_golden_costdcnet.py's shims (oracle/minkowski_lite.py stands in for the absent
MinkowskiEngine).  Prints one JSON object; bench.py carries the figures for 480x640 (VOID-1500) as constants with this script named.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT); sys.path.insert(0, GOLD)
from oracle import minkowski_lite as ML  # noqa: E402
ML.install(sys.modules)
import make_golden as MG  # noqa: E402
import make_golden_costdcnet as MC  # noqa: E402
from make_golden import synth  # noqa: E402


def main(h=480, w=640):
    torch.set_num_threads(8)
    ema, _ = MG.import_reference()
    model = ema.ExternalModel_Adapt('costdcnet', 0.1, MC.MAX_DEPTH, max_input_depth=None, device=torch.device('cpu'))
    model._prepare_head(MC.PREPARE)
    net = model.model.model
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.formula_state_dict_costdcnet(MC.PREPARE).items()})
    model.adapt_parameters(mode='meta_bn')
    rows = []                       # (name, elements, macs, in training-forward call order)
    names = {m: k for k, m in net.named_modules()}

    def dense_hook(m, inp, out):
        x = inp[0]
        wt = m.weight
        taps = int(np.prod(wt.shape[2:])) if wt.dim() > 2 else 1
        cin = wt.shape[1] if not isinstance(m, (nn.ConvTranspose2d, nn.ConvTranspose3d)) else wt.shape[0]
        cout = out.shape[1] if out.dim() > 2 else out.shape[-1]
        pos = out.numel() // cout
        rows.append((names[m], x.numel() + out.numel() + wt.numel(), pos * taps * cin * cout))

    def kernel_map_pairs(x, out, kernel_size):
        """(output voxel, offset) pairs with an existing input voxel = the products a generalized sparse convolution performs"""
        ts_in = torch.tensor(x.tensor_stride, dtype=torch.long)
        dims = [int(x.C[:, a].max()) + 2 * ML._S + 1 for a in range(1, 4)]
        kin = torch.sort(ML._keys(x.C, dims))[0]
        n = 0
        for off in ML.kernel_offsets(kernel_size):
            q = out.C.clone()
            q[:, 1:] += torch.tensor(off, dtype=torch.long) * ts_in
            ok = ((q[:, 1:] + ML._S) >= 0).all(1) & ((q[:, 1:] + ML._S) < torch.tensor(dims)).all(1)
            kq = ML._keys(q, dims)
            pos = torch.searchsorted(kin, kq).clamp(max=kin.numel() - 1)
            n += int((ok & (kin[pos] == kq)).sum())
        return n

    def sparse_hook(m, inp, out):
        x = inp[0]
        k = m.kernel if m.kernel.dim() == 3 else m.kernel[None]      # (K, Cin, Cout)
        rows.append((names[m], x.F.numel() + out.F.numel() + k.numel(), kernel_map_pairs(x, out, m.kernel_size) * k.shape[1] * k.shape[2]))
    hs = []
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            hs.append(m.register_forward_hook(dense_hook))
        elif isinstance(m, ML.MinkowskiConvolution):
            hs.append(m.register_forward_hook(sparse_hook))
    raw, image1, sparse_np = MC.costdc_frame(0, h, w, 1, 1500.0 / (h * w))
    model.train()
    model.forward(image=torch.from_numpy(image1), sparse_depth=torch.from_numpy(sparse_np), intrinsics=torch.eye(3)[None], loss_type=MC.LOSS_TYPE)
    for x in hs:
        x.remove()
    fwd_el = sum(r[1] for r in rows); fwd_mac = sum(r[2] for r in rows)
    # minimal backward: first occurrence of every layer = the grad pass (the reference runs it first, CostDCNet_adapt.py:207-256);
    # the proxy pass's repeats and the no-grad heads (proj / pred on the detached proxy rows) carry no gradient
    seen, bwd_el, bwd_mac, ev_el, ev_mac = set(), 0, 0, 0, 0
    for name, el, mac in rows:
        if name in seen:
            continue
        seen.add(name)
        if not name.startswith(('proj', 'pred')):          # the eval forward (src/tta_main.py:729-736): one pass, no heads
            ev_el += el; ev_mac += mac
        if name.startswith(('enc3d', 'proj.', 'pred.')) or name == 'enc2d.conv1':
            continue
        bwd_el += el; bwd_mac += mac
        if name == 'conv1_rgb_meta':                       # + its weight gradient
            bwd_el += el; bwd_mac += mac
    out = {'frame': [h, w], 'layers_forward': len(rows), 'forward_elements': fwd_el, 'forward_gmac': fwd_mac / 1e9,
           'backward_min_elements': bwd_el, 'backward_min_gmac': bwd_mac / 1e9,
           'eval_forward_elements': ev_el, 'eval_forward_gmac': ev_mac / 1e9,
           'step_elements': fwd_el + bwd_el, 'step_bytes_fp32': 4 * (fwd_el + bwd_el), 'step_gmac': (fwd_mac + bwd_mac) / 1e9}
    print(json.dumps(out))
    return out


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:3]])

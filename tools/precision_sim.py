#!/usr/bin/env python3
"""CPU SIMULATION of per-class arithmetic / storage widths for the MSG_CHN step (design aid, round 5).

Not the product and not a parity check: the PyTorch-CPU oracle with operand rounding injected per tensor class, to see
which classes tolerate single-bf16 arithmetic ("x1") and bf16 storage before kernels are written.  The measured budget of
the REAL kernels is tools/accuracy_report.py / profiles/r05_precision_budget.txt.

Classes (network_exp_msg_chn_adapt.py):
  real   the grad pass's convolutions (:479-506)             -- decides depth_train / depth_eval directly
  proxy  the no_grad zero-image pass (:509-532)              -- reaches the depth only through L_cos's gradient
  heads  the three MLP applications (:551-554), forward
  bwd    every data gradient of loss.backward() (tta_main.py:632), incl. the meta layer's weight gradient operands
per class:  x3 (fp32-faithful), x1 (both operands rounded to bf16), xa (activation operand rounded only), xw (weights only);
            suffix 's' = outputs STORED in bf16 as well.
  python tools/precision_sim.py --size 256x320 --steps 3 --cfg proxy=x1s,heads=x1s,bwd=x1
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as RF  # noqa: E402
from oracle import proxytta_oracle as O  # noqa: E402
from proxytta import synth  # noqa: E402

MODE = 'meta_selfsup_seq_1layer_ema'
CFG = {'real': 'x3', 'proxy': 'x3', 'heads': 'x3', 'bwd': 'x3'}
STATE = {'cls': 'real'}


def q(t):
    return t.to(torch.bfloat16).to(torch.float32)


def ops(mode, x, w):
    m = mode.rstrip('s')
    if m == 'x1':
        return q(x), q(w)
    if m == 'xa':
        return q(x), w
    if m == 'xw':
        return x, q(w)
    return x, w


class QConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, transposed, fmode, bmode):
        xq, wq = ops(fmode, x, w)
        if transposed:
            y = RF.conv_transpose2d(xq, wq, b, stride=stride, padding=1, output_padding=1)
        else:
            y = RF.conv2d(xq, wq, b, stride=stride, padding=1)
        if fmode.endswith('s'):
            y = q(y)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, transposed, bmode, b is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, transposed, bmode, has_b = ctx.cfg
        gq, wq = ops(bmode, g, w)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if transposed:
                gx = RF.conv2d(gq, wq, None, stride=stride, padding=1)
            else:
                gx = torch.nn.grad.conv2d_input(x.shape, wq, gq, stride=stride, padding=1)
            if bmode.endswith('s'):
                gx = q(gx)
        if ctx.needs_input_grad[1]:
            gq2, xq2 = ops(bmode, g, x)
            assert not transposed
            gw = torch.nn.grad.conv2d_weight(xq2, w.shape, gq2, stride=stride, padding=1)
        if has_b and ctx.needs_input_grad[2]:
            gb = g.sum(dim=(0, 2, 3))
        return gx, gw, gb, None, None, None, None


class QLin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, fmode, bmode):
        xq, wq = ops(fmode, x, w)
        y = RF.linear(xq, wq, b)
        if fmode.endswith('s'):
            y = q(y)
        ctx.save_for_backward(w)
        ctx.bmode = bmode
        return y

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        gq, wq = ops(ctx.bmode, g, w)
        gx = gq @ wq
        if ctx.bmode.endswith('s'):
            gx = q(gx)
        return gx, None, None, None, None


class ShimF:
    def __getattr__(self, k):
        return getattr(RF, k)

    @staticmethod
    def conv2d(x, w, b=None, stride=1, padding=1):
        return QConv.apply(x, w, b, stride, False, CFG[STATE['cls']], CFG['bwd'])

    @staticmethod
    def conv_transpose2d(x, w, b=None, stride=2, padding=1, output_padding=1):
        return QConv.apply(x, w, b, stride, True, CFG[STATE['cls']], CFG['bwd'])

    @staticmethod
    def linear(x, w, b=None):
        return QLin.apply(x, w, b, CFG['heads'], CFG['bwd'])


def network_forward_q(P, image, d, training, prepare_mode=MODE):
    STATE['cls'] = 'real'
    depth, feat = O.backbone(P, image, d, training, prepare_mode)
    if not training:
        return depth
    with torch.no_grad():
        STATE['cls'] = 'proxy'
        _, feat_zero = O.backbone(P, torch.zeros_like(image), d, training, prepare_mode, stop_at_encoder3=True)
        STATE['cls'] = 'real'
    flat = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    emb = O.mlp(P, 'pred', O.mlp(P, 'proj', flat(feat_zero).detach()))
    ref = O.mlp(P, 'proj', flat(feat))
    return depth, emb, ref


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().mean() / b.abs().mean())


def run(size, steps, cfg, w_cos, head_bias, frame0, threads, gain=1.0):
    h, w = size
    torch.set_num_threads(threads)
    hp = dict(max_input_depth=80.0, lr=1e-3, w_sd=1.0, w_sm=2.0, w_cos=w_cos)
    sd = synth.formula_state_dict(MODE, gain, head_bias)
    ref = O.MsgChnOracle(sd, MODE, **hp)
    tst = O.MsgChnOracle(sd, MODE, **hp)
    rows = []
    for s in range(steps):
        image, sparse = synth.synthetic_frame(frame0 + s, h, w, 1)
        ic, sc = torch.from_numpy(image), torch.from_numpy(sparse)
        for k in CFG:
            CFG[k] = 'x3'
        O.F, O.network_forward = RF, network_forward_q
        r = ref.step(ic, sc)
        e_ref = ref.forward_eval(ic, sc)
        CFG.update(cfg)
        O.F = ShimF()
        t = tst.step(ic, sc)
        e_tst = tst.forward_eval(ic, sc)
        O.F = RF
        name = ref.names[0]
        g_r, g_t = r['grads'][name], t['grads'][name]
        li_r = np.array([r['loss_info'][k] for k in ('loss', 'loss_smooth', 'loss_sparse_depth', 'loss_cos')])
        li_t = np.array([t['loss_info'][k] for k in ('loss', 'loss_smooth', 'loss_sparse_depth', 'loss_cos')])
        row = {'step': s, 'depth_train': rel(t['depth'], r['depth']), 'depth_eval': rel(e_tst, e_ref),
               'depth_move': rel(e_ref, r['depth']),
               'emb': rel(t['emb'], r['emb']), 'ref': rel(t['ref'], r['ref']),
               'grad_w_relmax': float((g_t - g_r).abs().max() / g_r.abs().max()), 'grad_w_relmae': rel(g_t, g_r),
               'sign_flips': float((torch.sign(g_t) != torch.sign(g_r)).float().mean()),
               'param_w_vs_lr': float((tst.P[name].detach() - ref.P[name].detach()).abs().mean() / 1e-3),
               'loss_info_rel': [float(x) for x in np.abs(li_t - li_r) / np.maximum(np.abs(li_r), 1e-12)],
               'loss_cos': float(li_r[3])}
        rows.append(row)
        print('  s%d dtrain %.1e deval %.1e (move %.1e) emb %.1e ref %.1e grad relmax %.1e flips %.3f param/lr %.3f loss_info %s' % (
            s, row['depth_train'], row['depth_eval'], row['depth_move'], row['emb'], row['ref'], row['grad_w_relmax'], row['sign_flips'],
            row['param_w_vs_lr'], ' '.join('%.0e' % x for x in row['loss_info_rel'])), flush=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', default='128x256')
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--cfg', default='proxy=x1s,heads=x1s')
    ap.add_argument('--w-cos', type=float, default=0.1)
    ap.add_argument('--head-bias', type=float, default=0.0)
    ap.add_argument('--frame0', type=int, default=0)
    ap.add_argument('--threads', type=int, default=8)
    a = ap.parse_args()
    cfg = dict(kv.split('=') for kv in a.cfg.split(',') if kv)
    print('cfg', cfg, 'size', a.size, 'w_cos', a.w_cos, flush=True)
    run([int(x) for x in a.size.split('x')], a.steps, cfg, a.w_cos, a.head_bias, a.frame0, a.threads)


if __name__ == '__main__':
    main()

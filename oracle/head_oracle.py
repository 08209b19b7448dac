"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's stage-2 head trainer for MSG_CHN.
Only tests/ may import this.  Pinned by tests/golden/head_*.npz (the real reference run on CPU, make_golden_head.py).

One `step()` = src/head_main.py:464-480 with loss_type 'head_selfsup_seq_ema[_reverse]':
  * `_update_head()`  proj_t <- tau * proj_t + (1 - tau) * proj over parameters()   NET:701-703, called at NET:682 / :691
  * both backbone passes under no_grad, stopping at depth_encoder3                       NET:626-676
  * reverse:     emb = pred(proj(feat_zero).detach()),  ref = proj(feat).detach()        NET:691-694
    not reverse: emb = pred(proj(feat)),                ref = proj(feat_zero).detach()   NET:681-684
  * prepare_loss = mean(2 - 2 <normalize(emb), normalize(ref)>)                          src/external_model_adapt.py:524-541
  * Adam over prepare_parameters('head_selfsup_ema') = proj.* and pred.* parameters (no proj_t), src/msg_chn_model_adapt.py:297-304;
    parameters that received no gradient (reverse: all of proj) are skipped by torch.optim.Adam.
NET = external_src/MSG_CHN/workspace/exp_msg_chn/network_exp_msg_chn_adapt.py
"""
import torch
import torch.nn.functional as F

from . import proxytta_oracle as O

_BUF = ('running_mean', 'running_var', 'num_batches_tracked')


def head_names(P):
    return [k for k in P if k.startswith(('proj.', 'pred.')) and not k.endswith(_BUF)]


def update_head(P, tau=0.999):
    with torch.no_grad():
        for k in P:
            if k.startswith('proj_t.') and not k.endswith(_BUF):
                s = P['proj.' + k[len('proj_t.'):]]
                P[k].copy_(P[k] * tau + s * (1.0 - tau))


def prepare_loss(embedding, reference):
    e = F.normalize(embedding, dim=-1, p=2)
    r = F.normalize(reference, dim=-1, p=2)
    return (2 - 2 * (e * r).sum(-1)).mean()


def head_forward(P, image, sparse_depth, reverse, max_input_depth=None, prepare_mode='meta_selfsup_seq_1layer_ema', tau=0.999):
    if max_input_depth is not None:
        sparse_depth = torch.clamp(sparse_depth, 0, max_input_depth)           # src/external_model_adapt.py:103-108
    update_head(P, tau)
    with torch.no_grad():
        _, feat = O.backbone(P, image, sparse_depth, True, prepare_mode, stop_at_encoder3=True)
        _, feat_zero = O.backbone(P, torch.zeros_like(image), sparse_depth, True, prepare_mode, stop_at_encoder3=True)
    flat = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    if reverse:
        emb = O.mlp(P, 'pred', O.mlp(P, 'proj', flat(feat_zero)).detach())
        ref = O.mlp(P, 'proj', flat(feat)).detach()
    else:
        emb = O.mlp(P, 'pred', O.mlp(P, 'proj', flat(feat)))
        ref = O.mlp(P, 'proj', flat(feat_zero)).detach()
    return emb, ref


class HeadTrainerOracle:
    def __init__(self, state_dict, loss_type='head_selfsup_seq_ema_reverse', max_input_depth=None, lr=2e-4, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=0.0, tau=0.999, prepare_mode='meta_selfsup_seq_1layer_ema'):
        assert 'head' in loss_type and 'ema' in loss_type and 'adapt' not in loss_type
        self.reverse = 'reverse' in loss_type
        self.P = {k: torch.as_tensor(v).clone() for k, v in state_dict.items()}
        self.names = head_names(self.P)
        for k in self.names:
            self.P[k].requires_grad_(True)
        self.max_input_depth, self.tau = max_input_depth, tau
        self.prepare_mode = prepare_mode
        self.hp = (lr, betas, eps, weight_decay)
        self.opt = None

    def step(self, image, sparse_depth):
        emb, ref = head_forward(self.P, image, sparse_depth, self.reverse, self.max_input_depth, prepare_mode=self.prepare_mode, tau=self.tau)
        loss = prepare_loss(emb, ref)
        params = [self.P[k] for k in self.names]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        live = [(k, p, g) for k, p, g in zip(self.names, params, grads) if g is not None]
        if self.opt is None:            # Adam state exists only for parameters that ever received a gradient
            self.opt = O.AdamState([p for _, p, _ in live], *self.hp)
            self.live = [k for k, _, _ in live]
        assert self.live == [k for k, _, _ in live]
        self.opt.step([p for _, p, _ in live], [g for _, _, g in live])
        return {'loss': float(loss.detach()), 'emb': emb.detach(), 'ref': ref.detach(), 'grads': {k: g for k, _, g in live}}

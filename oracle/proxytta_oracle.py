"""ORACLE — test infrastructure only, never shipped, never on the product path.

CPU (PyTorch fp32) restatement of the ProxyTTA per-frame step of seobbro/TTA-depth-completion
for the MSG_CHN backbone.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the product (``proxytta`` + ``libptta_hip.so``)
fails loudly when its HIP library is missing and never falls back to this code.

Parity pin: ``tests/golden/make_golden.py`` imports the real reference from /root/reference
(CPU shims only) in the build container, drives ``ExternalModel_Adapt`` end to end and commits
its inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against those vectors (the reference ships no tests of its own).

Each function cites the reference file:line it follows (paths relative to the reference root).
The network graph is written from the layer semantics, as a flat functional program over a
``{name: tensor}`` state dict, not as a copy of the reference's nn.Module classes.
"""
import math

import torch
import torch.nn.functional as F

NET = 'external_src/MSG_CHN/workspace/exp_msg_chn/network_exp_msg_chn_adapt.py'


def _up2(x):
    # F.interpolate(scale_factor=2, bilinear, align_corners=True): NET:200-209
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)


def _conv(P, name, x, stride=1, relu_in=True):
    if relu_in:
        x = F.relu(x)
    return F.conv2d(x, P[name + '.weight'], P[name + '.bias'], stride=stride, padding=1)


def _convT(P, name, x):
    # ReLU -> ConvTranspose2d(32,32,3,stride=2,padding=1,output_padding=1): NET:273-274
    return F.conv_transpose2d(F.relu(x), P[name + '.weight'], P[name + '.bias'],
                              stride=2, padding=1, output_padding=1)


def rgb_encoder(P, image):
    """RGBEncoder.forward, NET:252-264: five pre-activation maps at 1/1 .. 1/16."""
    pre = 'rgb_encoder.'
    x0 = _conv(P, pre + 'init.2', _conv(P, pre + 'init.0', image, relu_in=False))
    outs = [x0]
    x = x0
    for e in ('enc1', 'enc2', 'enc3', 'enc4'):
        x = _conv(P, pre + e + '.3', _conv(P, pre + e + '.1', x, stride=2))
        outs.append(x)
    return outs


def meta_layer(P, x, training, prepare_mode):
    """conv1_rgb_meta: NET:1065-1071 (1layer: plain Conv2d(32,32,3,1,1), no activation) or
    NET:28-36,1073-1077 (2layers: Res_Conv(32,128))."""
    if '2layers' in prepare_mode:
        p = 'conv1_rgb_meta.conv1_meta.'
        h = F.conv2d(x, P[p + '0.0.weight'], None, padding=1)
        h = F.batch_norm(h, P[p + '0.1.running_mean'], P[p + '0.1.running_var'],
                         P[p + '0.1.weight'], P[p + '0.1.bias'], training, 0.1, 1e-5)
        h = F.leaky_relu(h, 0.2)
        h = F.conv2d(h, P[p + '1.weight'], P[p + '1.bias'], padding=1)
        h = F.batch_norm(h, P[p + '2.running_mean'], P[p + '2.running_var'],
                         P[p + '2.weight'], P[p + '2.bias'], training, 0.1, 1e-5)
        return h + x
    return F.conv2d(x, P['conv1_rgb_meta.weight'], P['conv1_rgb_meta.bias'], padding=1)


def depth_encoder(P, idx, inp, pre_x2=None, pre_x3=None, pre_x4=None):
    """DepthEncoder.forward, NET:192-211."""
    pre = 'depth_encoder%d.' % idx
    x0 = _conv(P, pre + 'init.2', _conv(P, pre + 'init.0', inp, relu_in=False))
    if pre_x4 is not None:
        x0 = x0 + _up2(pre_x4)
    x1 = _conv(P, pre + 'enc1.3', _conv(P, pre + 'enc1.1', x0, stride=2))
    if pre_x3 is not None:
        x1 = x1 + _up2(pre_x3)
    x2 = _conv(P, pre + 'enc2.3', _conv(P, pre + 'enc2.1', x1, stride=2))
    if pre_x2 is not None:
        x2 = x2 + _up2(pre_x2)
    return x0, x1, x2


def depth_decoder(P, idx, dx, cx):
    """DepthDecoder.forward, NET:296-311: returns (x2, x3, x4, prediction)."""
    pre = 'depth_decoder%d.' % idx
    x2 = dx[2] + cx[2]
    x1 = dx[1] + cx[1]
    x0 = dx[0] + cx[0]
    x3 = _conv(P, pre + 'dec2.3', _convT(P, pre + 'dec2.1', x2))
    x4 = _conv(P, pre + 'dec1.3', _convT(P, pre + 'dec1.1', x1 + x3))
    out = _conv(P, pre + 'prdct.3', _conv(P, pre + 'prdct.1', x4 + x0))
    return x2, x3, x4, out


def sparse_pool(d, k):
    # NET:487,492: avg_pool(d)/(avg_pool(d>0)+1e-4)
    c = (d > 0).float()
    return F.avg_pool2d(d, k, k) / (F.avg_pool2d(c, k, k) + 0.0001)


def backbone(P, image, d, training, prepare_mode, stop_at_encoder3=False):
    """One pass of _rgbd_meta_contrast's cascade (grad pass NET:479-506, proxy pass :509-532)."""
    enc_c = rgb_encoder(P, image)
    enc_c[2] = meta_layer(P, enc_c[2], training, prepare_mode)
    e1 = depth_encoder(P, 1, sparse_pool(d, 4))
    d1 = depth_decoder(P, 1, e1, enc_c[2:5])
    p12 = _up2(d1[3])
    e2 = depth_encoder(P, 2, torch.cat((sparse_pool(d, 2), p12), 1), d1[0], d1[1], d1[2])
    d2 = depth_decoder(P, 2, e2, enc_c[1:4])
    p11 = _up2(d2[3] + p12)
    e3 = depth_encoder(P, 3, torch.cat((d, p11), 1), d2[0], d2[1], d2[2])
    if stop_at_encoder3:
        return None, e3[2]
    d3 = depth_decoder(P, 3, e3, enc_c[0:3])
    return d3[3] + p11, e3[2]


def mlp(P, prefix, x, training=True):
    """Linear - BatchNorm1d - ReLU - Linear, NET:1089-1098; BN1d in train mode uses batch
    statistics and updates running stats in place (momentum 0.1, eps 1e-5)."""
    h = F.linear(x, P[prefix + '.0.weight'], P[prefix + '.0.bias'])
    h = F.batch_norm(h, P[prefix + '.1.running_mean'], P[prefix + '.1.running_var'],
                     P[prefix + '.1.weight'], P[prefix + '.1.bias'], training, 0.1, 1e-5)
    if training and (prefix + '.1.num_batches_tracked') in P:
        P[prefix + '.1.num_batches_tracked'] += 1
    return F.linear(F.relu(h), P[prefix + '.3.weight'], P[prefix + '.3.bias'])


def network_forward(P, image, d, training, prepare_mode='meta_selfsup_seq_1layer_ema'):
    """_rgbd_meta_contrast with mode = [adapt, reverse, seq, ema], NET:463-557."""
    depth, feat = backbone(P, image, d, training, prepare_mode)
    if not training:
        return depth
    with torch.no_grad():
        _, feat_zero = backbone(P, torch.zeros_like(image), d, training, prepare_mode,
                                stop_at_encoder3=True)
    flat = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    emb = mlp(P, 'pred', mlp(P, 'proj', flat(feat_zero).detach()))     # NET:553
    ref = mlp(P, 'proj', flat(feat))                                     # NET:554
    return depth, emb, ref


def _pad16(n):
    return 0 if n % 16 == 0 else (n // 16 + 1) * 16 - n


def adapter_forward(P, image, d, training, prepare_mode='meta_selfsup_seq_1layer_ema'):
    """MsgChnModel_Adapt.forward, src/msg_chn_model_adapt.py:54-200: shapes not divisible by 16
    are zero-padded twice (top/right and bottom/left), run as a doubled batch, cropped, averaged."""
    h, w = image.shape[-2:]
    pt, pr = _pad16(h), _pad16(w)
    if pt or pr:
        image = torch.cat([F.pad(image, (0, pr, pt, 0)), F.pad(image, (pr, 0, 0, pt))], 0)
        d = torch.cat([F.pad(d, (0, pr, pt, 0)), F.pad(d, (pr, 0, 0, pt))], 0)
    out = network_forward(P, image, d, training, prepare_mode)
    depth = out[0] if training else out
    if pt or pr:
        o0, o1 = torch.chunk(depth, 2, 0)
        hh, ww = o0.shape[-2:]
        o0 = o0[:, :, pt:, :ww - pr]
        o1 = o1[:, :, :hh - pt, pr:]
        depth = torch.mean(torch.stack([o0, o1], 1), 1)
    return (depth, out[1], out[2]) if training else depth


def model_forward(P, image, sparse_depth, training, max_input_depth=None,
                  prepare_mode='meta_selfsup_seq_1layer_ema'):
    """ExternalModel_Adapt.forward, src/external_model_adapt.py:82-114 (clamp at :108)."""
    if max_input_depth is not None:
        sparse_depth = torch.clamp(sparse_depth, 0, max_input_depth)
    return adapter_forward(P, image, sparse_depth, training, prepare_mode)


def smoothness_loss(predict, image):
    """src/loss_utils.py:139-169 with gradient_yx :624-638."""
    pdx = predict[:, :, :, :-1] - predict[:, :, :, 1:]
    pdy = predict[:, :, :-1, :] - predict[:, :, 1:, :]
    idx = image[:, :, :, :-1] - image[:, :, :, 1:]
    idy = image[:, :, :-1, :] - image[:, :, 1:, :]
    wx = torch.exp(-torch.mean(torch.abs(idx), dim=1, keepdim=True))
    wy = torch.exp(-torch.mean(torch.abs(idy), dim=1, keepdim=True))
    return torch.mean(wx * torch.abs(pdx)) + torch.mean(wy * torch.abs(pdy))


def sparse_depth_loss(src, tgt, w):
    """src/loss_utils.py:116-137 (no eps: NaN when a sample has no valid point)."""
    loss = torch.sum(w * torch.abs(tgt - src), dim=[1, 2, 3])
    return torch.mean(loss / torch.sum(w, dim=[1, 2, 3]))


def adapt_loss(input_rgb, output_depth, sparse_depth, validity_map, embedding, reference,
               w_sd, w_sm, w_cos, max_input_depth=None):
    """compute_loss(loss_type='adapt') -> adapt_loss, src/external_model_adapt.py:191-203,
    :371-441, including the data-dependent gate ``loss_cos < 0.3 => w_cos = 0`` (:424-425)."""
    if max_input_depth is not None:
        sparse_depth = torch.clamp(sparse_depth, 0, max_input_depth)
    l_sm = smoothness_loss(output_depth, input_rgb)
    l_sd = sparse_depth_loss(output_depth, sparse_depth, validity_map)
    e = F.normalize(embedding, dim=-1, p=2)
    r = F.normalize(reference, dim=-1, p=2)
    l_cos = (2 - 2 * (e * r).sum(dim=-1)).mean()
    if l_cos < 0.3:
        w_cos = 0
    loss = w_sd * l_sd + w_sm * l_sm + w_cos * l_cos
    return loss, {'loss': loss, 'loss_smooth': l_sm, 'loss_sparse_depth': l_sd, 'loss_cos': l_cos}


def remove_outliers(sparse_depth, validity_map, kernel_size=7, threshold=1.5):
    """OutlierRemoval.remove_outliers, src/net_utils.py:766-811."""
    max_value = 10 * torch.max(sparse_depth)
    filled = torch.where(validity_map <= 0, torch.full_like(sparse_depth, max_value), sparse_depth)
    p = kernel_size // 2
    filled = F.pad(filled, (p, p, p, p), value=float(max_value))
    mins = -F.max_pool2d(-filled, kernel_size, 1, 0)
    clean = torch.where(mins < sparse_depth - threshold,
                        torch.zeros_like(validity_map), torch.ones_like(validity_map))
    clean = validity_map * clean
    return sparse_depth * clean, clean


def adapted_names(P, adapt_mode='meta'):
    """adapt_parameters(mode='meta'): every parameter whose name contains 'meta'
    (src/msg_chn_model_adapt.py:392-396); buffers are not parameters."""
    assert adapt_mode == 'meta'
    return [k for k in P if 'meta' in k and not k.endswith(('running_mean', 'running_var',
                                                              'num_batches_tracked'))]


class AdamState:
    """torch.optim.Adam semantics (src/tta_main.py:341-346): L2 weight decay folded into the
    gradient, bias-corrected, eps added after the sqrt(v)/sqrt(1-b2^t) division."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        # torch keys its state by parameter: a tensor listed twice (the CostDCNet DDP list names ResBlock.norm3 twice) shares ONE
        # (exp_avg, exp_avg_sq, step) and is updated once per occurrence, each with its own step count
        self._state = {}
        for p in params:
            self._state.setdefault(id(p), [torch.zeros_like(p), torch.zeros_like(p), 0])
        self.m = [self._state[id(p)][0] for p in params]
        self.v = [self._state[id(p)][1] for p in params]

    def step(self, params, grads):
        b1, b2 = self.betas
        self.t += 1
        with torch.no_grad():
            for p, g in zip(params, grads):
                st = self._state[id(p)]
                m, v = st[0], st[1]
                st[2] += 1
                bc1 = 1 - b1 ** st[2]
                bc2 = 1 - b2 ** st[2]
                if self.wd != 0:
                    g = g + self.wd * p
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(m, denom, value=-self.lr / bc1)


class MsgChnOracle:
    """Stateful wrapper: one object = model + Adam, ``step()`` = src/tta_main.py:583-633,
    ``forward_eval`` = :729-736."""

    def __init__(self, state_dict, prepare_mode='meta_selfsup_seq_1layer_ema', adapt_mode='meta',
                 max_input_depth=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 w_sd=1.0, w_sm=1.0, w_cos=1.0):
        self.P = {k: torch.as_tensor(v).clone() for k, v in state_dict.items()}
        self.prepare_mode = prepare_mode
        self.max_input_depth = max_input_depth
        self.names = adapted_names(self.P, adapt_mode)
        for k in self.names:
            self.P[k].requires_grad_(True)
        self.opt = AdamState([self.P[k] for k in self.names], lr, betas, eps, weight_decay)
        self.w = (w_sd, w_sm, w_cos)

    def forward_train(self, image, sparse_depth):
        return model_forward(self.P, image, sparse_depth, True, self.max_input_depth,
                             self.prepare_mode)

    def forward_eval(self, image, sparse_depth):
        with torch.no_grad():
            return model_forward(self.P, image, sparse_depth, False, self.max_input_depth,
                                 self.prepare_mode)

    def step(self, image, sparse_depth, validity_map=None, loss_image=None):
        """One TTA step. ``image`` feeds the network, ``loss_image`` (default: image) feeds the
        smoothness weights (src/tta_main.py:610 vs :620)."""
        if validity_map is None:
            validity_map = torch.where(sparse_depth > 0, torch.ones_like(sparse_depth), sparse_depth)
        if loss_image is None:
            loss_image = image
        depth, emb, ref = self.forward_train(image, sparse_depth)
        loss, info = adapt_loss(loss_image, depth, sparse_depth, validity_map, emb, ref,
                                *self.w, max_input_depth=self.max_input_depth)
        params = [self.P[k] for k in self.names]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for p, g in zip(params, grads)]
        self.opt.step(params, grads)
        return {'depth': depth.detach(), 'emb': emb.detach(), 'ref': ref.detach(),
                'loss_info': {k: float(v.detach()) for k, v in info.items()},
                'grads': {k: g for k, g in zip(self.names, grads)}}


def eval_metrics(output_depth, ground_truth, min_evaluate_depth, max_evaluate_depth):
    """src/tta_main.py:779-798 with src/eval_utils.py:117-174 (eps = 1e-9, eval_utils.py:24):
    MAE / RMSE on 1000x (mm), iMAE / iRMSE on 0.001x (1/km), over the validity mask."""
    eps = 1e-9
    mask = torch.where(ground_truth > 0, torch.ones_like(ground_truth), torch.zeros_like(ground_truth))
    mask[ground_truth < min_evaluate_depth] = 0.0
    mask[ground_truth > max_evaluate_depth] = 0.0
    idx = mask.nonzero(as_tuple=True)
    o, g = output_depth[idx], ground_truth[idx]
    mae = torch.mean(torch.abs(1000.0 * g - 1000.0 * o))
    rmse = torch.sqrt(torch.mean((1000.0 * g - 1000.0 * o) ** 2))
    imae = torch.mean(torch.abs(1.0 / (0.001 * g + eps) - 1.0 / (0.001 * o + eps)))
    irmse = torch.sqrt(torch.mean((1.0 / (0.001 * g + eps) - 1.0 / (0.001 * o + eps)) ** 2))
    return torch.stack([mae, rmse, imae, irmse])


# ---- modulated deformable convolution (NLSPN's native extension) --------------------------------------
# parity unpinned by the reference itself: its CPU path is an AT_ERROR stub and the CUDA build cannot run
# here; this restatement follows external_src/NLSPN/src/model/deformconv/src/cuda/
# modulated_deform_im2col_cuda.cuh:25-54 (bilinear), :128-194 (im2col, boundary rule :180) and
# modulated_deform_conv_cuda.cu:19-121 (per-group GEMM + bias), and is pinned by the properties of the
# reference's own deformconv/test.py (zero offset == conv, :69-110; gradients via autograd of this
# differentiable formulation vs the analytic kernels of :57-125,:197-328).
def mdconv_forward(x, weight, bias, offset, mask, stride=1, pad=1, dil=1, group=1, dg=1):
    B, C, H, W = x.shape
    Co, cpg, kh, kw = weight.shape
    K = kh * kw
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    ys = (torch.arange(Ho, dtype=x.dtype) * stride - pad).view(1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * stride - pad).view(1, 1, Wo)
    cpd = C // dg
    cols = []
    for k in range(K):
        i, j = k // kw, k % kw
        per_dg = []
        for d in range(dg):
            h = ys + i * dil + offset[:, d * 2 * K + 2 * k]
            w = xs + j * dil + offset[:, d * 2 * K + 2 * k + 1]
            inside = ((h > -1) & (w > -1) & (h < H) & (w < W)).to(x.dtype)
            h0, w0 = torch.floor(h).detach(), torch.floor(w).detach()
            lh, lw = h - h0, w - w0
            h0, w0 = h0.long(), w0.long()
            xd = x[:, d * cpd:(d + 1) * cpd]                               # (B, cpd, H, W)
            flat = xd.reshape(B, cpd, H * W)

            def corner(hh, ww):
                ok = ((hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)).to(x.dtype)
                idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1)).view(B, 1, Ho * Wo).expand(B, cpd, Ho * Wo)
                return flat.gather(2, idx).view(B, cpd, Ho, Wo) * ok.unsqueeze(1)
            v = ((1 - lh) * (1 - lw)).unsqueeze(1) * corner(h0, w0) + ((1 - lh) * lw).unsqueeze(1) * corner(h0, w0 + 1) + \
                (lh * (1 - lw)).unsqueeze(1) * corner(h0 + 1, w0) + (lh * lw).unsqueeze(1) * corner(h0 + 1, w0 + 1)
            per_dg.append(v * (inside * mask[:, d * K + k]).unsqueeze(1))
        cols.append(torch.cat(per_dg, 1))                                   # (B, C, Ho, Wo)
    col = torch.stack(cols, 2)                                              # (B, C, K, Ho, Wo)
    opg, cg = Co // group, C // group
    outs = []
    for g in range(group):
        wg = weight[g * opg:(g + 1) * opg].reshape(opg, cg * K)
        cg_col = col[:, g * cg:(g + 1) * cg].reshape(B, cg * K, Ho * Wo)
        outs.append(torch.einsum('ok,bkp->bop', wg, cg_col).view(B, opg, Ho, Wo))
    out = torch.cat(outs, 1)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out

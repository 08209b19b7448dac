"""ORACLE — test infrastructure only, never shipped, never on the product path.

CPU (PyTorch fp32) restatement of the ProxyTTA per-frame step for the NLSPN backbone (SURVEY.md §8 row a16,
BASELINE config 3).  Same import rule as ``oracle/proxytta_oracle.py``: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.

Parity pin: ``tests/golden/make_golden_nlspn.py`` imports the real reference from /root/reference in the build
container (CPU shims; torchvision's ResNet34 rebuilt from the reference's own BasicBlock; module ``DCN`` backed by
``proxytta_oracle.mdconv_forward`` because the reference's deformable convolution has no CPU implementation —
SURVEY.md §8c) and commits inputs/outputs as ``tests/golden/nlspn_*.npz``; ``tests/test_oracle_golden.py`` checks
this restatement against them.  The propagation arithmetic itself (modulated deformable convolution) is therefore
pinned by the property tests of the reference's ``deformconv/test.py`` only, not by reference outputs.

Paths below are relative to the reference root; NM = external_src/NLSPN/src/model/nlspnmodel_adapt.py,
CM = external_src/NLSPN/src/model/common.py, AD = src/nlspn_model_adapt.py.
The network is a flat functional program over a ``{name: tensor}`` state dict.
"""
import torch
import torch.nn.functional as F

from oracle.proxytta_oracle import AdamState, adapt_loss, mdconv_forward

BN_EPS = 1e-5


BN_RUNNING = [False]       # stage-2 head trainer only (head_forward below): BatchNorm2d in eval mode, from the loaded running statistics


def _bn(P, pre, x):
    """BatchNorm2d after adapt_parameters('meta_bn') (AD:322-337): running statistics dropped, so batch statistics
    are used in train AND eval mode.  Under `train(prepare=True)` (AD:360-368, stage 2) the module is in eval mode instead."""
    if BN_RUNNING[0]:
        return F.batch_norm(x, P[pre + '.running_mean'], P[pre + '.running_var'], P[pre + '.weight'], P[pre + '.bias'], False, 0.1, BN_EPS)
    return F.batch_norm(x, None, None, P[pre + '.weight'], P[pre + '.bias'], True, 0.1, BN_EPS)


def _cbr(P, pre, x, stride=1, bn=True, relu=True):
    """conv_bn_relu (CM:45-61): Conv2d(k=3, pad=1, bias=not bn) [-BN] [-LeakyReLU(0.2)]."""
    x = F.conv2d(x, P[pre + '.0.weight'], None if bn else P[pre + '.0.bias'], stride=stride, padding=1)
    if bn:
        x = _bn(P, pre + '.1', x)
    return F.leaky_relu(x, 0.2) if relu else x


def _ctbr(P, pre, x):
    """convt_bn_relu (CM:64-80): ConvTranspose2d(3, stride 2, pad 1, output_padding 1, no bias)-BN-LeakyReLU(0.2)."""
    x = F.conv_transpose2d(x, P[pre + '.0.weight'], None, stride=2, padding=1, output_padding=1)
    return F.leaky_relu(_bn(P, pre + '.1', x), 0.2)


def _basic_block(P, pre, x, stride):
    """BasicBlock.forward (NM:98-116): conv-bn-relu-conv-bn (+ 1x1/bn downsample) + identity, relu."""
    out = F.conv2d(x, P[pre + '.conv1.weight'], None, stride=stride, padding=1)
    out = F.relu(_bn(P, pre + '.bn1', out))
    out = _bn(P, pre + '.bn2', F.conv2d(out, P[pre + '.conv2.weight'], None, padding=1))
    if (pre + '.downsample.0.weight') in P:
        x = _bn(P, pre + '.downsample.1', F.conv2d(x, P[pre + '.downsample.0.weight'], None, stride=stride))
    return F.relu(out + x)


RESNET34 = ((2, 3, 1), (3, 4, 2), (4, 6, 2), (5, 3, 2))        # (conv index, blocks, stride of the first block)


def encoder(P, image, sparse_depth):
    """fe1 .. fe6 (NM:866-877): conv1_rgb -> conv1_rgb_meta (the adapted 48->48 conv at full resolution, NM:1371),
    conv1_dep (conv1_dep_meta is Identity), cat -> ResNet34 layer1..4 -> conv6."""
    rgb = F.leaky_relu(F.conv2d(image, P['conv1_rgb.0.weight'], P['conv1_rgb.0.bias'], padding=1), 0.2)
    rgb = F.conv2d(rgb, P['conv1_rgb_meta.weight'], P['conv1_rgb_meta.bias'], padding=1)
    dep = F.leaky_relu(F.conv2d(sparse_depth, P['conv1_dep.0.weight'], P['conv1_dep.0.bias'], padding=1), 0.2)
    fe = [torch.cat((rgb, dep), 1)]
    x = fe[0]
    for idx, nblocks, stride in RESNET34:
        for b in range(nblocks):
            x = _basic_block(P, 'conv%d.%d' % (idx, b), x, stride if b == 0 else 1)
        fe.append(x)
    fe.append(_cbr(P, 'conv6', x, stride=2))
    return fe                                                   # [fe1, fe2, fe3, fe4, fe5, fe6]


def _concat(fd, fe):
    """NM:474-490: crop the decoder map to the encoder map, then cat."""
    return torch.cat((fd[:, :, :fe.shape[2], :fe.shape[3]], fe), 1)


def decoder(P, fe):
    """Shared decoder and the three heads (NM:878-897): initial depth, guidance (8 ch), confidence (sigmoid)."""
    fe1, fe2, fe3, fe4, fe5, fe6 = fe
    fd5 = _ctbr(P, 'dec5', fe6)
    fd4 = _ctbr(P, 'dec4', _concat(fd5, fe5))
    fd3 = _ctbr(P, 'dec3', _concat(fd4, fe4))
    fd2 = _ctbr(P, 'dec2', _concat(fd3, fe3))
    x = _concat(fd2, fe2)
    pred_init = _cbr(P, 'id_dec0', _concat(_cbr(P, 'id_dec1', x), fe1), bn=False, relu=True)
    guide = _cbr(P, 'gd_dec0', _concat(_cbr(P, 'gd_dec1', x), fe1), bn=False, relu=False)
    cf = _concat(_cbr(P, 'cf_dec1', x), fe1)
    confidence = torch.sigmoid(F.conv2d(cf, P['cf_dec0.0.weight'], P['cf_dec0.0.bias'], padding=1))
    return pred_init, guide, confidence


def offset_affinity(P, guidance, confidence, legacy=False, k_f=3):
    """NLSPN._get_offset_affinity (NM:255-338) for affinity='TGASS', conf_prop=True."""
    B, _, H, W = guidance.shape
    num = k_f * k_f - 1
    oa = F.conv2d(guidance, P['prop_layer.conv_offset_aff.weight'], P['prop_layer.conv_offset_aff.bias'], padding=1)
    o1, o2, aff = torch.chunk(oa, 3, dim=1)
    # offsets are stored as (dy, dx) pairs per neighbour: view(B, num, 2, H, W) of cat(o1, o2) (NM:262-267) --
    # i.e. pair n = channels (2n, 2n+1) of the 16-channel cat, NOT (o1[n], o2[n]); a zero pair is inserted for the
    # centre tap
    off = torch.cat((o1, o2), 1).view(B, num, 2, H, W)
    zero = torch.zeros((B, 1, 2, H, W), dtype=off.dtype)
    off = torch.cat((off[:, :num // 2], zero, off[:, num // 2:]), 1).view(B, -1, H, W)
    aff = torch.tanh(aff) / (P['prop_layer.aff_scale_const'] + 1e-8)
    # confidence of each neighbour: 1x1 modulated deformable gather at the (detached) offset (NM:287-311)
    ones = torch.ones((B, 1, H, W), dtype=off.dtype)
    confs = []
    for idx in range(num + 1):
        ww, hh = idx % k_f, idx // k_f
        if ww == (k_f - 1) / 2 and hh == (k_f - 1) / 2:
            continue
        o = off[:, 2 * idx:2 * idx + 2].detach().clone()
        if legacy:
            o[:, 0] = o[:, 0] + hh - (k_f - 1) / 2
            o[:, 1] = o[:, 1] + ww - (k_f - 1) / 2
        confs.append(mdconv_forward(confidence, P['prop_layer.w_conf'], P['prop_layer.b'], o, ones, 1, 0, 1, 1, 1))
    aff = aff * torch.cat(confs, 1)
    # normalisation (NM:313-328): divide by max(sum|aff| + 1e-4, 1); centre weight = 1 - sum
    s = torch.sum(torch.abs(aff), dim=1, keepdim=True) + 1e-4
    s = torch.where(s < 1.0, torch.ones_like(s), s)
    aff = aff / s
    ref = 1.0 - torch.sum(aff, dim=1, keepdim=True)
    aff = torch.cat((aff[:, :num // 2], ref, aff[:, num // 2:]), 1)
    return off, aff


def propagate(P, feat_init, guidance, confidence, feat_fix, prop_time=18, legacy=False):
    """NLSPN.forward (NM:340-373): prop_time x {re-impose the sparse input, 3x3 modulated deformable conv with
    weight = ones, bias = 0 (NM:239-244)}."""
    off, aff = offset_affinity(P, guidance, confidence, legacy)
    mask_fix = (torch.sum(feat_fix > 0.0, dim=1, keepdim=True).detach() > 0.0).type_as(feat_fix)
    feat = feat_init
    for _ in range(prop_time):
        feat = (1.0 - mask_fix) * feat + mask_fix * feat_fix
        feat = mdconv_forward(feat, P['prop_layer.w'], P['prop_layer.b'], off, aff, 1, 1, 1, 1, 1)
    return feat, off, aff


def mlp(P, prefix, x):
    """MLP (NM:1398-1404): Linear-BatchNorm1d(train: batch statistics)-ReLU-Linear."""
    h = F.linear(x, P[prefix + '.0.weight'], P[prefix + '.0.bias'])
    h = F.batch_norm(h, None, None, P[prefix + '.1.weight'], P[prefix + '.1.bias'], True, 0.1, BN_EPS)
    return F.linear(F.relu(h), P[prefix + '.3.weight'], P[prefix + '.3.bias'])


def network_forward(P, image, sparse_depth, training, prop_time=18, legacy=False, want_intermediates=False):
    """NLSPNModel_Adapt._rgbd_meta_contrast (NM:850-944), mode = ['adapt', 'seq', 'reverse', 'ema'] (NM:587-608)."""
    fe = encoder(P, image, sparse_depth)
    pred_init, guide, confidence = decoder(P, fe)
    y, off, aff = propagate(P, pred_init, guide, confidence, sparse_depth, prop_time, legacy)
    depth = torch.clamp(y, min=0)
    if not training:
        return depth
    with torch.no_grad():
        fe_n = encoder(P, torch.zeros_like(image), sparse_depth)            # proxy pass (NM:907-916)
    dim = fe[5].shape[1]
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, dim)
    emb = mlp(P, 'pred', mlp(P, 'proj', rows(fe_n[5]).detach()))            # NM:932-933
    ref = mlp(P, 'proj_t', rows(fe[5]))                                      # NM:934 (carries the gradient)
    if want_intermediates:
        return depth, emb, ref, {'pred_init': pred_init, 'guide': guide, 'confidence': confidence, 'offset': off,
                                 'aff': aff, 'fe6': fe[5]}
    return depth, emb, ref


def model_forward(P, image, sparse_depth, training, max_input_depth=None, prop_time=18, legacy=False):
    """ExternalModel_Adapt.forward (src/external_model_adapt.py:82-114) + NLSPNModel_Adapt.forward (AD:88-128).
    The eval branch's biharmonic hole filling (AD:124-127, skimage, CPU) only acts when the clamped output
    contains exact zeros; it is outside this restatement (callers check for zeros)."""
    if max_input_depth is not None:
        sparse_depth = torch.clamp(sparse_depth, 0, max_input_depth)
    return network_forward(P, image, sparse_depth, training, prop_time, legacy)


def adapted_names(P, syncbn=False):
    """adapt_parameters('meta_bn') (AD:322-337) without SyncBatchNorm conversion: every parameter whose name
    contains 'meta', then weight/bias of every BatchNorm2d in module order (the heads' BatchNorm1d are not
    BatchNorm2d; after convert_syncbn (src/tta_main.py:326) they would be SyncBatchNorm and join the list)."""
    names = [k for k in P if 'meta' in k]
    for k in P:
        if k.endswith('.running_mean') and (syncbn or not k.startswith(('proj', 'pred'))):
            pre = k[:-len('.running_mean')]
            names += [pre + '.weight', pre + '.bias']
    return names


class NlspnOracle:
    """model + Adam; ``step()`` = src/tta_main.py:583-633, ``forward_eval`` = :729-736."""

    def __init__(self, state_dict, max_input_depth=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 w_sd=1.0, w_sm=1.0, w_cos=1.0, prop_time=18, legacy=False, syncbn_adapted=False):
        self.P = {k: torch.as_tensor(v).clone() for k, v in state_dict.items()}
        self.max_input_depth = max_input_depth
        self.prop_time, self.legacy = prop_time, legacy
        self.names = adapted_names(self.P, syncbn_adapted)        # True: the 94-tensor list of the reference's DDP run
        for k in self.names:
            self.P[k].requires_grad_(True)
        self.opt = AdamState([self.P[k] for k in self.names], lr, betas, eps, weight_decay)
        self.w = (w_sd, w_sm, w_cos)

    def forward_train(self, image, sparse_depth):
        return model_forward(self.P, image, sparse_depth, True, self.max_input_depth, self.prop_time, self.legacy)

    def forward_eval(self, image, sparse_depth):
        with torch.no_grad():
            return model_forward(self.P, image, sparse_depth, False, self.max_input_depth, self.prop_time, self.legacy)

    def step(self, image, sparse_depth, validity_map=None, loss_image=None):
        if validity_map is None:
            validity_map = torch.where(sparse_depth > 0, torch.ones_like(sparse_depth), sparse_depth)
        if loss_image is None:
            loss_image = image
        depth, emb, ref = self.forward_train(image, sparse_depth)
        loss, info = adapt_loss(loss_image, depth, sparse_depth, validity_map, emb, ref, *self.w,
                                max_input_depth=self.max_input_depth)
        params = [self.P[k] for k in self.names]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for p, g in zip(params, grads)]
        self.opt.step(params, grads)
        return {'depth': depth.detach(), 'emb': emb.detach(), 'ref': ref.detach(),
                'loss_info': {k: float(v.detach()) for k, v in info.items()},
                'grads': {k: g for k, g in zip(self.names, grads)}}


# ---- stage-2 head trainer: src/head_main.py:464-480 on NLSPNModel_Adapt._rgbd_meta_contrast_prepare (NM:1014-1060) ----
_BUF = ('running_mean', 'running_var', 'num_batches_tracked')


def head_names(P):
    return [k for k in P if k.startswith(('proj.', 'pred.')) and not k.endswith(_BUF)]


def update_head(P, tau=0.999):
    """_update_head (NM:1314-1316): proj_t <- tau proj_t + (1 - tau) proj over parameters()."""
    with torch.no_grad():
        for k in P:
            if k.startswith('proj_t.') and not k.endswith(_BUF):
                P[k].copy_(P[k] * tau + P['proj.' + k[len('proj_t.'):]] * (1.0 - tau))


def mlp_train(P, prefix, x):
    """MLP with its BatchNorm1d in train mode AND tracked: batch statistics, running statistics updated in place (momentum 0.1)."""
    h = F.linear(x, P[prefix + '.0.weight'], P[prefix + '.0.bias'])
    h = F.batch_norm(h, P[prefix + '.1.running_mean'], P[prefix + '.1.running_var'], P[prefix + '.1.weight'], P[prefix + '.1.bias'], True, 0.1, BN_EPS)
    P[prefix + '.1.num_batches_tracked'] += 1
    return F.linear(F.relu(h), P[prefix + '.3.weight'], P[prefix + '.3.bias'])


def head_forward(P, image, sparse_depth, reverse, max_input_depth=None, tau=0.999, features=None):
    """NM:1028-1058.  `features(P, image, sparse)` -> rows tensor [R, C] of one backbone pass (default: fe6 of this file's encoder, BatchNorm2d in
    eval mode as train(prepare=True) leaves it, AD:360-368); the zero-image pass is the same function on zeros_like(image)."""
    if max_input_depth is not None:
        sparse_depth = torch.clamp(sparse_depth, 0, max_input_depth)
    if features is None:
        def features(P_, img, sd):
            BN_RUNNING[0] = True
            try:
                fe6 = encoder(P_, img, sd)[5]
            finally:
                BN_RUNNING[0] = False
            return fe6.permute(0, 2, 3, 1).reshape(-1, fe6.shape[1])
    with torch.no_grad():
        rows = features(P, image, sparse_depth)
        rows_n = features(P, torch.zeros_like(image), sparse_depth)
    update_head(P, tau)
    a, b = (rows_n, rows) if reverse else (rows, rows_n)
    emb = mlp_train(P, 'pred', mlp_train(P, 'proj', a.detach()))
    with torch.no_grad():
        ref = mlp_train(P, 'proj_t', b)
    return emb, ref


class HeadTrainerOracle:
    """Adam over prepare_parameters('head_selfsup_ema') (AD:261-265): the twelve proj.* / pred.* tensors; all of them receive a gradient in
    both directions (the reference branch goes through proj_t)."""

    def __init__(self, state_dict, loss_type='head_selfsup_seq_ema_reverse', max_input_depth=None, lr=2e-4, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, tau=0.999, features=None):
        self.reverse = 'reverse' in loss_type
        self.P = {k: torch.as_tensor(v).clone() for k, v in state_dict.items()}
        self.names = head_names(self.P)
        for k in self.names:
            self.P[k].requires_grad_(True)
        self.max_input_depth, self.tau, self.features = max_input_depth, tau, features
        self.opt = AdamState([self.P[k] for k in self.names], lr, betas, eps, weight_decay)

    def step(self, image, sparse_depth):
        emb, ref = head_forward(self.P, image, sparse_depth, self.reverse, self.max_input_depth, self.tau, self.features)
        e, r = F.normalize(emb, dim=-1, p=2), F.normalize(ref, dim=-1, p=2)
        loss = (2 - 2 * (e * r).sum(-1)).mean()                       # prepare_loss, src/external_model_adapt.py:524-541
        params = [self.P[k] for k in self.names]
        grads = torch.autograd.grad(loss, params)
        self.opt.step(params, grads)
        return {'loss': float(loss.detach()), 'emb': emb.detach(), 'ref': ref.detach(), 'grads': dict(zip(self.names, grads))}

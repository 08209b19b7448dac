"""ORACLE — test infrastructure only, never shipped, never on the product path.

CPU (PyTorch fp32) restatement of the ProxyTTA per-frame step for the CostDCNet backbone (SURVEY.md §8 row a17,
BASELINE config 5).  Same import rule as ``oracle/proxytta_oracle.py``: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file.

Parity pin: ``tests/golden/make_golden_costdcnet.py`` imports the REAL reference from /root/reference in the build container
(CPU shims + ``oracle/minkowski_lite.py`` registered as the absent third-party ``MinkowskiEngine``) and commits
``tests/golden/costdcnet_*.npz``; ``tests/test_oracle_golden.py`` checks this restatement against them.  Everything that is
plain torch in the reference (Encoder2D, the meta conv, fusion, the P3D UNet3D, pixel-shuffle softmax regression, heads,
loss, Adam, dual-corner padding) is therefore pinned by reference outputs; the sparse encoder's arithmetic (MinkowskiEngine)
is **parity unpinned** — both sides run minkowski_lite's restatement of its published semantics.

Paths are relative to the reference root; CD = external_src/costdcnet/CostDCNet_adapt.py,
E2 = external_src/costdcnet/models/encoder2d.py, E3 = .../models/encoder3d.py, U3 = .../models/unet3d.py,
AD = src/costdcnet_model_adapt.py.  The network is a flat functional program over a ``{name: tensor}`` state dict.
"""
import torch
import torch.nn.functional as F

from oracle import minkowski_lite as ML
from oracle.proxytta_oracle import AdamState, adapt_loss

BN_EPS = 1e-5
RES, UP_SCALE = 16, 4            # AD:47-52 (args.res, args.up_scale)


BN2D_RUNNING = [False]     # stage-2 head trainer only (head_rows below): BatchNorm2d in eval mode, from the loaded running statistics


def _bn2d(P, pre, x):
    """BatchNorm2d after adapt_parameters('meta_bn') (AD:357-378): running statistics dropped -> batch statistics in
    train AND eval mode.  Under `train(prepare=True)` (AD:418-430, stage 2) the module is in eval mode with its buffers intact."""
    if BN2D_RUNNING[0]:
        return F.batch_norm(x, P[pre + '.running_mean'], P[pre + '.running_var'], P[pre + '.weight'], P[pre + '.bias'], False, 0.1, BN_EPS)
    return F.batch_norm(x, None, None, P[pre + '.weight'], P[pre + '.bias'], True, 0.1, BN_EPS)


def _bn_tracked(P, pre, x, training):
    """BatchNorm3d / BatchNorm1d that keep their running statistics: batch statistics + momentum-0.1 update in train mode,
    running statistics in eval mode."""
    if P.get('__syncbn_adapted__', False):
        # the DDP run (adapted_names(P, syncbn=True)): converted to SyncBatchNorm and caught by adapt_parameters('meta_bn'), the layer
        # lost its running statistics -> batch statistics in train AND eval mode, nothing to update (AD:364-372)
        return F.batch_norm(x, None, None, P[pre + '.weight'], P[pre + '.bias'], True, 0.1, BN_EPS)
    if training:
        P[pre + '.num_batches_tracked'] += 1
    return F.batch_norm(x, P[pre + '.running_mean'], P[pre + '.running_var'], P[pre + '.weight'], P[pre + '.bias'],
                        training, 0.1, BN_EPS)


def _resblock(P, pre, x, stride):
    """ResBlock.forward (E2:44-52): y = relu(bn(conv)); y = relu(bn(conv)); x = downsample(x); relu(x + y)."""
    y = F.relu(_bn2d(P, pre + '.norm1', F.conv2d(x, P[pre + '.conv1.weight'], P[pre + '.conv1.bias'], stride=stride, padding=1)))
    y = F.relu(_bn2d(P, pre + '.norm2', F.conv2d(y, P[pre + '.conv2.weight'], P[pre + '.conv2.bias'], padding=1)))
    if stride != 1:
        x = _bn2d(P, pre + '.norm3', F.conv2d(x, P[pre + '.downsample.0.weight'], P[pre + '.downsample.0.bias'], stride=stride))
    return F.relu(x + y)


def encoder2d(P, x):
    """Encoder2D.forward (E2:88-102) with meta_bn_rgb / conv1_rgb_meta unset inside the encoder (CD:485-489)."""
    x = F.relu(_bn2d(P, 'enc2d.norm1', F.conv2d(x, P['enc2d.conv1.weight'], P['enc2d.conv1.bias'], padding=1)))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _resblock(P, 'enc2d.layer%d.0' % li, x, stride)
        x = _resblock(P, 'enc2d.layer%d.1' % li, x, 1)
    return F.conv2d(x, P['enc2d.conv2.weight'], P['enc2d.conv2.bias'])


def depth2mdp(dep, z_step):
    """CostDCNet.depth2MDP (CD:356-388): one voxel (plane index, y, x) per pixel whose rounded plane index is non-zero,
    feature = residual to the plane in plane units."""
    idx = torch.round(dep / z_step).long().clamp(0, RES - 1)
    res_map = (dep - idx * z_step) / z_step
    B, _, H, W = dep.shape
    gy, gx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    b = torch.arange(B).view(B, 1, 1).expand(B, H, W)
    m = idx[:, 0] != 0
    coords = torch.stack([b[m], idx[:, 0][m], gy.expand(B, H, W)[m], gx.expand(B, H, W)[m]], 1).float()
    return ML.TensorField(features=res_map[:, 0][m].reshape(-1, 1), coordinates=coords).sparse()


def _me_bn(P, pre, x, training):
    return x._like(_bn_tracked(P, pre + '.bn', x.F, training))


def _me_block(P, pre, x, stride, training):
    """MinkowskiEngine BasicBlock as Encoder3D._make_layer builds it (E3:52-91)."""
    out = ML.sparse_conv(x, P[pre + '.conv1.kernel'], 3, stride)
    out = _me_bn(P, pre + '.norm1', out, training)
    out = out._like(torch.relu(out.F))
    out = _me_bn(P, pre + '.norm2', ML.sparse_conv(out, P[pre + '.conv2.kernel'], 3, 1), training)
    res = x
    if pre + '.downsample.0.kernel' in P:
        res = _me_bn(P, pre + '.downsample.1', ML.sparse_conv(x, P[pre + '.downsample.0.kernel'], 1, stride), training)
    out = out + res
    return out._like(torch.relu(out.F))


def encoder3d(P, x, training):
    """Encoder3D.forward (E3:93-103)."""
    out = _me_bn(P, 'enc3d.bn0', ML.sparse_conv(x, P['enc3d.conv1.kernel'], 3, 1), training)
    out = out._like(torch.relu(out.F))
    out = _me_block(P, 'enc3d.block1.0', out, 1, training)
    out = _me_block(P, 'enc3d.block2.0', out, (1, 2, 2), training)
    out = _me_block(P, 'enc3d.block3.0', out, (1, 2, 2), training)
    return ML.sparse_conv(out, P['enc3d.conv2.kernel'], 1, 1)


def fusion(sout, feat2d):
    """CostDCNet.fusion (CD:390-406): dense 3-D features at the origin of a (B,16,res,H/4,W/4) volume; image features
    broadcast over the planes and gated by the occupancy mask (mask + 1 - number of occupied planes of the pixel)."""
    B0, C0, H0, W0 = feat2d.shape
    dense, _, _ = sout.dense()
    dense = dense[:, :, :RES, :H0, :W0]
    B, C, D, H, W = dense.shape
    feat3d = torch.zeros((B0, C0, RES, H0, W0))
    feat3d[:B, :, :D, :H, :W] += dense
    mask = (torch.sum(feat3d != 0, dim=1, keepdim=True) != 0).float()
    mask_ = mask + (1 - torch.sum(mask, dim=2, keepdim=True).repeat(1, 1, mask.size(2), 1, 1))
    return torch.cat([feat2d.unsqueeze(2).repeat(1, 1, RES, 1, 1) * mask_, feat3d], dim=1)


def _p3d(P, pre, x, training):
    """P3D.forward (U3:76-84): 1x3x3 conv - BN3d - ELU - 3x1x1 conv - BN3d - ELU (no conv bias)."""
    x = F.elu(_bn_tracked(P, pre + '.bn1', F.conv3d(x, P[pre + '.conv1.weight'], None, padding=(0, 1, 1)), training))
    return F.elu(_bn_tracked(P, pre + '.bn2', F.conv3d(x, P[pre + '.conv2.weight'], None, padding=(1, 0, 0)), training))


def _double(P, pre, x, training):
    return _p3d(P, pre + '.double_conv.1', _p3d(P, pre + '.double_conv.0', x, training), training)


def unet3d(P, x, training):
    """UNet3D.forward(return_feature=True) (U3:29-47): Down = MaxPool3d(2) + DoubleConv, Up = nearest interpolate to the
    skip's size + cat([skip, up]) + DoubleConv."""
    x1 = _double(P, 'unet3d.inc', x, training)
    x2 = _double(P, 'unet3d.down1.maxpool_conv.1', F.max_pool3d(x1, 2), training)
    x3 = _double(P, 'unet3d.down2.maxpool_conv.1', F.max_pool3d(x2, 2), training)
    feat = _double(P, 'unet3d.down3.maxpool_conv.1', F.max_pool3d(x3, 2), training)
    x = feat
    for name, skip in (('up2', x3), ('up3', x2), ('up4', x1)):
        x = F.interpolate(x, size=skip.shape[2:], mode='nearest')
        x = _double(P, 'unet3d.%s.conv' % name, torch.cat([skip, x], dim=1), training)
    return F.conv3d(x, P['unet3d.classif0.weight'], P['unet3d.classif0.bias']), feat


def upsampling(cost):
    """CostDCNet.upsampling + disparity_regression (CD:408-424): per-plane pixel shuffle x4, softmax over the planes,
    expected plane index."""
    b, c, d, h, w = cost.shape
    cost = F.pixel_shuffle(cost.transpose(1, 2).reshape(b, -1, h, w), UP_SCALE)
    prop = F.softmax(cost, dim=1)
    return torch.sum(prop * torch.arange(0, RES, dtype=cost.dtype).view(1, RES, 1, 1), 1, keepdim=True)


def mlp(P, pre, x, training=True):
    """CostDCNet.MLP (CD:498-504): Linear - BatchNorm1d - ReLU - Linear; BN1d in train mode (batch statistics)."""
    x = F.linear(x, P[pre + '.0.weight'], P[pre + '.0.bias'])
    x = F.relu(_bn_tracked(P, pre + '.1', x, training))
    return F.linear(x, P[pre + '.3.weight'], P[pre + '.3.bias'])


def network_forward(P, image, sparse_depth, training, max_depth, want_intermediates=False):
    """CostDCNet._rgbd_meta_contrast (CD:207-256), mode = adapt / seq / reverse / ema, prepare_mode 1layer."""
    z_step = max_depth / (RES - 1)
    in_3d = depth2mdp(sparse_depth, z_step)
    feat2d = F.conv2d(encoder2d(P, torch.cat([image, sparse_depth], 1)), P['conv1_rgb_meta.weight'], P['conv1_rgb_meta.bias'], padding=1)
    feat3d = encoder3d(P, in_3d, training)
    vol = fusion(feat3d, feat2d)
    cost, feat = unet3d(P, vol, training)
    pred = upsampling(cost) * z_step
    if not training:
        return pred
    with torch.no_grad():                                    # proxy pass: zero image, same sparse depth (CD:233-240)
        f2z = F.conv2d(encoder2d(P, torch.cat([torch.zeros_like(image), sparse_depth], 1)), P['conv1_rgb_meta.weight'],
                       P['conv1_rgb_meta.bias'], padding=1)
        f3z = encoder3d(P, in_3d, training)
        _, feat_zero = unet3d(P, fusion(f3z, f2z), training)
    b, c, d, h, w = feat.shape
    rows = lambda t: t.reshape(b, c * d, h, w).permute(0, 2, 3, 1).reshape(-1, c * d)
    emb = mlp(P, 'pred', mlp(P, 'proj', rows(feat_zero))).detach()           # CD:250
    ref = mlp(P, 'proj_t', rows(feat))                                       # CD:251
    if want_intermediates:
        return pred, emb, ref, {'feat2d': feat2d, 'vol': vol, 'cost': cost, 'feat': feat}
    return pred, emb, ref


def model_forward(P, image, sparse_depth, training, max_depth, max_input_depth=None):
    """ExternalModel_Adapt.forward clamp (src/external_model_adapt.py:108) + CostDCNetModel_Adapt.forward (AD:62-114):
    dual-corner zero padding to multiples of 16 (batch doubling), network, crop both and average."""
    if max_input_depth is not None:
        sparse_depth = torch.clamp(sparse_depth, 0.0, max_input_depth)
    H, W = image.shape[-2:]
    pt, pr = (-H) % 16, (-W) % 16
    if pt or pr:
        image = torch.cat([F.pad(image, (0, pr, pt, 0)), F.pad(image, (pr, 0, 0, pt))], 0)
        sparse_depth = torch.cat([F.pad(sparse_depth, (0, pr, pt, 0)), F.pad(sparse_depth, (pr, 0, 0, pt))], 0)
    out = network_forward(P, image, sparse_depth, training, max_depth)
    depth = out[0] if training else out
    if pt or pr:
        o0, o1 = torch.chunk(depth, 2, 0)
        depth = (o0[:, :, pt:, :W] + o1[:, :, :H, pr:]) / 2.0
    return (depth, out[1], out[2]) if training else depth


def adapted_names(P, syncbn=False):
    """adapt_parameters('meta_bn') (AD:357-378): parameters whose name contains 'meta', then weight / bias of every
    BatchNorm2d in module order (Encoder2D only: BatchNorm3d / BatchNorm1d / MinkowskiBatchNorm are not BatchNorm2d);
    ResBlock.norm3 also sits inside `downsample` but a module is visited once.
    syncbn: the reference's DDP run calls convert_syncbn() first (src/tta_main.py:326,339): every BatchNorm is a SyncBatchNorm by then and
    matches the isinstance test -- Encoder2D's, the BatchNorm1d inside every MinkowskiBatchNorm, UNet3D's BatchNorm3d, the heads' BatchNorm1d --
    116 entries in module order.  convert_sync_batchnorm builds one new module per visited ATTRIBUTE, so ResBlock.norm3 and its alias
    downsample[1] become two modules sharing one Parameter: that parameter is listed twice (and Adam updates it twice per step)."""
    names = [k for k in P if 'meta' in k and not k.startswith('__')]
    for k in P:
        if not k.endswith('.running_mean'):
            continue
        pre = k[:-len('.running_mean')]
        if not syncbn:
            if k.startswith('enc2d.') and '.downsample.1.' not in k:
                names += [pre + '.weight', pre + '.bias']
            continue
        if pre.startswith('enc2d.') and '.downsample.1' in pre:
            pre = pre.replace('.downsample.1', '.norm3')          # the alias: same tensors, second listing
        names += [pre + '.weight', pre + '.bias']
    return names


class CostDcnOracle:
    """model + Adam; ``step()`` = src/tta_main.py:583-633, ``forward_eval`` = :729-736."""

    def __init__(self, state_dict, max_depth=8.0, max_input_depth=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 w_sd=1.0, w_sm=1.0, w_cos=1.0, syncbn=False):
        self.P = {k: torch.as_tensor(v).clone() for k, v in state_dict.items()}
        self.syncbn = syncbn
        for k in list(self.P):          # one BatchNorm behind two names (ResBlock.norm3 / downsample.1)
            if k.startswith('enc2d.') and '.downsample.1.' in k:
                self.P[k] = self.P[k.replace('.downsample.1.', '.norm3.')]
        self.max_depth, self.max_input_depth = max_depth, max_input_depth
        self.names = adapted_names(self.P, syncbn)
        if syncbn:
            self.P['__syncbn_adapted__'] = True
        for k in self.names:
            self.P[k].requires_grad_(True)
        self.opt = AdamState([self.P[k] for k in self.names], lr, betas, eps, weight_decay)
        self.w = (w_sd, w_sm, w_cos)

    def forward_train(self, image, sparse_depth):
        return model_forward(self.P, image, sparse_depth, True, self.max_depth, self.max_input_depth)

    def forward_eval(self, image, sparse_depth):
        with torch.no_grad():
            return model_forward(self.P, image, sparse_depth, False, self.max_depth, self.max_input_depth)

    def step(self, image, sparse_depth, validity_map=None, loss_image=None):
        if validity_map is None:
            validity_map = torch.where(sparse_depth > 0, torch.ones_like(sparse_depth), sparse_depth)
        if loss_image is None:
            loss_image = image
        depth, emb, ref = self.forward_train(image, sparse_depth)
        loss, info = adapt_loss(loss_image, depth, sparse_depth, validity_map, emb, ref, *self.w, max_input_depth=self.max_input_depth)
        params = [self.P[k] for k in self.names]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for p, g in zip(params, grads)]
        self.opt.step(params, grads)
        return {'depth': depth.detach(), 'emb': emb.detach(), 'ref': ref.detach(),
                'loss_info': {k: float(torch.as_tensor(v).detach()) for k, v in info.items()},
                'grads': {k: g for k, g in zip(self.names, grads)}}


# ---- stage-2 head trainer: src/head_main.py:464-480 on CostDCNet._rgbd_meta_contrast_prepare (CD:258-303) ----
def head_rows(P, image, sparse_depth, max_depth):
    """One no-gradient backbone pass up to the UNet3D bottleneck (CD:268-277), as rows [B * H/32 * W/32, 80 * 2].  `train(prepare=True)` (AD:418-430)
    puts the BatchNorm2d's (Encoder2D) into eval mode; BatchNorm3d and the sparse encoder's BatchNorm1d are not BatchNorm2d: train mode, batch
    statistics, running statistics updated by every pass."""
    z_step = max_depth / (RES - 1)
    BN2D_RUNNING[0] = True
    try:
        feat2d = F.conv2d(encoder2d(P, torch.cat([image, sparse_depth], 1)), P['conv1_rgb_meta.weight'], P['conv1_rgb_meta.bias'], padding=1)
    finally:
        BN2D_RUNNING[0] = False
    feat3d = encoder3d(P, depth2mdp(sparse_depth, z_step), True)
    _, feat = unet3d(P, fusion(feat3d, feat2d), True)
    b, c, d, h, w = feat.shape
    return feat.reshape(b, c * d, h, w).permute(0, 2, 3, 1).reshape(-1, c * d)


def make_head_trainer(state_dict, loss_type, max_depth=8.0, **kw):
    """oracle.nlspn_oracle.HeadTrainerOracle (the same EMA / heads / loss / Adam program: CD:283-303 = NM:1048-1058) over this backbone's rows."""
    from oracle.nlspn_oracle import HeadTrainerOracle
    sd = dict(state_dict)
    for k in list(sd):          # one BatchNorm behind two names (ResBlock.norm3 / downsample.1)
        if k.startswith('enc2d.') and '.downsample.1.' in k:
            sd[k] = sd[k.replace('.downsample.1.', '.norm3.')]
    return HeadTrainerOracle(sd, loss_type, features=lambda P, img, sp: head_rows(P, img, sp, max_depth), **kw)

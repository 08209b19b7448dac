"""Deterministic synthetic weights and frames for the ProxyTTA hot path.

Everything here is numpy integer hashing (splitmix64), so the same bytes come out on the
survey container, on the GPU box and inside the oracle: no 5.8 MB state_dict has to be
committed and no torch RNG (CPU and GPU generators differ) is involved.

The key/shape table mirrors the state_dict of the reference network after
``_prepare_head('meta_selfsup_seq_{1layer,2layers}_ema')``
(external_src/MSG_CHN/workspace/exp_msg_chn/network_exp_msg_chn_adapt.py:314-336, :1022-1087).
"""
import zlib

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def hash_uniform(tag, n):
    """n float64 values in [0,1), a pure function of (tag, index)."""
    seed = np.uint64(zlib.crc32(tag.encode('utf-8')))
    with np.errstate(over='ignore'):
        z = np.arange(n, dtype=np.uint64) * _GOLD + seed * _M1
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def _conv_keys(prefix, seq_idx, cout, cin):
    return [('%s.%d.weight' % (prefix, seq_idx), (cout, cin, 3, 3)),
            ('%s.%d.bias' % (prefix, seq_idx), (cout,))]


def _mlp_keys(prefix, din, dh, dout):
    return [(prefix + '.0.weight', (dh, din)), (prefix + '.0.bias', (dh,)),
            (prefix + '.1.weight', (dh,)), (prefix + '.1.bias', (dh,)),
            (prefix + '.1.running_mean', (dh,)), (prefix + '.1.running_var', (dh,)),
            (prefix + '.1.num_batches_tracked', ()),
            (prefix + '.3.weight', (dout, dh)), (prefix + '.3.bias', (dout,))]


def msg_chn_keys(prepare_mode='meta_selfsup_seq_1layer_ema'):
    """Ordered (name, shape) list of the MSG_CHN TTA network's state_dict."""
    keys = []
    keys += _conv_keys('rgb_encoder.init', 0, 32, 3) + _conv_keys('rgb_encoder.init', 2, 32, 32)
    for e in ('enc1', 'enc2', 'enc3', 'enc4'):
        keys += _conv_keys('rgb_encoder.' + e, 1, 32, 32) + _conv_keys('rgb_encoder.' + e, 3, 32, 32)
    for s, cin in ((1, 1), (2, 2), (3, 2)):
        enc = 'depth_encoder%d' % s
        keys += _conv_keys(enc + '.init', 0, 32, cin) + _conv_keys(enc + '.init', 2, 32, 32)
        for e in ('enc1', 'enc2'):
            keys += _conv_keys(enc + '.' + e, 1, 32, 32) + _conv_keys(enc + '.' + e, 3, 32, 32)
        dec = 'depth_decoder%d' % s
        for d in ('dec2', 'dec1'):
            # .1 is the ConvTranspose2d: weight is (Cin, Cout, 3, 3) = (32, 32, 3, 3)
            keys += _conv_keys(dec + '.' + d, 1, 32, 32) + _conv_keys(dec + '.' + d, 3, 32, 32)
        keys += _conv_keys(dec + '.prdct', 1, 32, 32) + _conv_keys(dec + '.prdct', 3, 1, 32)
    keys += _mlp_keys('proj', 32, 512, 512)
    keys += _mlp_keys('proj_t', 32, 512, 512)
    keys += _mlp_keys('pred', 512, 512, 512)
    if '2layers' in prepare_mode:
        p = 'conv1_rgb_meta.conv1_meta'
        keys += [(p + '.0.0.weight', (128, 32, 3, 3)),
                 (p + '.0.1.weight', (128,)), (p + '.0.1.bias', (128,)),
                 (p + '.0.1.running_mean', (128,)), (p + '.0.1.running_var', (128,)),
                 (p + '.0.1.num_batches_tracked', ()),
                 (p + '.1.weight', (32, 128, 3, 3)), (p + '.1.bias', (32,)),
                 (p + '.2.weight', (32,)), (p + '.2.bias', (32,)),
                 (p + '.2.running_mean', (32,)), (p + '.2.running_var', (32,)),
                 (p + '.2.num_batches_tracked', ())]
    else:
        keys += [('conv1_rgb_meta.weight', (32, 32, 3, 3)), ('conv1_rgb_meta.bias', (32,))]
    return keys


def _bn_keys(prefix, ch):
    return [(prefix + '.weight', (ch,)), (prefix + '.bias', (ch,)), (prefix + '.running_mean', (ch,)),
            (prefix + '.running_var', (ch,)), (prefix + '.num_batches_tracked', ())]


def nlspn_keys(prepare_mode='meta_selfsup_seq_1layer_ema'):
    """Ordered (name, shape) list of the NLSPN TTA network's state_dict: NLSPNModel_Adapt.__init__
    (external_src/NLSPN/src/model/nlspnmodel_adapt.py:376-452) with the ResNet34 stages [3,4,6,3] of
    BasicBlocks (:70-116), the propagation layer (:189-253) and the heads / meta layer of
    _prepare_head (:1333-1376)."""
    if '1layer' not in prepare_mode:
        raise NotImplementedError('NLSPN key table covers the canonical 1layer meta layer only')
    keys = [('conv1_rgb.0.weight', (48, 3, 3, 3)), ('conv1_rgb.0.bias', (48,)),
            ('conv1_dep.0.weight', (16, 1, 3, 3)), ('conv1_dep.0.bias', (16,))]
    inpl = 64
    for stage, (planes, nblocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
        for b in range(nblocks):
            pre = 'conv%d.%d' % (stage + 2, b)
            keys += [(pre + '.conv1.weight', (planes, inpl, 3, 3))] + _bn_keys(pre + '.bn1', planes)
            keys += [(pre + '.conv2.weight', (planes, planes, 3, 3))] + _bn_keys(pre + '.bn2', planes)
            if b == 0 and (stride != 1 or inpl != planes):
                keys += [(pre + '.downsample.0.weight', (planes, inpl, 1, 1))] + _bn_keys(pre + '.downsample.1', planes)
            inpl = planes
    keys += [('conv6.0.weight', (512, 512, 3, 3))] + _bn_keys('conv6.1', 512)
    for name, cin, cout in (('dec5', 512, 256), ('dec4', 768, 128), ('dec3', 384, 64), ('dec2', 192, 64)):
        keys += [(name + '.0.weight', (cin, cout, 3, 3))] + _bn_keys(name + '.1', cout)       # ConvTranspose2d: (Cin, Cout, 3, 3)
    keys += [('id_dec1.0.weight', (64, 128, 3, 3))] + _bn_keys('id_dec1.1', 64)
    keys += [('id_dec0.0.weight', (1, 128, 3, 3)), ('id_dec0.0.bias', (1,))]
    keys += [('gd_dec1.0.weight', (64, 128, 3, 3))] + _bn_keys('gd_dec1.1', 64)
    keys += [('gd_dec0.0.weight', (8, 128, 3, 3)), ('gd_dec0.0.bias', (8,))]
    keys += [('cf_dec1.0.weight', (32, 128, 3, 3))] + _bn_keys('cf_dec1.1', 32)
    keys += [('cf_dec0.0.weight', (1, 96, 3, 3)), ('cf_dec0.0.bias', (1,))]
    keys += [('prop_layer.aff_scale_const', (1,)), ('prop_layer.w', (1, 1, 3, 3)), ('prop_layer.b', (1,)),
             ('prop_layer.w_conf', (1, 1, 1, 1)),
             ('prop_layer.conv_offset_aff.weight', (24, 8, 3, 3)), ('prop_layer.conv_offset_aff.bias', (24,))]
    keys += _mlp_keys('proj', 512, 1024, 1024)
    keys += _mlp_keys('proj_t', 512, 1024, 1024)
    keys += _mlp_keys('pred', 1024, 1024, 1024)
    keys += [('conv1_rgb_meta.weight', (48, 48, 3, 3)), ('conv1_rgb_meta.bias', (48,))]
    return keys


def formula_state_dict_nlspn(prepare_mode='meta_selfsup_seq_1layer_ema', gain=1.0):
    sd = {k: formula_tensor(k, s, gain) for k, s in nlspn_keys(prepare_mode)}
    # fixed (non-trainable) constants of the propagation layer (nlspnmodel_adapt.py:227-247)
    sd['prop_layer.aff_scale_const'] = np.full((1,), 0.5 * 8, dtype=np.float32)
    sd['prop_layer.w'] = np.ones((1, 1, 3, 3), dtype=np.float32)
    sd['prop_layer.b'] = np.zeros((1,), dtype=np.float32)
    sd['prop_layer.w_conf'] = np.ones((1, 1, 1, 1), dtype=np.float32)
    # Conditioning of the heads, so the synthetic network behaves like a trained one instead of diverging in the 18
    # propagation sweeps: metres-scale sparse depth is brought to O(1) by conv1_dep, the initial depth sits at a
    # positive level (keeps the clamped output free of exact zeros, which the reference's eval path would send to
    # skimage's biharmonic inpainting), offsets are O(1) pixel and affinities mostly positive.
    sd['conv1_dep.0.weight'] = (sd['conv1_dep.0.weight'] / 40.0).astype(np.float32)
    sd['id_dec0.0.bias'] = np.full((1,), 10.0, dtype=np.float32)
    sd['gd_dec0.0.weight'] = (sd['gd_dec0.0.weight'] * 0.25).astype(np.float32)
    sd['cf_dec0.0.weight'] = (sd['cf_dec0.0.weight'] * 0.5).astype(np.float32)
    sd['prop_layer.conv_offset_aff.weight'] = (sd['prop_layer.conv_offset_aff.weight'] * 0.5).astype(np.float32)
    b = sd['prop_layer.conv_offset_aff.bias'].copy()
    b[16:] += 0.5
    sd['prop_layer.conv_offset_aff.bias'] = b
    return sd


def formula_tensor(name, shape, gain=1.0):
    """One state_dict entry, Kaiming-scaled so activations stay O(1) through ~50 layers."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(name, n) * 2.0 - 1.0
    if name.endswith('num_batches_tracked'):
        return np.zeros((), dtype=np.int64)
    if name.endswith('running_mean'):
        v = 0.1 * u
    elif name.endswith('running_var'):
        v = 1.0 + 0.2 * np.abs(u)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        v = u * np.sqrt(3.0) * gain * np.sqrt(2.0 / fan_in)
    elif name.endswith('bias'):
        # conv biases 0.01 + jitter (reference init: constant 0.01), BN beta small
        v = 0.01 + 0.02 * u
    else:
        v = 1.0 + 0.1 * u      # BN gamma
    return v.reshape(shape).astype(np.float32)


def formula_state_dict(prepare_mode='meta_selfsup_seq_1layer_ema', gain=1.0, head_bias=0.0):
    """head_bias != 0: the output biases of `proj` and `pred` become the SAME vector of that scale, so every embedding / reference
    row is dominated by one direction and L_cos is small -- the side of the `loss_cos < 0.3` gate (src/external_model_adapt.py:424-425)
    that trained heads sit on; plain formula weights give L_cos ~ 2."""
    sd = {k: formula_tensor(k, s, gain) for k, s in msg_chn_keys(prepare_mode)}
    if head_bias:
        v = (head_bias * (hash_uniform('head_bias', 512) * 2.0 - 1.0)).astype(np.float32)
        sd['proj.3.bias'] = v.copy()
        sd['pred.3.bias'] = v.copy()
    return sd


def synthetic_frame(frame_idx, height, width, n=1, density=0.05, dmin=1.0, dmax=80.0):
    """KITTI-shaped synthetic frame: image U[0,1) (n,3,H,W) and a sparse depth
    map with Bernoulli(density) support and U[dmin,dmax) metres (n,1,H,W), both float32."""
    tag = 'frame%d_%dx%dx%d' % (frame_idx, n, height, width)
    image = hash_uniform(tag + '/image', n * 3 * height * width)
    keep = hash_uniform(tag + '/mask', n * height * width) < density
    depth = dmin + (dmax - dmin) * hash_uniform(tag + '/depth', n * height * width)
    sparse = np.where(keep, depth, 0.0)
    return (image.reshape(n, 3, height, width).astype(np.float32),
            sparse.reshape(n, 1, height, width).astype(np.float32))


# ---- CostDCNet ----------------------------------------------------------------------------------
def _me_bn_keys(prefix, ch):
    return _bn_keys(prefix + '.bn', ch)           # MinkowskiBatchNorm wraps an nn.BatchNorm1d called `bn`


def _p3d_keys(prefix, cin, cout):
    """P3D block (external_src/costdcnet/models/unet3d.py:66-84): 1x3x3 conv - BN3d - ELU - 3x1x1 conv - BN3d - ELU."""
    return ([(prefix + '.conv1.weight', (cout, cin, 1, 3, 3))] + _bn_keys(prefix + '.bn1', cout) +
            [(prefix + '.conv2.weight', (cout, cout, 3, 1, 1))] + _bn_keys(prefix + '.bn2', cout))


def costdcnet_keys(prepare_mode='meta_selfsup_seq_1layer_ema'):
    """Ordered (name, shape) list of the CostDCNet TTA network's state_dict: CostDCNet.__init__
    (external_src/costdcnet/CostDCNet_adapt.py:13-23: Encoder2D(4,16), Encoder3D(1,16,planes=(32,48,64)),
    UNet3D(32,16,f_maps=[32,48,64,80])) plus the heads / meta layer of _prepare_head (:426-496, 'selfsup'+'ema'+'1layer')."""
    if '1layer' not in prepare_mode:
        raise NotImplementedError('CostDCNet key table covers the canonical 1layer meta layer only')
    keys = [('enc2d.conv1.weight', (64, 4, 3, 3)), ('enc2d.conv1.bias', (64,))] + _bn_keys('enc2d.norm1', 64)
    inpl = 64
    for li, (planes, stride) in enumerate([(64, 1), (96, 2), (128, 2)]):
        for b in range(2):
            pre = 'enc2d.layer%d.%d' % (li + 1, b)
            s = stride if b == 0 else 1
            keys += [(pre + '.conv1.weight', (planes, inpl, 3, 3)), (pre + '.conv1.bias', (planes,)),
                     (pre + '.conv2.weight', (planes, planes, 3, 3)), (pre + '.conv2.bias', (planes,))]
            keys += _bn_keys(pre + '.norm1', planes) + _bn_keys(pre + '.norm2', planes)
            if s != 1:          # norm3 is registered twice (attribute + inside `downsample`): both key sets exist, same tensors
                keys += _bn_keys(pre + '.norm3', planes)
                keys += [(pre + '.downsample.0.weight', (planes, inpl, 1, 1)), (pre + '.downsample.0.bias', (planes,))]
                keys += _bn_keys(pre + '.downsample.1', planes)
            inpl = planes
    keys += [('enc2d.conv2.weight', (16, 128, 1, 1)), ('enc2d.conv2.bias', (16,))]
    # sparse encoder (models/encoder3d.py:33-103); MinkowskiConvolution kernels are (27, Cin, Cout), no bias
    keys += [('enc3d.conv1.kernel', (27, 1, 32))] + _me_bn_keys('enc3d.bn0', 32)
    inpl = 32
    for bi, planes in enumerate((32, 48, 64)):
        pre = 'enc3d.block%d.0' % (bi + 1)
        keys += [(pre + '.conv1.kernel', (27, inpl, planes))] + _me_bn_keys(pre + '.norm1', planes)
        keys += [(pre + '.conv2.kernel', (27, planes, planes))] + _me_bn_keys(pre + '.norm2', planes)
        if bi > 0:
            keys += [(pre + '.downsample.0.kernel', (1, inpl, planes))] + _me_bn_keys(pre + '.downsample.1', planes)
        inpl = planes
    keys += [('enc3d.conv2.kernel', (64, 16))]
    # UNet3D (models/unet3d.py:7-47): DoubleConv = two P3D blocks
    f = (32, 48, 64, 80)
    keys += _p3d_keys('unet3d.inc.double_conv.0', 32, f[0]) + _p3d_keys('unet3d.inc.double_conv.1', f[0], f[0])
    for i in range(3):          # Down: MaxPool3d(2) + DoubleConv(in, out, mid=in)
        pre = 'unet3d.down%d.maxpool_conv.1.double_conv' % (i + 1)
        keys += _p3d_keys(pre + '.0', f[i], f[i]) + _p3d_keys(pre + '.1', f[i], f[i + 1])
    for name, cin, cout in (('up2', f[3] + f[2], f[2]), ('up3', f[2] + f[1], f[1]), ('up4', f[1] + f[0], f[0])):
        pre = 'unet3d.%s.conv.double_conv' % name       # Up: DoubleConv(in, out, mid=out)
        keys += _p3d_keys(pre + '.0', cin, cout) + _p3d_keys(pre + '.1', cout, cout)
    keys += [('unet3d.classif0.weight', (16, 32, 1, 1, 1)), ('unet3d.classif0.bias', (16,))]
    keys += _mlp_keys('proj', 160, 512, 512) + _mlp_keys('proj_t', 160, 512, 512) + _mlp_keys('pred', 512, 512, 512)
    keys += [('conv1_rgb_meta.weight', (16, 16, 3, 3)), ('conv1_rgb_meta.bias', (16,))]
    return keys


def formula_state_dict_costdcnet(prepare_mode='meta_selfsup_seq_1layer_ema', gain=1.0):
    sd = {k: formula_tensor(k, s, gain) for k, s in costdcnet_keys(prepare_mode)}
    for k in list(sd):          # the reference holds ONE BatchNorm behind both names
        if '.downsample.1.' in k and k.startswith('enc2d.'):
            sd[k] = sd[k.replace('.downsample.1.', '.norm3.')]
        if k.endswith('.kernel'):       # (K, Cin, Cout): fan-in = K * Cin, not formula_tensor's prod(shape[1:])
            s = sd[k].shape
            fan_in = s[0] * s[1] if len(s) == 3 else s[0]
            u = hash_uniform(k, sd[k].size) * 2.0 - 1.0
            sd[k] = (u * np.sqrt(3.0) * gain * np.sqrt(2.0 / fan_in)).reshape(s).astype(np.float32)
    # sparse inputs are residuals in [-0.5, 0.5] plane units on a handful of voxels: lift the first sparse layer so the
    # 3-D branch carries signal comparable to the image branch (a trained network does)
    sd['enc3d.conv1.kernel'] = (sd['enc3d.conv1.kernel'] * 4.0).astype(np.float32)
    return sd

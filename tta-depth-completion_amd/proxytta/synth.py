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


def formula_state_dict(prepare_mode='meta_selfsup_seq_1layer_ema', gain=1.0):
    return {k: formula_tensor(k, s, gain) for k, s in msg_chn_keys(prepare_mode)}


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

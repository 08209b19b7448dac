"""TEST INFRASTRUCTURE ONLY (oracle): a CPU stand-in for the subset of MinkowskiEngine that CostDCNet uses.

MinkowskiEngine is a third-party dependency of the reference that is absent from /root/reference and pinned nowhere
(requirements omit it; call sites: external_src/costdcnet/CostDCNet_adapt.py:381-388 TensorField / UNWEIGHTED_AVERAGE /
.sparse(), :394 .dense(); models/encoder3d.py:28,42-50,59-91,93-103 MinkowskiConvolution k=3 / k=1, stride [1,2,2],
MinkowskiBatchNorm, MinkowskiReLU, modules.resnet_block.BasicBlock; src/costdcnet_model_adapt.py:368
MinkowskiSyncBatchNorm).  It cannot be built or imported here, so its PUBLISHED semantics (MinkowskiEngine 0.5 docs /
"4D Spatio-Temporal ConvNets", CVPR'19) are restated below.  **Parity unpinned**: no reference output exists for this
arithmetic; everything downstream of `SparseTensor.dense()` is pinned by running the real reference on top of this module
(tests/golden/make_golden_costdcnet.py).  The one choice the documentation leaves open (the kernel-offset order, below) is pinned by
the reference's pretrained weights, which also give a functional check of the rest: with them the real network on top of this module
completes a synthetic indoor scene to 7.5 mm MAE (tools/costdcnet_kernel_order.py).

Semantics restated
  * generalized sparse convolution: out[u] = sum_{i in N(u, K)} W_i x[u + i * tensor_stride] over EXISTING inputs only;
    stride-1 convolutions keep the input coordinates, a stride-s convolution has output coordinates
    unique(floor(c / (ts*s)) * (ts*s)) and tensor stride ts*s; odd kernels are centred.
  * kernel offsets are enumerated with the first spatial axis fastest: k = (d0+1) + 3*(d1+1) + 9*(d2+1)  [pinned by the reference's
    PRETRAINED weights: with external_src/costdcnet/weights/*.pth the real network completes a synthetic indoor scene to 7.5 mm MAE
    under this order, 99 mm under the opposite one, 125 mm with the offsets shuffled -- tools/costdcnet_kernel_order.py,
    tests/test_oracle_golden.py::test_sparse_kernel_order_matches_pretrained_weights];
    kernel tensor (K, Cin, Cout), or (Cin, Cout) when kernel volume = 1 and stride = 1 (the layout of the shipped
    external_src/costdcnet/weights/enc3d.pth: conv2.kernel (64,16), downsample.0.kernel (1,32,48)); no bias.
  * a kernel_size-1, stride-s convolution only sees inputs that sit exactly on an output coordinate.
  * MinkowskiBatchNorm = nn.BatchNorm1d over the feature rows; MinkowskiReLU = relu on the features.
  * TensorField(...).sparse() with UNWEIGHTED_AVERAGE: coordinates floored to integers, duplicates averaged.
  * SparseTensor.dense(): coordinates divided by the tensor stride, no shift when min_coordinate is None (negative
    coordinates are an error), shape = (max batch + 1, C, max coordinate + 1 per axis); returns
    (dense, min_coordinate, tensor_stride).
"""
import types

import torch
import torch.nn as nn

_S = 8          # coordinate bias so that neighbour keys of border voxels stay non-negative


def _keys(C, dims):
    """Linear key of (b, c0, c1, c2) rows; dims = per-axis extents (after bias)."""
    k = C[:, 0]
    for a in range(1, C.shape[1]):
        k = k * dims[a - 1] + (C[:, a] + _S)
    return k


class SparseTensor(object):
    def __init__(self, features, coordinates, tensor_stride=(1, 1, 1)):
        self.F = features
        self.C = coordinates.long()
        self.tensor_stride = tuple(int(t) for t in tensor_stride)

    def _like(self, F):
        return SparseTensor(F, self.C, self.tensor_stride)

    def __add__(self, other):
        assert self.C.shape == other.C.shape and bool((self.C == other.C).all()), 'sparse tensors on different coordinate maps'
        return self._like(self.F + other.F)

    __iadd__ = __add__

    def dense(self, shape=None, min_coordinate=None, contract_stride=True):
        assert min_coordinate is None and shape is None
        coords = self.C[:, 1:]
        if not bool((coords >= 0).all()):
            raise ValueError('Coordinate has a negative value')
        ts = torch.tensor(self.tensor_stride, dtype=torch.long)
        if contract_stride:
            coords = coords // ts
        size = coords.max(0)[0] + 1
        nb = int(self.C[:, 0].max()) + 1
        out = torch.zeros((nb, self.F.shape[1]) + tuple(int(s) for s in size), dtype=self.F.dtype)
        out[self.C[:, 0], :, coords[:, 0], coords[:, 1], coords[:, 2]] = self.F
        return out, torch.zeros(1, coords.shape[1], dtype=torch.int32), ts.int()


class TensorField(object):
    def __init__(self, features, coordinates, quantization_mode=None, minkowski_algorithm=None, device=None):
        self.F, self.C = features, coordinates

    def sparse(self):
        C = torch.floor(self.C).long()
        uniq, inv = torch.unique(C, dim=0, return_inverse=True)
        F = torch.zeros((uniq.shape[0], self.F.shape[1]), dtype=self.F.dtype).index_add_(0, inv, self.F)
        cnt = torch.zeros(uniq.shape[0], dtype=self.F.dtype).index_add_(0, inv, torch.ones(C.shape[0], dtype=self.F.dtype))
        return SparseTensor(F / cnt[:, None], uniq, (1, 1, 1))


def _triple(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v, v, v)


def kernel_offsets(kernel_size):
    if kernel_size == 1:
        return [(0, 0, 0)]
    r = range(-(kernel_size // 2), kernel_size // 2 + 1)
    return [(d0, d1, d2) for d2 in r for d1 in r for d0 in r]          # first axis fastest (pinned: module docstring)


def sparse_conv(x, kernel, kernel_size, stride):
    """Generalized sparse convolution (see module docstring).  kernel: (K, Cin, Cout) or (Cin, Cout)."""
    stride = _triple(stride)
    ts_in = torch.tensor(x.tensor_stride, dtype=torch.long)
    ts_out = ts_in * torch.tensor(stride, dtype=torch.long)
    if all(s == 1 for s in stride):
        C_out = x.C
    else:
        q = torch.cat([x.C[:, :1], (x.C[:, 1:] // ts_out) * ts_out], 1)
        C_out = torch.unique(q, dim=0)
    dims = [int(x.C[:, a].max()) + 2 * _S + 1 for a in range(1, 4)]
    kin = _keys(x.C, dims)
    order = torch.argsort(kin)
    kin_sorted = kin[order]
    W = kernel if kernel.dim() == 3 else kernel[None]
    out = torch.zeros((C_out.shape[0], W.shape[2]), dtype=x.F.dtype)
    for k, off in enumerate(kernel_offsets(kernel_size)):
        q = C_out.clone()
        q[:, 1:] += torch.tensor(off, dtype=torch.long) * ts_in
        ok = ((q[:, 1:] + _S) >= 0).all(1) & ((q[:, 1:] + _S) < torch.tensor(dims)).all(1)
        kq = _keys(q, dims)
        pos = torch.searchsorted(kin_sorted, kq).clamp(max=kin_sorted.numel() - 1)
        hit = ok & (kin_sorted[pos] == kq)
        j = hit.nonzero(as_tuple=True)[0]
        if j.numel():
            out = out.index_add(0, j, x.F[order[pos[j]]] @ W[k])
    return SparseTensor(out, C_out, tuple(int(t) for t in ts_out))


class MinkowskiConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, kernel_generator=None,
                 expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__()
        assert dimension == 3 and dilation == 1 and not bias
        self.kernel_size, self.stride = kernel_size, _triple(stride)
        vol = kernel_size ** 3
        shape = (in_channels, out_channels) if (vol == 1 and all(s == 1 for s in self.stride)) else (vol, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.zeros(shape))
        nn.init.normal_(self.kernel, std=0.05)

    def forward(self, x):
        return sparse_conv(x, self.kernel, self.kernel_size, self.stride)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats)

    def forward(self, x):
        return x._like(self.bn(x.F))


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    pass


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x._like(torch.relu(x.F))


class BasicBlock(nn.Module):
    """MinkowskiEngine.modules.resnet_block.BasicBlock: conv3-bn-relu-conv3-bn, + (downsampled) input, relu."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out = out + residual
        return self.relu(out)


def _batched_coordinates(coords, dtype=torch.int32, device=None):
    return torch.cat([torch.cat([torch.full((c.shape[0], 1), b, dtype=c.dtype), c], 1) for b, c in enumerate(coords)], 0).to(dtype)


def _kaiming_normal_(tensor, a=0, mode='fan_in', nonlinearity='leaky_relu'):
    with torch.no_grad():
        return tensor.normal_(0, 0.05)


def install(sys_modules):
    """Register this module as `MinkowskiEngine` (+ `MinkowskiEngine.modules.resnet_block`) so the reference imports it."""
    me = types.ModuleType('MinkowskiEngine')
    for k in ('SparseTensor', 'TensorField', 'MinkowskiConvolution', 'MinkowskiBatchNorm', 'MinkowskiSyncBatchNorm', 'MinkowskiReLU'):
        setattr(me, k, globals()[k])
    me.SparseTensorQuantizationMode = types.SimpleNamespace(UNWEIGHTED_AVERAGE='unweighted_average')
    me.MinkowskiAlgorithm = types.SimpleNamespace(SPEED_OPTIMIZED='speed')
    me.utils = types.SimpleNamespace(batched_coordinates=_batched_coordinates, kaiming_normal_=_kaiming_normal_)
    mods = types.ModuleType('MinkowskiEngine.modules')
    rb = types.ModuleType('MinkowskiEngine.modules.resnet_block')
    rb.BasicBlock = BasicBlock
    mods.resnet_block = rb
    me.modules = mods
    sys_modules['MinkowskiEngine'] = me
    sys_modules['MinkowskiEngine.modules'] = mods
    sys_modules['MinkowskiEngine.modules.resnet_block'] = rb
    return me

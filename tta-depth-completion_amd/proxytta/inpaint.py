"""Eval-time hole filling of the NLSPN adapter (src/nlspn_model_adapt.py:124-127 -> src/data_utils.py:327-354 `inpainting`):
pixels whose predicted depth is exactly 0 (the clamp at nlspnmodel_adapt.py:371) are filled by biharmonic inpainting.  The
reference does this on the CPU with scikit-image's `restoration.inpaint_biharmonic` -- a third-party dependency absent from the
reference tree and from this image (PARITY UNPINNED): its published algorithm is restated here on scipy, host-side like the
reference's (the depth map is on its way to the CPU metrics anyway, src/tta_main.py:769).

skimage.restoration.inpaint_biharmonic(image, mask) for a single-channel N-d image:
  * the mask is split into independent regions: connected components (full connectivity) of the mask dilated by one pixel
    (cross-shaped structuring element), restricted to the mask;
  * for every masked point the biharmonic stencil is laplace(laplace(delta)) evaluated on the radius-2 neighbourhood box clipped
    to the image (scipy.ndimage.laplace, 'reflect' boundaries); coefficients on masked points form the unknown matrix, on known
    points the right-hand side;
  * the sparse system is solved directly and the result clipped to [min, max] of the known pixels.
The reference passes each sample as a (1, H, W) array, i.e. a 3-D image one voxel thick: the clipped stencil has extent 1 along
that axis and reduces to the 2-D biharmonic operator.
"""
import numpy as np
import scipy.ndimage as ndi
from scipy import sparse
from scipy.sparse.linalg import spsolve


def _inpaint_region(mask, out, limits):
    pts = np.stack(np.where(mask), axis=-1)
    index = -np.ones(mask.shape, dtype=np.int64)
    index[mask] = np.arange(len(pts))
    shape = np.array(out.shape)
    rows_u, cols_u, vals_u = [], [], []
    rhs = np.zeros(len(pts), dtype=np.float64)
    cache = {}
    for n, pt in enumerate(pts):
        lo, hi = np.maximum(pt - 2, 0), np.minimum(pt + 3, shape)
        key = (tuple(pt - lo), tuple(hi - lo))
        coef = cache.get(key)
        if coef is None:                                  # the stencil depends only on the box and the point's place in it
            delta = np.zeros(hi - lo)
            delta[tuple(pt - lo)] = 1
            coef = ndi.laplace(ndi.laplace(delta))
            cache[key] = coef
        box = tuple(slice(a, b) for a, b in zip(lo, hi))
        idx, known = index[box], out[box]
        nz = coef != 0
        unk = nz & (idx >= 0)
        rows_u += [n] * int(unk.sum()); cols_u += idx[unk].tolist(); vals_u += coef[unk].tolist()
        kn = nz & (idx < 0)
        rhs[n] = -float((coef[kn] * known[kn].astype(np.float64)).sum())
    A = sparse.csr_matrix((vals_u, (rows_u, cols_u)), shape=(len(pts), len(pts)))
    res = np.clip(np.asarray(spsolve(A.tocsc(), rhs)).ravel(), *limits)
    out[mask] = res
    return out


def inpaint_biharmonic(image, mask):
    """image, mask: same-shape arrays (any dimensionality); returns a copy with the masked points filled."""
    image = np.asarray(image)
    mask = np.asarray(mask).astype(bool)
    if image.shape != mask.shape:
        raise ValueError('Input arrays have to be the same shape')
    out = np.copy(image)
    if not mask.any():
        return out
    if mask.all():
        raise ValueError('nothing to inpaint from: every pixel is masked')
    known = image[~mask]
    limits = (float(known.min()), float(known.max()))
    dil = ndi.binary_dilation(mask, structure=ndi.generate_binary_structure(mask.ndim, 1))
    labeled, num = ndi.label(dil, structure=np.ones((3,) * mask.ndim))
    labeled = labeled * mask
    for r in range(1, num + 1):
        region = labeled == r
        if region.any():
            _inpaint_region(region, out, limits)
    return out


def inpainting(depth_map):
    """src/data_utils.py:327-354: depth_map N x 1 x H x W (numpy, modified in place and returned); holes = exact zeros."""
    if not (depth_map == 0).any():
        return depth_map
    for i in range(depth_map.shape[0]):
        depth = depth_map[i, ...]
        depth_map[i] = inpaint_biharmonic(depth, depth == 0)
    return depth_map

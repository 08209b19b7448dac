"""ctypes binding of libptta_hip.so (C-ABI: include/ptta.h).

The product path has no CPU fallback: if the shared library is missing or cannot be loaded this
module raises, and every call that needs a GPU raises when no HIP device is present.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

# The binding does not touch the process environment.  The hosting process should export HIP_FORCE_DEV_KERNARG=1 before the HIP runtime
# initialises (kernel arguments in device memory; 0 costs 7 % of the replayed MSG_CHN step): bench.py does and reports it, INTEGRATION.md.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libptta_hip.so')

PTTA_BACKBONE_MSG_CHN = 0
PTTA_BACKBONE_NLSPN = 1
PTTA_BACKBONE_COSTDCNET = 2
PTTA_NLSPN_LEGACY_OFFSET = 0x100
PTTA_NLSPN_SYNCBN_ADAPT = 0x200
PTTA_SYNCBN_ADAPT = 0x200
PTTA_META_1LAYER = 0
PTTA_META_2LAYERS = 1
PTTA_DTYPE_F32 = 0
PTTA_DTYPE_MIXED = 1
PTTA_ABI_VERSION = 2           # include/ptta.h; checked against ptta_version() at load
CONV_S1, CONV_S2, CONV_T2 = 0, 1, 2


class Hparams(ctypes.Structure):
    _fields_ = [('lr', c_float), ('beta1', c_float), ('beta2', c_float), ('eps', c_float),
                ('weight_decay', c_float), ('w_sparse_depth', c_float), ('w_smoothness', c_float),
                ('w_cos', c_float), ('max_input_depth', c_float), ('max_predict_depth', c_float)]


# every symbol include/ptta.h declares: (name, restype, argtypes)
_P = c_void_p
SIGNATURES = [
    ('ptta_create', c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, POINTER(Hparams)]),
    ('ptta_destroy', None, [_P]),
    ('ptta_last_error', c_char_p, [_P]),
    ('ptta_set_hparams', c_int, [_P, POINTER(Hparams), _P]),
    ('ptta_load_weights', c_int, [_P, c_char_p, _P, POINTER(c_int64), c_int, _P]),
    ('ptta_bind_adapted', c_int, [_P, c_char_p, _P, _P, _P]),
    ('ptta_set_adam_step', c_int, [_P, c_int, _P]),
    ('ptta_get_adam_step', c_int, [_P, POINTER(c_int), _P]),
    ('ptta_forward_train', c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ('ptta_embedding_rows', c_int64, [_P]),
    ('ptta_forward_eval', c_int, [_P, _P, _P, _P, _P]),
    ('ptta_loss_forward', c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, _P, _P]),
    ('ptta_loss_backward', c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P]),
    ('ptta_backward', c_int, [_P, _P, _P, _P, _P, _P]),
    ('ptta_set_grad', c_int, [_P, c_char_p, _P, c_int64, _P]),
    ('ptta_get_grad', c_int, [_P, c_char_p, _P, c_int64, _P]),
    ('ptta_adapted_count', c_int, [_P]),
    ('ptta_adapted_name', c_char_p, [_P, c_int, POINTER(c_int64)]),
    ('ptta_adapted_repeat', c_int, [_P, c_int]),
    ('ptta_adam_step', c_int, [_P, _P, _P, _P]),
    ('ptta_step', c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    ('ptta_step_pipelined', c_int, [_P] * 5 + [c_uint64] + [_P] * 2 + [c_uint64] + [_P] * 3),
    ('ptta_pipeline_stream', c_int, [_P, POINTER(c_void_p)]),
    ('ptta_forward_eval_last', c_int, [_P, _P, _P]),
    ('ptta_outlier_removal', c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P]),
    ('ptta_eval_metrics', c_int, [_P, _P, c_int64, c_float, c_float, _P, _P, _P]),
    ('ptta_mdconv_forward', c_int, [_P] * 6 + [c_int] * 15 + [_P]),
    ('ptta_mdconv_backward', c_int, [_P] * 11 + [c_int] * 15 + [_P]),
    ('ptta_set_image_norm', c_int, [_P, c_float, POINTER(c_float), POINTER(c_float)]),
    ('ptta_head_bind', c_int, [_P, c_char_p, _P, _P, _P]),
    ('ptta_head_set_hparams', c_int, [_P, c_float, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    ('ptta_head_reload', c_int, [_P, _P]),
    ('ptta_head_forward', c_int, [_P, _P, _P, c_int, _P, _P, _P]),
    ('ptta_head_backward', c_int, [_P, _P, _P]),
    ('ptta_head_adam_step', c_int, [_P, _P]),
    ('ptta_head_step', c_int, [_P, _P, _P, c_int, _P, _P]),
    ('ptta_head_get_grad', c_int, [_P, c_char_p, _P, c_int64, POINTER(c_int), _P]),
    ('ptta_crop_flip', c_int, [_P, _P] + [c_int] * 6 + [_P] * 5),
    ('ptta_rotate', c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    ('ptta_resize_crop', c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    ('ptta_photometric', c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    ('ptta_photometric_full', c_int, [_P, _P, c_int, c_int, c_int] + [_P] * 12),
    ('ptta_add_noise', c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, c_float, c_int, _P]),
    ('ptta_remove_patches', c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    ('ptta_crop_pad', c_int, [_P, _P, c_int, c_int, c_int, c_int] + [_P] * 7 + [c_int, c_float, _P]),
    ('ptta_resize_pad', c_int, [_P, _P, c_int, c_int, c_int, c_int] + [_P] * 5 + [c_int, c_int, c_float, _P]),
    ('ptta_set_stat_sync', c_int, [_P, _P, _P, _P, c_int64, c_int]),
    ('ptta_rccl_unique_id', c_int, [_P]),
    ('ptta_rccl_comm_create', c_int, [_P, c_int, c_int, POINTER(c_void_p)]),
    ('ptta_rccl_comm_destroy', c_int, [_P]),
    ('ptta_rccl_allreduce_mean_f32', c_int, [_P, _P, c_int64, _P]),
    ('ptta_rccl_last_error', c_char_p, []),
    ('ptta_set_stat_sync_rccl', c_int, [_P, _P, _P, c_int64, c_int]),
    ('ptta_set_grad_sync_rccl', c_int, [_P, _P]),
    ('ptta_set_graph', c_int, [_P, c_int]),
    ('ptta_set_option', c_int, [_P, c_char_p, c_int]),
    ('ptta_get_option', c_int, [_P, c_char_p, POINTER(c_int)]),
    ('ptta_profile', c_int, [_P, c_int]),
    ('ptta_profile_read', c_int, [_P, c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                  POINTER(c_int64), _P]),
    ('ptta_debug_tensor', c_int, [_P, c_char_p, _P, c_int64, POINTER(c_int64), _P]),
    ('ptta_op_conv32', c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    ('ptta_op_conv32_chain', c_int, [_P] * 6 + [c_int] * 7 + [POINTER(c_float), _P]),
    ('ptta_version', c_int, []),
]

_lib = None


def load():
    """Load libptta_hip.so (built by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64: import it FIRST so that this library binds to the HIP runtime torch uses
    # (loading libptta_hip.so first would pull /opt/rocm's copy into the process and torch would then see no device)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libptta_hip.so not found at %s: build it with `python -c "import '
                           '__graft_entry__ as g; g.build()"` — there is no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    have = lib.ptta_version()
    if have != PTTA_ABI_VERSION:
        raise RuntimeError('%s has ABI version %d, this binding is written against %d (include/ptta.h PTTA_ABI_VERSION): rebuild it with '
                           '`make -C tta-depth-completion_amd/csrc`' % (LIB_PATH, have, PTTA_ABI_VERSION))
    _lib = lib
    return lib


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.ptta_last_error(handle).decode() if handle else ''
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg))


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return c_void_p(t.data_ptr())


__all__ = ['load', 'check', 'ptr', 'Hparams', 'SIGNATURES', 'LIB_PATH', 'byref']

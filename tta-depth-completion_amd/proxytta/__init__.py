"""proxytta — MI355X-native ProxyTTA test-time-adaptation step (MSG_CHN backbone).

Python here is plumbing (tensors, streams, torch.distributed); the step itself is libptta_hip.so
(hand-written HIP for gfx950, C-ABI in include/ptta.h).  Importing this package never touches the
GPU; constructing a model or an Engine does, and raises if the library or the device is missing.
"""
from . import synth  # noqa: F401

__all__ = ['synth', 'ExternalModel_Adapt', 'ExternalModelAdapt', 'MsgChnModel_Adapt', 'Engine']


def __getattr__(name):
    if name in ('ExternalModel_Adapt', 'ExternalModelAdapt', 'MsgChnModel_Adapt', 'CANONICAL_LOSS_TYPE'):
        from . import model
        return getattr(model, name)
    if name == 'Engine':
        from .engine import Engine
        return Engine
    raise AttributeError(name)

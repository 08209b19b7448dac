"""Multi-GPU plumbing for the ProxyTTA step: one process per GPU, torch.distributed
(backend 'nccl' = RCCL over xGMI on ROCm, 'gloo' in the CPU tests).

Two modes:
  * independent frame streams (BASELINE config 4): frame i of the stream goes to rank i % world;
    every rank adapts its own copy of the adapted parameters on its sub-stream.  No collective on
    the data path.
  * shared-parameter batched TTA (what the reference's DDP does, src/msg_chn_model_adapt.py:476-480,
    src/tta_main.py:354): every rank runs forward+loss+backward on its slice of the batch, then ONE
    collective: mean all-reduce of the adapted-parameter gradients only (37 KB for the MSG_CHN 1layer
    meta conv, 160 KB for NLSPN's 88 meta_bn tensors; the reference all-reduces every gradient and
    discards most of them), then the same Adam update everywhere.  The reference also converts BatchNorm to
    SyncBatchNorm (src/tta_main.py:326): `Engine.enable_stat_sync()` (ptta_set_stat_sync) exchanges each BatchNorm's
    partial sums over the ranks, so two ranks with half a batch each reproduce the single-process run on the whole batch
    (tests/test_gpu_syncbn.py).  Without it every rank normalises with its own slice's statistics (a warning is issued).
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """Indices of the frames this rank owns (round-robin, as a streaming source would deal them)."""
    return list(range(rank, n_frames, world))


def allreduce_adapted_grads(grads, group=None):
    """Mean all-reduce of a list of gradient tensors as ONE flat message (latency-bound payload:
    a single small collective beats one per tensor).  Returns the reduced tensors (in place)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grads
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return grads


import atexit
import weakref

_rccl_default = []                          # communicator of the default process group (group=None)
_rccl_groups = weakref.WeakKeyDictionary()  # keyed on the group OBJECT: a collected group's id() can be reused by another one


def _rccl_destroy(comm):
    from . import _lib
    try:
        _lib.load().ptta_rccl_comm_destroy(comm)
    except Exception:
        pass


def rccl_communicator(group=None):
    """(comm handle, world size) of the library-owned RCCL communicator of this process (include/ptta.h ptta_rccl_*): rank 0
    draws the unique id, torch.distributed (any backend: it only carries 128 bytes, once) broadcasts it, every rank joins.
    Without an initialised process group: a one-rank communicator.  One communicator per process group, destroyed when the
    group object is collected or at interpreter exit."""
    import ctypes
    from . import _lib
    if group is None and _rccl_default:
        return _rccl_default[0]
    if group is not None and group in _rccl_groups:
        return _rccl_groups[group]
    lib = _lib.load()
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    ident = ctypes.create_string_buffer(128)
    if rank == 0:
        rc = lib.ptta_rccl_unique_id(ident)
        if rc:
            raise RuntimeError('ptta_rccl_unique_id failed (%d): %s' % (rc, lib.ptta_rccl_last_error().decode()))
    if world > 1:
        box = [bytes(ident.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = ctypes.create_string_buffer(box[0], 128)
    comm = ctypes.c_void_p()
    rc = lib.ptta_rccl_comm_create(ident, rank, world, ctypes.byref(comm))
    if rc:
        raise RuntimeError('ptta_rccl_comm_create failed (%d): %s' % (rc, lib.ptta_rccl_last_error().decode()))
    entry = (comm.value, world)
    if group is None:
        _rccl_default.append(entry)
        atexit.register(_rccl_destroy, comm.value)
    else:
        _rccl_groups[group] = entry
        weakref.finalize(group, _rccl_destroy, comm.value)
    return entry


_warned = []


def shared_parameter_step(engine, image, sparse, validity=None, loss_image=None, group=None, w=None):
    """Batched TTA across ranks with shared adapted parameters: local forward / loss / backward
    through the library, one gradient all-reduce, fused Adam with the reduced gradients.
    `w` = (w_sparse_depth, w_smoothness, w_cos); default: the engine's hyper-parameters."""
    if getattr(engine, '_rccl_grads', False) and w is None:
        # library-owned RCCL communicator: statistics exchange, gradient all-reduce and Adam are all inside ONE fused step
        return engine.step(image, sparse, validity=validity, loss_image=loss_image, want_depth=True)
    if w is None:
        w = (engine.hp.w_sparse_depth, engine.hp.w_smoothness, engine.hp.w_cos)
    if (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1 and not _warned
            and 'stat_sync' not in getattr(engine, '_keep', {})):
        import warnings
        _warned.append(1)
        warnings.warn('shared_parameter_step: BatchNorm statistics are per rank (the reference converts to SyncBatchNorm, '
                      'src/tta_main.py:326); call engine.enable_stat_sync() / model.convert_syncbn() to exchange them')
    depth, emb, ref = engine.forward_train(image, sparse)
    if validity is None:
        validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
    if loss_image is None:
        loss_image = image
    info = engine.loss_forward(loss_image, depth, sparse, validity, emb, ref, *w)
    gd, gr = engine.loss_backward(loss_image, depth, sparse, validity, emb, ref)
    if getattr(engine, 'backbone', 'msg_chn') == 'msg_chn' and len(engine.adapted) == 2:
        gw, gb = engine.backward(gd, gr)
        allreduce_adapted_grads([gw, gb], group)
        engine.adam_step(gw, gb)
        return info, depth
    # any adapted set (MSG_CHN 2layers: 7 tensors, NLSPN meta_bn: 88): gradients stay in the library; they are read,
    # reduced as ONE flat message, written back and consumed by the on-device Adam
    like = {k: torch.empty(engine.adapted_numel[k], device=depth.device) for k in engine.adapted}
    grads = engine.backward_all(gd, gr, like)
    allreduce_adapted_grads(grads, group)
    for k, g in zip(engine.adapted, grads):
        engine.set_grad(k, g)
    engine.adam_step()
    return info, depth

"""Thin Python owner of one libptta_hip handle: device memory, streams and tensors come from
PyTorch-ROCm, all arithmetic of the step runs inside the HIP library."""
import ctypes
from ctypes import byref, c_int, c_int64, c_void_p

import torch

from . import _lib
from ._lib import Hparams, check, ptr

ADAPTED = ('conv1_rgb_meta.weight', 'conv1_rgb_meta.bias')          # 1layer meta layer
_P2 = 'conv1_rgb_meta.conv1_meta.'
ADAPTED_2LAYERS = (_P2 + '0.0.weight', _P2 + '0.1.weight', _P2 + '0.1.bias', _P2 + '1.weight', _P2 + '1.bias',
                   _P2 + '2.weight', _P2 + '2.bias')


HEAD_PARAMS = tuple('%s.%s.%s' % (m, l, t) for m in ('proj', 'pred') for l in ('0', '1', '3') for t in ('weight', 'bias'))
HEAD_TARGETS = tuple('proj_t' + k[4:] for k in HEAD_PARAMS[:6])


def adapted_names(meta='1layer'):
    return ADAPTED_2LAYERS if meta == '2layers' else ADAPTED
_BOUND_SUFFIX = ('running_mean', 'running_var', 'num_batches_tracked')


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def image_norm_constants(normalized_image_range):
    """(divisor, mean[3], std[3]) of `normalized_image_range` as interpreted by
    Transforms.normalize_images (src/transforms.py:681-708)."""
    r = list(normalized_image_range)
    standard = any(isinstance(v, (tuple, list)) for v in r)
    if r == [0, 1]:
        return 255.0, [0.0] * 3, [1.0] * 3
    if standard:
        mean, std = list(r[0]), list(r[1])
        if len(mean) != 3 or len(std) != 3:
            raise ValueError('Unsupported normalization range: {}'.format(normalized_image_range))
        return 255.0, mean, std
    if r == [-1, 1]:
        return 255.0, [0.5] * 3, [0.5] * 3
    if r == [0, 255]:
        return 1.0, [0.0] * 3, [1.0] * 3
    raise ValueError('Unsupported normalization range: {}'.format(normalized_image_range))


class Engine:
    """One (batch, height, width, dtype) instance of the MSG_CHN ProxyTTA step on the current GPU."""

    def __init__(self, n, height, width, dtype='fp32', lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=1.0, w_cos=1.0,
                 max_input_depth=None, meta='1layer', backbone='msg_chn', legacy_offset=False, max_predict_depth=None,
                 syncbn_adapted=False, keep=(), options=None):
        if not torch.cuda.is_available():
            raise RuntimeError('proxytta needs a HIP device (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        self.lib = _lib.load()
        self.n, self.h, self.w = int(n), int(height), int(width)
        self.dtype = dtype
        self.meta = meta
        self.backbone = backbone
        self.emb_dim = 1024 if backbone == 'nlspn' else 512
        if backbone == 'costdcnet' and not max_predict_depth:
            raise ValueError('CostDCNet needs max_predict_depth (the far plane of its cost volume, ExternalModel_Adapt(max_predict_depth))')
        self.adapted = adapted_names(meta)
        self.hp = Hparams(lr, betas[0], betas[1], eps, weight_decay, w_sparse_depth, w_smoothness,
                          w_cos, -1.0 if max_input_depth is None else float(max_input_depth), float(max_predict_depth or 0.0))
        self.handle = c_void_p()
        if dtype not in ('fp32', 'mixed'):
            raise ValueError("dtype must be 'fp32' (fp32 maps, bf16x3 products) or 'mixed' (fp32 / bf16x3 for the real frames' forward, narrow "
                             "bf16 maps and single-MFMA products for the no_grad proxy pass and the backward: include/ptta.h PTTA_DTYPE_MIXED)")
        code = {'fp32': _lib.PTTA_DTYPE_F32, 'mixed': _lib.PTTA_DTYPE_MIXED}[dtype]
        # precision budget only: classes of the mixed mode kept at fp32 / bf16x3 (include/ptta.h PTTA_MIXED_KEEP_*)
        for k in keep:
            code |= {'proxy': 0x100, 'backward': 0x200, 'heads': 0x400, 'bwd_rounded_w': 0x800}[k]
        rc = self.lib.ptta_create(byref(self.handle),
                                  {'nlspn': _lib.PTTA_BACKBONE_NLSPN, 'costdcnet': _lib.PTTA_BACKBONE_COSTDCNET}.get(backbone, _lib.PTTA_BACKBONE_MSG_CHN),
                                  (_lib.PTTA_META_2LAYERS if meta == '2layers' else _lib.PTTA_META_1LAYER) |
                                  (_lib.PTTA_NLSPN_LEGACY_OFFSET if (legacy_offset and backbone == 'nlspn') else 0) |
                                  (_lib.PTTA_SYNCBN_ADAPT if (syncbn_adapted and backbone in ('nlspn', 'costdcnet')) else 0),
                                  self.n, self.h, self.w, code, byref(self.hp))
        if rc != 0:
            why = {-19: 'no HIP device', -12: 'out of device memory', -22: 'invalid size / argument',
                   -38: 'configuration not on the accelerated path (NLSPN needs fp32 and '
                        'the 1layer meta conv; MSG_CHN needs the 1layer or 2layers meta layer)'}.get(rc, 'see include/ptta.h')
            raise RuntimeError('ptta_create failed (%d): %s' % (rc, why))
        for k, v in (options or {}).items():       # include/ptta.h ptta_set_option
            self.set_option(k, v)
        self.rows = int(self.lib.ptta_embedding_rows(self.handle))
        # adapted tensors in the reference's order, from the library (MSG_CHN 1layer: 2, 2layers: 7, NLSPN meta_bn: 88,
        # src/nlspn_model_adapt.py:322-337)
        names, self.adapted_numel, self.adapted_repeat = [], {}, {}
        for i in range(int(self.lib.ptta_adapted_count(self.handle))):
            n = c_int64(0)
            name = self.lib.ptta_adapted_name(self.handle, i, byref(n)).decode()
            names.append(name)
            self.adapted_numel[name] = int(n.value)
            # entries of the reference's parameter list behind this tensor (CostDCNet's DDP list names four tensors twice: include/ptta.h)
            self.adapted_repeat[name] = int(self.lib.ptta_adapted_repeat(self.handle, i))
        if backbone in ('nlspn', 'costdcnet'):
            self.adapted = names
        else:
            assert names == list(self.adapted), (names, self.adapted)
        self._keep = {}           # tensors whose storage the library borrows
        self.device = torch.device('cuda', torch.cuda.current_device())

    def close(self):
        if getattr(self, 'handle', None) and self.handle.value:
            torch.cuda.synchronize()
            self.lib.ptta_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        check(self.lib, self.handle, rc, what)

    # ---- weights ---------------------------------------------------------------------------
    def load_state_dict(self, state):
        """state: {reference state_dict key: cuda tensor}.  Adapted parameters are skipped (bind
        them with bind_adapted); BatchNorm buffers are bound by pointer and updated in place."""
        for k, t in state.items():
            if k in self.adapted:
                continue
            assert t.is_cuda and t.is_contiguous(), k
            if k.endswith(_BOUND_SUFFIX):
                self._keep[k] = t
            shape = (c_int64 * max(t.dim(), 1))(*([int(x) for x in t.shape] or [1]))
            self._chk(self.lib.ptta_load_weights(self.handle, k.encode(), ptr(t), shape, max(t.dim(), 1),
                                                 _stream()), 'ptta_load_weights(%s)' % k)

    def bind_adapted(self, name, param, exp_avg, exp_avg_sq):
        for t in (param, exp_avg, exp_avg_sq):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        self._keep['adapt/' + name] = (param, exp_avg, exp_avg_sq)
        self._chk(self.lib.ptta_bind_adapted(self.handle, name.encode(), ptr(param), ptr(exp_avg),
                                             ptr(exp_avg_sq)), 'ptta_bind_adapted')

    def set_hparams(self, **kw):
        for k, v in kw.items():
            if k == 'betas':
                self.hp.beta1, self.hp.beta2 = v
            elif k == 'max_input_depth':
                self.hp.max_input_depth = -1.0 if v is None else float(v)
            else:
                setattr(self.hp, k, v)
        self._chk(self.lib.ptta_set_hparams(self.handle, byref(self.hp), _stream()), 'ptta_set_hparams')

    def set_adam_step(self, step):
        self._chk(self.lib.ptta_set_adam_step(self.handle, int(step), _stream()), 'ptta_set_adam_step')

    def adam_step_count(self):
        out = c_int(0)
        self._chk(self.lib.ptta_get_adam_step(self.handle, byref(out), _stream()), 'ptta_get_adam_step')
        return out.value

    # ---- the path ----------------------------------------------------------------------------
    def _f32(self, t, shape):
        assert t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.contiguous()

    def forward_train(self, image, sparse, want_emb=True):
        image = self._f32(image, (self.n, 3, self.h, self.w))
        sparse = self._f32(sparse, (self.n, 1, self.h, self.w))
        depth = torch.empty((self.n, 1, self.h, self.w), device=image.device, dtype=torch.float32)
        emb = ref = None
        if want_emb:
            emb = torch.empty((self.rows, self.emb_dim), device=image.device, dtype=torch.float32)
            ref = torch.empty_like(emb)
        self._chk(self.lib.ptta_forward_train(self.handle, ptr(image), ptr(sparse), ptr(depth), ptr(emb),
                                              ptr(ref), _stream()), 'ptta_forward_train')
        return depth, emb, ref

    def forward_eval(self, image, sparse):
        image = self._f32(image, (self.n, 3, self.h, self.w))
        sparse = self._f32(sparse, (self.n, 1, self.h, self.w))
        depth = torch.empty((self.n, 1, self.h, self.w), device=image.device, dtype=torch.float32)
        self._chk(self.lib.ptta_forward_eval(self.handle, ptr(image), ptr(sparse), ptr(depth), _stream()),
                  'ptta_forward_eval')
        return depth

    def loss_forward(self, loss_image, depth, sparse, validity, emb, ref, w_sd, w_sm, w_cos):
        info = torch.empty(4, device=depth.device, dtype=torch.float32)
        rows = 0 if emb is None else emb.shape[0]
        self._chk(self.lib.ptta_loss_forward(self.handle, ptr(loss_image.contiguous()), ptr(depth.contiguous()),
                                             ptr(sparse.contiguous()), ptr(validity.contiguous()),
                                             ptr(None if emb is None else emb.contiguous()),
                                             ptr(None if ref is None else ref.contiguous()), rows,
                                             float(w_sd), float(w_sm), float(w_cos), ptr(info), _stream()),
                  'ptta_loss_forward')
        return info

    def loss_backward(self, loss_image, depth, sparse, validity, emb, ref):
        """Gradients of the LAST loss_forward w.r.t. depth and ref (same inputs must be passed)."""
        gd = torch.empty_like(depth)
        gr = None if ref is None else torch.empty_like(ref)
        rows = 0 if emb is None else emb.shape[0]
        self._chk(self.lib.ptta_loss_backward(self.handle, ptr(loss_image.contiguous()), ptr(depth.contiguous()),
                                              ptr(sparse.contiguous()), ptr(validity.contiguous()),
                                              ptr(None if emb is None else emb.contiguous()),
                                              ptr(None if ref is None else ref.contiguous()), rows,
                                              ptr(gd), ptr(gr), _stream()), 'ptta_loss_backward')
        return gd, gr

    def grad(self, name, like):
        """Gradient of an adapted parameter after backward()/step(), shaped like `like`."""
        out = torch.empty_like(like)
        self._chk(self.lib.ptta_get_grad(self.handle, name.encode(), ptr(out), out.numel(), _stream()), 'ptta_get_grad')
        return out

    def set_grad(self, name, value):
        """Overwrite the stored gradient of an adapted parameter (reduced over ranks) before adam_step()."""
        value = value.contiguous()
        self._chk(self.lib.ptta_set_grad(self.handle, name.encode(), ptr(value), value.numel(), _stream()), 'ptta_set_grad')

    def backward_all(self, grad_depth, grad_ref, params):
        """loss.backward() for any meta layer: returns the gradients of `params` ({name: tensor})."""
        self._chk(self.lib.ptta_backward(self.handle, ptr(grad_depth.contiguous()),
                                         ptr(None if grad_ref is None else grad_ref.contiguous()), None, None, _stream()),
                  'ptta_backward')
        return [self.grad(k, params[k]) for k in self.adapted]

    def backward(self, grad_depth, grad_ref):
        gw = torch.empty((32, 32, 3, 3), device=grad_depth.device, dtype=torch.float32)
        gb = torch.empty((32,), device=grad_depth.device, dtype=torch.float32)
        self._chk(self.lib.ptta_backward(self.handle, ptr(grad_depth.contiguous()),
                                         ptr(None if grad_ref is None else grad_ref.contiguous()),
                                         ptr(gw), ptr(gb), _stream()), 'ptta_backward')
        return gw, gb

    def adam_step(self, gw=None, gb=None):
        self._chk(self.lib.ptta_adam_step(self.handle, ptr(gw), ptr(gb), _stream()), 'ptta_adam_step')

    @staticmethod
    def _frame_key(image, sparse):
        # what identifies a frame's CONTENT on the torch side: the storage and torch's in-place modification counters (a staging slot that is
        # refilled with `copy_` keeps its address and gets a new version)
        # (inference-mode tensors have no version counter and raise on the read: they get a key that never matches, i.e. a fresh token per
        # call -- correct, merely unpipelined; such callers pass explicit tokens)
        try:
            return (image.data_ptr(), image._version, sparse.data_ptr(), sparse._version)
        except RuntimeError:
            return object()

    def step(self, image, sparse, validity=None, loss_image=None, want_depth=False, next_frame=None, frame_token=None, next_token=None):
        """forward + loss + backward + Adam in one enqueue (src/tta_main.py:610-633).
        Returns (loss_info[4] device tensor, depth or None).
        next_frame = (image, sparse) of the frame the NEXT call will pass: the part of its forward upstream of the adapted layer then runs
        beside this step (ptta_step_pipelined, include/ptta.h); same results.  The library recognises the announced frame by a TOKEN, which
        this wrapper derives from the tensors (address + torch's in-place version counter: a refilled staging buffer is a new frame); a
        caller that writes device buffers behind torch's back passes explicit non-zero `frame_token` / `next_token` integers instead."""
        image = self._f32(image, (self.n, 3, self.h, self.w))
        sparse = self._f32(sparse, (self.n, 1, self.h, self.w))
        info = torch.empty(4, device=image.device, dtype=torch.float32)
        depth = torch.empty((self.n, 1, self.h, self.w), device=image.device, dtype=torch.float32) if want_depth else None
        if next_frame is not None:
            nimg = self._f32(next_frame[0], (self.n, 3, self.h, self.w))
            nsp = self._f32(next_frame[1], (self.n, 1, self.h, self.w))
            key, nkey = self._frame_key(image, sparse), self._frame_key(nimg, nsp)
            ann = getattr(self, '_announced', None)
            if frame_token is None:
                frame_token = ann[0] if (ann is not None and ann[1] == key) else self._new_token()
            if next_token is None:
                next_token = frame_token if nkey == key else self._new_token()
            self._announced = (next_token, nkey)
            self._keep['next_frame'] = (nimg, nsp, image, sparse)
            self._chk(self.lib.ptta_step_pipelined(self.handle, ptr(image), ptr(None if loss_image is None else loss_image.contiguous()),
                                                   ptr(sparse), ptr(None if validity is None else validity.contiguous()), int(frame_token),
                                                   ptr(nimg), ptr(nsp), int(next_token), ptr(depth), ptr(info), _stream()), 'ptta_step_pipelined')
            return info, depth
        self._announced = None
        self._chk(self.lib.ptta_step(self.handle, ptr(image), ptr(None if loss_image is None else loss_image.contiguous()),
                                     ptr(sparse), ptr(None if validity is None else validity.contiguous()),
                                     ptr(depth), ptr(info), _stream()), 'ptta_step')
        return info, depth

    def _new_token(self):
        self._token = getattr(self, '_token', 0) + 1
        return self._token

    def forward_eval_last(self):
        """The scored eval forward of the frame the last step(..., next_frame=...) call adapted, reusing that frame's parameter-independent
        prefix (ptta_forward_eval_last)."""
        depth = torch.empty((self.n, 1, self.h, self.w), device=self.device, dtype=torch.float32)
        self._chk(self.lib.ptta_forward_eval_last(self.handle, ptr(depth), _stream()), 'ptta_forward_eval_last')
        return depth

    def prefix_stream(self):
        """torch view of the stream the next frame's prefix runs on (ptta_pipeline_stream): order an asynchronous producer of that frame
        (FrameStager's H2D copy) on it with `.wait_event(...)` before the step call that announces the frame."""
        if getattr(self, '_prefix_stream', None) is None:
            sp = c_void_p()
            self._chk(self.lib.ptta_pipeline_stream(self.handle, byref(sp)), 'ptta_pipeline_stream')
            self._prefix_stream = torch.cuda.ExternalStream(sp.value, device=self.device)
        return self._prefix_stream

    def set_image_norm(self, normalized_image_range):
        """Fuse Transforms.normalize_images (src/transforms.py:668-710) into the first convolution: the
        engine then takes RAW images.  Accepts the reference's `normalized_image_range` values: [0, 1],
        [-1, 1], [0, 255], or [mean, std] with 3-element lists; anything else raises ValueError like the
        reference does."""
        div, mean, std = image_norm_constants(normalized_image_range)
        m = (ctypes.c_float * 3)(*mean)
        s = (ctypes.c_float * 3)(*std)
        self._chk(self.lib.ptta_set_image_norm(self.handle, float(div), m, s), 'ptta_set_image_norm')

    def enable_stat_sync(self, group=None):
        """model.convert_syncbn() for the one-process-per-GPU run (src/tta_main.py:326): train-mode BatchNorm statistics of
        the GLOBAL batch.  The library hands a float64 buffer of partial sums to this callback, which sums it over the ranks
        with torch.distributed (RCCL on ROCm; gloo works too) on the current stream."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if world <= 1:
            self._chk(self.lib.ptta_set_stat_sync(self.handle, None, None, None, 0, 1), 'ptta_set_stat_sync')
            return
        buf = torch.zeros(4 * 2 * 1024, device=self.device, dtype=torch.float64)

        def _allreduce(user, ptr_, count, stream):
            try:
                # the collective must be ordered on the stream the library is enqueuing on (it may be an internal one)
                ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
                with torch.cuda.stream(ext):
                    dist.all_reduce(buf[:count], op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception:                       # never raise through the C frame
                import traceback
                traceback.print_exc()
                return -5
        cb = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, ctypes.c_longlong, c_void_p)(_allreduce)
        self._keep['stat_sync'] = (buf, cb)
        self._chk(self.lib.ptta_set_stat_sync(self.handle, ctypes.cast(cb, c_void_p), None, c_void_p(buf.data_ptr()), buf.numel(), world),
                  'ptta_set_stat_sync')

    def enable_rccl_sync(self, comm=None, world=None, stats=True, grads=True):
        """The reference's DDP + SyncBatchNorm run (src/tta_main.py:326-354) on the library's own RCCL communicator: the
        BatchNorm statistics exchange (`stats`) and the mean all-reduce of the adapted gradients between backward and Adam of the
        fused step (`grads`) are enqueued by the library on the step's stream -- no Python frame per collective, and the MSG_CHN
        step keeps replaying its hipGraph.  `comm`: a handle from proxytta.distributed.rccl_communicator() (default: the
        process-wide one).  None/None with no process group = a one-rank communicator (plumbing test)."""
        from . import distributed as D
        if comm is None:
            comm, world = D.rccl_communicator()
        self._keep['rccl'] = comm
        if stats:
            buf = torch.zeros(4 * 2 * 1024, device=self.device, dtype=torch.float64)
            self._keep['stat_sync'] = (buf, None)
            self._chk(self.lib.ptta_set_stat_sync_rccl(self.handle, c_void_p(comm), c_void_p(buf.data_ptr()), buf.numel(), int(world)), 'ptta_set_stat_sync_rccl')
        if grads:
            self._chk(self.lib.ptta_set_grad_sync_rccl(self.handle, c_void_p(comm)), 'ptta_set_grad_sync_rccl')
        self._rccl_grads = bool(grads)

    def disable_rccl_sync(self):
        self._chk(self.lib.ptta_set_stat_sync_rccl(self.handle, None, None, 0, 1), 'ptta_set_stat_sync_rccl')
        self._chk(self.lib.ptta_set_grad_sync_rccl(self.handle, None), 'ptta_set_grad_sync_rccl')
        self._rccl_grads = False
        self._keep.pop('stat_sync', None)

    # ---- stage-2 head trainer (include/ptta.h "Stage-2 head trainer"; src/head_main.py:464-480) ------------
    def bind_head(self, name, param, exp_avg=None, exp_avg_sq=None):
        ts = [t for t in (param, exp_avg, exp_avg_sq) if t is not None]
        for t in ts:
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        self._keep['head/' + name] = ts
        self._chk(self.lib.ptta_head_bind(self.handle, name.encode(), ptr(param), ptr(exp_avg), ptr(exp_avg_sq)), 'ptta_head_bind')

    def set_head_hparams(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, tau=0.999, adam_step=-1):
        self._chk(self.lib.ptta_head_set_hparams(self.handle, lr, betas[0], betas[1], eps, weight_decay, tau, int(adam_step), _stream()),
                  'ptta_head_set_hparams')

    def head_reload(self):
        self._chk(self.lib.ptta_head_reload(self.handle, _stream()), 'ptta_head_reload')

    def head_forward(self, image, sparse, reverse=True, want=True):
        image = self._f32(image, (self.n, 3, self.h, self.w))
        sparse = self._f32(sparse, (self.n, 1, self.h, self.w))
        emb = ref = None
        if want:
            emb = torch.empty((self.rows, self.emb_dim), device=image.device, dtype=torch.float32)
            ref = torch.empty_like(emb)
        self._chk(self.lib.ptta_head_forward(self.handle, ptr(image), ptr(sparse), int(bool(reverse)), ptr(emb), ptr(ref), _stream()),
                  'ptta_head_forward')
        return emb, ref

    def head_backward(self):
        loss = torch.empty(1, device=self.device, dtype=torch.float32)
        self._chk(self.lib.ptta_head_backward(self.handle, ptr(loss), _stream()), 'ptta_head_backward')
        return loss

    def head_adam_step(self):
        self._chk(self.lib.ptta_head_adam_step(self.handle, _stream()), 'ptta_head_adam_step')

    def head_step(self, image, sparse, reverse=True):
        image = self._f32(image, (self.n, 3, self.h, self.w))
        sparse = self._f32(sparse, (self.n, 1, self.h, self.w))
        loss = torch.empty(1, device=image.device, dtype=torch.float32)
        self._chk(self.lib.ptta_head_step(self.handle, ptr(image), ptr(sparse), int(bool(reverse)), ptr(loss), _stream()), 'ptta_head_step')
        return loss

    def head_grad(self, name, like):
        """Gradient of a head parameter after head_backward(), or None when it is outside the graph (reverse: all of proj)."""
        has = c_int(0)
        out = torch.empty_like(like)
        self._chk(self.lib.ptta_head_get_grad(self.handle, name.encode(), ptr(out), out.numel(), byref(has), _stream()), 'ptta_head_get_grad')
        return out if has.value else None

    def set_graph(self, enable):
        self._chk(self.lib.ptta_set_graph(self.handle, int(bool(enable))), 'ptta_set_graph')

    def set_option(self, key, value):
        """Per-handle switch (include/ptta.h ptta_set_option: graph, aux_stream, thru, fuse_first, fuse_head_bwd, fuse_heads, heads_v2,
        cos_in_gemm, mask_bits, stamps)."""
        self._chk(self.lib.ptta_set_option(self.handle, key.encode(), int(value)), 'ptta_set_option(%s)' % key)

    def get_option(self, key):
        v = ctypes.c_int(0)
        self._chk(self.lib.ptta_get_option(self.handle, key.encode(), byref(v)), 'ptta_get_option(%s)' % key)
        return v.value

    def profile(self, enable):
        self._chk(self.lib.ptta_profile(self.handle, int(bool(enable))), 'ptta_profile')

    def profile_read(self, klass):
        """(ms_total, algorithmic_bytes, macs, launches) of conv32 kernel class `klass` since profile(True)."""
        ms, by, mc, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0), c_int64(0)
        self._chk(self.lib.ptta_profile_read(self.handle, int(klass), byref(ms), byref(by), byref(mc), byref(n), _stream()),
                  'ptta_profile_read')
        return ms.value, by.value, mc.value, n.value

    def debug_tensor(self, name):
        n = c_int64(0)
        self._chk(self.lib.ptta_debug_tensor(self.handle, name.encode(), None, 0, byref(n), _stream()), 'ptta_debug_tensor')
        out = torch.empty(n.value, device=self.device, dtype=torch.float32)
        self._chk(self.lib.ptta_debug_tensor(self.handle, name.encode(), ptr(out), n.value, byref(n), _stream()),
                  'ptta_debug_tensor')
        return out


def op_conv32(x_nhwc, weight, bias, mode, relu_in=False, in_major=False, flip=False, dtype='fp32', naive=False, x3=False):
    """Test hook: run the hot-path 32->32 3x3 kernel on an fp32 NHWC tensor."""
    lib = _lib.load()
    b, h, w, _ = x_nhwc.shape
    ho, wo = {0: (h, w), 1: (h // 2, w // 2), 2: (2 * h, 2 * w)}[mode]
    out = torch.empty((b, ho, wo, 32), device=x_nhwc.device, dtype=torch.float32)
    rc = lib.ptta_op_conv32(ptr(x_nhwc.contiguous()), ptr(weight.contiguous()), ptr(bias), ptr(out), b, h, w, mode,
                            int(relu_in), int(in_major), int(flip), 0 if dtype == 'fp32' else 1, int(bool(naive)) | (2 if x3 else 0), _stream())
    if rc != 0:
        raise RuntimeError('ptta_op_conv32 failed (%d)' % rc)
    return out

"""Host-side mirror of the reference's plugin surface for the ProxyTTA hot path.

``ExternalModel_Adapt`` keeps the method names, argument meaning and return values of
src/external_model_adapt.py:29-660 and of the per-backbone adapter src/msg_chn_model_adapt.py
(forward / compute_loss / _prepare_head / adapt_parameters / parameters / train / eval / to /
restore_model / save_model / convert_syncbn / distributed_data_parallel), so a tta_main-style
driver runs unchanged; underneath, forward, loss, backward and Adam are libptta_hip kernels.
``step()`` and ``adapt()`` are the fused entry points the reference only has inline
(src/tta_main.py:579-636 and :504-804).
"""
import math

import torch
import torch.nn as nn

from . import synth
from .engine import ADAPTED, HEAD_TARGETS, Engine, adapted_names

CANONICAL_LOSS_TYPE = 'adapt_meta_selfsup_seq_ema_reverse'


class _Tree(nn.Module):
    """Parameter container with the reference's state_dict keys (no forward of its own)."""

    def __init__(self):
        super().__init__()

    def _leaf(self, dotted, tensor, is_param):
        mod = self
        parts = dotted.split('.')
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Tree())
            mod = getattr(mod, p)
        if is_param:
            mod.register_parameter(parts[-1], nn.Parameter(tensor))
        else:
            mod.register_buffer(parts[-1], tensor)


def _init_tensor(name, shape):
    """Shape-faithful default init (xavier for the backbone convs with bias 0.01, kaiming fan_out
    for the meta conv, PyTorch defaults for Linear/BatchNorm); real runs restore a checkpoint."""
    if name.endswith('num_batches_tracked'):
        return torch.zeros((), dtype=torch.int64)
    if name.endswith('running_mean'):
        return torch.zeros(shape)
    if name.endswith('running_var'):
        return torch.ones(shape)
    t = torch.empty(shape)
    if len(shape) == 4:
        if 'meta' in name:
            nn.init.kaiming_normal_(t, mode='fan_out', nonlinearity='relu')
        else:
            nn.init.xavier_normal_(t)
    elif len(shape) == 2:
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
    elif name.endswith('bias'):
        if '.1.' in name and ('proj' in name or 'pred' in name):
            t.zero_()
        elif 'proj' in name or 'pred' in name:
            t.uniform_(-0.04, 0.04)
        else:
            t.fill_(0.01)
    else:
        t.fill_(1.0)
    return t


class _ForwardFn(torch.autograd.Function):
    """(depth, emb, ref) = network(image, sparse); backward -> grads of the adapted parameters."""

    @staticmethod
    def forward(ctx, adapter, image, sparse, *adapted):
        eng = adapter._engine(image)
        depth, emb, ref = eng.forward_train(image, sparse)
        ctx.eng = eng
        ctx.params = dict(zip(eng.adapted, adapted))
        ctx.mark_non_differentiable(emb)
        return depth, emb, ref

    @staticmethod
    def backward(ctx, g_depth, g_emb, g_ref):
        if g_depth is None:
            g_depth = torch.zeros((ctx.eng.n, 1, ctx.eng.h, ctx.eng.w), device=ctx.eng.device)
        grads = ctx.eng.backward_all(g_depth, g_ref, ctx.params)
        return (None, None, None) + tuple(grads)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, image, depth, sparse, validity, emb, ref, w_sd, w_sm, w_cos):
        info = eng.loss_forward(image, depth, sparse, validity, emb, ref, w_sd, w_sm, w_cos)
        # the gate and the gradient scalars live on the device until the next loss call
        ctx.eng, ctx.args = eng, (image, depth.detach(), sparse, validity, None if emb is None else emb.detach(),
                                  None if ref is None else ref.detach())
        return info[0].clone(), info

    @staticmethod
    def backward(ctx, g_loss, g_info):
        gd, gr = ctx.eng.loss_backward(*ctx.args)
        gd = gd * g_loss
        if gr is not None:
            gr = gr * g_loss
        return None, None, gd, None, None, None, gr, None, None, None


class _PrepareLossFn(torch.autograd.Function):
    """prepare_loss (src/external_model_adapt.py:524-541) of the last head forward; its backward hands the head parameters
    the gradients ptta_head_backward already produced (parameters outside the graph get None, as in the reference)."""

    @staticmethod
    def forward(ctx, eng, names, *head_params):
        loss = eng.head_backward()
        ctx.grads = [eng.head_grad(k, p) for k, p in zip(names, head_params)]
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g_loss):
        return (None, None) + tuple(None if g is None else g * g_loss for g in ctx.grads)


class MsgChnModel_Adapt(object):
    """Counterpart of src/msg_chn_model_adapt.py:11-556 on libptta_hip."""

    def __init__(self, max_predict_depth=5.0, inpainting=False, device=torch.device('cuda'), dtype='fp32',
                 max_input_depth=None):
        self.max_predict_depth = max_predict_depth
        self.max_input_depth = max_input_depth
        self.device = device
        self.dtype = dtype
        self.training = True
        self.prepare_mode = None
        self.model = _Tree()
        for k, s in synth.msg_chn_keys():                     # heads + meta layer come with _prepare_head
            if not k.startswith(('rgb_encoder', 'depth_')):
                continue
            self.model._leaf(k, _init_tensor(k, s), not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
        self._engines = {}
        self._opt_state = {}
        self._adam_t = 0              # torch.optim.Adam's state['step'], shared by every per-shape engine
        self.sync_bn = False
        self.hparams = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                            w_sparse_depth=1.0, w_smoothness=1.0, w_cos=1.0)
        self.total_time = self.train_time = self.eval_time = 0.0
        self.to(device)

    def _clear_engines(self):
        """Release every per-shape engine (workspaces are GBs at full size: do not wait for __del__)."""
        engines, self._engines = getattr(self, '_engines', {}), {}
        for eng in engines.values():
            eng.close()

    def _timed(self, fn, loss_type):
        """The reference's wall-clock hook: `'time' in loss_type` brackets the network forward with
        torch.cuda.synchronize() + time.time() and accumulates train_time / eval_time / total_time
        (network_exp_msg_chn_adapt.py:338-340,407-414); read back by forward(loss_type='get_time')."""
        if 'time' not in loss_type:
            return fn()
        import time
        torch.cuda.synchronize()
        t0 = time.time()
        out = fn()
        torch.cuda.synchronize()
        dt = time.time() - t0
        if self.training:
            self.train_time += dt
        else:
            self.eval_time += dt
        self.total_time += dt
        return out

    # ---- reference surface -----------------------------------------------------------------
    def _prepare_head(self, mode=''):
        """network_adapt._prepare_head (network_exp_msg_chn_adapt.py:1022-1087)."""
        if 'meta' not in mode or 'selfsup' not in mode or 'ema' not in mode or 'seq' not in mode:
            raise NotImplementedError('hot path covers prepare_mode meta_selfsup_seq_{1layer}_ema, got %r' % mode)
        if '1layer' not in mode and '2layers' not in mode:
            raise NotImplementedError('meta layer of %r is not on the accelerated path (1layer / 2layers are)' % mode)
        self.prepare_mode = mode
        self.meta = '2layers' if '2layers' in mode else '1layer'
        self.adapted = adapted_names(self.meta)
        for k, s in synth.msg_chn_keys(mode):
            if k.startswith(('proj', 'pred', 'conv1_rgb_meta')):
                self.model._leaf(k, _init_tensor(k, s).to(self.device),
                                 not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
        self._clear_engines()

    def parameters(self):
        return list(self.model.parameters())

    def adapt_parameters(self, mode=None):
        """mode 'meta': every parameter whose name contains 'meta' (msg_chn_model_adapt.py:392-396)."""
        if mode != 'meta':
            raise NotImplementedError("adapt_mode %r: only 'meta' is on the accelerated path" % mode)
        return nn.ParameterList([p for n, p in self.model.named_parameters() if 'meta' in n])

    # ---- stage 2 (src/head_main.py:259-276, 464-480) ------------------------------------------------------
    def prepare_parameters(self, mode=''):
        """mode containing 'head' (head_main.py:268 passes 'head_selfsup_ema'): the heads are RE-CREATED -- the reference calls
        `_prepare_head(mode)` again here (src/msg_chn_model_adapt.py:295-298), stage 2 trains them from scratch -- and the
        proj / pred parameters without proj_t are returned (:300-304)."""
        if 'head' not in mode or 'selfsup' not in mode or 'ema' not in mode:
            raise NotImplementedError("prepare_parameters(%r): only the head trainer's 'head_selfsup_ema' is on the accelerated path" % mode)
        if self.prepare_mode is None:
            raise RuntimeError('_prepare_head(prepare_mode) first (head_main.py:259)')
        params = dict(self.model.named_parameters())
        with torch.no_grad():
            for k, shape in self._state_keys():
                if k.startswith(('proj.', 'pred.')):
                    t = params.get(k, None)
                    tgt = t if t is not None else self.model.state_dict()[k]
                    tgt.copy_(_init_tensor(k, shape).to(tgt.device))
            state = self.model.state_dict()
            for k in state:
                if k.startswith('proj_t.'):
                    state[k].copy_(state['proj.' + k[len('proj_t.'):]])          # copy.deepcopy(self.proj)
        self._clear_engines()
        self._head_names = [k for k, _ in self.model.named_parameters() if ('proj' in k or 'pred' in k) and '_t' not in k]
        return [params[k] for k in self._head_names]

    # (key, shape) table of this backbone's state dict and whether `reverse` leaves proj without a gradient (MSG_CHN: its reference branch is
    # proj itself, network_exp_msg_chn_adapt.py:691-694; NLSPN / CostDCNet: the EMA target proj_t, so proj trains in both directions)
    heads_reverse_freezes_proj = True

    def _state_keys(self):
        return synth.msg_chn_keys(self.prepare_mode)

    def bind_head_optimizer(self, optimizer, tau=0.999):
        """Share Adam state with a torch.optim.Adam built on prepare_parameters(): `head_step` then updates its exp_avg /
        exp_avg_sq / step in place, so optimizer.state_dict() (save_model) stays meaningful."""
        params = dict(self.model.named_parameters())
        g = optimizer.param_groups[0]
        self._head_hp = dict(lr=g['lr'], betas=tuple(g['betas']), eps=g['eps'], weight_decay=g['weight_decay'], tau=tau)
        self._head_opt = optimizer
        self._head_state = {}
        t = 0
        for k in self._head_names:
            st = optimizer.state[params[k]]
            if 'exp_avg' not in st:
                st['step'] = torch.tensor(0.0)
                st['exp_avg'] = torch.zeros_like(params[k].data)
                st['exp_avg_sq'] = torch.zeros_like(params[k].data)
            self._head_state[k] = st
            t = max(t, int(float(st['step'])))
        self._head_t = t
        self._clear_engines()

    def _head_engine(self, image):
        eng = self._engine(image)
        if not getattr(eng, '_heads_bound', False):
            if getattr(self, '_head_names', None) is None:
                raise RuntimeError('prepare_parameters(\'head_selfsup_ema\') must be called before a head forward (head_main.py:268)')
            params = dict(self.model.named_parameters())
            state = getattr(self, '_head_state', None) or {}
            for k in self._head_names:
                st = state.get(k) or {'exp_avg': torch.zeros_like(params[k].data), 'exp_avg_sq': torch.zeros_like(params[k].data)}
                state[k] = st
                eng.bind_head(k, params[k].data, st['exp_avg'], st['exp_avg_sq'])
            self._head_state = state
            for k in HEAD_TARGETS:
                eng.bind_head(k, params[k].data)
            hp = getattr(self, '_head_hp', None) or dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, tau=0.999)
            eng.set_head_hparams(adam_step=getattr(self, '_head_t', 0), **hp)
            eng._head_t = getattr(self, '_head_t', 0)
            eng._heads_bound = True
        return eng

    def _sync_head_step(self, eng):
        """Counterpart of _sync_adam_step for the stage-2 heads: every (batch, height, width) engine has its own device-side
        step count, torch.optim.Adam has ONE `step` per parameter -- a second engine (the last short batch of a loader) must
        continue the bias correction where the first one stands."""
        t = getattr(self, '_head_t', 0)
        if getattr(eng, '_head_t', None) != t:
            hp = getattr(self, '_head_hp', None) or dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, tau=0.999)
            eng.set_head_hparams(adam_step=t, **hp)
            eng._head_t = t

    def head_forward(self, image, sparse_depth, loss_type):
        """forward(loss_type='head_selfsup_seq_ema[_reverse]') in training mode: (None, embedding, reference)
        (network_exp_msg_chn_adapt.py:610-699); the EMA of proj_t runs first (:682,:691)."""
        if 'ema' not in loss_type or 'adapt' in loss_type:
            raise NotImplementedError('loss_type %r is not on the accelerated path' % loss_type)
        eng = self._head_engine(image)
        eng.head_reload()                 # a torch optimizer may have stepped the bound parameters since the last call
        emb, ref = eng.head_forward(image, sparse_depth, 'reverse' in loss_type)
        self._head_last = eng
        return None, emb, ref

    def prepare_loss(self, embedding, reference):
        """compute_loss(loss_type='prepare'): the loss of the LAST head forward (embedding / reference are its outputs)."""
        eng = getattr(self, '_head_last', None)
        if eng is None:
            raise RuntimeError('compute_loss(loss_type=\'prepare\') follows a head forward')
        params = dict(self.model.named_parameters())
        loss = _PrepareLossFn.apply(eng, tuple(self._head_names), *[params[k] for k in self._head_names])
        return loss, {'loss': loss}

    def head_step(self, image, sparse_depth, loss_type):
        """Fused stage-2 step (head_main.py:464-480): EMA, forward, prepare loss, backward, Adam -- one library call."""
        eng = self._head_engine(image)
        # the library keeps ONE step count for the head tensors, torch.optim.Adam one per parameter: in `reverse` mode only
        # `pred` steps, so a run that switches between the two loss types would give proj pred's count -> refused
        rev = 'reverse' in loss_type
        if self.heads_reverse_freezes_proj and getattr(self, '_head_mode', rev) != rev and getattr(self, '_head_t', 0) > 0:
            raise NotImplementedError('head_step: switching between reverse and non-reverse loss types inside one run is not supported '
                                      '(per-parameter Adam step counts would diverge)')
        self._head_mode = rev
        self._sync_head_step(eng)
        loss = eng.head_step(image, sparse_depth, rev)
        self._head_t = getattr(self, '_head_t', 0) + 1
        eng._head_t = self._head_t
        live = self._head_names if (not rev or not self.heads_reverse_freezes_proj) else [k for k in self._head_names if k.startswith('pred')]
        for k in live:
            st = self._head_state[k]
            if 'step' in st:
                st['step'] += 1
        return loss

    def train(self):
        self.training = True
        self.model.train()

    train_prepare = train
    train_meta = train

    def eval(self):
        self.training = False
        self.model.eval()

    def to(self, device):
        self.device = device
        self.model.to(device)
        self._clear_engines()

    def data_parallel(self):
        raise NotImplementedError('one process per GPU; use proxytta.distributed')

    def distributed_data_parallel(self, rank):
        """The reference wraps in DDP (msg_chn_model_adapt.py:476-480) and all-reduces every
        gradient; here only the adapted-parameter gradients are reduced: drive the step through
        proxytta.distributed.shared_parameter_step (one flat all-reduce), not through loss.backward()."""
        self.ddp_rank = rank

    def convert_syncbn(self, apex=False):
        """SyncBatchNorm.convert_sync_batchnorm (msg_chn_model_adapt.py:547-556): from now on train-mode BatchNorm
        statistics are those of the GLOBAL batch.  With one rank that is what the engine computes anyway; with
        several ranks the statistics are exchanged inside proxytta.distributed.shared_parameter_step."""
        self.sync_bn = True
        for eng in self._engines.values():
            eng.enable_stat_sync()

    def restore_model(self, restore_path, optimizer=None):
        ckpt = torch.load(restore_path, map_location=self.device)
        self.load_state_dict(ckpt['net'])
        if optimizer is not None and 'optimizer' in ckpt:
            optimizer.load_state_dict(ckpt['optimizer'])
            if getattr(self, '_optimizer', None) is optimizer:
                self.bind_optimizer(optimizer)      # load_state_dict REPLACED exp_avg / exp_avg_sq / step: rebind them
        return optimizer, ckpt.get('train_step', 0)

    def save_model(self, checkpoint_path, step, optimizer, meanvar=None):
        torch.save({'net': self.model.state_dict(), 'optimizer': optimizer.state_dict() if optimizer else {},
                    'train_step': step}, checkpoint_path)

    def load_state_dict(self, state):
        with torch.no_grad():
            own = self.model.state_dict()
            missing = [k for k in own if k not in state]
            if missing:
                raise KeyError('missing keys: %s' % missing[:5])
            for k, v in own.items():
                v.copy_(torch.as_tensor(state[k]).to(v.device))
        for eng in self._engines.values():
            eng.load_state_dict(self.model.state_dict())

    # ---- engine plumbing --------------------------------------------------------------------
    def _engine(self, image):
        n, _, h, w = image.shape
        key = (n, h, w)
        eng = self._engines.get(key)
        if eng is None:
            if self.prepare_mode is None:
                raise RuntimeError('_prepare_head(mode) must be called before forward (tta_main.py:322)')
            eng = Engine(n, h, w, dtype=self.dtype, max_input_depth=self.max_input_depth, meta=self.meta, **self.hparams)
            eng.load_state_dict(self.model.state_dict())
            params = dict(self.model.named_parameters())
            for name in self.adapted:
                p = params[name]
                st = self._opt_state.setdefault(name, {'exp_avg': torch.zeros_like(p.data), 'exp_avg_sq': torch.zeros_like(p.data)})
                eng.bind_adapted(name, p.data, st['exp_avg'], st['exp_avg_sq'])
            if getattr(self, '_image_norm', None) is not None:
                eng.set_image_norm(self._image_norm)
            eng._t = 0
            if self.sync_bn:
                eng.enable_stat_sync()          # no-op with one rank
            self._engines[key] = eng
        return eng

    def _sync_adam_step(self, eng):
        """Every (batch, height, width) has its own engine and device-side step counter; torch.optim.Adam has ONE
        `step` per parameter, so the bias correction must continue across shape changes (last short batch of a
        loader with drop_last=False, src/tta_main.py:281)."""
        if eng._t != self._adam_t:
            eng.set_adam_step(self._adam_t)
            eng._t = self._adam_t

    def set_image_norm(self, normalized_image_range):
        """Take RAW images from now on: Transforms.normalize_images (src/transforms.py:668-710, called at
        src/tta_main.py:454-464,652) is fused into the first convolution's loads.  The loss keeps seeing the
        raw image, as in the reference (tta_main.py:610 vs :620)."""
        from .engine import image_norm_constants
        image_norm_constants(normalized_image_range)        # validate now (raises ValueError like the reference)
        self._image_norm = normalized_image_range
        for eng in self._engines.values():
            eng.set_image_norm(normalized_image_range)

    def set_hparams(self, **kw):
        self.hparams.update({k: v for k, v in kw.items() if k in self.hparams})
        if 'max_input_depth' in kw:
            self.max_input_depth = kw['max_input_depth']
        for eng in self._engines.values():
            eng.set_hparams(**kw)

    def bind_optimizer(self, optimizer):
        """Share Adam state with a torch.optim.Adam built on adapt_parameters(): its exp_avg /
        exp_avg_sq tensors become the buffers the fused step updates, so optimizer.state_dict()
        (save_model) stays meaningful."""
        params = dict(self.model.named_parameters())
        g = optimizer.param_groups[0]
        self.set_hparams(lr=g['lr'], betas=tuple(g['betas']), eps=g['eps'], weight_decay=g['weight_decay'])
        for name in self.adapted:
            p = params[name]
            st = optimizer.state[p]
            if 'exp_avg' not in st:
                st['step'] = torch.tensor(0.0)
                st['exp_avg'] = torch.zeros_like(p.data)
                st['exp_avg_sq'] = torch.zeros_like(p.data)
            self._opt_state[name] = st
        # the library keeps ONE count of optimizer.step() calls; a tensor the list names twice shows twice that in torch's state
        self._adam_t = int(float(optimizer.state[params[self.adapted[0]]]['step']))
        self._clear_engines()
        self._optimizer = optimizer

    # ---- forward / loss -----------------------------------------------------------------------
    def forward(self, image, sparse_depth, intrinsics=None, crop_mask=None, loss_type='pretrain'):
        if 'head' in loss_type and 'init_meta' not in loss_type and loss_type != 'prepare':
            if not self.training:
                raise NotImplementedError('the head forward is a training-mode call (head_main.py:441)')
            return self.head_forward(image, sparse_depth, loss_type)
        if not ('meta' in loss_type and 'selfsup' in loss_type):
            raise NotImplementedError('loss_type %r is not on the accelerated path' % loss_type)
        if self.training and 'adapt' in loss_type:
            params = dict(self.model.named_parameters())
            return self._timed(lambda: _ForwardFn.apply(self, image, sparse_depth, *[params[k] for k in self.adapted]), loss_type)
        with torch.no_grad():
            return self._timed(lambda: self._engine(image).forward_eval(image, sparse_depth), loss_type)

    def step(self, image, sparse_depth, validity_map=None, loss_image=None, want_depth=False, next_frame=None):
        """The fused step.  next_frame = (image, sparse_depth) of the frame the NEXT call will pass (the data loader's look-ahead): the part of
        its forward upstream of the adapted layer then runs beside this step (Engine.step / ptta_step_pipelined); same results."""
        eng = self._engine(image)
        self._sync_adam_step(eng)
        if next_frame is not None and (getattr(eng, 'backbone', 'msg_chn') != 'msg_chn' or tuple(next_frame[0].shape) != tuple(image.shape)):
            next_frame = None                       # other backbones have nothing upstream of their adapted layers; a new shape is a new engine
        info, depth = eng.step(image, sparse_depth, validity_map, loss_image, want_depth, **({'next_frame': next_frame} if next_frame is not None else {}))
        self._adam_t += 1
        eng._t = self._adam_t
        opt = getattr(self, '_optimizer', None)
        if opt is not None:
            never = getattr(self, '_never_stepped', None)
            skip = {id(p) for k, p in self.model.named_parameters() if k in never} if never else ()
            for p in opt.param_groups[0]['params']:             # as listed: a tensor named twice is stepped twice
                if 'step' in opt.state[p] and id(p) not in skip:
                    opt.state[p]['step'] += 1
        return info, depth


def eng_is_padded(eng):
    """Frame sizes not divisible by 16 run the dual-corner padding and the plain step (no pipelining state to reuse)."""
    return (eng.h % 16) != 0 or (eng.w % 16) != 0


class ExternalModel_Adapt(object):
    """Counterpart of src/external_model_adapt.py:29 (model_name 'msg_chn', 'nlspn', 'costdcnet')."""

    def __init__(self, model_name, min_predict_depth, max_predict_depth, max_input_depth=None, offset=False,
                 from_scratch=False, dataset_name=None, device=torch.device('cuda'), dtype='fp32'):
        self.model_name = model_name
        self.dataset_name = dataset_name
        self.device = device
        self.max_predict_depth = max_predict_depth
        self.max_input_depth = max_input_depth
        if model_name == 'msg_chn':
            self.model = MsgChnModel_Adapt(device=device, max_predict_depth=max_predict_depth, dtype=dtype,
                                           max_input_depth=max_input_depth)
        elif model_name == 'nlspn':
            from .nlspn import NlspnModel_Adapt
            self.model = NlspnModel_Adapt(device=device, max_depth=max_predict_depth, offset=offset, dataset_name=dataset_name,
                                          from_scratch=from_scratch, max_input_depth=max_input_depth)
        elif 'costdcnet' in model_name:
            from .costdcnet import CostDCNetModel_Adapt
            self.model = CostDCNetModel_Adapt(device=device, max_depth=max_predict_depth, max_input_depth=max_input_depth)
        else:
            raise ValueError('Unsupported depth completion model: {}'.format(model_name))

    def forward(self, image, sparse_depth, crop_mask=None, intrinsics=None, loss_type='pretrain'):
        if loss_type == 'get_time':
            m = self.model
            return [m.total_time, m.train_time, m.eval_time]
        # the clamp of external_model_adapt.py:108 happens inside the library (prep kernel)
        return self.model.forward(image=image, sparse_depth=sparse_depth, intrinsics=intrinsics,
                                  crop_mask=crop_mask, loss_type=loss_type)

    def compute_loss(self, input_rgb=None, output_depth=None, sparse_depth=None, validity_map=None,
                     embedding=None, reference=None, w_loss_sparse_depth=1.0, w_loss_smoothness=1.0,
                     w_loss_cos=1.0, loss_type='adapt', **unused):
        if 'prepare' in loss_type:                  # src/external_model_adapt.py:155-158
            return self.model.prepare_loss(embedding, reference)
        if 'adapt' not in loss_type:
            raise NotImplementedError('loss_type %r is not on the accelerated path' % loss_type)
        eng = self.model._engine(input_rgb)
        loss, info = _LossFn.apply(eng, input_rgb, output_depth, sparse_depth, validity_map, embedding, reference,
                                   w_loss_sparse_depth, w_loss_smoothness, w_loss_cos)
        info = info.detach()
        return loss, {'loss': loss, 'loss_smooth': info[1], 'loss_sparse_depth': info[2], 'loss_cos': info[3]}

    def _prepare_head(self, mode=''):
        self.model._prepare_head(mode=mode)

    def parameters(self):
        return self.model.parameters()

    def adapt_parameters(self, mode=''):
        return self.model.adapt_parameters(mode=mode)

    def prepare_parameters(self, mode=''):
        return self.model.prepare_parameters(mode)

    # (key, shape) table of this backbone's state dict and whether `reverse` leaves proj without a gradient (MSG_CHN: its reference branch is
    # proj itself, network_exp_msg_chn_adapt.py:691-694; NLSPN / CostDCNet: the EMA target proj_t, so proj trains in both directions)
    heads_reverse_freezes_proj = True

    def _state_keys(self):
        return synth.msg_chn_keys(self.prepare_mode)

    def bind_head_optimizer(self, optimizer, tau=0.999):
        self.model.bind_head_optimizer(optimizer, tau)

    def head_step(self, image, sparse_depth, loss_type='head_selfsup_seq_ema_reverse'):
        """One stage-2 training step (src/head_main.py:464-480) in one library call; returns the loss (1 device float)."""
        return self.model.head_step(image, sparse_depth, loss_type)

    def train(self, meta=False, prepare=False):
        self.model.train()

    def eval(self):
        self.model.eval()

    def to(self, device):
        self.device = device
        self.model.to(device)

    def data_parallel(self):
        self.model.data_parallel()

    def distributed_data_parallel(self, rank):
        self.model.distributed_data_parallel(rank)

    def restore_model(self, restore_path, optimizer=None, learning_schedule=None, learning_rates=None,
                      n_step_per_epoch=None):
        return self.model.restore_model(restore_path=restore_path, optimizer=optimizer)

    def save_model(self, checkpoint_path, step, optimizer, meanvar=None):
        self.model.save_model(checkpoint_path, step, optimizer, meanvar)

    def convert_syncbn(self, apex=False):
        self.model.convert_syncbn(apex)

    # ---- fused entry points -----------------------------------------------------------------
    def set_image_norm(self, normalized_image_range):
        self.model.set_image_norm(normalized_image_range)

    def step(self, image, sparse_depth, validity_map=None, loss_image=None, want_depth=False, next_frame=None):
        """One TTA step (src/tta_main.py:610-633) in one library call; returns (loss_info[4], depth).  next_frame: see the adapter's step."""
        if next_frame is not None and isinstance(self.model, MsgChnModel_Adapt) and type(self.model) is MsgChnModel_Adapt:
            return self.model.step(image, sparse_depth, validity_map, loss_image, want_depth, next_frame=next_frame)
        return self.model.step(image, sparse_depth, validity_map, loss_image, want_depth)

    def adapt(self, image, sparse_depth, inner_iter=1, validity_map=None, loss_image=None, next_frame=None):
        """Per-frame adaptation = inner_iter steps then the scored eval forward
        (src/tta_main.py:579-636 and :729-736).  Returns (depth, loss_info of the last step).
        next_frame = (image, sparse_depth) the next call will pass: announced to the LAST step of this frame (frame pipelining)."""
        info = None
        self.model.train()
        piped = False
        for it in range(inner_iter):
            if it == inner_iter - 1 and next_frame is not None:
                info, _ = self.step(image, sparse_depth, validity_map, loss_image, next_frame=next_frame)
                piped = type(self.model) is MsgChnModel_Adapt and tuple(next_frame[0].shape) == tuple(image.shape)
                continue
            info, _ = self.model.step(image, sparse_depth, validity_map, loss_image)
        self.model.eval()
        if piped:                    # the scored forward from the adapted frame's own prefix (ptta_forward_eval_last)
            eng = self.model._engine(image)
            if getattr(eng.lib, 'ptta_forward_eval_last', None) is not None and not eng_is_padded(eng):
                return eng.forward_eval_last(), info
        depth = self.model.forward(image, sparse_depth, loss_type=CANONICAL_LOSS_TYPE)
        return depth, info


ExternalModelAdapt = ExternalModel_Adapt


class OutlierRemoval(object):
    """Counterpart of src/net_utils.py:750-811 on libptta_hip (one fused stencil instead of 6 ATen kernels)."""

    def __init__(self, kernel_size=7, threshold=1.5):
        self.kernel_size = kernel_size
        self.threshold = threshold
        self._scratch = None

    def remove_outliers(self, sparse_depth, validity_map):
        from . import _lib
        from .engine import _stream
        from ._lib import ptr
        lib = _lib.load()
        assert sparse_depth.is_cuda and sparse_depth.dtype == torch.float32 and sparse_depth.shape == validity_map.shape
        n, _, h, w = sparse_depth.shape
        if self._scratch is None or self._scratch.device != sparse_depth.device:
            self._scratch = torch.empty(1024, device=sparse_depth.device, dtype=torch.float32)
        sd, vm = sparse_depth.contiguous(), validity_map.contiguous()
        sd_out, vm_out = torch.empty_like(sd), torch.empty_like(vm)
        rc = lib.ptta_outlier_removal(ptr(sd), ptr(vm), ptr(sd_out), ptr(vm_out), n, h, w, int(self.kernel_size),
                                      float(self.threshold), ptr(self._scratch), _stream())
        if rc != 0:
            raise RuntimeError('ptta_outlier_removal failed (%d)' % rc)
        return sd_out, vm_out


_metric_scratch = {}


def eval_metrics(output_depth, ground_truth, min_evaluate_depth=0.0, max_evaluate_depth=100.0):
    """MAE / RMSE (mm) and iMAE / iRMSE (1/km) of src/tta_main.py:779-798 as a 4-element device tensor."""
    from . import _lib
    from ._lib import ptr
    from .engine import _stream
    lib = _lib.load()
    dev = output_depth.device
    if dev not in _metric_scratch:
        _metric_scratch[dev] = torch.empty(2048, device=dev, dtype=torch.float64)
    out = torch.empty(4, device=dev, dtype=torch.float32)
    o, g = output_depth.contiguous().float(), ground_truth.contiguous().float()
    rc = lib.ptta_eval_metrics(ptr(o), ptr(g), o.numel(), float(min_evaluate_depth), float(max_evaluate_depth),
                               ptr(_metric_scratch[dev]), ptr(out), _stream())
    if rc != 0:
        raise RuntimeError('ptta_eval_metrics failed (%d)' % rc)
    return out

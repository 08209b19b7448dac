"""Host-side mirror of src/nlspn_model_adapt.py (NLSPNModel_Adapt, the per-backbone adapter the reference selects
with model_name='nlspn', src/external_model_adapt.py:61-63) on libptta_hip.

What is on the accelerated path: the canonical TTA flow of bash/adapt/adapt_nlspn_*.sh --
prepare_mode 'meta_selfsup_seq_1layer_ema', adapt_mode 'meta_bn', loss_type 'adapt_meta_selfsup_seq_ema_reverse':
the reference's own call sequence ``forward`` / ``compute_loss`` / ``loss.backward()`` / ``optimizer.step()``
(src/tta_main.py:610-633, autograd Functions over ptta_forward_train / ptta_loss_* / ptta_backward), the fused
``step()`` (the same in one library call, Adam on device for the 88 adapted tensors) and the eval ``forward()`` (:729-736).
Any frame size from 16 x 16 up: encoder maps of odd size make the decoder maps one row / column larger and they are cropped
before each concatenation as in nlspnmodel_adapt.py:474-490 (NYUv2's 228 x 304 is such a size).
The eval path's biharmonic hole filling (src/nlspn_model_adapt.py:124-127, skimage on the CPU) is not applied: exact
zeros of the clamped output are returned as zeros.
"""
import torch
import torch.nn as nn

from . import synth
from .engine import Engine
from .model import MsgChnModel_Adapt, _ForwardFn, _init_tensor, _Tree

_BUFFERS = ('running_mean', 'running_var', 'num_batches_tracked')


def nlspn_adapted_names(keys, syncbn=False):
    """adapt_parameters('meta_bn') (src/nlspn_model_adapt.py:322-337): parameters whose name contains 'meta', then
    weight / bias of every BatchNorm2d in module order (the heads' BatchNorm1d are not BatchNorm2d).  After
    convert_syncbn() (src/tta_main.py:326 runs it BEFORE adapt_parameters) every BatchNorm is a SyncBatchNorm and the
    isinstance test also matches the heads': proj.1, proj_t.1, pred.1 join the list (94 tensors)."""
    names = [k for k in keys if 'meta' in k]
    for k in keys:
        if k.endswith('.running_mean') and (syncbn or not k.startswith(('proj', 'pred'))):
            pre = k[:-len('.running_mean')]
            names += [pre + '.weight', pre + '.bias']
    return names


class NlspnModel_Adapt(MsgChnModel_Adapt):
    """Counterpart of src/nlspn_model_adapt.py:13-486."""

    def __init__(self, device=torch.device('cuda'), max_depth=100.0, inpainting=False, use_pretrained=False, dataset_name=None,
                 from_scratch=False, offset=False, max_input_depth=None):
        self.legacy = bool(offset)                            # args.legacy = offset (src/nlspn_model_adapt.py:62)
        self.max_predict_depth = max_depth
        self.max_depth = max_depth
        self.max_input_depth = max_input_depth
        self.device = device
        self.dtype = 'fp32'
        self.training = True
        self.prepare_mode = None
        self.model = _Tree()
        for k, s in synth.nlspn_keys():
            if k.startswith(('proj', 'pred', 'conv1_rgb_meta')):
                continue                                      # come with _prepare_head
            self.model._leaf(k, self._init(k, s), not k.endswith(_BUFFERS))
        self._engines = {}
        self._opt_state = {}
        self._adam_t = 0
        self.sync_bn = False
        self.hparams = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=1.0, w_cos=1.0)
        self.total_time = self.train_time = self.eval_time = 0.0
        self.fill_holes = True            # eval forward: biharmonic hole filling like the reference (False: the raw network output)
        self.to(device)

    @staticmethod
    def _init(name, shape):
        if name == 'prop_layer.aff_scale_const':
            return torch.full((1,), 0.5 * 8)
        if name in ('prop_layer.w', 'prop_layer.w_conf'):
            return torch.ones(shape)
        if name == 'prop_layer.b' or name.startswith('prop_layer.conv_offset_aff'):
            return torch.zeros(shape)                         # nlspnmodel_adapt.py:221-222: zero-initialised
        return _init_tensor(name, shape)

    heads_reverse_freezes_proj = False          # stage 2: ref = proj_t(...), proj trains in both directions (csrc/ghead.hip)

    def _state_keys(self):
        return [(k, s) for k, s in synth.nlspn_keys(self.prepare_mode) if not (k.startswith('enc2d.') and '.downsample.1.' in k)]

    def _prepare_head(self, mode=''):
        """NLSPNModel_Adapt._prepare_head (nlspnmodel_adapt.py:1333-1396)."""
        if 'meta' not in mode or 'selfsup' not in mode or 'ema' not in mode or 'seq' not in mode or '1layer' not in mode:
            raise NotImplementedError('hot path covers prepare_mode meta_selfsup_seq_1layer_ema, got %r' % mode)
        self.prepare_mode = mode
        self.meta = '1layer'
        for k, s in synth.nlspn_keys(mode):
            if k.startswith(('proj', 'pred', 'conv1_rgb_meta')):
                self.model._leaf(k, _init_tensor(k, s).to(self.device), not k.endswith(_BUFFERS))
        self.adapted = nlspn_adapted_names([k for k, _ in synth.nlspn_keys(mode)], self.sync_bn)
        self._clear_engines()

    def convert_syncbn(self, apex=False):
        super().convert_syncbn(apex)
        if self.prepare_mode is not None:           # heads exist: the adapted list grows to the 94 tensors of the DDP run
            self.adapted = nlspn_adapted_names([k for k, _ in synth.nlspn_keys(self.prepare_mode)], True)
            self._clear_engines()

    def adapt_parameters(self, mode=None):
        if mode != 'meta_bn':
            raise NotImplementedError("adapt_mode %r: only 'meta_bn' (the NLSPN scripts' mode) is on the accelerated path" % mode)
        params = dict(self.model.named_parameters())
        return nn.ParameterList([params[k] for k in self.adapted])

    def _engine(self, image):
        n, _, h, w = image.shape
        key = (n, h, w)
        eng = self._engines.get(key)
        if eng is None:
            if self.prepare_mode is None:
                raise RuntimeError('_prepare_head(mode) must be called before forward (tta_main.py:322)')
            eng = Engine(n, h, w, backbone='nlspn', legacy_offset=self.legacy, max_input_depth=self.max_input_depth,
                         syncbn_adapted=self.sync_bn, **self.hparams)
            assert eng.adapted == self.adapted, 'adapted parameter list drifted from the library'
            eng.load_state_dict(self.model.state_dict())     # the heads' BatchNorm1d buffers are bound and updated in place
            params = dict(self.model.named_parameters())
            for name in self.adapted:
                p = params[name]
                st = self._opt_state.setdefault(name, {'exp_avg': torch.zeros_like(p.data), 'exp_avg_sq': torch.zeros_like(p.data)})
                eng.bind_adapted(name, p.data, st['exp_avg'], st['exp_avg_sq'])
            if getattr(self, '_image_norm', None) is not None:
                eng.set_image_norm(self._image_norm)
            eng._t = 0
            if self.sync_bn:
                eng.enable_stat_sync()
            self._engines[key] = eng
        return eng

    def forward(self, image, sparse_depth, intrinsics=None, crop_mask=None, loss_type='pretrain'):
        if 'head' in loss_type and 'init_meta' not in loss_type and loss_type != 'prepare':       # stage 2 (src/head_main.py:464-468)
            if not self.training:
                raise NotImplementedError('the head forward is a training-mode call (head_main.py:441)')
            return self.head_forward(image, sparse_depth, loss_type)
        if self.training and 'adapt' in loss_type:
            # (depth, emb, ref) with autograd edges to the 88 adapted tensors: loss.backward() runs ptta_loss_backward +
            # ptta_backward and fills their .grad (src/tta_main.py:610-632)
            params = dict(self.model.named_parameters())
            return self._timed(lambda: _ForwardFn.apply(self, image, sparse_depth, *[params[k] for k in self.adapted]), loss_type)
        with torch.no_grad():
            out = self._timed(lambda: self._engine(image).forward_eval(image, sparse_depth), loss_type)
        # Fill in any holes with inpainting (src/nlspn_model_adapt.py:124-127): exact zeros left by the final clamp are filled by
        # biharmonic interpolation on the host, as in the reference (scikit-image there; proxytta/inpaint.py restates it on scipy).
        # One flag comes back first: a map without holes -- the usual case -- never leaves the device.
        if self.fill_holes and 'head' not in loss_type and bool((out == 0).any()):
            from .inpaint import inpainting
            out = torch.from_numpy(inpainting(out.detach().cpu().numpy())).to(out.device)
        return out

    def save_model(self, checkpoint_path, step, optimizer, meanvar=None):
        ckpt = {'net': self.model.state_dict(), 'optimizer': optimizer.state_dict() if optimizer else {}, 'train_step': step}
        if meanvar is not None:
            ckpt['meanvar'] = meanvar                         # src/nlspn_model_adapt.py:473-475
        torch.save(ckpt, checkpoint_path)

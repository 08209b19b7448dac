"""Host-side mirror of src/costdcnet_model_adapt.py (CostDCNetModel_Adapt, the per-backbone adapter the reference selects
with model_name='costdcnet', src/external_model_adapt.py:76-78) on libptta_hip.

On the accelerated path: the canonical TTA flow of bash/adapt/adapt_costdc_*.sh -- prepare_mode
'meta_selfsup_seq_1layer_ema', adapt_mode 'meta_bn', loss_type 'adapt_meta_selfsup_seq_ema_reverse': the reference's own
call sequence forward / compute_loss / loss.backward() / optimizer.step() (src/tta_main.py:610-633), the fused step() and
the eval forward (:729-736).  res = 16 depth planes, up_scale = 4 (src/costdcnet_model_adapt.py:47-52); frame sizes not
divisible by 16 run the reference's dual-corner padding inside the library.  The sparse 3-D encoder's arithmetic lives in
MinkowskiEngine in the reference (absent there, parity unpinned): see csrc/costdc_kernels.hip.
"""
import torch
import torch.nn as nn

from . import synth
from .engine import Engine
from .model import MsgChnModel_Adapt, _ForwardFn, _init_tensor, _Tree

_BUFFERS = ('running_mean', 'running_var', 'num_batches_tracked')


def costdcnet_adapted_names(keys, syncbn=False):
    """adapt_parameters('meta_bn') (src/costdcnet_model_adapt.py:357-378): parameters whose name contains 'meta', then
    weight / bias of every BatchNorm2d in module order = Encoder2D's (BatchNorm3d, BatchNorm1d and the sparse encoder's
    MinkowskiBatchNorm are not BatchNorm2d; ResBlock.norm3 also sits inside `downsample` but a module is visited once).
    syncbn: the list of the reference's DDP run.  convert_syncbn() runs BEFORE adapt_parameters (src/tta_main.py:326,339), every
    BatchNorm is a SyncBatchNorm by then and passes the isinstance test (:364-366): Encoder2D's, the BatchNorm1d inside each
    MinkowskiBatchNorm (`enc3d.<layer>.bn`), UNet3D's BatchNorm3d and the heads' BatchNorm1d -- 116 entries.
    convert_sync_batchnorm builds one new module per visited attribute, so ResBlock.norm3 and its alias downsample[1] are two
    modules around ONE Parameter: enc2d.layer{2,3}.0.norm3.{weight,bias} are listed twice (and stepped twice by Adam)."""
    names = [k for k in keys if 'meta' in k]
    for k in keys:
        if not k.endswith('.running_mean'):
            continue
        pre = k[:-len('.running_mean')]
        if not syncbn:
            if k.startswith('enc2d.') and '.downsample.1.' not in k:
                names += [pre + '.weight', pre + '.bias']
            continue
        if pre.startswith('enc2d.'):
            pre = pre.replace('.downsample.1', '.norm3')
        names += [pre + '.weight', pre + '.bias']
    return names


def _unique(names):
    seen, out = set(), []
    for k in names:
        if k not in seen:
            seen.add(k)
            out.append(k)
    return out


class CostDCNetModel_Adapt(MsgChnModel_Adapt):
    """Counterpart of src/costdcnet_model_adapt.py:31-556."""

    def __init__(self, device=torch.device('cuda'), max_depth=10.0, max_input_depth=None):
        self.max_predict_depth = max_depth
        self.max_depth = max_depth
        self.max_input_depth = max_input_depth
        self.device = device
        self.dtype = 'fp32'
        self.training = True
        self.prepare_mode = None
        self.model = _Tree()
        for k, s in synth.costdcnet_keys():
            if k.startswith(('proj', 'pred', 'conv1_rgb_meta')) or ('.downsample.1.' in k and k.startswith('enc2d.')):
                continue                                      # heads / meta layer come with _prepare_head; norm3 is one module
            self.model._leaf(k, _init_tensor(k, s), not k.endswith(_BUFFERS))
        self._engines = {}
        self._opt_state = {}
        self._adam_t = 0
        self.sync_bn = False
        self.hparams = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=1.0, w_cos=1.0)
        self.total_time = self.train_time = self.eval_time = 0.0
        self.to(device)

    heads_reverse_freezes_proj = False          # stage 2: ref = proj_t(...), proj trains in both directions (csrc/ghead.hip)

    def _state_keys(self):
        return [(k, s) for k, s in synth.costdcnet_keys(self.prepare_mode) if not (k.startswith('enc2d.') and '.downsample.1.' in k)]

    def _prepare_head(self, mode=''):
        """CostDCNet._prepare_head (CostDCNet_adapt.py:426-496)."""
        if 'meta' not in mode or 'selfsup' not in mode or 'ema' not in mode or 'seq' not in mode or '1layer' not in mode:
            raise NotImplementedError('hot path covers prepare_mode meta_selfsup_seq_1layer_ema, got %r' % mode)
        self.prepare_mode = mode
        self.meta = '1layer'
        for k, s in synth.costdcnet_keys(mode):
            if k.startswith(('proj', 'pred', 'conv1_rgb_meta')):
                self.model._leaf(k, _init_tensor(k, s).to(self.device), not k.endswith(_BUFFERS))
        self._set_adapted()

    def _set_adapted(self):
        # adapted_listed: the reference's parameter list (duplicates included); adapted: the tensors behind it, as the library binds them
        self.adapted_listed = costdcnet_adapted_names([k for k, _ in synth.costdcnet_keys(self.prepare_mode)], self.sync_bn)
        self.adapted = _unique(self.adapted_listed)
        # proj / pred feed the detached embedding (CD:249): their BatchNorm parameters never get a gradient, torch.optim.Adam skips them
        self._never_stepped = {k for k in self.adapted if k.startswith(('proj.1.', 'pred.1.'))} if self.sync_bn else set()
        self._clear_engines()

    def adapt_parameters(self, mode=None):
        if mode != 'meta_bn':
            raise NotImplementedError("adapt_mode %r: only 'meta_bn' (the CostDCNet scripts' mode) is on the accelerated path" % mode)
        params = dict(self.model.named_parameters())
        # the DDP list names four tensors twice: torch.optim.Adam then steps them twice per step(), like the reference's.  The reference
        # pins torch 1.10.1 (README.md:74: one entry at a time, two complete consecutive updates -- what step() does on device); with
        # torch >= 2.0 build the optimizer with foreach=False to get the same from optimizer.step()
        return nn.ParameterList([params[k] for k in self.adapted_listed])

    def convert_syncbn(self, apex=False):
        """SyncBatchNorm.convert_sync_batchnorm (src/costdcnet_model_adapt.py:535-546).  The reference converts BEFORE
        adapt_parameters('meta_bn') (src/tta_main.py:326,339): from here on every BatchNorm of the model is adapted and normalises
        with batch statistics in train AND eval mode (its running statistics are dropped, :364-372) -- the library's
        PTTA_SYNCBN_ADAPT engine, whose sparse encoder has a backward for its BatchNorm gamma / beta."""
        super().convert_syncbn(apex)
        if self.prepare_mode is not None:
            self._set_adapted()

    def state_dict(self):
        """The reference's state_dict lists ResBlock.norm3 a second time as downsample.1 (same tensors)."""
        sd = self.model.state_dict()
        out = {}
        for k, _ in synth.costdcnet_keys(self.prepare_mode or 'meta_selfsup_seq_1layer_ema'):
            src = k.replace('.downsample.1.', '.norm3.') if k.startswith('enc2d.') else k
            if src in sd:
                out[k] = sd[src]
        return out

    def load_state_dict(self, state):
        state = {k: v for k, v in state.items() if not (k.startswith('enc2d.') and '.downsample.1.' in k)}
        super().load_state_dict(state)

    def save_model(self, checkpoint_path, step, optimizer, meanvar=None):
        torch.save({'net': self.state_dict(), 'optimizer': optimizer.state_dict() if optimizer else {}, 'train_step': step}, checkpoint_path)

    def _engine(self, image):
        n, _, h, w = image.shape
        key = (n, h, w)
        eng = self._engines.get(key)
        if eng is None:
            if self.prepare_mode is None:
                raise RuntimeError('_prepare_head(mode) must be called before forward (tta_main.py:322)')
            eng = Engine(n, h, w, backbone='costdcnet', max_input_depth=self.max_input_depth, max_predict_depth=self.max_depth,
                         syncbn_adapted=self.sync_bn, **self.hparams)
            assert all(self.adapted_listed.count(k) == max(eng.adapted_repeat[k], 1) for k in eng.adapted), 'listing counts drifted'
            assert eng.adapted == self.adapted, 'adapted parameter list drifted from the library'
            eng.load_state_dict(self.model.state_dict())      # running statistics are bound by pointer and updated in place
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
            params = dict(self.model.named_parameters())
            return self._timed(lambda: _ForwardFn.apply(self, image, sparse_depth, *[params[k] for k in self.adapted]), loss_type)
        with torch.no_grad():
            return self._timed(lambda: self._engine(image).forward_eval(image, sparse_depth), loss_type)

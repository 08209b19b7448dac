"""Generate CostDCNet golden vectors by running the REAL reference (read-only at /root/reference) on CPU.

Runs only in the build container.  No reference file is modified or copied; it is imported in place with the harness-side
shims of make_golden.py (``.cuda()`` no-ops, ``torchvision`` stub, cwd = reference root) plus ONE stand-in:
``MinkowskiEngine`` — a third-party dependency that is absent from the reference tree and pinned nowhere — is provided by
``oracle/minkowski_lite.py`` (a restatement of its published semantics; see that file).  Everything that is plain torch in
the reference (Encoder2D, conv1_rgb_meta, fusion, the P3D UNet3D, pixel-shuffle softmax regression, MLP heads, adapt_loss,
dual-corner padding, torch.optim.Adam on adapt_parameters('meta_bn')) is therefore the reference's own arithmetic fed with
an identical dense 3-D feature volume on both sides; the sparse encoder's arithmetic is parity-unpinned.

Driven surface: ExternalModel_Adapt('costdcnet') -> _prepare_head, adapt_parameters('meta_bn'), forward, compute_loss
(loss_type='adapt'), backward, Adam.step, eval forward (src/tta_main.py:583-633, :729-736).
Usage:  python tests/golden/make_golden_costdcnet.py [case ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import minkowski_lite as ML  # noqa: E402
ML.install(sys.modules)
import make_golden as MG  # noqa: E402
from make_golden import synth  # noqa: E402
from make_golden_fullsize import pix_index, summarise  # noqa: E402

LOSS_TYPE = 'adapt_meta_selfsup_seq_ema_reverse'
PREPARE = 'meta_selfsup_seq_1layer_ema'
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
MAX_DEPTH = 8.0                  # --max_predict_depth of bash/adapt/adapt_costdc_scannet.sh
HP = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sd=1.0, w_sm=2.0, w_cos=0.1)


def costdc_frame(idx, h, w, n, density):
    """Raw 0..255 image -> ImageNet-normalised network input (adapt_costdc_scannet.sh:27) + the raw image for the loss;
    indoor sparse depth in [0.3, 7.5) m."""
    image01, sparse = synth.synthetic_frame(idx, h, w, n, density=density, dmin=0.3, dmax=7.5)
    raw = np.floor(image01 * 255.0).astype(np.float32)
    return raw, ((raw / np.float32(255.0) - MEAN) / STD).astype(np.float32), sparse


def install_cpu_syncbn():
    """Harness-side shim (like the `.cuda()` no-ops): torch.nn.SyncBatchNorm refuses CPU tensors.  In a single process without a
    process group it never synchronises anyway (`need_sync` is false): its forward reduces to F.batch_norm with the module's own
    training / running-statistics rules (torch/nn/modules/batchnorm.py SyncBatchNorm.forward), which is what this replacement runs."""
    import torch.nn as nn
    import torch.nn.functional as F

    def forward(self, input):
        bn_training = True if self.training else (self.running_mean is None and self.running_var is None)
        eaf = 0.0 if self.momentum is None else self.momentum
        use = not self.training or self.track_running_stats
        return F.batch_norm(input, self.running_mean if use else None, self.running_var if use else None, self.weight, self.bias, bn_training, eaf, self.eps)
    nn.SyncBatchNorm.forward = forward


def run_case(ema, name, h, w, n, steps, density=0.05, sampled=False, syncbn=False, light=False, alt=False, out=None, prefix=''):
    """syncbn: the reference's DDP run converts every BatchNorm to SyncBatchNorm BEFORE adapt_parameters('meta_bn') (src/tta_main.py:326,339):
    the isinstance test of src/costdcnet_model_adapt.py:364-366 then matches every BatchNorm of the model (116 entries; ResBlock.norm3 and its
    alias inside `downsample` become two modules sharing one Parameter, which therefore appears TWICE in the list and gets two Adam updates
    per step) and all of them lose their running statistics."""
    model = ema.ExternalModel_Adapt('costdcnet', 0.1, MAX_DEPTH, max_input_depth=None, device=torch.device('cpu'))
    model._prepare_head(PREPARE)
    net = model.model.model
    sd = synth.formula_state_dict_costdcnet(PREPARE)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(np.shape(v))) for k, v in sd.items()], 'key table drifted'
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    if syncbn:
        install_cpu_syncbn()
        model.convert_syncbn()
    params = model.adapt_parameters(mode='meta_bn')
    if prefix:       # the reference against ITSELF: one adapted weight one ulp off -- how far its own trajectory moves (the sensitivity floor)
        with torch.no_grad():
            params[0].view(-1)[0] = torch.nextafter(params[0].view(-1)[0], torch.tensor(float('inf')))
    pnames = {id(p): k for k, p in net.named_parameters()}
    names = [pnames[id(p)] for p in params]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                 # syncbn: "optimizer contains a parameter group with duplicate parameters"
        opt = torch.optim.Adam(params, lr=HP['lr'], betas=HP['betas'], eps=HP['eps'], weight_decay=HP['weight_decay'])
    own = out is None
    if own:
        out = {'meta': np.array([h, w, n, steps], dtype=np.int64), 'density': np.array(density),
               'hp': np.array([HP['lr'], HP['betas'][0], HP['betas'][1], HP['eps'], HP['weight_decay'], HP['w_sd'], HP['w_sm'], HP['w_cos'],
                               MAX_DEPTH], dtype=np.float64),
               'adapted_names': np.array(names)}
        if sampled:
            out['pix_idx'] = pix_index(n * h * w)
    for s in range(steps):
        raw, image1, sparse_np = costdc_frame(s, h, w, n, density)
        image, sparse, loss_image = torch.from_numpy(image1), torch.from_numpy(sparse_np), torch.from_numpy(raw)
        validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
        K = torch.eye(3)[None].repeat(n, 1, 1)           # the padding path concatenates the intrinsics (AD:182); values unused
        model.train()
        depth, emb, ref = model.forward(image=image, sparse_depth=sparse, intrinsics=K, loss_type=LOSS_TYPE)
        loss, info = model.compute_loss(
            input_rgb=loss_image, output_depth=depth, sparse_depth=sparse.detach(), validity_map=validity.detach(), embedding=emb,
            reference=ref, w_loss_sparse_depth=HP['w_sd'], w_loss_smoothness=HP['w_sm'], w_loss_cos=HP['w_cos'], loss_type='adapt')
        opt.zero_grad()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        assert syncbn or all(k in grads for k in names), 'an adapted tensor got no gradient'
        opt.step()
        model.eval()
        with torch.no_grad():
            depth_eval = model.forward(image=image, sparse_depth=sparse, intrinsics=K, loss_type=LOSS_TYPE)
        p = prefix + 's%d/' % s
        if prefix:          # the perturbed trajectory: the scored depth's sampled pixels only
            summarise(out, p + 'depth_eval', depth_eval, blk=False)
            continue
        if sampled:
            summarise(out, p + 'depth_train', depth, blk=not light)
            summarise(out, p + 'depth_eval', depth_eval, blk=not light)
        else:
            out[p + 'depth_train'] = depth.detach().numpy()
            out[p + 'depth_eval'] = depth_eval.numpy()
        out[p + 'loss_info'] = np.array([float(torch.as_tensor(info[k]).detach()) for k in ('loss', 'loss_smooth', 'loss_sparse_depth', 'loss_cos')])
        if light and s not in (0, steps - 1):
            continue
        e, r = emb.detach().numpy(), ref.detach().numpy()
        idx, out[p + 'emb_rows'] = MG.sample_rows(e)
        _, out[p + 'ref_rows'] = MG.sample_rows(r)
        out[p + 'row_idx'] = idx
        out[p + 'emb_shape'] = np.array(e.shape)
        out[p + 'loss_info'] = np.array([float(torch.as_tensor(info[k]).detach()) for k in ('loss', 'loss_smooth', 'loss_sparse_depth', 'loss_cos')])
        named = dict(net.named_parameters())
        for i, k in enumerate(names):
            out[p + 'grad/' + k] = grads[k].numpy() if k in grads else np.zeros(tuple(named[k].shape), np.float32)     # syncbn: proj / pred BatchNorm1d get no gradient
            out[p + 'param/' + k] = named[k].detach().numpy().copy()
            if named[k] in opt.state and 'exp_avg' in opt.state[named[k]]:
                out[p + 'exp_avg/' + k] = opt.state[named[k]]['exp_avg'].numpy().copy()
        for k, v in net.state_dict().items():          # buffers that the eval forward reads / save_model writes
            if k.endswith(('running_mean', 'running_var')) and not k.startswith(('enc2d.', 'proj_t.')):
                out[p + 'buf/' + k] = v.numpy().copy()
        print(name, 'step', s, 'loss_info', out[p + 'loss_info'], 'depth mean', float(depth.mean()), float(depth_eval.mean()),
              'points', int((sparse > 0).sum()), flush=True)
    if not own:
        return
    if alt:
        run_case(ema, name, h, w, n, steps, density, sampled, syncbn, light, out=out, prefix='alt/')
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), flush=True)


CASES = {
    'costdcnet_64x96': lambda e: run_case(e, 'costdcnet_64x96', 64, 96, 1, 2),
    'costdcnet_64x64_n2': lambda e: run_case(e, 'costdcnet_64x64_n2', 64, 64, 2, 1),
    'costdcnet_72x100_pad': lambda e: run_case(e, 'costdcnet_72x100_pad', 72, 100, 1, 1),       # dual-corner padding, odd pooled sizes
    'costdcnet_320x400': lambda e: run_case(e, 'costdcnet_320x400', 320, 400, 1, 1, density=0.012, sampled=True),   # the ScanNet script's frame
    'costdcnet_64x64_n2_syncbn': lambda e: run_case(e, 'costdcnet_64x64_n2_syncbn', 64, 64, 2, 2, syncbn=True),      # the DDP run's adapted set (116 entries)
    # ONE parameter set over 16 different frames (src/tta_main.py:504-804), with the reference's own one-ulp-perturbed trajectory beside it
    'costdcnet_96x128_seq16': lambda e: run_case(e, 'costdcnet_96x128_seq16', 96, 128, 1, 16, density=0.03, sampled=True, light=True, alt=True),
    'costdcnet_480x640': lambda e: run_case(e, 'costdcnet_480x640', 480, 640, 1, 1, density=1500.0 / (480 * 640), sampled=True),  # config 5
}

if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ema, _ = MG.import_reference()
    for c in (sys.argv[1:] or list(CASES)):
        CASES[c](ema)

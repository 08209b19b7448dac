"""Full-size golden vectors from the REAL reference (read-only at /root/reference), CPU, build container only.

SURVEY.md §8c asks for one 256x320 and one 352x1216 reference case stored as checksums + sampled pixels; this script
emits them for both MSG_CHN meta layers, plus a 10-step sequence (the depth-after-N-steps consequence of the loose
gradient bounds) and a fixture of the reference's evaluation metrics (src/eval_utils.py:117-174 driven the way
src/tta_main.py:779-798 drives them).  Same import shims as make_golden.py (imported from it, nothing copied).

Stored per step (full maps would be 1.7 MB each):
  * `pix_idx` + `depth_train_pix` / `depth_eval_pix`: 4096 pixels on a fixed low-discrepancy index set,
  * `*_sum`, `*_abs_mean` (float64 checksums over the whole map) and `*_blk`: 8x8 block means of the whole map,
  * loss_info, the FULL gradients / post-step values / Adam moments of every adapted tensor, BatchNorm buffers,
  * 24 sampled rows of the embedding and of the reference projection.
Usage:  python tests/golden/make_golden_fullsize.py [case ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from make_golden import HP, LOSS_TYPE, synth  # noqa: E402

NPIX = 4096


def pix_index(numel, k=NPIX):
    """Fixed, well-spread pixel subset: floor(frac(i * phi) * numel)."""
    i = np.arange(1, k + 1, dtype=np.float64)
    return np.unique(np.floor(np.modf(i * 0.6180339887498949)[0] * numel).astype(np.int64))


def summarise(out, key, t, blk=True, npix=NPIX):
    a = t.detach().numpy().astype(np.float32)
    flat = a.reshape(-1)
    idx = pix_index(flat.size, npix)
    out[key + '_pix'] = flat[idx]
    out[key + '_sum'] = np.array(flat.sum(dtype=np.float64))
    out[key + '_abs_mean'] = np.array(np.abs(flat).mean(dtype=np.float64))
    if not blk:
        return
    n, c, h, w = a.shape
    k = 8 if (h % 8 == 0 and w % 8 == 0) else 4          # 228 x 304 (NYUv2): 4 x 4 blocks
    out[key + '_blk'] = a.reshape(n, c, h // k, k, w // k, k).mean(axis=(3, 5), dtype=np.float64).astype(np.float32)


def run_case(ema, name, prepare_mode, h, w, n, steps, full_every=1, frame0=0, moments=True, light=False, alt=False, out=None, prefix=''):
    """light: a long-horizon sequence (the reference adapts ONE parameter set over a whole dataset, src/tta_main.py:504-636) -- per step only
    the sampled pixels + checksums of the scored depth and loss_info; the adapted tensors at every `full_every`-th step."""
    hp = dict(HP)
    model = ema.ExternalModel_Adapt('msg_chn', 0.0, 80.0, max_input_depth=hp['max_input_depth'], device=torch.device('cpu'))
    model._prepare_head(prepare_mode)
    net = model.model.model
    sd = synth.formula_state_dict(prepare_mode, 1.0)
    assert list(sd.keys()) == list(net.state_dict().keys()), 'key table drifted from reference'
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    params = model.adapt_parameters(mode='meta')
    if prefix:
        # the REFERENCE's own sensitivity: the same program from a start that differs by ONE unit in the last place of ONE adapted weight
        # (stored under `alt/`): how far two runs of the reference itself are apart after s steps is the floor any other fp32 program sits on
        with torch.no_grad():
            w0 = params[0].view(-1)
            w0[0] = torch.nextafter(w0[0], w0[0] * 2)
    opt = torch.optim.Adam(params, lr=hp['lr'], betas=hp['betas'], eps=hp['eps'], weight_decay=hp['weight_decay'])
    out_ = {'meta': np.array([h, w, n, steps, frame0], dtype=np.int64),
           'hp': np.array([hp['lr'], hp['betas'][0], hp['betas'][1], hp['eps'], hp['weight_decay'], hp['w_sd'], hp['w_sm'],
                           hp['w_cos'], hp['max_input_depth'], 1.0], dtype=np.float64),
           'pix_idx': pix_index(n * h * w, 1024 if light else NPIX)}
    out = out_ if out is None else out
    names = [k for k, _ in net.named_parameters() if 'meta' in k]
    for s in range(steps):
        image_np, sparse_np = synth.synthetic_frame(frame0 + s, h, w, n)
        image, sparse = torch.from_numpy(image_np), torch.from_numpy(sparse_np)
        validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
        model.train()
        depth, emb, ref = model.forward(image=image, sparse_depth=sparse, loss_type=LOSS_TYPE)
        loss, info = model.compute_loss(
            input_rgb=image.detach(), output_depth=depth, sparse_depth=sparse.detach(), validity_map=validity.detach(),
            embedding=emb, reference=ref, w_loss_sparse_depth=hp['w_sd'], w_loss_smoothness=hp['w_sm'],
            w_loss_cos=hp['w_cos'], loss_type='adapt')
        opt.zero_grad()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if 'meta' in k}
        opt.step()
        model.eval()
        with torch.no_grad():
            depth_eval = model.forward(image=image, sparse_depth=sparse, loss_type=LOSS_TYPE)
        p = prefix + 's%d/' % s
        if not light:
            summarise(out, p + 'depth_train', depth)
        summarise(out, p + 'depth_eval', depth_eval, blk=not light, npix=1024 if light else NPIX)
        out[p + 'loss_info'] = np.array([float(info[k].detach()) for k in ('loss', 'loss_smooth', 'loss_sparse_depth', 'loss_cos')])
        full = ((s % full_every == 0) or s == steps - 1) and not prefix
        if full:
            if not light:
                e, r = emb.detach().numpy(), ref.detach().numpy()
                idx, out[p + 'emb_rows'] = MG.sample_rows(e)
                _, out[p + 'ref_rows'] = MG.sample_rows(r)
                out[p + 'row_idx'] = idx
                out[p + 'emb_shape'] = np.array(e.shape)
            state = opt.state_dict()['state']
            for i, k in enumerate(names):
                out[p + 'grad/' + k] = grads[k].numpy()
                out[p + 'param/' + k] = dict(net.named_parameters())[k].detach().numpy().copy()
                if moments:
                    out[p + 'exp_avg/' + k] = state[i]['exp_avg'].numpy().copy()
                    out[p + 'exp_avg_sq/' + k] = state[i]['exp_avg_sq'].numpy().copy()
            for k, v in net.state_dict().items():
                if 'running_' in k and ('proj.' in k or 'pred.' in k or 'meta' in k):
                    out[p + 'buf/' + k] = v.numpy().copy()
        print(name, 'step', s, 'loss_info', out[p + 'loss_info'], 'depth mean', float(depth.mean()), flush=True)
    if prefix:
        return
    if alt:
        run_case(ema, name, prepare_mode, h, w, n, steps, full_every, frame0, moments, light, out=out, prefix='alt/')
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes', flush=True)


def run_eval_metrics():
    """The four numbers src/tta_main.py:779-798 accumulates per batch (before its `* batch_size / 1000` bookkeeping),
    computed by the reference's own src/eval_utils.py on hash-formula inputs (nothing large is stored)."""
    import eval_utils
    n, h, w = 2, 64, 96
    u = lambda tag: synth.hash_uniform(tag, n * h * w).reshape(n, 1, h, w).astype(np.float32)
    gt = (u('em/gt') * 90.0).astype(np.float32)
    gt[u('em/mask') < 0.7] = 0.0
    outd = np.maximum(gt + (u('em/noise') - 0.5) * 3.0, 0.1).astype(np.float32) + (gt == 0) * 5.0
    outd = outd.astype(np.float32)
    res = {}
    for lo, hi in ((0.0, 100.0), (0.5, 80.0), (1e-3, 655.0)):
        o, g = torch.squeeze(torch.from_numpy(outd)), torch.squeeze(torch.from_numpy(gt))
        mask = torch.where(g > 0, torch.ones_like(g), torch.zeros_like(g))
        mask[g < lo] = 0.0
        mask[g > hi] = 0.0
        o, g = o[mask.nonzero(as_tuple=True)], g[mask.nonzero(as_tuple=True)]
        res['%g_%g' % (lo, hi)] = np.array([
            float(eval_utils.torch_mean_abs_err(1000.0 * o, 1000.0 * g)), float(eval_utils.torch_root_mean_sq_err(1000.0 * o, 1000.0 * g)),
            float(eval_utils.torch_inv_mean_abs_err(0.001 * o, 0.001 * g)), float(eval_utils.torch_inv_root_mean_sq_err(0.001 * o, 0.001 * g))])
        print('eval metrics', lo, hi, res['%g_%g' % (lo, hi)])
    np.savez_compressed(os.path.join(HERE, 'eval_metrics.npz'), meta=np.array([n, h, w]), **res)


CASES = {
    'msgchn_1layer_256x320': lambda e: run_case(e, 'msgchn_1layer_256x320', 'meta_selfsup_seq_1layer_ema', 256, 320, 1, 2),
    'msgchn_2layers_256x320': lambda e: run_case(e, 'msgchn_2layers_256x320', 'meta_selfsup_seq_2layers_ema', 256, 320, 1, 1, moments=False),
    'msgchn_1layer_352x1216': lambda e: run_case(e, 'msgchn_1layer_352x1216', 'meta_selfsup_seq_1layer_ema', 352, 1216, 1, 2),
    'msgchn_2layers_352x1216': lambda e: run_case(e, 'msgchn_2layers_352x1216', 'meta_selfsup_seq_2layers_ema', 352, 1216, 1, 1, moments=False),
    'msgchn_1layer_64x96_seq10': lambda e: run_case(e, 'msgchn_1layer_64x96_seq10', 'meta_selfsup_seq_1layer_ema', 64, 96, 1, 10,
                                                    full_every=9, frame0=100),
    # the reference's operating point is n_batch // ngpus frames per rank (bash/adapt/adapt_msgchn_vkitti.sh:21, src/tta_main.py:224)
    'msgchn_1layer_352x1216_n2': lambda e: run_case(e, 'msgchn_1layer_352x1216_n2', 'meta_selfsup_seq_1layer_ema', 352, 1216, 2, 2, frame0=20),
    'msgchn_1layer_352x1216_n4': lambda e: run_case(e, 'msgchn_1layer_352x1216_n4', 'meta_selfsup_seq_1layer_ema', 352, 1216, 4, 2, frame0=30),
    # N = 8: beyond the 1,024 cosine-partial slots that N >= 5 overran before round 6 (loss.hip loss_cb) -- the reference's own operating range
    'msgchn_1layer_352x1216_n8': lambda e: run_case(e, 'msgchn_1layer_352x1216_n8', 'meta_selfsup_seq_1layer_ema', 352, 1216, 8, 2, frame0=40),
    # long horizons: one parameter set adapted over a stream of frames (src/tta_main.py:504-636)
    'msgchn_1layer_64x96_seq200': lambda e: run_case(e, 'msgchn_1layer_64x96_seq200', 'meta_selfsup_seq_1layer_ema', 64, 96, 1, 200,
                                                     full_every=20, frame0=1000, light=True, moments=False, alt=True),
    'msgchn_1layer_256x320_seq150': lambda e: run_case(e, 'msgchn_1layer_256x320_seq150', 'meta_selfsup_seq_1layer_ema', 256, 320, 1, 150,
                                                       full_every=25, frame0=2000, light=True, moments=False, alt=True),
    # the headline size (BASELINE config 2) over 120 frames: ~25 min of CPU for the two trajectories
    'msgchn_1layer_352x1216_seq120': lambda e: run_case(e, 'msgchn_1layer_352x1216_seq120', 'meta_selfsup_seq_1layer_ema', 352, 1216, 1, 120,
                                                        full_every=30, frame0=4000, light=True, moments=False, alt=True),
    # the other meta mode (conv1_rgb_meta + its BatchNorm2d, network_exp_msg_chn_adapt.py 2layers) over 80 frames
    'msgchn_2layers_256x320_seq80': lambda e: run_case(e, 'msgchn_2layers_256x320_seq80', 'meta_selfsup_seq_2layers_ema', 256, 320, 1, 80,
                                                       full_every=20, frame0=6000, light=True, moments=False, alt=True),
    # N frames per call over a horizon: 3 frames per step, 60 steps (the reference's operating point is n_batch // ngpus frames per rank)
    'msgchn_1layer_96x128_n3_seq60': lambda e: run_case(e, 'msgchn_1layer_96x128_n3_seq60', 'meta_selfsup_seq_1layer_ema', 96, 128, 3, 60,
                                                        full_every=30, frame0=8000, light=True, moments=False, alt=True),
    'eval_metrics': lambda e: run_eval_metrics(),
}

if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ema, _ = MG.import_reference()
    for c in (sys.argv[1:] or list(CASES)):
        CASES[c](ema)

#!/usr/bin/env python3
"""bench.py — ProxyTTA step throughput on MI355X (metric of BASELINE.json).

  python bench.py --gpus N --steps K --warmup W [--dtype mixed|fp32]

One "step" = one pass of the hot path (forward [grad + proxy pass + heads] + loss + backward +
Adam, src/tta_main.py:610-633) over one synthetic 352x1216 KITTI-shaped frame, MSG_CHN backbone,
prepare_mode meta_selfsup_seq_1layer_ema, batch 1 per GPU, inputs resident in HBM.
N > 1: one process per GPU (torch.distributed.run), independent frame streams sharded over the
ranks, no data-path collective ("weak" scaling); barrier + max-over-ranks timing.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     dominant kernel class = the stride-1 3x3 32->32 convolutions with ReLU on load
               (conv32_s1_x3_kernel<true, ...> on large maps + conv32_s1_small_kernel<true, ...>):
               algorithmic bytes/MACs of its launches (SURVEY.md §8d counting rule: input + output +
               weight elements per layer) / their duration measured with hipEvents on the launch
               stream (ptta_profile); roofline_by_class: the same for every launch class of the step.
  timing       SURVEY.md §8d / BASELINE.md §4: whatever --steps says, >= 10 blocks of K steps so that
               >= 200 steps are timed (each block bracketed by barrier + synchronize, MAX over ranks);
               ms_per_step = MEDIAN over the blocks, min / max beside it.
  cpu_baseline the oracle (PyTorch CPU restatement, oracle/proxytta_oracle.py) timed on this
               box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

# Kernel arguments in device memory: the bench process asks for it BEFORE the HIP runtime exists (forced to 0 the replayed step is 7 % slower:
# every launch then fetches its arguments across the host link).  A user's own setting is kept; the JSON line reports
# what the run had (`config.hip_force_dev_kernarg`).  The library and its Python binding never touch the environment.
KERNARG_PRESET = os.environ.get('HIP_FORCE_DEV_KERNARG')
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

import numpy as np  # noqa: E402
import torch  # noqa: E402

MODE = 'meta_selfsup_seq_1layer_ema'
H, W = 352, 1216
# SURVEY.md §8d / BASELINE.md §3 (measured with hooks on the reference): 905.0 M layer-I/O elements
# and 107.2 GMAC per step for MSG_CHN 1layer at 352x1216, batch 1
ALG_ELEMENTS_PER_STEP = 905.0e6
# Of those, the proxy pass's RGB encoder (zero image through frozen weights: 10 conv layers, in + out + weight elements
# = 173.375 elements per pixel + 83,808 weights) is a constant of the handle and is NOT executed per step any more
# (computed once at the first step after ptta_load_weights): the step is priced on the elements it still moves.
ALG_ELEMENTS_HOISTED = 173.375 * H * W + 9 * 9216 + 864
ALG_FLOP_PER_STEP = 214.5e9
HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK = 157.3e12   # fp32 matrix peak
MFMA_BF16_PEAK = 2.5e15    # dense bf16
HP = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=2.0,
          w_cos=0.1, max_input_depth=80.0)
DTYPE_TEXT = {
    'fp32': 'f32 storage, bf16x3 MFMA arithmetic (fp32 accumulate)',
    'mixed': ('mixed: real frames\' forward (the scored depth) f32 storage + bf16x3 MFMA arithmetic; no_grad proxy pass and all data gradients bf16 '
              'storage + single bf16 MFMA; fp32 accumulate everywhere; loss / Adam / adapted parameters fp32 (include/ptta.h PTTA_DTYPE_MIXED)'),
}


def cpu_model_string():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(warm=3, steps=10):
    """the PyTorch-CPU restatement (pinned to the reference by tests/golden) on this box's host cores,
    3 warm-up + 10 timed steps, forward / loss / backward / Adam split."""
    from oracle import proxytta_oracle as O
    from proxytta import synth
    # PyTorch's CPU convolutions at batch 1 stop scaling (and then regress badly) past a few dozen
    # threads; 16 is where the oracle is fastest on the GPU box's host (99 s/step with all 256; the container's cgroup quota is 16 CPUs)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    o = O.MsgChnOracle(synth.formula_state_dict(MODE), MODE, max_input_depth=80.0, lr=1e-3, w_sd=1.0, w_sm=2.0, w_cos=0.1)
    frames = [[torch.from_numpy(x) for x in synth.synthetic_frame(i, H, W, 1)] for i in range(2)]
    split = {'forward': 0.0, 'loss': 0.0, 'backward': 0.0, 'adam': 0.0}
    per_step = []
    for i in range(warm + steps):
        image, sparse = frames[i % 2]
        validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)
        t0 = time.perf_counter()
        depth, emb, ref = o.forward_train(image, sparse)
        t1 = time.perf_counter()
        loss, _ = O.adapt_loss(image, depth, sparse, validity, emb, ref, *o.w, max_input_depth=o.max_input_depth)
        t2 = time.perf_counter()
        params = [o.P[k] for k in o.names]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        t3 = time.perf_counter()
        o.opt.step(params, [torch.zeros_like(p) if g is None else g for p, g in zip(params, grads)])
        t4 = time.perf_counter()
        if i >= warm:
            for k, dt in zip(split, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                split[k] += dt / steps
            per_step.append(t4 - t0)
    dt = float(np.median(per_step))
    return {'value': 1.0 / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'cpu_model': cpu_model_string(), 'host_cpus_visible': os.cpu_count(),
            'threads_reason': 'torch CPU convolutions at batch 1 are fastest at 16 threads on this host (all cores: ~99 s/step; cgroup quota 16 CPUs)',
            's_per_step_median': dt, 's_per_step_min': float(min(per_step)), 's_per_step_max': float(max(per_step)),
            'split_s': {k: round(v, 4) for k, v in split.items()},
            'sample': '%d timed steps of the same 352x1216 workload after %d warm-up steps (PyTorch-CPU oracle, fp32, %.2f s/step median; the '
                      'oracle runs autograd only into the adapted tensors)' % (steps, warm, dt)}


def msgchn_2layers_workload(steps=30):
    """MSG_CHN with the `2layers` meta layer (Res_Conv(32,128) + 2 BatchNorm2d, 7 adapted tensors): the recipe of
    bash/adapt/adapt_msgchn_vkitti.sh:26.  Same 352x1216 synthetic frames as the headline (1layer) configuration; both precision modes."""
    from proxytta import synth
    from proxytta.engine import Engine
    mode = 'meta_selfsup_seq_2layers_ema'
    pipe = os.environ.get('PTTA_PIPELINE', '1') != '0'           # frames as a stream, like the headline loop
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(i, H, W, 1)] for i in range(4)]

    def run(dtype):
        eng = Engine(1, H, W, meta='2layers', dtype=dtype, **HP)
        sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(mode).items()}
        eng.load_state_dict(sd)
        for name in eng.adapted:
            eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
        for i in range(5):
            eng.step(*frames[i % 4], next_frame=frames[(i + 1) % 4] if pipe else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            info, _ = eng.step(*frames[(5 + i) % 4], next_frame=frames[(6 + i) % 4] if pipe else None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        fin = bool(torch.isfinite(info).all().item())
        eng.close()
        return dt, fin

    dt, fin = run('fp32')
    dtm, finm = run('mixed')
    return {'workload': 'MSG_CHN 2layers meta (Res_Conv(32,128), 7 adapted tensors), 352x1216, 1 TTA step/frame, batch 1, frame pipelining ' + ('on' if pipe else 'off'),
            'ms_per_step': 1e3 * dt, 'frames_per_s': 1.0 / dt, 'finite': fin,
            'mixed_mode': {'ms_per_step': 1e3 * dtm, 'frames_per_s': 1.0 / dtm, 'finite': finm}}


def msgchn_adapt_loop_workload(frames_n=60):
    """The per-frame loop of the reference (src/tta_main.py:579-636 + :729-736): ONE TTA step, then the scored eval forward of the same frame,
    frames fed as a stream (frame pipelining; the eval forward reuses the adapted frame's prefix).  Reported beside the metric."""
    from proxytta import synth
    from proxytta.engine import ADAPTED, Engine
    eng = Engine(1, H, W, **HP)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict('meta_selfsup_seq_1layer_ema').items()}
    eng.load_state_dict(sd)
    for name in ADAPTED:
        eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(i, H, W, 1)] for i in range(4)]
    out = {'workload': 'MSG_CHN 1layer, 352x1216, per frame: 1 TTA step + the scored eval forward, batch 1'}
    for mode in ('pipelined', 'call_by_call'):
        for phase, count in (('warm', 5), ('timed', frames_n)):
            if phase == 'timed':
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            for i in range(count):
                if mode == 'pipelined':
                    eng.step(*frames[i % 4], next_frame=frames[(i + 1) % 4])
                    d = eng.forward_eval_last()
                else:
                    eng.step(*frames[i % 4])
                    d = eng.forward_eval(*frames[i % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames_n
        out[mode] = {'ms_per_frame': 1e3 * dt, 'frames_per_s': 1.0 / dt, 'finite': bool(torch.isfinite(d).all().item())}
    eng.close()
    return out


def nlspn_macs(h, w, n=1):
    """Multiply-accumulates of one NLSPN TTA step from the architecture (nlspnmodel_adapt.py:385-452, ResNet34
    BasicBlocks :70-116): (training forward incl. the proxy pass and the heads, eval forward, minimal backward).
    The training forward is 1,113 GMAC at 352x1216 -- the number SURVEY.md §8d measured with hooks on the reference."""
    px = lambda s: n * (h // s) * (w // s)
    enc = [(px(1), 3, 48, 9), (px(1), 48, 48, 9), (px(1), 1, 16, 9)]
    inpl, s = 64, 1
    for planes, nb, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for b in range(nb):
            st = stride if b == 0 else 1
            s *= st
            enc += [(px(s), inpl, planes, 9), (px(s), planes, planes, 9)]
            if st != 1 or inpl != planes:
                enc.append((px(s), inpl, planes, 1))
            inpl = planes
    enc.append((px(16), 512, 512, 9))
    dec = [(px(8), 512, 256, 2.25), (px(4), 768, 128, 2.25), (px(2), 384, 64, 2.25), (px(1), 192, 64, 2.25),
           (px(1), 128, 64, 9), (px(1), 128, 1, 9), (px(1), 128, 64, 9), (px(1), 128, 8, 9), (px(1), 128, 32, 9),
           (px(1), 96, 1, 9), (px(1), 8, 24, 9)]
    m = lambda layers: sum(p * ci * co * t for p, ci, co, t in layers)
    mlp = px(16) * 512 * 1024 + px(16) * 1024 * 1024
    fwd_train = 2 * m(enc) + m(dec) + 2 * mlp + 2 * px(16) * 1024 * 1024
    fwd_eval = m(enc) + m(dec)
    bwd = (m(enc) - m(enc[:3])) + m(dec) + mlp + px(1) * 48 * 48 * 9
    return fwd_train, fwd_eval, bwd


def nlspn_cpu_baseline(inner_iter=3):
    """The NLSPN oracle (PyTorch CPU restatement, oracle/nlspn_oracle.py) on this box's host cores: one eval forward as
    warm-up, then ONE timed TTA step of the same 352x1216 workload (about 10 s on 16 threads)."""
    from oracle import nlspn_oracle as NO
    from proxytta import synth
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
    image01, sparse = synth.synthetic_frame(0, H, W, 1)
    raw = np.floor(image01 * 255.0).astype(np.float32)
    im, sp, rawt = torch.from_numpy(((raw / np.float32(255.0) - mean) / std).astype(np.float32)), torch.from_numpy(sparse), torch.from_numpy(raw)
    o = NO.NlspnOracle(synth.formula_state_dict_nlspn(), max_input_depth=80.0, lr=3e-4, w_sd=1.0, w_sm=0.0, w_cos=0.0, legacy=True)
    o.forward_eval(im, sp)
    t0 = time.time()
    o.step(im, sp, loss_image=rawt)
    dt = time.time() - t0
    return {'value': 1.0 / (inner_iter * dt), 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '1 TTA step of the same 352x1216 workload after one eval forward (PyTorch-CPU oracle, fp32, %.1f s/step; %d steps/frame)' % (dt, inner_iter)}


CLASSES = ['s1_relu_large', 's1_relu_small', 's1_plain_large', 's1_plain_small', 'strided_large', 'strided_small', 'heads', 'in_out_convs', 'rest']


def msgchn_batch_workload(ns=(1, 2, 4, 8, 16), dtype='mixed', steps=20, blocks=5, warmup=6, by_class=True):
    """Beside the metric, never `value` (the headline stays batch 1 per GPU): the SAME step with N frames per call -- the reference's own
    operating point is `n_batch // ngpus` frames per rank (bash/adapt/adapt_msgchn_vkitti.sh:21 --n_batch 16, src/tta_main.py:224: 2 frames per
    GPU on 8 GPUs, 16 on one).  Same protocol as the headline: frames resident in HBM, ptta_step_pipelined (every call names the next batch),
    median block; then the same steps kernel by kernel on one stream for the per-class table.  Algorithmic bytes at the stored width."""
    from proxytta import synth
    from proxytta.engine import ADAPTED, Engine
    out = {}
    for n in ns:
        eng = Engine(n, H, W, dtype=dtype, **HP)
        sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(MODE).items()}
        eng.load_state_dict(sd)
        for name in ADAPTED:
            eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
        nfr = 3
        frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(3000 + 16 * i, H, W, n)] for i in range(nfr)]
        for i in range(warmup):
            info, _ = eng.step(*frames[i % nfr], next_frame=frames[(i + 1) % nfr])
        it, bl = warmup, []
        for _ in range(blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                info, _ = eng.step(*frames[it % nfr], next_frame=frames[(it + 1) % nfr])
                it += 1
            torch.cuda.synchronize()
            bl.append((time.perf_counter() - t0) / steps)
        ms = 1e3 * float(np.median(bl))
        rec = {'ms_per_step': ms, 'frames_per_s': 1e3 * n / ms, 'ms_per_frame': ms / n, 'finite': bool(torch.isfinite(info).all().item()),
               'pipelined_active': eng.get_option('pipelined_active') if hasattr(eng, 'get_option') else None}
        if by_class:
            eng.profile(True)
            torch.cuda.synchronize()
            for i in range(steps):
                eng.step(*frames[i % nfr])
            torch.cuda.synchronize()
            prof = [eng.profile_read(k) for k in range(len(CLASSES))]
            eng.profile(False)
            stored = sum(p[1] for p in prof) / steps
            rec['alg_bytes_per_step'] = stored
            rec['step_hbm_frac'] = stored / (ms * 1e-3) / HBM_PEAK
            rec['roofline_by_class'] = {name: {'launches_per_step': cn / steps, 'us_per_step': 1e3 * cms / steps, 'us_per_frame': 1e3 * cms / steps / n,
                                               'hbm_frac': (cby / (cms * 1e-3) / HBM_PEAK) if (cms > 0 and cby > 0) else None}
                                        for name, (cms, cby, cmc, cn) in zip(CLASSES, prof)}
            rec['serial_sum_us_per_frame'] = sum(1e3 * p[0] for p in prof) / steps / n
        out[str(n)] = rec
        eng.close()
        del eng, sd, frames
        torch.cuda.empty_cache()
    base = out.get('1')
    if base:
        for k, v in out.items():
            v['speedup_vs_batch_1'] = v['frames_per_s'] / base['frames_per_s']
    out['_note'] = ('N frames per call on ONE GPU, %s, ptta_step_pipelined, %d blocks x %d steps, median block; step_hbm_frac = stored algorithmic bytes of '
                    'the step / time / 8 TB/s; roofline_by_class from the kernel-by-kernel one-stream leg (hipEvents)' % (dtype, blocks, steps))
    return out


def two_streams_workload(steps=100, nstreams=2, dtype='mixed'):
    """Beside the metric, never `value`: the config-4 sharding (independent frame streams, each with its own adapted parameters and
    Adam state) applied INSIDE one GPU -- two handles, two HIP streams, the headline's dtype and call (ptta_step_pipelined: every stream
    names its next frame).  What the second stream gains is the share of the single-stream step that is latency (small launches, serial
    chains), not bandwidth."""
    from proxytta import synth
    from proxytta.engine import Engine
    engs, keep = [], []
    for _ in range(nstreams):
        eng = Engine(1, H, W, dtype=dtype, **HP)
        sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(MODE).items()}
        eng.load_state_dict(sd)
        for name in eng.adapted:
            keep.append((sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name])))
            eng.bind_adapted(name, *keep[-1])
        engs.append(eng)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    frames = [[[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(900 + 10 * j + i, H, W, 1)] for i in range(4)] for j in range(nstreams)]

    def run(k):
        for it in range(k):
            for j, (e, st) in enumerate(zip(engs, streams)):
                with torch.cuda.stream(st):
                    info, _ = e.step(*frames[j][it % 4], next_frame=frames[j][(it + 1) % 4])
        return info
    run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {'workload': '%d independent MSG_CHN 1layer frame streams on one GPU (own parameters / Adam state each), 352x1216, batch 1, %s, ptta_step_pipelined' % (nstreams, dtype),
           'frames_per_s': steps * nstreams / dt, 'ms_per_frame': 1e3 * dt / (steps * nstreams), 'finite': bool(torch.isfinite(info).all().item())}
    for e in engs:
        e.close()
    return out


def nlspn_workload(frames=10, inner_iter=3, dtype='fp32', with_mixed=True):
    """BASELINE config 3 (not the headline metric): NLSPN backbone, 352x1216, 3 TTA steps per frame + the scored eval
    forward, adapt_mode meta_bn (88 adapted tensors), batch 1, inputs resident in HBM.  Reported beside the metric."""
    from proxytta import synth
    from proxytta.engine import Engine
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
    eng = Engine(1, H, W, backbone='nlspn', dtype=dtype, legacy_offset=True, lr=3e-4, w_sparse_depth=1.0, w_smoothness=0.0, w_cos=0.0, max_input_depth=80.0)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
    eng.load_state_dict({k: v for k, v in sd.items() if v.dtype == torch.float32})
    keep = []
    for k in eng.adapted:
        keep.append((sd[k].clone().contiguous(), torch.zeros_like(sd[k]), torch.zeros_like(sd[k])))
        eng.bind_adapted(k, *keep[-1])
    data = []
    for i in range(2):
        image01, sparse = synth.synthetic_frame(i, H, W, 1)
        data.append((torch.from_numpy(((np.floor(image01 * 255) / 255 - mean) / std).astype(np.float32)).cuda(), torch.from_numpy(sparse).cuda()))
    eng.step(*data[0]); eng.forward_eval(*data[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        for _ in range(inner_iter):
            info, _ = eng.step(*data[f % 2])
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / (frames * inner_iter)
    t0 = time.perf_counter()
    for f in range(frames):
        d = eng.forward_eval(*data[f % 2])
    torch.cuda.synchronize()
    t_eval = (time.perf_counter() - t0) / frames
    out = {'workload': 'NLSPN (ResNet34 + 18-sweep propagation), 352x1216, %d TTA steps/frame, meta_bn (88 adapted tensors), legacy offsets as src/tta_main.py:317, batch 1' % inner_iter,
           'ms_per_step': 1e3 * t_step, 'frames_per_s': 1.0 / (inner_iter * t_step), 'eval_forward_ms': 1e3 * t_eval,
           'step_roofline': (lambda f, e, b: {'bound': 'mfma', 'alg_gmac_per_step': (f + b) / 1e9, 'alg_gmac_eval_forward': e / 1e9,
                                                'achieved': 3 * 2.0 * (f + b) / t_step / 1e12, 'peak': MFMA_BF16_PEAK / 1e12, 'unit': 'TFLOP/s',
                                                'frac': 3 * 2.0 * (f + b) / t_step / MFMA_BF16_PEAK,
                                                'frac_useful': 2.0 * (f + b) / t_step / MFMA_BF16_PEAK,
                                                'note': 'bf16x3: three bf16 MFMAs per fp32 product; frac counts all three, frac_useful = frac / 3 is the fp32-equivalent rate against the same bf16 peak'})(*nlspn_macs(H, W)),
           'finite': bool(torch.isfinite(info).all().item() and torch.isfinite(d).all().item())}
    eng.close()
    if dtype == 'fp32' and with_mixed:
        # the generic engine's mixed mode beside it (fp32 storage; one bf16 MFMA per product for the proxy frames, two -- hi + lo weights --
        # for the data gradients; scored depth 3.5e-5 after the three steps: tests/test_gpu_mixed.py, profiles/r06_nlspn_costdcnet_mixed.txt)
        m = nlspn_workload(frames, inner_iter, dtype='mixed')
        out['mixed_mode'] = {'ms_per_step': m['ms_per_step'], 'frames_per_s': m['frames_per_s'], 'eval_forward_ms': m['eval_forward_ms'], 'finite': m['finite']}
    return out


def head_stage2_workload(steps=30):
    """Stage-2 head trainer step (src/head_main.py:464-480) at the headline shape, both loss types; beside the metric."""
    from proxytta import synth
    from proxytta.engine import HEAD_PARAMS, HEAD_TARGETS, Engine
    eng = Engine(1, H, W, dtype='fp32', **HP)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(MODE).items()}
    eng.load_state_dict(sd)
    for name in eng.adapted:
        eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
    keep = []
    for k in HEAD_PARAMS:
        keep.append((torch.zeros_like(sd[k]), torch.zeros_like(sd[k])))
        eng.bind_head(k, sd[k], *keep[-1])
    for k in HEAD_TARGETS:
        eng.bind_head(k, sd[k])
    eng.set_head_hparams(lr=2e-4, adam_step=0)
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(700 + i, H, W, 1)] for i in range(4)]
    out = {'workload': 'stage-2 head trainer step (EMA target, prepare loss, Linear weight gradients, Adam), MSG_CHN 1layer, 352x1216, batch 1'}
    for name, reverse in (('head_selfsup_seq_ema_reverse', True), ('head_selfsup_seq_ema', False)):
        for i in range(3):
            eng.head_step(*frames[i % 4], reverse)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = eng.head_step(*frames[i % 4], reverse)
        torch.cuda.synchronize()
        out[name] = {'ms_per_step': 1e3 * (time.perf_counter() - t0) / steps, 'finite': bool(torch.isfinite(loss).all().item())}
    eng.close()
    # the same trainer on the op-list engine (csrc/ghead.hip): NLSPN, 352x1216 -- two no-gradient ResNet34 encoder passes from running
    # statistics + MLP(512, 1024, 1024) heads on 1,672 rows
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
    eng = Engine(1, H, W, backbone='nlspn', legacy_offset=True, lr=3e-4, w_sparse_depth=1.0, w_smoothness=0.0, w_cos=0.0, max_input_depth=80.0)
    sdn = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
    eng.load_state_dict({k: v for k, v in sdn.items() if v.dtype == torch.float32})
    keep = [(sdn[k].clone().contiguous(), torch.zeros_like(sdn[k]), torch.zeros_like(sdn[k])) for k in eng.adapted]
    for k, t in zip(eng.adapted, keep):
        eng.bind_adapted(k, *t)
    gp = ['%s.%s.%s' % (m, l, t) for m in ('proj', 'pred') for l in ('0', '1', '3') for t in ('weight', 'bias')]
    for k in gp:
        keep.append((torch.zeros_like(sdn[k]), torch.zeros_like(sdn[k])))
        eng.bind_head(k, sdn[k], *keep[-1])
    for k in gp[:6]:
        eng.bind_head('proj_t' + k[4:], sdn['proj_t' + k[4:]])
    eng.set_head_hparams(lr=2e-4, adam_step=0)
    image01, sparse = synth.synthetic_frame(700, H, W, 1)
    fr = (torch.from_numpy(((np.floor(image01 * 255) / 255 - mean) / std).astype(np.float32)).cuda(), torch.from_numpy(sparse).cuda())
    for i in range(2):
        eng.head_step(*fr, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        loss = eng.head_step(*fr, True)
    torch.cuda.synchronize()
    out['nlspn_head_selfsup_seq_ema_reverse'] = {'ms_per_step': 1e3 * (time.perf_counter() - t0) / 10, 'finite': bool(torch.isfinite(loss).all().item())}
    eng.close()
    return out


def costdcnet_shared(args, rank, world, dist):
    """BASELINE config 5: CostDCNet, 480x640, batched TTA with SHARED adapted parameters -- every rank adapts its own frame of
    the global batch, BatchNorm statistics are those of the global batch (SyncBatchNorm, src/tta_main.py:326), ONE flat
    all-reduce of the 112 adapted gradients of the DDP list per step (RCCL over xGMI on a multi-GPU node), identical Adam on every rank.
    Optional workload (`--workload costdcnet-shared`); the headline metric stays the default one."""
    from proxytta import synth
    from proxytta.distributed import shared_parameter_step
    from proxytta.engine import Engine
    h, w = 480, 640
    # syncbn_adapted: the adapted list of the reference's DDP run (convert_syncbn before adapt_parameters, src/tta_main.py:326,339):
    # every BatchNorm adapted, 116 listed entries = 112 tensors, the sparse encoder with its backward
    eng = Engine(1, h, w, backbone='costdcnet', max_predict_depth=8.0, lr=3e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, syncbn_adapted=True)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_costdcnet().items()}
    eng.load_state_dict(sd)
    keep = []
    for k in eng.adapted:
        keep.append((sd[k].clone().contiguous(), torch.zeros_like(sd[k]), torch.zeros_like(sd[k])))
        eng.bind_adapted(k, *keep[-1])
    mode = 'single rank'
    if world > 1:
        if dist.get_backend() == 'nccl':
            # the library's own RCCL communicator: BatchNorm statistics exchange + the gradient all-reduce are enqueued by
            # the library inside ONE fused step (include/ptta.h ptta_set_stat_sync_rccl / ptta_set_grad_sync_rccl)
            eng.enable_rccl_sync()
            mode = 'library-owned RCCL communicator (ncclAllReduce enqueued by libptta_hip on the step stream)'
        else:
            eng.enable_stat_sync()
            mode = 'torch.distributed callback (%s)' % dist.get_backend()
    frames = costdcnet_frames(2, h, w)                          # every rank its own two frames of the stream
    data = [[torch.from_numpy(np.roll(x, 7 * rank, axis=-1)).cuda() for x in f] for f in frames]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for i in range(max(args.warmup, 3)):
        shared_parameter_step(eng, data[i % 2][1], data[i % 2][2], loss_image=data[i % 2][0])
    # start-up self-check of the shared-parameter protocol: after the warm-up steps (different frames on every rank, one statistics exchange
    # per BatchNorm and one gradient all-reduce per step) every rank must hold BITWISE the parameters of rank 0
    flat = torch.cat([k[0].reshape(-1) for k in keep])
    same = True
    if dist is not None:
        ref = flat.clone() if dist.get_backend() == 'nccl' else flat.cpu().clone()
        dist.broadcast(ref, 0)
        ok = torch.tensor([1 if torch.equal(ref.to(flat.device), flat) else 0], device=ref.device, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        same = bool(ok.item())
    if rank == 0:
        print('costdcnet-shared: rccl_ranks=%d exchange=%s adapted parameters bitwise equal across ranks after %d steps: %s'
              % (world, mode, max(args.warmup, 3), same), file=sys.stderr)
    if not same:
        raise SystemExit('costdcnet-shared: the ranks hold different adapted parameters after the warm-up steps')
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        info, _ = shared_parameter_step(eng, data[i % 2][1], data[i % 2][2], loss_image=data[i % 2][0])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device='cuda' if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(info).all().item())
    eng.close()
    if rank == 0:
        print(json.dumps({
            'metric': 'TTA frames/sec (fwd+loss+bwd+Adam), CostDCNet 480x640, shared adapted parameters', 'value': world * args.steps / elapsed,
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 storage, bf16x6 (forward of the real frames) / bf16x3 MFMA arithmetic (fp32 accumulate)', 'data': 'synthetic',
            'config': {'workload': 'CostDCNet, 480x640 VOID-shaped synthetic (1500 points), global batch = n_gpus frames, 1 TTA step, meta_bn with the '
                                   'adapted list of the reference\'s DDP run (convert_syncbn before adapt_parameters: every BatchNorm incl. BatchNorm3d / '
                                   'BatchNorm1d / the sparse encoder\'s, 116 listed entries = 112 tensors, src/costdcnet_model_adapt.py:364-366)',
                       'parallelism': 'dp%d: SyncBatchNorm statistics exchange per BatchNorm (forward and backward) + one flat gradient all-reduce '
                                      '(11,888 floats) per step' % world,
                       'exchange': mode, 'rccl_ranks': world, 'params_bitwise_equal_across_ranks_after_warmup': same, 'finite': finite,
                       'launch_form': 'ptta_step call by call (the generic engine has no pipelined form: ptta_get_option "pipelined_active" = 0)'}}))
    if dist is not None:
        dist.destroy_process_group()


def msgchn_multi_stream(args, rank, world, dist, affinity):
    """Config 4 with S independent frame streams PER GPU (--streams-per-gpu S): every stream has its own engine (adapted parameters, Adam
    state, hipGraphs) and HIP stream; frames of one stream stay sequential.  One stream's step leaves ~20 % of the chip's time to kernel
    prologues, tails and launch floors, which a second stream fills (DESIGN_LOG.md section 13).  NOT the headline: `value` there is one stream
    per GPU; this line says so in config.streams_per_gpu."""
    from proxytta import synth
    from proxytta.engine import Engine
    S = args.streams_per_gpu
    engs, keep = [], []
    for k in range(S):
        eng = Engine(1, H, W, dtype=args.dtype, **HP)
        sd = {kk: torch.from_numpy(np.asarray(v)).cuda() for kk, v in synth.formula_state_dict(MODE).items()}
        eng.load_state_dict(sd)
        for name in eng.adapted:
            keep.append((sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name])))
            eng.bind_adapted(name, *keep[-1])
        engs.append(eng)
    streams = [torch.cuda.Stream() for _ in range(S)]
    frames = [[[torch.from_numpy(x).cuda() for x in synth.synthetic_frame((rank * S + k) * 1000 + i, H, W, 1)] for i in range(4)] for k in range(S)]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n, it0):
        info = None
        for it in range(n):
            for k, (e, st) in enumerate(zip(engs, streams)):
                with torch.cuda.stream(st):
                    info, _ = e.step(*frames[k][(it0 + it) % 4], next_frame=frames[k][(it0 + it + 1) % 4])
        return info
    run(args.warmup, 0)
    nblocks = 1 if args.single_block else max(10, -(-200 // max(args.steps, 1)))
    block_s, own_block_s, it = [], [], args.warmup
    for blk in range(nblocks):
        barrier()
        t0 = time.perf_counter()
        info = run(args.steps, it)
        it += args.steps
        barrier()
        dt = time.perf_counter() - t0
        own_block_s.append(dt)
        if dist is not None:
            t = torch.tensor([dt], device='cuda' if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        block_s.append(dt)
    elapsed = float(np.median(block_s))
    # per-rank view of the same blocks (each rank's OWN median block, gathered): tells a slow rank from a slow barrier in a SCALE run
    per_rank_ms = [1e3 * float(np.median(own_block_s)) / args.steps]
    if dist is not None:
        t = torch.zeros(world, device='cuda' if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        t[rank] = per_rank_ms[0]
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(x) for x in t.cpu()]
    finite = bool(torch.isfinite(info).all().item())
    for e in engs:
        e.close()
    if rank == 0:
        print(json.dumps({
            'metric': 'TTA frames/sec (fwd+loss+bwd+Adam) at 352x1216', 'value': world * S * args.steps / elapsed, 'unit': 'frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / (S * args.steps), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPE_TEXT[args.dtype], 'data': 'synthetic',
            'config': {'workload': 'MSG_CHN 1layer meta, 352x1216 KITTI-shaped synthetic, 1 TTA step/frame, batch 1 per stream',
                       'launch': 'ptta_step_pipelined on every stream (each call names its stream\'s next frame)',
                       'parallelism': 'independent frame streams, dp%d x %d streams per GPU, no collectives' % (world, S), 'streams_per_gpu': S,
                       'note': 'NOT the headline configuration (one stream per GPU): ms_per_step is per frame over all streams of a GPU',
                       'cpu_affinity': affinity, 'finite': finite},
            'timing': {'blocks': nblocks, 'steps_timed': nblocks * args.steps * S, 'ms_per_frame_min': 1e3 * min(block_s) / (S * args.steps),
                       'ms_per_frame_max': 1e3 * max(block_s) / (S * args.steps),
                       'per_rank_ms_per_step': {'min': min(per_rank_ms), 'median': float(np.median(per_rank_ms)), 'max': max(per_rank_ms), 'ranks': per_rank_ms,
                                                'note': "each rank's own median block / steps (S frames per step), before the MAX over ranks"}}}))
    if dist is not None:
        dist.destroy_process_group()


def plumbing_only(args, rank, world):
    """The multi-rank skeleton of main() without device work (tests/test_distributed_cpu.py): rendezvous on 127.0.0.1,
    warm-up, barrier, K timed no-op steps, barrier, MAX over ranks, rank 0 prints the JSON line with value null."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({'metric': 'TTA frames/sec (fwd+loss+bwd+Adam) at 352x1216', 'value': None, 'unit': 'frames/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': 'none', 'data': 'none',
                          'config': {'workload': 'plumbing only: no device work', 'parallelism': 'independent frame streams, dp%d, no collectives' % world}}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N child processes, one per GPU (as
    src/tta.py:189-269 spawns one adapt_ddp per GPU), BEFORE this process touches the GPU, forward rank 0's JSON line and
    exit with the worst child code.  Under torch.distributed.run (RANK set) this is never reached."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=os.environ.get('MASTER_PORT', str(port)))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for line in out0.decode().splitlines():          # ONE JSON line on stdout; library chatter (e.g. gloo's) to stderr
        print(line, file=sys.stdout if line.startswith('{') else sys.stderr)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def render_minor_of_hip_device(index):
    """DRM render minor of HIP device `index` WITHOUT touching the GPU: HIP enumerates the KFD topology's GPU nodes in node order
    (/sys/class/kfd/kfd/topology/nodes/<n>/properties: simd_count > 0, drm_render_minor), filtered by ROCR_VISIBLE_DEVICES and then
    HIP_VISIBLE_DEVICES when those are plain index lists.  Returns (minor, how) or (None, reason): renderD(128 + index) is NOT assumed --
    the DRM minors follow PCI probe order, which need not be the runtime's order."""
    base = '/sys/class/kfd/kfd/topology/nodes'
    try:
        minors = []
        for n in sorted(int(x) for x in os.listdir(base) if x.isdigit()):
            props = dict(l.split(None, 1) for l in open('%s/%d/properties' % (base, n)).read().splitlines() if ' ' in l)
            if int(props.get('simd_count', '0')) > 0 and int(props.get('drm_render_minor', '0')) > 0:
                minors.append(int(props['drm_render_minor']))
        for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
            v = os.environ.get(var, '').strip()
            if v:
                if not all(t.strip().isdigit() for t in v.split(',')):
                    return None, '%s is not an index list' % var
                minors = [minors[int(t)] for t in v.split(',') if int(t) < len(minors)]
        if index < len(minors):
            return minors[index], 'kfd topology'
        return None, 'kfd topology lists %d GPU(s)' % len(minors)
    except (OSError, ValueError, IndexError) as e:
        return None, 'kfd topology unreadable (%s)' % type(e).__name__


def pin_rank_to_local_cpus(local_rank, ranks_on_node):
    """Before any HIP call: restrict this rank (its Python launch loop, the staging copies, the runtime's helper threads) to the CPUs next
    to its GPU -- the NUMA node sysfs names for the GPU's render node (render_minor_of_hip_device; local_cpulist), otherwise an even slice
    of the CPUs this process may use.  Eight launch loops migrating over one host is where the >= 6x of config 4 would be lost first.
    Returns a description for the JSON line."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return 'unsupported'
    if ranks_on_node <= 1 or len(allowed) < 2 * ranks_on_node:
        return 'unchanged (%d CPUs, %d rank(s))' % (len(allowed), ranks_on_node)
    cpus, how = None, 'even slice'

    def cpulist(rank):
        minor, _ = render_minor_of_hip_device(rank)
        return None if minor is None else open('/sys/class/drm/renderD%d/device/local_cpulist' % minor).read().strip()
    try:
        minor, why = render_minor_of_hip_device(local_rank)
        if minor is None:
            how = 'even slice (%s)' % why
        else:
            txt = cpulist(local_rank)
            near = set()
            for part in txt.split(','):
                a, _, b = part.partition('-')
                near.update(range(int(a), int(b or a) + 1))
            near = sorted(near & set(allowed))
            # several GPUs share a NUMA node: slice that node's CPUs among them by local rank
            if len(near) >= 2:
                peers = [r for r in range(ranks_on_node) if cpulist(r) == txt]
                k, n = (peers.index(local_rank), len(peers)) if local_rank in peers else (0, 1)
                step = max(len(near) // n, 1)
                cpus, how = near[k * step:(k + 1) * step] or near, 'NUMA-local to renderD%d (%s), slice %d/%d' % (minor, txt, k, n)
    except (OSError, ValueError):
        pass
    if not cpus:
        step = len(allowed) // ranks_on_node
        cpus = allowed[local_rank * step:(local_rank + 1) * step]
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        return 'failed: %s' % e
    return '%s: %d CPUs [%d..%d]' % (how, len(cpus), cpus[0], cpus[-1])


def costdcnet_frames(count, h, w):
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
    from proxytta import synth
    out = []
    for i in range(count):
        image01, sparse = synth.synthetic_frame(i, h, w, 1, density=1500.0 / (h * w), dmin=0.3, dmax=7.5)      # VOID-1500
        raw = np.floor(image01 * 255.0).astype(np.float32)
        out.append((raw, ((raw / np.float32(255.0) - mean) / std).astype(np.float32), sparse))
    return out


# one CostDCNet TTA step at 480x640 with 1500 points, counted with forward hooks on the REAL reference by tools/costdcnet_count.py
# (SURVEY.md section 8d's rule: per conv / linear / sparse-conv layer input + output + weight elements; minimal backward)
COSTDC_ALG = {'forward_elements': 1079523516, 'forward_gmac': 212.040342912, 'backward_min_elements': 517284352, 'backward_min_gmac': 105.2539392,
              'eval_forward_elements': 538339550, 'eval_forward_gmac': 105.838309056}


def costdcnet_workload(frames=30):
    """BASELINE config 5's per-GPU work (not the headline metric): CostDCNet, 480x640 VOID-shaped frame with 1500 sparse points,
    1 TTA step per frame (bash/adapt/adapt_costdc_*.sh: inner_iter 1) + the scored eval forward, adapt_mode meta_bn (32 adapted
    tensors), batch 1, inputs resident in HBM."""
    from proxytta import synth
    from proxytta.engine import Engine
    h, w = 480, 640
    eng = Engine(1, h, w, backbone='costdcnet', max_predict_depth=8.0, lr=3e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_costdcnet().items()}
    eng.load_state_dict(sd)
    keep = []
    for k in eng.adapted:
        keep.append((sd[k].clone().contiguous(), torch.zeros_like(sd[k]), torch.zeros_like(sd[k])))
        eng.bind_adapted(k, *keep[-1])
    data = [[torch.from_numpy(x).cuda() for x in f] for f in costdcnet_frames(2, h, w)]
    eng.step(data[0][1], data[0][2], loss_image=data[0][0]); eng.forward_eval(data[0][1], data[0][2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        info, _ = eng.step(data[f % 2][1], data[f % 2][2], loss_image=data[f % 2][0])
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / frames
    t0 = time.perf_counter()
    for f in range(frames):
        d = eng.forward_eval(data[f % 2][1], data[f % 2][2])
    torch.cuda.synchronize()
    t_eval = (time.perf_counter() - t0) / frames
    out = {'workload': 'CostDCNet (Encoder2D + sparse 3-D encoder + P3D UNet3D, 16 planes), 480x640 VOID-shaped frame with 1500 points, '
                       '1 TTA step/frame, meta_bn (32 adapted tensors), batch 1',
           'ms_per_step': 1e3 * t_step, 'frames_per_s': 1.0 / t_step, 'eval_forward_ms': 1e3 * t_eval,
           'finite': bool(torch.isfinite(info).all().item() and torch.isfinite(d).all().item())}
    alg_bytes = 4.0 * (COSTDC_ALG['forward_elements'] + COSTDC_ALG['backward_min_elements'])
    alg_gmac = COSTDC_ALG['forward_gmac'] + COSTDC_ALG['backward_min_gmac']
    out['step_roofline'] = {
        'bound': 'hbm', 'alg_bytes_per_step': alg_bytes, 'alg_gmac_per_step': alg_gmac, 'achieved': alg_bytes / t_step / 1e9, 'peak': HBM_PEAK / 1e9,
        'unit': 'GB/s', 'frac': alg_bytes / t_step / HBM_PEAK,
        'eval_forward': {'alg_bytes': 4.0 * COSTDC_ALG['eval_forward_elements'], 'alg_gmac': COSTDC_ALG['eval_forward_gmac'],
                         'frac': 4.0 * COSTDC_ALG['eval_forward_elements'] / t_eval / HBM_PEAK},
        'mfma_frac_useful': 2e9 * alg_gmac / t_step / MFMA_BF16_PEAK,
        'note': 'algorithmic bytes / MACs measured with forward hooks on the reference (tools/costdcnet_count.py, SURVEY 8d rule: BatchNorm / ELU / '
                'pooling / fusion traffic counted as fused = 0); arithmetic: bf16x6 forward for the real frames, bf16x3 elsewhere; '
                'AI = 99 FLOP/B < ridge 312: HBM is the binding roof'}
    eng.close()
    return out


def costdcnet_cpu_baseline():
    """The CostDCNet oracle (PyTorch CPU restatement, oracle/costdcnet_oracle.py) on this box's host cores: ONE timed TTA step
    of the same 480x640 workload after one eval forward."""
    from oracle import costdcnet_oracle as CO
    from proxytta import synth
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    raw, im, sp = [torch.from_numpy(x) for x in costdcnet_frames(1, 480, 640)[0]]
    o = CO.CostDcnOracle(synth.formula_state_dict_costdcnet(), max_depth=8.0, lr=3e-3, w_sd=1.0, w_sm=2.0, w_cos=0.1)
    o.forward_eval(im, sp)
    t0 = time.time()
    o.step(im, sp, loss_image=raw)
    dt = time.time() - t0
    return {'value': 1.0 / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '1 TTA step of the same 480x640 workload after one eval forward (PyTorch-CPU oracle, fp32, %.1f s/step)' % dt}


def pipelined_self_check(dtype, k, rank=0, options=None, keep=()):
    """The headline is timed on ptta_step_pipelined; this replays the first `k` frames of the timed stream on TWO fresh handles -- one through
    the pipelined call (next frame announced), one through plain ptta_step -- from the same initial parameters and compares the adapted
    parameters, the Adam moments and every step's loss_info BIT FOR BIT (what tests/test_gpu_staging_augment.py asserts under pytest, here
    inside the run that produced the number)."""
    from proxytta import synth
    from proxytta.engine import ADAPTED, Engine
    nframes = 4
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(rank * 1000 + i, H, W, 1)] for i in range(nframes)]
    res = []
    for mode in ('pipelined', 'plain'):
        eng = Engine(1, H, W, dtype=dtype, options=options, keep=keep, **HP)
        sd = {kk: torch.from_numpy(np.asarray(v)).cuda() for kk, v in synth.formula_state_dict(MODE).items()}
        eng.load_state_dict(sd)
        st = {name: (sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name])) for name in ADAPTED}
        for name in ADAPTED:
            eng.bind_adapted(name, *st[name])
        infos = []
        for i in range(k):
            nxt = frames[(i + 1) % nframes] if mode == 'pipelined' else None
            infos.append(eng.step(*frames[i % nframes], next_frame=nxt)[0].clone())
        torch.cuda.synchronize()
        res.append((torch.stack(infos).cpu(), {n_: [t.detach().cpu().clone() for t in v] for n_, v in st.items()}))
        eng.close()
    (ia, pa), (ib, pb) = res
    same = bool(torch.equal(ia, ib))
    for n_ in pa:
        for ta, tb in zip(pa[n_], pb[n_]):
            same = same and bool(torch.equal(ta, tb))
    return same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--dtype', default=os.environ.get('PTTA_BENCH_DTYPE', 'mixed'), choices=['fp32', 'mixed'],
                    help="'mixed' (default): BASELINE config 2 -- the scored depth's tensors fp32 / bf16x3, the no_grad proxy pass, the heads and the "
                         "data gradients bf16 / single MFMA (include/ptta.h PTTA_DTYPE_MIXED); 'fp32': every tensor fp32 / bf16x3")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--single-block', action='store_true', help='time ONE block of --steps steps (quick A/B runs) instead of >= 10 blocks / >= 200 steps')
    ap.add_argument('--no-self-check', action='store_true', help='skip the pipelined-vs-plain bitwise replay on fresh handles')
    ap.add_argument('--no-nlspn', action='store_true', help='skip the side measurements (2layers, NLSPN, CostDCNet)')
    ap.add_argument('--streams-per-gpu', type=int, default=1,
                    help='config 4 only (independent frame streams): this many streams -- own adapted parameters, Adam state and hipGraphs each -- '
                         'per GPU; the headline (and the default) is 1')
    ap.add_argument('--workload', default='msg_chn', choices=['msg_chn', 'costdcnet-shared'],
                    help="'costdcnet-shared': BASELINE config 5 (shared adapted parameters, all-reduce + SyncBatchNorm) instead of the headline metric")
    ap.add_argument('--plumbing-only', action='store_true',
                    help='tests only: run launcher / rendezvous / barrier / max-over-ranks / JSON with NO device work '
                         '(value is null, "data": "none"); lets the multi-rank path run on a CPU box with gloo')
    args = ap.parse_args()

    if 'RANK' not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    kernarg = {'value': os.environ.get('HIP_FORCE_DEV_KERNARG'), 'set_by': 'caller' if KERNARG_PRESET is not None else 'bench.py, before the HIP runtime was loaded',
               'note': 'kernel arguments in device memory; 0 measured 7 % slower on the replayed step (DESIGN.md); the library never sets it'}
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)' % (args.gpus, world))
    affinity = pin_rank_to_local_cpus(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))      # before anything touches the GPU
    if args.plumbing_only:
        return plumbing_only(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device: the product path has no CPU fallback')
    torch.cuda.set_device(local_rank % torch.cuda.device_count())     # (modulo only matters for plumbing tests on a 1-GPU box)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # 'nccl' IS RCCL on ROCm; PTTA_BENCH_BACKEND=gloo exists only to exercise this path on one GPU
        dist.init_process_group(os.environ.get('PTTA_BENCH_BACKEND', 'nccl'), rank=rank, world_size=world)

    if args.workload == 'costdcnet-shared':
        return costdcnet_shared(args, rank, world, dist)
    if args.streams_per_gpu > 1:
        return msgchn_multi_stream(args, rank, world, dist, affinity)
    from proxytta import synth
    from proxytta.engine import ADAPTED, Engine
    keep = tuple(k for k in os.environ.get('PTTA_BENCH_KEEP', '').split(',') if k)      # precision-budget runs only (tools/precision_budget.sh)
    # A/B runs only (tools/exp_ab.sh): per-handle switches of include/ptta.h ptta_set_option as "key=value,key=value"
    options = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in os.environ.get('PTTA_BENCH_OPTIONS', '').split(',') if kv}
    eng = Engine(1, H, W, dtype=args.dtype, keep=keep, options=options, **HP)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(MODE).items()}
    eng.load_state_dict(sd)
    for name in ADAPTED:
        eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
    nframes = 4     # this rank's slice of the frame stream, resident in HBM before the timed region
    frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(rank * 1000 + i, H, W, 1)] for i in range(nframes)]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # A STREAM of frames: every call names the frame the next call will pass, and the part of that frame's forward that does not depend on
    # the adapted parameters (sparse-depth pooling, the frozen RGB encoder, the depth-only head of the stage-1 encoder) runs on its own
    # stream beside this frame's step (ptta_step_pipelined; identical results, tests/test_gpu_staging_augment.py).  K timed calls = K such
    # prefixes + K remainders: the first timed frame's prefix is started by the last warm-up call, the last call starts one for frame K+1.
    pipe = os.environ.get('PTTA_PIPELINE', '1') != '0'

    def nxt(i):
        return frames[(i + 1) % nframes] if pipe else None
    for i in range(args.warmup):
        eng.step(*frames[i % nframes], next_frame=nxt(i))
    # >= 10 blocks of K steps, >= 200 timed steps in all; every block is the contract's timed region: barrier + synchronize on
    # both sides, MAX over ranks; the reported figure is the MEDIAN block.  (With --warmup 0 the first timed call computes its own prefix in line.)
    nblocks = 1 if args.single_block else max(10, -(-200 // max(args.steps, 1)))
    block_s, own_block_s = [], []
    info = None
    it = args.warmup
    for blk in range(nblocks):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            info, _ = eng.step(*frames[it % nframes], next_frame=nxt(it))
            it += 1
        barrier()
        dt = time.perf_counter() - t0
        own_block_s.append(dt)
        if dist is not None:
            t = torch.tensor([dt], device='cuda' if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        block_s.append(dt)
    elapsed = float(np.median(block_s))
    # per-rank view of the same blocks (each rank's OWN median block, gathered): tells a slow rank from a slow barrier in a SCALE run
    per_rank_ms = [1e3 * float(np.median(own_block_s)) / args.steps]
    if dist is not None:
        t = torch.zeros(world, device='cuda' if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        t[rank] = per_rank_ms[0]
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(x) for x in t.cpu()]
    # the same K steps call by call with no next frame announced (plain ptta_step): reported beside the metric
    plain_ms = None
    if pipe and world == 1:
        for i in range(3):
            eng.step(*frames[i % nframes])
        pb = []
        for blk in range(nblocks):
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for i in range(args.steps):
                eng.step(*frames[i % nframes])
            torch.cuda.synchronize()
            pb.append(1e3 * (time.perf_counter() - tp) / args.steps)
        plain_ms = float(np.median(pb))
    # the all-fp32 mode beside the headline (same protocol, fresh handle): what the mixed mode buys on this box in this run
    other_dtype_ms, graph_ms = None, None

    def side_run(dt, opts):
        e2 = Engine(1, H, W, dtype=dt, options=opts, **HP)
        sd2 = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(MODE).items()}
        e2.load_state_dict(sd2)
        for name in ADAPTED:
            e2.bind_adapted(name, sd2[name], torch.zeros_like(sd2[name]), torch.zeros_like(sd2[name]))
        for i in range(max(args.warmup, 3)):
            e2.step(*frames[i % nframes], next_frame=nxt(i))
        ob = []
        for blk in range(min(nblocks, 5)):
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for i in range(args.steps):
                e2.step(*frames[i % nframes], next_frame=nxt(i))
            torch.cuda.synchronize()
            ob.append(1e3 * (time.perf_counter() - tp) / args.steps)
        e2.close()
        return float(np.median(ob)), len(ob)
    if world == 1 and not args.no_self_check:
        od = 'fp32' if args.dtype == 'mixed' else 'mixed'
        ms_, nb_ = side_run(od, options or None)
        other_dtype_ms = {'dtype': od, 'ms_per_step': ms_, 'blocks': nb_}
        ms_, nb_ = side_run(args.dtype, dict(options, graph=0 if options.get('graph') else 1))
        graph_ms = {'graph': 0 if options.get('graph') else 1, 'ms_per_step': ms_, 'blocks': nb_,
                    'note': 'the same stream of frames with the other launch form (graph = 1: every call replays captured hipGraphs; 0: direct launches)'}
    # Second figure: the same K steps with every frame starting in PAGEABLE HOST memory, as the reference's
    # dataloader hands it over (src/tta_main.py:519-523): pinned triple buffer + copy stream, frame k+1 travels while frame k
    # is adapted.  Never `value` (that one is HBM-resident by contract).
    from_host = None
    if world == 1:
        from proxytta.staging import FrameStager
        host_frames = [synth.synthetic_frame(500 + i, H, W, 1) for i in range(nframes)]
        st = FrameStager(1, H, W)
        st.submit(*host_frames[0])
        for phase, count in (('warmup', max(args.warmup, 2)), ('timed', args.steps)):
            if phase == 'timed':
                torch.cuda.synchronize()
                th = time.perf_counter()
            for i in range(count):
                st.submit(*host_frames[(i + 1) % nframes])
                im, sp = st.acquire()
                # frame pipelining on top of the staging: the next slot's H2D copy is ordered on the library's prefix stream
                eng.step(im, sp, next_frame=st.peek_next(eng.prefix_stream()) if pipe else None)
                st.release()
        torch.cuda.synchronize()
        host_elapsed = time.perf_counter() - th
        st.acquire(); st.release()
        # the reference's way on the same engine: blocking copy from pageable memory, then the step
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i in range(args.steps):
            im = torch.from_numpy(host_frames[i % nframes][0]).to('cuda')
            sp = torch.from_numpy(host_frames[i % nframes][1]).to('cuda')
            eng.step(im, sp)
        torch.cuda.synchronize()
        blocking_elapsed = time.perf_counter() - tb
        from_host = {'value': args.steps / host_elapsed, 'unit': 'frames/s', 'ms_per_step': 1e3 * host_elapsed / args.steps,
                     'h2d_bytes_per_frame': st.bytes_per_frame,
                     'staging': 'pageable host frame -> pinned slot (host memcpy) -> async H2D on a copy stream, three slots, host-ordered by events',
                     'blocking_to_device_ms_per_step': 1e3 * blocking_elapsed / args.steps}
    # Roofline leg: the timed region above replays a hipGraph, inside which kernels cannot be
    # bracketed by events, so the SAME K steps are re-run kernel by kernel right here with every
    # launch of the dominant kernel class bracketed by hipEvents on its launch stream.
    pipe_active = eng.get_option('pipelined_active')          # (read before the profiling leg switches the handle to one stream)
    eng.profile(True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        eng.step(*frames[i % nframes])
    torch.cuda.synchronize()
    instrumented_ms = 1e3 * (time.perf_counter() - t1) / args.steps
    prof = [eng.profile_read(k) for k in range(len(CLASSES))]
    # dominant kernel = conv32_s1_x3_kernel<T, RELU, *> (+ its fused first-layer form): the stride-1 convolutions with ReLU on load on maps above
    # 1/4 resolution -- the class with the largest share of the step (roofline_by_class).  Rounds 1-4 priced it TOGETHER with the small-map
    # kernel (conv32_s1_small_kernel, a different kernel bound by launch latency, not by HBM): that figure stays beside it for continuity.
    ms, abytes, macs, launches = prof[0]
    ms_b, abytes_b, macs_b, launches_b = [sum(x) for x in zip(prof[0], prof[1])]
    eng.profile(False)
    finite = bool(torch.isfinite(info).all().item())
    eq_plain = None
    if pipe and world == 1 and not args.no_self_check:
        eq_plain = pipelined_self_check(args.dtype, min(args.steps, 12), rank, options or None, keep)

    if rank == 0:
        mixed = args.dtype == 'mixed'
        steps_per_s = world * args.steps / elapsed
        # fp32 storage, bf16x3 arithmetic: 3 x 2 x MACs bf16 FLOP against 256 B of fp32 I/O per pixel -> 216 FLOP/B < ridge 312: HBM is the roof
        # (narrow launches: 1 x 2 x MACs against 128 B -> 144 FLOP/B: HBM again)
        achieved = abytes / (ms * 1e-3) / 1e9
        roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': achieved / (HBM_PEAK / 1e9)}
        roof.update({'kernel': ('stride-1 3x3 32->32 convolutions with ReLU on load on maps above 1/4 resolution: conv32_s1_x3_kernel<T, RELU = true, *> and its '
                                'fused first-layer form conv32_s1_first_kernel<T, *>' +
                                ('; T = float for the real frames, unsigned short (bf16) for the proxy frames -- both counted, bytes at the stored width'
                                 if mixed else '; T = float')),
                     'with_small_map_kernel': {'note': 'the definition of rounds 1-4: the same layers on maps of <= 256 tiles (conv32_s1_small_kernel<T, true, *>, '
                                                       'latency-bound one-tile blocks) counted into the class',
                                               'launches': launches_b, 'avg_launch_us': 1e3 * ms_b / max(launches_b, 1),
                                               'frac': abytes_b / (ms_b * 1e-3) / HBM_PEAK},
                     'measured': 'hipEvents around each launch, same K steps re-run kernel by kernel on ONE stream (%.3f ms/step)' % instrumented_ms,
                     'launches': launches, 'avg_launch_us': 1e3 * ms / max(launches, 1),
                     'alg_bytes_per_launch': abytes / max(launches, 1), 'traffic': None})
        tpath = os.path.join(ROOT, 'profiles', 'traffic_%s.json' % args.dtype)
        if os.path.exists(tpath):       # HBM bytes per launch from the separate rocprofv3 --pmc passes: NOT measured in this run
            tj = json.load(open(tpath))
            roof['traffic'] = tj.get('hbm_bytes_per_launch')
            roof['traffic_provenance'] = {'file': 'profiles/traffic_%s.json' % args.dtype, 'measured_in_this_run': False,
                                          'source': tj.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/traffic_from_pmc.py)'),
                                          'commit': tj.get('commit'), 'round': tj.get('round')}
        by_class = {}
        for name, (cms, cby, cmc, cn) in zip(CLASSES, prof):
            per = args.steps
            by_class[name] = {'launches_per_step': cn / per, 'us_per_step': 1e3 * cms / per, 'alg_bytes_per_step': cby / per,
                              'hbm_frac': (cby / (cms * 1e-3) / HBM_PEAK) if cms > 0 else None,
                              'mfma_bf16x3_frac': (3 * 2.0 * cmc / (cms * 1e-3) / MFMA_BF16_PEAK) if cms > 0 else None}
        by_class['_note'] = ('hipEvent leg (kernel by kernel, one stream, every launch bracketed): us_per_step sums the bracketed launches of the class; '
                             'algorithmic bytes by SURVEY 8d (0 for rest); heads priced as the 6 + 2 linear layers the reference executes')
        stored = sum(v['alg_bytes_per_step'] for k, v in by_class.items() if isinstance(v, dict))
        if mixed:
            # bytes at the width each tensor is STORED at, as the launches of the instrumented leg counted them (SURVEY 8d rule per layer)
            step_bytes = stored
            step_note = ('alg_bytes_per_step = sum over the step\'s launches of (input + output + weight elements) x the STORED element size of each '
                         '(fp32 for the real frames\' forward maps, bf16 for the proxy pass and the gradient maps): a narrower tensor lowers the numerator, '
                         'the fraction cannot inflate; the constant zero-image RGB-encoder pass is hoisted out of the step')
        else:
            step_bytes = (ALG_ELEMENTS_PER_STEP - ALG_ELEMENTS_HOISTED) * 4
            step_note = 'alg_bytes_per_step excludes the constant zero-image RGB-encoder pass, hoisted out of the step'
        out = {
            'metric': 'TTA frames/sec (fwd+loss+bwd+Adam) at 352x1216', 'value': steps_per_s, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE_TEXT[args.dtype],
            'data': 'synthetic',
            'config': {'workload': 'MSG_CHN 1layer meta, 352x1216 KITTI-shaped synthetic, 1 TTA step/frame, batch 1 per GPU',
                       'parallelism': 'independent frame streams, dp%d, no collectives' % world, 'finite': finite,
                       'cpu_affinity': affinity, 'hip_force_dev_kernarg': kernarg,
                       'other_dtype_same_run': other_dtype_ms, 'other_launch_form_same_run': graph_ms, 'options': options or 'defaults',
                       'launch': ('hipGraph replay (option graph = 1)' if options.get('graph') else
                                  'direct launches on the caller\'s stream + the handle\'s second and prefix streams (option graph = 0, the default)'),
                       'frame_pipelining': ('on: every call names the next frame of the stream; the part of its forward upstream of the adapted layer (frozen RGB '
                                            'encoder, sparse-depth pooling, stage-1/4 cascade down to decoder 1\'s last transposed conv) runs on a second stream '
                                            'beside the current step; K timed calls = K prefixes + K remainders; identical results' if pipe else 'off'),
                       'ms_per_step_without_frame_pipelining': plain_ms,
                       # the timed path against plain ptta_step on fresh handles, bit for bit (null: not run -- N > 1 or --no-self-check)
                       'pipelined_equals_plain': eq_plain,
                       # 1: ptta_step_pipelined ran as itself on the timed handle; 0: it degraded to ptta_step call by call (include/ptta.h)
                       'pipelined_active': pipe_active,
                       'rccl_ranks': world if (dist is not None and dist.get_backend() == 'nccl') else (0 if dist is None else None),
                       'collective_backend': None if dist is None else dist.get_backend()},
            'step_roofline': {'alg_bytes_per_step': step_bytes, 'alg_flop_per_step': ALG_FLOP_PER_STEP,
                              'alg_bytes_reference_executes': ALG_ELEMENTS_PER_STEP * 4,
                              'alg_bytes_fp32_storage': (ALG_ELEMENTS_PER_STEP - ALG_ELEMENTS_HOISTED) * 4,
                              'alg_bytes_bf16_storage': (ALG_ELEMENTS_PER_STEP - ALG_ELEMENTS_HOISTED) * 2,
                              'note': step_note,
                              'hbm_frac_per_gpu': step_bytes * (steps_per_s / world) / HBM_PEAK,
                              'hbm_frac_per_gpu_on_bf16_bytes': (ALG_ELEMENTS_PER_STEP - ALG_ELEMENTS_HOISTED) * 2 * (steps_per_s / world) / HBM_PEAK,
                              # bf16 matrix work ISSUED per step (SURVEY 8d MACs: grad-pass convs 29.34 G, proxy-pass convs 19.11 G of which the
                              # RGB encoder's 6.93 G are hoisted, heads 28.93 G, minimal backward 29.86 G): fp32 mode 3 MFMAs per product
                              # everywhere; mixed mode 3 for the real frames' forward only, 1 for the proxy pass, the heads and the data gradients
                              'mfma_frac_per_gpu': (2.0 * ((3 * 29.34e9 + (19.11e9 - 6.93e9) + 28.93e9 + 29.86e9) if mixed else 3 * (107.2e9 - 6.93e9))
                                                    * (steps_per_s / world) / MFMA_BF16_PEAK),
                              'mfma_frac_note': 'issued bf16 MFMA FLOP (bf16x3 launches counted 3x, single-MFMA launches 1x) / 2.5 PFLOP/s dense'},
            'roofline': roof,
            'roofline_by_class': by_class,
            'timing': {'protocol': '%d blocks x %d steps, each block bracketed by barrier + synchronize, MAX over ranks; ms_per_step = median block' % (nblocks, args.steps),
                       'blocks': nblocks, 'steps_timed': nblocks * args.steps, 'ms_per_step_median': 1e3 * elapsed / args.steps,
                       'ms_per_step_min': 1e3 * min(block_s) / args.steps, 'ms_per_step_max': 1e3 * max(block_s) / args.steps,
                       'per_rank_ms_per_step': {'min': min(per_rank_ms), 'median': float(np.median(per_rank_ms)), 'max': max(per_rank_ms),
                                                'ranks': per_rank_ms, 'note': "each rank's own median block, before the MAX over ranks"}},
        }
        if from_host is not None:
            out['from_host_memory'] = from_host
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
    eng.close()
    if rank == 0:
        if world == 1 and not args.no_nlspn:
            out['other_workloads'] = {'msg_chn_2layers': msgchn_2layers_workload(), 'nlspn': nlspn_workload(), 'costdcnet': costdcnet_workload(),
                                      'head_stage2': head_stage2_workload(), 'msg_chn_two_streams_per_gpu': two_streams_workload(dtype=args.dtype),
                                      'msg_chn_batch': msgchn_batch_workload(dtype=args.dtype),
                                      'msg_chn_step_plus_scored_eval': msgchn_adapt_loop_workload()}
            if not args.no_cpu_baseline:
                out['other_workloads']['nlspn']['cpu_baseline'] = nlspn_cpu_baseline()
                out['other_workloads']['costdcnet']['cpu_baseline'] = costdcnet_cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Long-horizon drift of the HIP step against the REAL reference's trajectory (tests/golden/msgchn_1layer_64x96_seq200.npz: 200 steps on 200
frames; msgchn_1layer_256x320_seq150.npz: 30 steps): ONE parameter set adapted over a stream of frames, as src/tta_main.py:504-636 does.
Per step: relative MAE of the scored depth (depth_eval, north_star bound 1e-3) on the sampled pixels, worst loss_info term, and -- where the
fixture holds them -- the adapted parameters.  Both precision modes, plain and pipelined calls.  python tools/drift_report.py > profiles/r06_drift.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from proxytta import synth  # noqa: E402
from tests.util import golden_hp, make_engine, rel_mae  # noqa: E402
from tests.test_oracle_golden import reference_floor  # noqa: E402

GD = os.path.join(ROOT, 'tests', 'golden')


def run(name, dtype, pipelined, keep=(), options=None):
    g = np.load(os.path.join(GD, name + '.npz'))
    h, w, n, steps, frame0 = [int(x) for x in g['meta']]
    hp, gain = golden_hp(g)
    eng, sd, adapted = make_engine(n, h, w, dtype, hp, gain, None, keep=keep, options=options, meta='2layers' if '2layers' in name else '1layer')
    frame = lambda s: [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(frame0 + s, h, w, n)]
    cur = frame(0)
    de, li, rows = [], [], []
    for s in range(steps):
        nxt = frame(s + 1)
        info, _ = eng.step(cur[0], cur[1], next_frame=nxt if pipelined else None)
        d_eval = eng.forward_eval_last() if pipelined else eng.forward_eval(*cur)
        p = 's%d/' % s
        de.append(rel_mae(d_eval.detach().float().cpu().numpy().reshape(-1)[g['pix_idx']], g[p + 'depth_eval_pix']))
        li.append(float((np.abs(info.cpu().numpy() - g[p + 'loss_info']) / np.maximum(np.abs(g[p + 'loss_info']), 1e-12)).max()))
        prm = [rel_mae(t[0], g[p + 'param/' + k]) for k, t in adapted.items() if p + 'param/' + k in g.files]
        if prm:
            rows.append((s, de[-1], li[-1], prm))
        cur = nxt
    eng.close()
    de, li = np.array(de), np.array(li)
    floor = reference_floor(g, steps)
    if keep:
        dtype = dtype + ' with ' + ' + '.join(keep) + ' kept at fp32 / bf16x3'
    if options:
        dtype = dtype + ' ' + ' '.join('%s=%d' % kv for kv in options.items())
    print('%s  %s  %s: depth_eval rel MAE  max %.2e at step %d | mean %.2e | last %.2e   loss_info max %.2e' % (
        name, dtype, 'pipelined + forward_eval_last' if pipelined else 'ptta_step + ptta_forward_eval', de.max(), int(de.argmax()), de.mean(), de[-1], li.max()))
    q = max(1, steps // 10)
    print('   per window of %d steps (max depth_eval): ' % q + ' '.join('%.1e' % de[i:i + q].max() for i in range(0, steps, q)))
    print('   the reference against ITSELF, one weight 1 ulp off:  ' + ' '.join('%.1e' % floor[min(i + q, steps) - 1] for i in range(0, steps, q)))
    print('   worst ratio to that floor (+5e-5): %.2f' % float((de / (floor + 5e-5)).max()))
    for s, d, l, prm in rows:
        print('   step %3d depth_eval %.2e loss_info %.1e  weight %.2e bias %.2e' % (s, d, l, prm[0], prm[1]))
    return de.max()


if __name__ == '__main__':
    print('north_star tolerance on the scored depth: 1e-3 relative MAE')
    if len(sys.argv) > 1 and sys.argv[1] == '--classes':
        # which narrow class of the mixed mode carries its long-horizon distance (include/ptta.h PTTA_MIXED_KEEP_*)
        for name in (sys.argv[2:] or ['msgchn_1layer_256x320_seq150']):
            for keep in ((), ('backward',), ('proxy',), ('heads',), ('proxy', 'heads'), ('backward', 'heads'), ('backward', 'proxy')):
                run(name, 'mixed', False, keep)
        raise SystemExit(0)
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            run(name, 'fp32', True)
            run(name, 'mixed', True)
            run(name, 'mixed', True, options={'bwd_w2': 0})        # round 5's form of the narrow data gradients (bf16-rounded weights)
        raise SystemExit(0)
    for name in ('msgchn_1layer_64x96_seq200', 'msgchn_1layer_256x320_seq150', 'msgchn_1layer_352x1216_seq120'):
        run(name, 'fp32', True)
        run(name, 'mixed', False)
        run(name, 'mixed', True)
        run(name, 'mixed', True, options={'bwd_w2': 0})            # round 5's form of the narrow data gradients (bf16-rounded weights)

"""Measured gradient / parameter deviations on the gate fixtures (bounds of tests/test_gpu_parity.py::test_fused_step_on_both_sides_of_the_cosine_gate = 2x these)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')]
from proxytta import synth
from tests.util import golden_hp, make_engine, rel_mae
for name in ('msgchn_1layer_64x96_gate_below', 'msgchn_1layer_64x96_gate_above'):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    hp, gain = golden_hp(g)
    for impl in ('exact', None):
        eng, sd, adapted = make_engine(n, h, w, 'fp32', hp, gain, impl, head_bias=float(g['head_bias']))
        for s in range(steps):
            image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(s, h, w, n)]
            p = 's%d/' % s
            info, depth = eng.step(image, sparse, want_depth=True)
            gw, gb = eng.debug_tensor('gW').view(32, 32, 3, 3), eng.debug_tensor('gB')
            d_eval = eng.forward_eval(image, sparse)
            row = [rel_mae(gw, g[p + 'grad/conv1_rgb_meta.weight']), rel_mae(gb, g[p + 'grad/conv1_rgb_meta.bias'])]
            for k, (prm, m, v) in adapted.items():
                row += [rel_mae(prm, g[p + 'param/' + k]), rel_mae(m, g[p + 'exp_avg/' + k]), rel_mae(v, g[p + 'exp_avg_sq/' + k])]
            print(name, impl, s, 'gW %.2e gB %.2e | W: p %.2e m %.2e v %.2e | b: p %.2e m %.2e v %.2e | depth %.2e eval %.2e' % (
                *row, rel_mae(depth, g[p + 'depth_train']), rel_mae(d_eval, g[p + 'depth_eval'])))
        eng.close()

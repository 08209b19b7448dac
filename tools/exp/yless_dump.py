"""three NLSPN steps + CostDCNet step on the loaded library; gradients, loss terms and depths to an .npz (tools/exp/yless_bitwise.sh compares two builds bitwise)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np, torch
sys.argv = [sys.argv[0]] + sys.argv[1:]
out = sys.argv[1]
import importlib.util
spec = importlib.util.spec_from_file_location('gmr', os.path.join(ROOT, 'tools', 'generic_mixed_report.py'))
src = open(os.path.join(ROOT, 'tools', 'generic_mixed_report.py')).read().split("\nfor nm in ('nlspn_352x1216_legacy_inner3'")[0]
g = {'__file__': os.path.join(ROOT, 'tools', 'generic_mixed_report.py'), '__name__': 'gmr'}
exec(compile(src, 'gmr', 'exec'), g)
res = {}
for backbone, name in (('nlspn', 'nlspn_228x304_legacy'), ('costdcnet', 'costdcnet_320x400')):
    gold = np.load(os.path.join(g['GD'], name + '.npz'))
    h, w, n, steps = [int(x) for x in gold['meta']]
    v = [float(x) for x in gold['hp']]
    hp = dict(lr=v[0], betas=(v[1], v[2]), eps=v[3], weight_decay=v[4], w_sparse_depth=v[5], w_smoothness=v[6], w_cos=v[7], max_input_depth=v[8] if backbone == 'nlspn' else None)
    eng, adapted = g['build'](backbone, n, h, w, hp, 'fp32', ())
    fr = g['nlspn_frame'](0, h, w, n) if backbone == 'nlspn' else g['costdc_frame'](0, h, w, n, float(gold['density']))
    raw, image1, sparse = [torch.from_numpy(x).cuda() for x in fr]
    for s in range(3):
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        res['%s/s%d/info' % (backbone, s)] = info.cpu().numpy(); res['%s/s%d/depth' % (backbone, s)] = depth.cpu().numpy()
        if s == 0:
            for k in adapted: res['%s/grad/%s' % (backbone, k)] = eng.grad(k, adapted[k][0]).cpu().numpy()
    res['%s/eval' % backbone] = eng.forward_eval(image1, sparse).cpu().numpy()
    for k in adapted: res['%s/param/%s' % (backbone, k)] = adapted[k][0].cpu().numpy()
    eng.close()
np.savez(out, **res)
print('wrote', out, len(res), 'arrays')

"""Time the NLSPN TTA step (BASELINE config 3: 1x3x352x1216) on one GPU.  python tools/bench_nlspn.py [H W steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
import numpy as np
import torch

from proxytta import synth
from proxytta.engine import Engine

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 352
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1216
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    n = 1
    eng = Engine(n, h, w, backbone='nlspn', lr=3e-4, w_sparse_depth=1.0, w_smoothness=0.0, w_cos=0.0, max_input_depth=80.0)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
    eng.load_state_dict({k: v for k, v in sd.items() if v.dtype == torch.float32})
    keep = []
    for k in eng.adapted:
        p = sd[k].clone().contiguous()
        keep.append((p, torch.zeros_like(p), torch.zeros_like(p)))
        eng.bind_adapted(k, *keep[-1])
    image01, sparse = synth.synthetic_frame(0, h, w, n)
    image = torch.from_numpy(((np.floor(image01 * 255) / 255 - MEAN) / STD).astype(np.float32)).cuda()
    sparse = torch.from_numpy(sparse).cuda()
    print('alloc GB', torch.cuda.mem_get_info()[0] / 2**30, 'free of', torch.cuda.mem_get_info()[1] / 2**30, flush=True)
    for name, fn in (('step', lambda: eng.step(image, sparse)), ('eval', lambda: eng.forward_eval(image, sparse))):
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        print('%s: %.2f ms' % (name, (time.time() - t0) / steps * 1e3), flush=True)
    d = eng.forward_eval(image, sparse)
    print('depth finite', bool(torch.isfinite(d).all()), float(d.min()), float(d.max()))


if __name__ == '__main__':
    main()

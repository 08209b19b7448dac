"""Host enqueue time vs GPU time of one step (is a workload launch-bound on the host?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from proxytta import synth
from proxytta.engine import Engine
which = sys.argv[1]
if which == 'costdcnet':
    h, w = 480, 640
    eng = Engine(1, h, w, backbone='costdcnet', lr=3e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_predict_depth=8.0)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_costdcnet().items()}
else:
    h, w = 352, 1216
    eng = Engine(1, h, w, backbone='nlspn', legacy_offset=True, lr=3e-4, w_sparse_depth=1.0, w_smoothness=0.0, w_cos=0.0, max_input_depth=80.0)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict_nlspn().items()}
eng.load_state_dict({k: v for k, v in sd.items() if v.dtype == torch.float32} if which != 'costdcnet' else sd)
keep = []
for k in eng.adapted:
    keep.append((sd[k].clone().contiguous(), torch.zeros_like(sd[k]), torch.zeros_like(sd[k]))); eng.bind_adapted(k, *keep[-1])
im, sp = (torch.from_numpy(a).cuda() for a in synth.synthetic_frame(0, h, w, 1, density=0.005 if which == 'costdcnet' else 0.05, dmin=0.3, dmax=7.5))
for _ in range(2): eng.step(im, sp)
torch.cuda.synchronize()
enq = []; tot = []
for _ in range(5):
    t0 = time.perf_counter(); eng.step(im, sp); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
print(which, 'enqueue ms', [round(x, 2) for x in enq], 'total ms', [round(x, 2) for x in tot])

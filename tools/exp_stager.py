"""Staged frame loop timing variants (tools/exp_h2d.py measures the raw copies)."""
import sys, time
sys.path.insert(0, 'tta-depth-completion_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from proxytta import synth
from proxytta.staging import FrameStager
from tests.util import make_engine
H, W = 352, 1216
HP = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
eng, _, _ = make_engine(1, H, W, 'fp32', HP)
frames = [synth.synthetic_frame(i, H, W, 1) for i in range(4)]
res = [[torch.from_numpy(a).cuda() for a in f] for f in frames]
which = sys.argv[1]
st = FrameStager(1, H, W)
def loop(n, step=True, stage=True):
    if stage: st.submit(*frames[0])
    for i in range(n):
        if stage:
            t0 = time.perf_counter(); st.submit(*frames[(i + 1) % 4]); t1 = time.perf_counter()
            im, sp = st.acquire()
        else:
            im, sp = res[i % 4]; t0 = t1 = 0
        if step: eng.step(im, sp)
        t2 = time.perf_counter()
        if stage: st.release()
        if i == n - 1: print('  last iter: submit %.3f ms, acquire+step enqueue %.3f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
    if stage: st.acquire(); st.release()
for name, kw in (('step only', dict(stage=False)), ('stage only', dict(step=False)), ('both', dict())):
    if which not in name: continue
    loop(5, **kw); torch.cuda.synchronize(); t0 = time.perf_counter(); loop(30, **kw); torch.cuda.synchronize()
    print(name, 1e3 * (time.perf_counter() - t0) / 30, 'ms/iter', flush=True)

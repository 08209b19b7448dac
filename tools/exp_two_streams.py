"""Two independent TTA frame streams (own handles, own adapted parameters, own hipGraphs) enqueued on two HIP streams of ONE
GPU versus one stream: how much of the step is latency that a second stream can fill."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
import numpy as np, torch
import bench
from proxytta import synth
from proxytta.engine import Engine

def make():
    eng = Engine(1, bench.H, bench.W, dtype='fp32', **bench.HP)
    sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(bench.MODE).items()}
    eng.load_state_dict(sd)
    keep = []
    for name in eng.adapted:
        keep.append((sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name])))
        eng.bind_adapted(name, *keep[-1])
    return eng, keep

frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(i, bench.H, bench.W, 1)] for i in range(4)]
for nstreams in (1, 2, 3):
    engs = [make() for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    def run(k):
        for it in range(k):
            for (e, _), s in zip(engs, streams):
                with torch.cuda.stream(s):
                    e.step(*frames[it % 4])
    run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 100; run(K); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('streams %d: %.3f ms per frame, %.1f frames/s' % (nstreams, 1e3 * dt / (K * nstreams), K * nstreams / dt))
    for e, _ in engs: e.close()

#!/usr/bin/env python3
"""Where the branches of one replayed, pipelined step start and end WITHOUT a profiler attached: option "stamps" (include/ptta.h) adds
one-thread nodes to the step's and the prefix's graphs that write wall_clock64() ticks since the step's first node.  rocprofv3's kernel
trace slows the host enough to move the next frame's prefix to the end of the step and to serialise the step's two chains; this does not.
  python3 tools/step_stamps.py [mixed|fp32] [key=value ...]     (ptta_set_option keys)"""
import os
import sys
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
from proxytta import synth
from proxytta.engine import ADAPTED, Engine

dtype = sys.argv[1] if len(sys.argv) > 1 and '=' not in sys.argv[1] else 'mixed'
opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in sys.argv[1:] if '=' in kv}
opts['stamps'] = 1
eng = Engine(1, bench.H, bench.W, dtype=dtype, options=opts, **bench.HP)
sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.formula_state_dict(bench.MODE).items()}
eng.load_state_dict(sd)
for name in ADAPTED:
    eng.bind_adapted(name, sd[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]))
frames = [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(i, bench.H, bench.W, 1)] for i in range(4)]
khz = 100000.0      # wall_clock64: 100 MHz constant clock on MI300-class parts
names = ['graph start', 'real chain start', 'real chain end', 'proxy branch start', 'proxy chain end', 'heads part 1 end', 'heads part 2 end',
         'decoder 3 end', 'heads backward end', 'backward + weight gradient done (Adam inside its reduction)', 'after Adam (own launch: option adam_in_wgrad = 0)', "next frame's prefix start", 'its RGB encoder end', 'its end']
acc = []
k = 0
for rep in range(24):
    # steady state: the host runs ahead of the GPU (no wait between steps) and the stamps read are those of the LAST step of a burst --
    # with a wait after every step the direct-launch form shows the host's enqueue times instead (the prefix "starts" when it is queued)
    for j in range(8):
        eng.step(*frames[k % 4], next_frame=frames[(k + 1) % 4]); k += 1
    torch.cuda.synchronize()
    if rep >= 4:
        acc.append(eng.debug_tensor('stamps').cpu().numpy()[:14] / khz * 1e3)
t = np.median(np.stack(acc), axis=0)
print('dtype', dtype, 'options', opts, '(microseconds since the first node of the step graph; median over 20 bursts of 8 steps, last step of each)')
for j, (n, v) in enumerate(zip(names, t)):
    if dtype == 'mixed' or j not in (1, 2, 3, 4, 5, 6):        # (the fp32 mode has one chain: no proxy branch, heads in one piece)
        print('  %-26s %8.1f us' % (n, v))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200):
    eng.step(*frames[i % 4], next_frame=frames[(i + 1) % 4])
torch.cuda.synchronize()
print('  ms_per_step (with the stamp nodes): %.4f' % ((time.perf_counter() - t0) / 200 * 1e3))

#!/usr/bin/env python3
"""Race soak of the timed path: K frames of a stream through ptta_step_pipelined (next frame announced, its prefix on its own stream beside the
step, two buffer sets, cross-stream events) and the same K frames through plain ptta_step on a second handle, both modes; every step's loss_info,
the adapted parameters and both Adam moments compared BIT FOR BIT (bench.pipelined_self_check with a long stream; tests hold 6 - 12 frames).
  python tools/soak_pipelined.py [K = 2000]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
import bench

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for dtype in ('mixed', 'fp32'):
    t0 = time.perf_counter()
    same = bench.pipelined_self_check(dtype, k)
    print('%s: %d frames at %dx%d, pipelined vs plain: adapted parameters, Adam moments and every loss_info bitwise equal: %s (%.1f s)'
          % (dtype, k, bench.H, bench.W, same, time.perf_counter() - t0), flush=True)
    assert same

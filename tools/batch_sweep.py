#!/usr/bin/env python3
"""Frames/s of the fused step as a function of the per-GPU batch (the reference's scripts use n_batch 16): bench.py's
msg_chn_batch side workload on its own.   python tools/batch_sweep.py [--dtype mixed|fp32] [N ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
args = sys.argv[1:]
dtype = 'mixed'
if args and args[0] == '--dtype':
    dtype, args = args[1], args[2:]
ns = [int(x) for x in (args or ['1', '2', '4', '8', '16'])]
res = bench.msgchn_batch_workload(ns, dtype=dtype)
for k in [str(n) for n in ns]:
    r = res[k]
    print('batch %2s: %7.3f ms/step %7.1f frames/s  x%.2f  step_hbm_frac %.3f  serial %.0f us/frame' % (
        k, r['ms_per_step'], r['frames_per_s'], r.get('speedup_vs_batch_1', 0), r['step_hbm_frac'], r['serial_sum_us_per_frame']), flush=True)
    print('   ' + '  '.join('%s %.0f/%s' % (c, v['us_per_frame'], ('%.2f' % v['hbm_frac']) if v['hbm_frac'] else '-') for c, v in r['roofline_by_class'].items()))
print(json.dumps(res))

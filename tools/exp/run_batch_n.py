"""one N-frames-per-call measurement of bench.msgchn_batch_workload under a profiler: python3 tools/exp/run_batch_n.py N [dtype]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd'))
import bench
n = int(sys.argv[1]); dt = sys.argv[2] if len(sys.argv) > 2 else 'mixed'
r = bench.msgchn_batch_workload(ns=(n,), dtype=dt, steps=10, blocks=2, warmup=3, by_class=False)
print({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a in ('ms_per_step', 'frames_per_s')}) for k, v in r.items()})

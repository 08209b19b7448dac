"""bench.py's side paths under -m gpu: every leg that is not the headline still has to run (round 5's review found --streams-per-gpu
raising a NameError that no test covered)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args):
    r = subprocess.run([sys.executable, 'bench.py'] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])


def test_streams_per_gpu_leg_runs_and_reports_per_rank_time():
    out = _bench(['--streams-per-gpu', '2', '--steps', '4', '--warmup', '2', '--single-block'])
    assert out['config']['streams_per_gpu'] == 2 and out['config']['finite'] and out['value'] > 0
    assert out['timing']['per_rank_ms_per_step']['ranks'] and out['dtype'].startswith('mixed')


def test_batch_workload_reports_every_n():
    sys.path.insert(0, ROOT)
    import bench
    res = bench.msgchn_batch_workload(ns=(1, 2), steps=3, blocks=1, warmup=2)
    for k in ('1', '2'):
        r = res[k]
        assert r['finite'] and r['pipelined_active'] == 1 and r['frames_per_s'] > 0 and 0 < r['step_hbm_frac'] < 1
        assert set(r['roofline_by_class']) == set(bench.CLASSES)
    assert abs(res['2']['alg_bytes_per_step'] / res['1']['alg_bytes_per_step'] - 2.0) < 0.02

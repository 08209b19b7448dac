"""world_size-2 gloo tests of the N>1 plumbing (runs on CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from proxytta import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    gw = torch.full((32, 32, 3, 3), float(rank + 1))
    gb = torch.arange(32, dtype=torch.float32) * (rank + 1)
    D.allreduce_adapted_grads([gw, gb])
    ok = torch.allclose(gw, torch.full_like(gw, 1.5)) and torch.allclose(gb, torch.arange(32, dtype=torch.float32) * 1.5)
    # bench.py's timing protocol: barrier, then MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t) == 2.0
    out[rank] = (ok, D.shard_frames(7, rank, world))
    dist.destroy_process_group()


def test_gloo_world2_grad_allreduce_and_sharding():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0][0] and out[1][0]
    assert out[0][1] == [0, 2, 4, 6] and out[1][1] == [1, 3, 5]
    assert sorted(out[0][1] + out[1][1]) == list(range(7))      # disjoint cover of the stream


def test_single_process_is_identity():
    g = [torch.ones(3), torch.zeros(2)]
    assert D.allreduce_adapted_grads(g) is g


def _bench(cmd, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable] + cmd, cwd=root, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_one_rank_per_gpu():
    """`python bench.py --gpus 2` (no launcher around it) starts 2 ranks before touching any device, rendezvous on
    127.0.0.1, barrier + MAX-over-ranks timing, ONE JSON line from rank 0 with n_gpus = 2 (device work stubbed out)."""
    out = _bench(['bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1', '--plumbing-only'])
    assert out['n_gpus'] == 2 and out['steps'] == 4 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['value'] is None and out['data'] == 'none' and out['ms_per_step'] >= 1.0


def test_bench_under_torch_distributed_run():
    """The driver's launch line: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2."""
    out = _bench(['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                  '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--plumbing-only'])
    assert out['n_gpus'] == 2 and out['steps'] == 3


def test_bench_refuses_mismatched_world():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--plumbing-only'], cwd=root, capture_output=True, text=True,
                       env=dict(os.environ, RANK='0', WORLD_SIZE='4'), timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr


def _syncbn_worker(rank, world, port, out):
    """The SyncBatchNorm exchange protocol of ptta_set_stat_sync, in torch on CPU: per-rank {sum, sum of squares} -> SUM
    all-reduce -> finalize with world x the local row count == BatchNorm over the concatenated batch."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    x_all = torch.randn(2 * 96, 32, generator=g) * 3 + 1.5
    x = x_all[rank * 96:(rank + 1) * 96]
    sums = torch.stack([x.double().sum(0), (x.double() ** 2).sum(0)])       # "collapse": double sums of the local partials
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)                             # the callback
    R = world * x.shape[0]                                                  # finalize with world x local rows
    mean = sums[0] / R
    var = sums[1] / R - mean ** 2
    y = (x - mean.float()) / torch.sqrt(var.float() + 1e-5)
    ref = torch.nn.functional.batch_norm(x_all, None, None, None, None, True, 0.0, 1e-5)[rank * 96:(rank + 1) * 96]
    out[rank] = bool(torch.allclose(y, ref, atol=2e-5))
    dist.destroy_process_group()


def test_syncbn_exchange_protocol_equals_global_batchnorm():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_syncbn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] and out[1]

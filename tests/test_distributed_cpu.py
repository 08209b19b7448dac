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

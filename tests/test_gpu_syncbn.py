"""Shared-parameter multi-rank mode = the reference's actual DDP run (src/tta_main.py:326-354): SyncBatchNorm statistics over
the global batch (ptta_set_stat_sync) + mean all-reduce of the adapted-parameter gradients.  Two ranks (two processes on
the one GPU of the test box, gloo backend: the collectives are the same calls as with RCCL) with HALF a batch each must
reproduce the reference's single-process batch-2 golden run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, backbone, out):
    for p in (ROOT, os.path.join(ROOT, 'tta-depth-completion_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from proxytta import distributed as D
    from proxytta import synth
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    res = {}
    if backbone == 'msg_chn':
        from tests.util import golden_hp, make_engine
        g = np.load(os.path.join(GOLD, 'msgchn_1layer_32x48_n2.npz'))
        h, w, n, steps = [int(x) for x in g['meta']]
        hp, gain = golden_hp(g)
        eng, sd, adapted = make_engine(1, h, w, 'fp32', hp, gain, None)
        eng.enable_stat_sync()
        for s in range(steps):
            image, sparse = [torch.from_numpy(x[rank:rank + 1]).cuda() for x in synth.synthetic_frame(s, h, w, n)]
            info, depth = D.shared_parameter_step(eng, image, sparse)
            res['s%d/info' % s] = info.cpu().numpy()
            res['s%d/depth' % s] = depth.cpu().numpy()
            for k, (prm, m, v) in adapted.items():
                res['s%d/param/%s' % (s, k)] = prm.cpu().numpy().copy()
                res['s%d/grad/%s' % (s, k)] = eng.grad(k, prm).cpu().numpy()
            for k in ('proj.1.running_mean', 'pred.1.running_var'):
                res['s%d/buf/%s' % (s, k)] = sd[k].cpu().numpy().copy()
            res['s%d/eval' % s] = eng.forward_eval(image, sparse).cpu().numpy()
    elif backbone == 'costdcnet_ddp':
        # the reference's DDP run: every BatchNorm adapted (PTTA_SYNCBN_ADAPT), statistics + gradients over both ranks
        from tests.test_gpu_costdcnet import costdc_frame
        from tests.test_gpu_costdcnet_syncbn import golden_hp as hp_of, make as make_ddp
        g = np.load(os.path.join(GOLD, 'costdcnet_64x64_n2_syncbn.npz'))
        h, w, n, steps = [int(x) for x in g['meta']]
        eng, sd, adapted = make_ddp(1, h, w, hp_of(g))
        eng.enable_stat_sync()
        raw, image1, sparse = [torch.from_numpy(x[rank:rank + 1]).cuda() for x in costdc_frame(0, h, w, n, float(g['density']))]
        info, depth = D.shared_parameter_step(eng, image1, sparse, loss_image=raw)
        res['depth'] = depth.cpu().numpy()
        res['info'] = info.cpu().numpy()
        for k, (prm, m, v) in adapted.items():
            res['param/' + k] = prm.cpu().numpy().copy()
            res['grad/' + k] = eng.grad(k, prm).cpu().numpy()
        res['eval'] = eng.forward_eval(image1, sparse).cpu().numpy()
        # two more shared steps on further frames (every rank its own): the adapted parameters must stay BITWISE equal across the ranks
        for s in (1, 2):
            raw, image1, sparse = [torch.from_numpy(x[rank:rank + 1]).cuda() for x in costdc_frame(s, h, w, n, float(g['density']))]
            D.shared_parameter_step(eng, image1, sparse, loss_image=raw)
            for k, (prm, m, v) in adapted.items():
                res['s%d/param/%s' % (s, k)] = prm.cpu().numpy().copy()
    else:
        from tests.test_gpu_costdcnet import costdc_frame, make_costdc
        g = np.load(os.path.join(GOLD, 'costdcnet_64x64_n2.npz'))
        h, w, n, steps = [int(x) for x in g['meta']]
        eng, sd, adapted = make_costdc(1, h, w)
        eng.enable_stat_sync()
        raw, image1, sparse = [torch.from_numpy(x[rank:rank + 1]).cuda() for x in costdc_frame(0, h, w, n, float(g['density']))]
        depth, emb, ref = eng.forward_train(image1, sparse)
        res['depth'] = depth.cpu().numpy()
        res['buf'] = sd['unet3d.inc.double_conv.0.bn1.running_mean'].cpu().numpy().copy()
    out[rank] = res
    eng.close()
    dist.destroy_process_group()


def _run(backbone):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), backbone, out), nprocs=2, join=True)
    return out[0], out[1]


def test_two_ranks_reproduce_the_batch2_reference_run():
    from tests.util import rel_mae
    r0, r1 = _run('msg_chn')
    g = np.load(os.path.join(GOLD, 'msgchn_1layer_32x48_n2.npz'))
    steps = int(g['meta'][3])
    for s in range(steps):
        p = 's%d/' % s
        depth = np.concatenate([r0[p + 'depth'], r1[p + 'depth']], 0)
        assert rel_mae(depth, g[p + 'depth_train']) < 1e-4                  # heads' BatchNorm does not touch the depth; the step count does
        # loss terms are means over the batch: the reference's batch-2 values are the averages of the two ranks' values
        np.testing.assert_allclose(0.5 * (r0[p + 'info'] + r1[p + 'info']), g[p + 'loss_info'], rtol=1e-4)
        for k in ('conv1_rgb_meta.weight', 'conv1_rgb_meta.bias'):
            assert np.array_equal(r0[p + 'param/' + k], r1[p + 'param/' + k])     # identical update on every rank
            # each rank holds the gradient of ITS local loss; DDP's mean over the ranks is the batch-2 gradient
            assert rel_mae(0.5 * (r0[p + 'grad/' + k] + r1[p + 'grad/' + k]), g[p + 'grad/' + k]) < 3e-2, k
            assert rel_mae(r0[p + 'param/' + k], g[p + 'param/' + k]) < 1e-3, k
        # SyncBatchNorm: running statistics of the heads are those of the GLOBAL batch, identical on both ranks
        for k in ('proj.1.running_mean', 'pred.1.running_var'):
            assert np.array_equal(r0[p + 'buf/' + k], r1[p + 'buf/' + k])
            assert rel_mae(r0[p + 'buf/' + k], g[p + 'buf/' + k]) < 1e-4, k
        assert rel_mae(np.concatenate([r0[p + 'eval'], r1[p + 'eval']], 0), g[p + 'depth_eval']) < 1e-4


def test_costdcnet_forward_with_global_batch_statistics():
    """Every BatchNorm of CostDCNet (BatchNorm2d, BatchNorm3d, the sparse encoder's voxel BatchNorm with its per-rank voxel
    counts) sees the global batch: two ranks with one frame each reproduce the depth and the tracked running statistics of
    the reference's batch-2 forward."""
    from tests.util import rel_mae
    r0, r1 = _run('costdcnet')
    g = np.load(os.path.join(GOLD, 'costdcnet_64x64_n2.npz'))
    assert rel_mae(np.concatenate([r0['depth'], r1['depth']], 0), g['s0/depth_train']) < 1e-3
    assert np.array_equal(r0['buf'], r1['buf'])
    assert rel_mae(r0['buf'], g['s0/buf/unet3d.inc.double_conv.0.bn1.running_mean']) < 2e-3


def test_costdcnet_two_ranks_reproduce_the_ddp_adapted_run():
    """Two ranks with one frame each, every BatchNorm adapted and synchronised (forward statistics AND the backward's two sums, the
    sparse encoder's with its per-rank voxel counts): the batch-2 run of the reference with the DDP adapted list."""
    from tests.util import rel_mae
    r0, r1 = _run('costdcnet_ddp')
    g = np.load(os.path.join(GOLD, 'costdcnet_64x64_n2_syncbn.npz'))
    assert rel_mae(np.concatenate([r0['depth'], r1['depth']], 0), g['s0/depth_train']) < 1e-4
    # identical parameters on both ranks after every one of the three shared steps (round-3 verdict: the invariant DDP gives the reference)
    keys = [k for k in r0 if '/param/' in k or k.startswith('param/')]
    assert len(keys) >= 3 * 100
    for k in keys:
        assert np.array_equal(r0[k], r1[k]), k
    names = [k[len('grad/'):] for k in r0 if k.startswith('grad/')]
    assert len(names) == 112
    worst = 0.0
    for k in names:
        assert np.array_equal(r0['param/' + k], r1['param/' + k]), k              # identical update on every rank
        if k.startswith(('proj.1.', 'pred.1.')):
            continue
        # after shared_parameter_step the stored gradient IS the mean over the ranks = the batch-2 gradient (measured 1.2e-4, as the
        # single-process run of tests/test_gpu_costdcnet_syncbn.py: bound 2x)
        e = rel_mae(r0['grad/' + k], g['s0/grad/' + k])
        worst = max(worst, e)
        assert e < 2.5e-4, (k, e)
        assert rel_mae(r0['param/' + k], g['s0/param/' + k]) < 1.5e-5, k
    print('worst gradient deviation', worst)
    # the eval forward normalises with LOCAL batch statistics (SyncBatchNorm only synchronises in training mode): each rank's frame
    # alone differs from the batch-2 eval of the golden run, so only the training-side quantities are compared


# ---- the library-owned RCCL communicator (ptta_rccl_*, ptta_set_stat_sync_rccl, ptta_set_grad_sync_rccl) ----------------------------
# One GPU = one rank: the communicator has ONE rank (RCCL refuses two ranks on one device), so these tests pin the plumbing --
# librccl resolves, the communicator initialises, the collectives are enqueued by the library on the step's stream, the MSG_CHN
# step keeps replaying its hipGraph with the exchange on -- and that a sum / mean over one rank leaves the results unchanged.
# The N > 1 exchange protocol itself is pinned by the two-rank gloo tests above; N > 1 RCCL remains unmeasured on hardware.
def test_rccl_one_rank_msg_chn_step_replays_graph_and_matches():
    from proxytta import distributed as D
    from proxytta import synth
    from tests.util import golden_hp, make_engine
    g = np.load(os.path.join(GOLD, 'msgchn_1layer_32x48_n2.npz'))
    h, w, n, steps = [int(x) for x in g['meta']]
    hp, gain = golden_hp(g)
    comm, world = D.rccl_communicator()
    assert comm and world == 1
    out = []
    for sync in (False, True):
        eng, sd, adapted = make_engine(n, h, w, 'fp32', hp, gain, None)
        if sync:
            eng.enable_rccl_sync()
            eng._chk(eng.lib.ptta_set_graph(eng.handle, 1), 'ptta_set_graph')      # graph replay is NOT refused with the RCCL exchange
        for s in range(steps):
            image, sparse = [torch.from_numpy(x).cuda() for x in synth.synthetic_frame(s, h, w, n)]
            info, depth = D.shared_parameter_step(eng, image, sparse) if sync else eng.step(image, sparse, want_depth=True)
        torch.cuda.synchronize()
        out.append((info.cpu().numpy(), depth.cpu().numpy(), {k: v[0].cpu().numpy().copy() for k, v in adapted.items()},
                    sd['proj.1.running_mean'].cpu().numpy().copy()))
        eng.close()
    (i0, d0, p0, b0), (i1, d1, p1, b1) = out
    # the exchange re-expresses a BatchNorm's partial sums as one double per channel: same statistics to fp32 rounding
    np.testing.assert_allclose(i1, i0, rtol=1e-5)
    assert np.abs(d1 - d0).mean() / np.abs(d0).mean() < 1e-6
    for k in p0:
        assert np.abs(p1[k] - p0[k]).max() < 1e-6, k
    np.testing.assert_allclose(b1, b0, rtol=1e-5, atol=1e-7)
    # the reference's batch-2 vectors, with the exchange on
    ref = g['s%d/depth_train' % (steps - 1)]
    assert np.abs(d1 - ref).mean() / np.abs(ref).mean() < 1e-3


@pytest.mark.parametrize('backbone', ['costdcnet', 'nlspn'])
def test_rccl_one_rank_generic_engine(backbone):
    """The op-list engine (CostDCNet: one collective per BatchNorm incl. the sparse encoder's voxel BatchNorm; NLSPN) with the
    library's RCCL exchange and the in-step gradient all-reduce: unchanged results on one rank."""
    from proxytta import distributed as D
    res = []
    for sync in (False, True):
        if backbone == 'costdcnet':
            from tests.test_gpu_costdcnet import costdc_frame, make_costdc
            n, h, w = 1, 64, 64
            eng, sd, adapted = make_costdc(n, h, w)
            raw, image1, sparse = [torch.from_numpy(x).cuda() for x in costdc_frame(0, h, w, n)]
        else:
            from tests.test_gpu_nlspn import make_nlspn, nlspn_frame
            n, h, w = 1, 32, 64
            eng, sd, adapted = make_nlspn(n, h, w)
            raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
        if sync:
            eng.enable_rccl_sync()
        info, depth = eng.step(image1, sparse, loss_image=raw, want_depth=True)
        torch.cuda.synchronize()
        res.append((info.cpu().numpy(), depth.cpu().numpy(), {k: v[0].cpu().numpy().copy() for k, v in adapted.items()}))
        eng.close()
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-4)
    assert np.abs(res[1][1] - res[0][1]).mean() / np.abs(res[0][1]).mean() < 1e-5
    # Adam's first step is +-lr: compare where the two runs agree on the sign (a sum re-expressed in double moves near-zero gradients)
    tot = same = 0
    for k in res[0][2]:
        d = np.abs(res[1][2][k] - res[0][2][k])
        tot += d.size; same += int((d < 1e-6).sum())
    assert same >= 0.995 * tot, (same, tot)

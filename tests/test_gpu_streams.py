"""Independent frame streams on ONE GPU (SURVEY.md 8e, config 4 applied inside a device): handles are self-contained -- two
engines enqueued concurrently on two HIP streams (hipGraph replay each) must reproduce, bit for bit, what one engine alone
computes on the default stream.  Guards the `msg_chn_two_streams_per_gpu` bench figure against shared scratch / static state."""
import numpy as np
import pytest
import torch

from proxytta import synth
from tests.util import make_engine

pytestmark = pytest.mark.gpu

HP = dict(lr=1e-3, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)


def _frames(n, h, w, count):
    return [[torch.from_numpy(x).cuda() for x in synth.synthetic_frame(40 + i, h, w, n)] for i in range(count)]


@pytest.mark.parametrize('meta', ['1layer', '2layers'])
def test_two_streams_equal_one_stream(meta):
    n, h, w, steps = 1, 64, 96, 4
    frames = _frames(n, h, w, steps)
    # reference run: one engine, default stream
    e0, _, ad0 = make_engine(n, h, w, hp=HP, meta=meta)
    ref_info = []
    for f in frames:
        info, _ = e0.step(*f)
        ref_info.append(info.clone())
    ref_depth = e0.forward_eval(*frames[0]).clone()
    torch.cuda.synchronize()
    # two engines with the same weights and inputs, interleaved on two streams
    engs = [make_engine(n, h, w, hp=HP, meta=meta) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    infos = [[], []]
    for f in frames:
        for k, ((e, _, _), st) in enumerate(zip(engs, streams)):
            with torch.cuda.stream(st):
                info, _ = e.step(*f)
                infos[k].append(info.clone())
    depths = []
    for (e, _, _), st in zip(engs, streams):
        with torch.cuda.stream(st):
            depths.append(e.forward_eval(*frames[0]).clone())
    torch.cuda.synchronize()
    for k in range(2):
        for a, b in zip(infos[k], ref_info):
            assert torch.equal(a, b)
        assert torch.equal(depths[k], ref_depth)
        for name in ad0:
            assert torch.equal(engs[k][2][name][0], ad0[name][0]), name
    for e, _, _ in engs:
        e.close()
    e0.close()

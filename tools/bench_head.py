"""Stage-2 head trainer step time at the KITTI shape (both modes) + per-kernel view when run under rocprofv3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tta-depth-completion_amd')); sys.path.insert(0, ROOT)
import torch
from proxytta import synth
from tests.test_gpu_head_trainer import make_head_engine
H, W = 352, 1216
eng, sd, _, _ = make_head_engine(1, H, W, dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0), 0.999)
frames = [[torch.from_numpy(a).cuda() for a in synth.synthetic_frame(i, H, W, 1)] for i in range(4)]
for reverse in (True, False):
    for i in range(5): eng.head_step(*frames[i % 4], reverse)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(30): loss = eng.head_step(*frames[i % 4], reverse)
    torch.cuda.synchronize()
    print('reverse' if reverse else 'forward', '%.3f ms/step' % (1e3 * (time.perf_counter() - t0) / 30), 'loss', float(loss), flush=True)

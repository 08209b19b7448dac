"""Where does the host -> HBM time go?  (pageable -> pinned memcpy, pinned -> device DMA, blocking .to())"""
import sys
import time
import numpy as np
import torch
H, W = 352, 1216
x = np.random.rand(1, 3, H, W).astype(np.float32)
pin = torch.empty((1, 3, H, W)).pin_memory()
dev = torch.empty((1, 3, H, W), device='cuda')
xt = torch.from_numpy(x)
def t(name, f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); print(name, 1e3 * (time.perf_counter() - t0) / n, 'ms', flush=True)
print('threads', torch.get_num_threads(), flush=True)
which = sys.argv[1:]
if 'a' in which: t('pageable->pinned copy_', lambda: pin.copy_(xt))
if 'b' in which:
    pn = pin.numpy(); t('pageable->pinned np.copyto', lambda: np.copyto(pn, x))
if 'c' in which:
    y = np.empty_like(x); t('pageable->pageable np.copyto', lambda: np.copyto(y, x))
if 'd' in which: t('pinned->dev non_blocking', lambda: dev.copy_(pin, non_blocking=True))
if 'e' in which: t('pageable .to(cuda)', lambda: xt.to('cuda'))
s = torch.cuda.Stream()
def on_stream():
    with torch.cuda.stream(s):
        dev.copy_(pin, non_blocking=True)
if 'f' in which: t('pinned->dev on side stream', on_stream)
ev = torch.cuda.Event()
def ev_sync():
    with torch.cuda.stream(s):
        dev.copy_(pin, non_blocking=True); ev.record(s)
    ev.synchronize()
if 'g' in which: t('pinned->dev + event sync', ev_sync)

"""Which part of the staged loop is slow? toggles: nosync nowait oneslot nohost"""
import sys, time
import numpy as np, torch
H, W = 352, 1216
opts = sys.argv[1:]
slots = 1 if 'oneslot' in opts else 2
host = [torch.empty((1, 4, H, W)).pin_memory() for _ in range(slots)]
dev = [torch.empty((1, 4, H, W), device='cuda') for _ in range(slots)]
src = [torch.rand(1, 4, H, W) for _ in range(4)]
cs = torch.cuda.Stream()
ready = [torch.cuda.Event() for _ in range(slots)]
cons = [torch.cuda.Event() for _ in range(slots)]
work = torch.empty(64 << 20, device='cuda')
def loop(n):
    for i in range(n):
        k = i % slots
        if i >= slots and 'nosync' not in opts: ready[k].synchronize()
        if 'nohost' not in opts: host[k].copy_(src[i % 4])
        with torch.cuda.stream(cs):
            if i >= slots and 'nowait' not in opts: cs.wait_event(cons[k])
            dev[k].copy_(host[k], non_blocking=True)
            ready[k].record(cs)
        torch.cuda.current_stream().wait_event(ready[k])
        if 'kernel' in opts: work.add_(1.0)           # ~0.1 ms of compute consuming the slot
        cons[k].record(torch.cuda.current_stream())
loop(4); torch.cuda.synchronize(); t0 = time.perf_counter(); loop(40); te = time.perf_counter(); torch.cuda.synchronize()
print(opts, 'enqueue %.3f ms/iter, total %.3f ms/iter' % (1e3 * (te - t0) / 40, 1e3 * (time.perf_counter() - t0) / 40), flush=True)

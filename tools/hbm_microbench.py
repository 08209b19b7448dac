#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU for the step's typical tensor sizes (calibration of the roofline's denominator):
fill, copy and read-only reduction of fp32 buffers, timed with events over 50 repetitions after warm-up."""
import torch

def timeit(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3      # us

for mb in (3.4, 13.7, 54.8, 109.6, 219.0, 1024.0):
    n = int(mb * 1e6 / 4)
    x = torch.empty(n, device='cuda'); y = torch.empty(n, device='cuda')
    x.normal_()
    t_fill = timeit(lambda: x.zero_())
    t_copy = timeit(lambda: y.copy_(x))
    t_sum = timeit(lambda: x.sum())
    t_add = timeit(lambda: torch.add(x, 1.0, out=y))
    print('%7.1f MB  fill %6.1f us %5.2f TB/s | copy %6.1f us %5.2f TB/s (r+w) | sum %6.1f us %5.2f TB/s | add %6.1f us %5.2f TB/s (r+w)' % (
        mb, t_fill, mb / t_fill, t_copy, 2 * mb / t_copy, t_sum, mb / t_sum, t_add, 2 * mb / t_add))

"""What the box's memory system gives: write-only (fill), read-only (sum) and copy rates at working sets from 32 MB to 1 GB (HIP events, median of 20)."""
import torch
dev = torch.device('cuda:0')
def timeit(f, n=20):
    for _ in range(3): f()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2] * 1e-3
for mb in (32, 75, 150, 300, 600, 1024):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    tw = timeit(lambda: x.fill_(1.0)); tr = timeit(lambda: x.sum()); tc = timeit(lambda: y.copy_(x))
    print('%5d MB: fill %.2f TB/s (%.1f us)   sum %.2f TB/s (%.1f us)   copy %.2f TB/s read+write (%.1f us)' % (mb, mb * 1.048576e-6 / tw, tw * 1e6, mb * 1.048576e-6 / tr, tr * 1e6, 2 * mb * 1.048576e-6 / tc, tc * 1e6))

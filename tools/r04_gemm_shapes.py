"""Time every GEMM shape of a config-5 training step in isolation (hip.gemm, HIP events, median of 20): python tools/r04_gemm_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ab_opt_amd import hip
dev = torch.device('cuda:0')
# (count per step, B, M, N, K, a transposed view, b transposed view, bias+relu)
SHAPES = [(4, 1, 64, 64, 1048576, 1, 1, 0), (1, 1, 64, 240, 1048576, 1, 1, 0), (1, 1, 64, 26, 1048576, 1, 1, 0),
          (6, 1, 4096, 2016, 128, 0, 0, 0), (6, 1, 4096, 128, 2016, 0, 1, 0), (6, 1, 2016, 128, 4096, 1, 1, 0),
          (6, 1, 4096, 1824, 128, 0, 1, 0), (6, 1, 128, 1824, 4096, 1, 1, 0),
          (6, 192, 256, 57, 256, 0, 1, 0), (6, 192, 256, 57, 256, 1, 1, 0), (6, 192, 256, 256, 56, 0, 0, 0), (6, 192, 256, 56, 256, 1, 1, 0),
          (1, 1, 4096, 256, 1413, 0, 0, 1), (1, 1, 4096, 1413, 256, 0, 1, 0), (1, 1, 256, 1413, 4096, 1, 1, 0),
          (6, 3, 128, 128, 4096, 1, 1, 0), (6, 1, 4096, 128, 128, 0, 1, 0), (6, 1, 128, 128, 4096, 1, 1, 0), (7, 1, 4096, 128, 128, 0, 0, 1),
          (3, 1, 4096, 128, 131, 0, 0, 1), (3, 1, 4096, 131, 128, 0, 1, 0), (3, 1, 128, 131, 4096, 1, 1, 0),
          (2, 1, 4096, 128, 256, 0, 0, 1), (2, 1, 4096, 256, 128, 0, 1, 0), (2, 1, 128, 256, 4096, 1, 1, 0)]
def timeit(f, n=20):
    for _ in range(3): f()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2] * 1e3
tot = 0.0
for cnt, B, M, N, K, at, bt, br in SHAPES:
    sh = lambda r, c, tr: (torch.randn(B, c, r, device=dev).transpose(1, 2) if tr else torch.randn(B, r, c, device=dev))
    a, b = sh(M, K, at), sh(N, K, bt)
    if B == 1: a, b = a[0], b[0]
    bias = torch.randn(N, device=dev) if br else None
    us = timeit(lambda: hip.gemm(a, b, bias=bias, relu=bool(br)))
    gf = 2e-9 * B * M * N * K
    byts = 4 * B * (M * K + N * K + M * N)
    tot += cnt * us
    print('%2d x B=%-3d M=%-5d N=%-5d K=%-7d %s%s%s  %7.1f us  %6.1f TF/s  %5.2f TB/s   step share %6.1f us' % (cnt, B, M, N, K, 'T' if at else 'n', 'T' if bt else 'n', ' br' if br else '   ', us, gf / us * 1e-3 * 1e3 / 1e3, byts / us * 1e-6, cnt * us))
print('sum over the step: %.2f ms' % (tot * 1e-3))

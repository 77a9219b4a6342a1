"""One GEMM shape in a loop (for rocprofv3 / PMC passes and same-box timing): python tools/r05/gemm_one.py B M N K at bt bias_relu [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ab_opt_amd import hip
B, M, N, K, at, bt, br = [int(x) for x in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 50
dev = torch.device('cuda:0')
sh = lambda r, c, tr: (torch.randn(B, c, r, device=dev).transpose(1, 2) if tr else torch.randn(B, r, c, device=dev))
a, b = sh(M, K, at), sh(N, K, bt)
if B == 1: a, b = a[0], b[0]
bias = torch.randn(N, device=dev) if br else None
for _ in range(3): hip.gemm(a, b, bias=bias, relu=bool(br))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): hip.gemm(a, b, bias=bias, relu=bool(br))
e1.record(); torch.cuda.synchronize()
print('B=%d M=%d N=%d K=%d %s%s: %.1f us per call (back to back), %.1f TF/s' % (B, M, N, K, 'T' if at else 'n', 'T' if bt else 'n', e0.elapsed_time(e1) / iters * 1e3, 2e-12 * B * M * N * K / (e0.elapsed_time(e1) / iters * 1e-3)))

"""Time model.encode() (residue + pair embedding) on the GPU: python tools/bench_encode.py [N] [L] [iters]."""
import sys, time
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from ab_opt_amd.utils.synth import build_model
from ab_opt_amd.utils.synth import make_batch, LAYOUT_256, LAYOUT_128

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(N, LAYOUT_256 if L == 256 else LAYOUT_128).items()}
with torch.no_grad():
    for _ in range(2):
        out = model.encode(batch, True, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = model.encode(batch, True, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
print(f'encode N={N} L={L}: {dt * 1e3:.2f} ms  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')

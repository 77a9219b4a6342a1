#!/usr/bin/env python
"""Micro-benchmark of the IPA-core kernel alone (through abopt_ga_block_forward) at the bench shape.
    python tools/bench_ipa.py [N] [L] [iters]
Prints the HIP-event average of the kernel and the implied algorithmic GB/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from ab_opt_amd import hip
from ab_opt_amd.modules import GABlock
from ab_opt_amd.utils import synth
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device('cuda:0')
blk = synth.fill_module_(GABlock(128, 64), seed=1).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
v = torch.randn(N, L, 3, device=dev, generator=g)
R = hip.so3_exp(v)
t = torch.randn(N, L, 3, device=dev, generator=g) * 2
x = torch.randn(N, L, 128, device=dev, generator=g)
z = torch.randn(N, L, L, 64, device=dev, generator=g)
mask = torch.ones(N, L, dtype=torch.bool, device=dev)
for _ in range(3):
    out = blk(R, t, x, z, mask)
torch.cuda.synchronize()
hip.prof_enable(True)
t0 = time.perf_counter()
for _ in range(iters):
    out = blk(R, t, x, z, mask)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
n, ms = hip.prof_collect()
hip.prof_enable(False)
per = ms / n
gbs = bench.ipa_algorithmic_bytes(N, L) / (per * 1e-3) / 1e9
print(f'N={N} L={L}: ipa_core {per*1e3:.1f} us/launch = {gbs:.0f} GB/s algorithmic ({gbs/80:.1f}% of 8 TB/s); whole GABlock {dt*1e3:.3f} ms; checksum {out.double().sum().item():.6f}')

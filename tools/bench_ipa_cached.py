#!/usr/bin/env python
"""IPA-core timing with the per-call pair-bias cache (the sampler's configuration): K steps of FullDPM._run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from ab_opt_amd import hip
args = [a for a in sys.argv[1:] if not a.startswith('--')]
N = int(args[0]) if len(args) > 0 else 32
L = int(args[1]) if len(args) > 1 else 256
K = int(args[2]) if len(args) > 2 else 5
dev = torch.device('cuda:0')
dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, N, L, 100, seed=1)
if '--shared' in sys.argv:        # one complex, N samples: context shared inside the kernels
    res_feat, pair_feat = res_feat[:1].contiguous(), pair_feat[:1].contiguous()
run = lambda n: dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, True, True, None, 1, 0, False, stop_after=n)
run(2); torch.cuda.synchronize()
hip.prof_enable(True)
run(K); torch.cuda.synchronize()
n, ms = hip.prof_collect()
print(f'cached path: ipa_core {ms / n * 1e3:.1f} us/launch = {bench.ipa_algorithmic_bytes(N, L) / (ms / n * 1e-3) / 1e9 / 80:.1f}% of 8 TB/s (SURVEY 8d bytes) over {n} launches')

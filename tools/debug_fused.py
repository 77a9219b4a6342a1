#!/usr/bin/env python
"""Fused core + tail against the two-launch form: max |difference| of the EpsilonNet outputs (0 expected)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from ab_opt_amd import hip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, N, L, 100, seed=1)
def run():
    out = dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, True, True, None, 1, 0, False, stop_after=1, graph=False)
    torch.cuda.synchronize()
    return [o[99].clone() for o in out[:3]]
os.environ['ABOPT_FUSE_TAIL'] = '0'; a = run()
os.environ['ABOPT_FUSE_TAIL'] = '1'; b = run()
for x, y, n in zip(a, b, 'vps'):
    print(n, 'max abs diff', (x.double() - y.double()).abs().max().item(), 'equal', torch.equal(x, y))

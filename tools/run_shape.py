"""Run K sampling steps at one shape (for rocprofv3 --kernel-trace --stats and for timing shapes other than the bench's):
    python tools/run_shape.py --n 1000 --l 48 --shared --flavour abdock --steps 10 [--graph]
prints sample-steps/s and ms per step (eager unless --graph)."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=1000)
ap.add_argument('--l', type=int, default=48)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--shared', action='store_true')
ap.add_argument('--flavour', default='abdock')
ap.add_argument('--graph', action='store_true')
ap.add_argument('--repeats', type=int, default=3)
a = ap.parse_args()
dev = torch.device('cuda:0')
abd = a.flavour == 'abdesign'
cdrs = None if a.l in (128, 256) else [(8, min(26, a.l))]
dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, a.n, a.l, 100, seed=5, abdesign=abd, shared_context=a.shared, cdrs=cdrs)
run = lambda n, g: dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, abd, True, None, 99, 0, False, stop_after=n, graph=g)
run(2, False)
if a.graph:
    run(a.steps, True)
torch.cuda.synchronize()
best = 1e9
for _ in range(a.repeats):
    t0 = time.perf_counter()
    run(a.steps, a.graph)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print('shape N=%d L=%d shared=%s %s graph=%s: %.1f sample-steps/s, %.4f ms per step' % (a.n, a.l, a.shared, a.flavour, a.graph, a.n * a.steps / best, best / a.steps * 1e3))

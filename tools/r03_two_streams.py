"""Experiment: two half batches replayed as two hipGraphs on two streams against one full batch on one stream.
The core is HBM-bound, the tail / projections are issue-bound: do they overlap when the halves run one phase apart?
    python tools/r03_two_streams.py [--n 32] [--steps 20]"""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=32)
ap.add_argument('--l', type=int, default=256)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--parts', type=int, default=2)
ap.add_argument('--offset', type=int, default=0, help='spin cycles (torch.cuda._sleep) ahead of part i: i * offset')
a = ap.parse_args()
dev = torch.device('cuda:0')


def make(n, seed):
    dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, n, a.l, 100, seed=seed, abdesign=True)
    return lambda g: dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, True, True, None, 99, 0, False, stop_after=a.steps, graph=g)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


full = make(a.n, 5)
full(False); full(True); torch.cuda.synchronize()
t_full = timed(lambda: full(True))
parts = [make(a.n // a.parts, 5 + i) for i in range(a.parts)]
for p in parts:
    p(False); p(True)
torch.cuda.synchronize()
t_seq = timed(lambda: [p(True) for p in parts])
streams = [torch.cuda.Stream() for _ in parts]


def conc():
    for i, (s, p) in enumerate(zip(streams, parts)):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            if a.offset and i:
                torch.cuda._sleep(i * a.offset)
            p(True)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)


ref = [[t.clone() for t in p(True)[:3]] for p in parts]
torch.cuda.synchronize()
t_conc = timed(conc)
outs = []
for s, p in zip(streams, parts):
    with torch.cuda.stream(s):
        outs.append([t.clone() for t in p(True)[:3]])
torch.cuda.synchronize()
same = all(torch.equal(x, y) for r, o in zip(ref, outs) for x, y in zip(r, o))
f = lambda t: '%.4f ms per step (%.0f sample-steps/s)' % (t / a.steps * 1e3, a.n * a.steps / t)
print('N=%d L=%d, %d steps: one graph %s | %d parts one after the other %s | %d parts on %d streams %s | concurrent results identical: %s'
      % (a.n, a.l, a.steps, f(t_full), a.parts, f(t_seq), a.parts, a.parts, f(t_conc), same))

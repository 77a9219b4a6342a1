"""Does a captured loop KEEP its speed (the state lives with its buffers' placement) or do all captures of a process move together (a state of the box)?
Six graphs of the same loop alive at once (K = 20 .. 25 steps: distinct cache keys, each with its own pool), timed round-robin for several passes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
dev = torch.device('cuda:0')
N, L = 32, 256
dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, N, L, 100, seed=5)
dpm.max_graphs = 16
run = lambda K, g: dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, True, True, None, 99, 0, False, stop_after=K, graph=g)
run(20, False)
Ks = [20, 21, 22, 23, 24, 25]
for K in Ks:
    run(K, True)
torch.cuda.synchronize()
for p in range(6):
    row = []
    for K in Ks:
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); run(K, True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e3)
        row.append(min(ts))
    print('pass %d: ' % p + '  '.join('K=%d %.4f' % (K, t) for K, t in zip(Ks, row)), flush=True)

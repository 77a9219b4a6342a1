"""Time series of the replayed 20-step loop in ONE process: back-to-back replays, then replays with idle gaps (is the slow state a matter of what preceded?)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
dev = torch.device('cuda:0')
N, L, K = 32, 256, 20
dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, N, L, 100, seed=5)
run = lambda g: dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, True, True, None, 99, 0, False, stop_after=K, graph=g)
run(False); run(True); torch.cuda.synchronize()
def one():
    t0 = time.perf_counter(); run(True); torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
t00 = time.perf_counter()
ser = [one() for _ in range(240)]
print('back to back, 240 replays over %.1f s:' % (time.perf_counter() - t00))
for i in range(0, 240, 20): print('  ' + ' '.join('%.3f' % x for x in ser[i:i + 20]))
for gap in (0.05, 0.5, 2.0):
    ser = []
    for _ in range(8):
        time.sleep(gap); ser.append(one()); ser.append(one())
    print('idle %.2f s before each pair: ' % gap + ' '.join('%.3f' % x for x in ser))
ser = [one() for _ in range(40)]
print('back to back again: ' + ' '.join('%.3f' % x for x in ser))

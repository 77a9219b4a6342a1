"""Is the process-to-process spread of the replayed step (1.12 / 1.17 / 1.25 ms for one binary on one box) a matter of where the big buffers land?
Re-capture the same loop several times in ONE process, with and without a dummy allocation that shifts the addresses of everything behind it."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
dev = torch.device('cuda:0')
N, L, K = 32, 256, 20
dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, N, L, 100, seed=5)
run = lambda g: dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, True, True, None, 99, 0, False, stop_after=K, graph=g)
from ab_opt_amd import hip
rec = {}
_pbc, _pt = hip.pair_bias_cache, hip.pair_terms
def pbc_(*a, **k):
    t = _pbc(*a, **k); rec['cache'] = t.data_ptr(); return t
def pt_(*a, **k):
    t = _pt(*a, **k); rec['terms'] = t.data_ptr(); return t
hip.pair_bias_cache, hip.pair_terms = pbc_, pt_
run(False)
keep = []
for trial in range(10):
    dpm.clear_graphs()
    torch.cuda.empty_cache()
    if trial >= 4:
        keep.append(torch.empty((37 + 61 * trial) * 1024 * 1024 // 4, device=dev))      # odd sizes: shifts what comes next
    run(True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e3)
    g = list(dpm._graphs.values())[-1]
    ws = [b.data_ptr() for b in g.keep]
    f = lambda a: '0x%x (mod 2^30: %4d MB)' % (a, (a >> 20) & 1023)
    print('trial %d: %.4f %.4f %.4f ms per step | pair_feat %s cache %s terms %s ws %s' % (trial, *ts, f(g.pair_feat.data_ptr()), f(rec['cache']), f(rec['terms']), ' '.join(f(a) for a in ws)), flush=True)

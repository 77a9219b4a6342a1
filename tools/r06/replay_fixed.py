"""What a replay of the K-step loop costs beyond its K steps (bench.py times K = 20 steps per region; loop100 times 100): timed regions like bench.py's
(synchronize on both sides) for K = 1, 2, 5, 10, 20, 50, 100, a straight-line fit, and the pieces of the fixed part: the input copies of _LoopGraph.replay, the
graph launch alone, the one-off builds.   python tools/r06/replay_fixed.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from ab_opt_amd import hip
    dev = torch.device('cuda:0')
    N, L, T = 32, 256, 100
    dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, N, L, T, seed=2022)
    run = lambda n: dpm._run(state, T, res_feat, pair_feat, gen, mres, True, True, True, None, 1234, 0, False, stop_after=n, graph=True)
    rows = []
    for K in (1, 2, 5, 10, 20, 50, 100):
        run(K); run(K)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(K)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ms = sorted(ts)[3] * 1e3
        # the graph alone (no input copies), and with the device idle for 20 ms before
        g = [v for k, v in dpm._graphs.items() if k[4] == K][0]
        tg = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.graph.replay()
            torch.cuda.synchronize()
            tg.append(time.perf_counter() - t0)
        # back to back: 3 replays in flight, per replay
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            g.graph.replay()
        torch.cuda.synchronize()
        bb = (time.perf_counter() - t0) / 3 * 1e3
        rows.append((K, ms, sorted(tg)[3] * 1e3, bb))
        print('K = %3d: region %.3f ms (%.4f per step) | graph.replay() alone %.3f ms | three replays back to back %.3f ms each' % (K, ms, ms / K, sorted(tg)[3] * 1e3, bb), flush=True)
    import numpy as np
    k = np.array([r[0] for r in rows], float)
    for name, col in (('region', 1), ('graph alone', 2), ('back to back', 3)):
        y = np.array([r[col] for r in rows])
        c, f = np.polyfit(k, y, 1)
        print('%-13s = %.4f ms x K + %.3f ms' % (name, c, f))
    # pieces
    def t_of(fn, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    arr = dpm.eps_net.encoder.packed_array()
    print('pair-bias cache build %.3f ms, pair terms %.3f ms' % (t_of(lambda: hip.pair_bias_cache(arr, 6, pair_feat)), t_of(lambda: hip.pair_terms(pair_feat))))
    g = [v for k, v in dpm._graphs.items() if k[4] == 20][0]
    print('input copies of replay(): state %.3f ms, res_feat %.3f, masks %.3f, seed H2D %.3f' % (
        t_of(lambda: [d.copy_(s_) for d, s_ in zip(g.state, state)]), t_of(lambda: g.res_feat.copy_(res_feat)),
        t_of(lambda: (g.mask_generate.copy_(gen), g.mask_res.copy_(mres))), t_of(lambda: g.seed_dev.copy_(torch.tensor([1, 2], dtype=torch.int64)))))
    print('empty synchronize %.4f ms' % t_of(lambda: torch.cuda.synchronize(), 50))


if __name__ == '__main__':
    main()

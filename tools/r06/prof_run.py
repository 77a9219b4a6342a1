"""cProfile of the host side of one replayed K = 20 region (dpm._run): where the 0.5 ms per replay outside the graph go -- the packed-weight fingerprint check (170 parameters),
torch.cuda.memory_stats of _bias_cache_fits, the input copies.   python tools/r06/prof_run.py"""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device('cuda:0')
dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, 32, 256, 100, seed=2022)
run = lambda n: dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, True, True, None, 1234, 0, False, stop_after=n, graph=True)
run(20); run(20); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    run(20)
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(22)

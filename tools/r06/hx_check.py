"""Developer check of a -DC32_HX variant: K sampling steps on a fixed seed; saves the last positions so that two libraries can be compared.
    ABOPT_LIB_PATH=... [ABOPT_DEV_ZTERMS=1] python tools/r06/hx_check.py <tag> [<other tag to compare with>]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from ab_opt_amd import hip
dev = torch.device('cuda:0')
N, L = 32, 256
dpm, state, res_feat, pair_feat, gen, mres = bench.build_workload(dev, N, L, 100, seed=5)
beta = dpm.trans_pos.var_sched.betas[100].expand([N]).contiguous()
tv, tp, ts, _, _ = dpm._run(state, 100, res_feat, pair_feat, gen, mres, True, True, True, None, 99, 0, False, stop_after=2, graph=False)
torch.cuda.synchronize()
out = dict(p99=tp[99].cpu(), v99=tv[99].cpu(), p98=tp[98].cpu())
os.makedirs('gpurun_out/hx', exist_ok=True)
torch.save(out, f'gpurun_out/hx/{sys.argv[1]}.pt')
print(sys.argv[1], 'finite', bool(torch.isfinite(out['p98']).all()), 'p99 abs max', float(out['p99'].abs().max()))
if len(sys.argv) > 2:
    ref = torch.load(f'gpurun_out/hx/{sys.argv[2]}.pt')
    for k in out:
        print(f'  {k}: max abs diff vs {sys.argv[2]}: {float((out[k] - ref[k]).abs().max()):.3e}')

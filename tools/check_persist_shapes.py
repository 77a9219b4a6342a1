"""Developer check: the persistent IPA core against the one-block kernel at long / ragged shapes (bit-identical expected)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_hip_parity import standalone_abdesign_dpm, _rand_eps_inputs, DEV
from ab_opt_amd import hip
for N, L in ((8, 1024), (5, 1000), (3, 2085)):
    d = standalone_abdesign_dpm(100, 2).to(DEV)
    lens = [L - (13 * i) % 97 for i in range(N)]
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lens, 7000 + N, [(5, 14), (22, 30)])
    beta = d.trans_pos.var_sched.betas[37].expand([N]).contiguous()
    arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
    big = hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=hip.pair_bias_cache(arr, 6, pf))
    sl = slice(N - 2, N)
    c = lambda a: a[sl].contiguous()
    small = hip.eps_net_forward(ew, c(v), c(p), c(s), c(rf), c(pf), c(beta), c(gen), c(mres), d.abdock, d.num_bins, False,
                                pair_bias_cache=hip.pair_bias_cache(arr, 6, c(pf)))
    ok = all(torch.isfinite(big[k]).all().item() and torch.equal(big[k][sl], small[k]) for k in ('R_next', 'eps_pos', 'c'))
    print(N, L, 'blocks', N * ((L + 15) // 16), 'bit-identical' if ok else 'MISMATCH')

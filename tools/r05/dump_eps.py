"""EpsilonNet outputs of the bench-geometry test inputs (N = 32, L = 256, ragged) with the library named by ABOPT_LIB_PATH -> a .pt file; with two
files given, compares them and each against the oracle in float64 on four samples.  python tools/r05/dump_eps.py out.pt | python tools/r05/dump_eps.py a.pt b.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for k in a:
        if k.startswith('ref_'): continue
        d = (a[k].double() - b[k].double()).abs()
        print('%-10s equal %s  max |a-b| %.3e' % (k, torch.equal(a[k], b[k]), d.max().item()), end='')
        if 'ref_' + k in a:
            r = a['ref_' + k].double(); ix = a['ids']
            print('   vs float64 oracle (4 samples): a %.3e  b %.3e' % ((a[k][ix].double() - r).abs().max().item(), (b[k][ix].double() - r).abs().max().item()), end='')
        print()
    sys.exit()
from test_hip_parity import _rand_eps_inputs, standalone_abdesign_dpm, DEV
from ab_opt_amd import hip
from oracle import dpm
N, L, T, t = 32, 256, 100, 63
d_cpu = standalone_abdesign_dpm(T, 2); d = standalone_abdesign_dpm(T, 2).to(DEV)
lengths = ([256] * 5 + [243, 256, 200]) * (N // 8)
v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lengths, 2000 + N, [(25, 33), (51, 57), (94, 106), (133, 144), (159, 166), (198, 207)])
beta = d.trans_pos.var_sched.betas[t].expand([N]).contiguous()
pbc = hip.pair_bias_cache(d.eps_net.encoder.packed_array(), 6, pf)
net = hip.eps_net_forward(d.eps_net.packed(), v, p, s, rf, pf, beta, gen, mres, False, 0, False, pair_bias_cache=pbc)
out = {k: a.cpu().clone() for k, a in net.items() if a is not None}
ids = [0, 5, N // 2 + 1, N - 1]
out['ids'] = torch.tensor(ids)
if os.environ.get('WITH_ORACLE'):
    sd = {k: x.cpu().double() for k, x in d_cpu.state_dict().items()}
    inv = d_cpu.trans_rot.angular_distrib_inv
    den = dpm.Denoiser(sd, num_steps=T, variant='abdesign', obj='pred_x0', mode='mm', pre='', tables=(None, dict(stddevs=inv.stddevs, approx_flag=inv.approx_flag, X=inv.X, Y=None)))
    c = lambda a: a[out['ids'].to(a.device)].cpu()
    ref = den._eps(c(v).double(), c(p).double(), c(s), c(rf).double(), c(pf).double(), c(beta).double(), c(gen), c(mres), False)
    out['ref_R_next'], out['ref_eps_pos'], out['ref_c'] = ref[1], ref[2], ref[3]
torch.save(out, sys.argv[1])
print('saved', sys.argv[1], os.environ.get('ABOPT_LIB_PATH', 'product'))

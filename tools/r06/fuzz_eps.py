"""Shape fuzz of EpsilonNet through the sampler's launch path against the oracle: random batch sizes, lengths (ragged, incl. L not a multiple of 16 / 32),
flavours and pair-feature sharing; the oracle runs on two samples per case.  Round 6: wherever the launch takes the 32-row kernels the pair terms
of DESIGN.md section 3.1b are handed over, as the sampler does (and half of the cases are drawn large enough to take them).   python tools/r06/fuzz_eps.py [cases] [seed]"""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_hip_parity import _rand_eps_inputs, standalone_abdesign_dpm, DEV
from conftest import build_model
from ab_opt_amd import hip
from oracle import dpm
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
models = {}
def model(flavour):
    if flavour not in models:
        if flavour == 'abdesign':
            models[flavour] = (standalone_abdesign_dpm(100, 2), standalone_abdesign_dpm(100, 2).to(DEV))
        else:
            models[flavour] = (build_model(100, 2).diffusion, build_model(100, 2, device=DEV).diffusion)
    return models[flavour]
worst = {}
t00 = time.time()
for case in range(ncases):
    flavour = rnd.choice(['abdesign', 'abdock'])
    L = rnd.choice([rnd.randint(1, 40), rnd.randint(41, 130), rnd.randint(131, 300), 256, 48, 64, 200])
    group = rnd.choice([0, 0, 1, 4])                       # 0 distinct pair features | 1 one complex for all | g samples per complex
    maxn = max(1, min(72, 40000 // (L * L // 64 + 1)))
    N = rnd.randint(1, maxn)
    if case % 2 == 0 and L >= 33:                            # a shape that fills the chip: the 32-row kernels and the term path
        N = max(N, min(maxn, -(-256 // ((L + 31) // 32)) + rnd.randint(0, 6)))
    if group > 1: N = max(group, N // group * group)
    lengths = [max(1, L - rnd.randint(0, L // 3)) if rnd.random() < 0.5 else L for _ in range(N)]
    d_cpu, d = model(flavour)
    a, b = sorted(rnd.sample(range(L + 1), 2))
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lengths, 5000 + case, [(a, b)])
    Nc = N if group == 0 else (1 if group == 1 else N // group)
    if group > 1:                                           # samples of a complex share its residue mask
        mres = mres[::group].repeat_interleave(group, 0).contiguous(); gen = gen & mres
    pfc = pf[:Nc].contiguous()
    t = rnd.choice([100, 63, 21, 2, 1])
    beta = d.trans_pos.var_sched.betas[t].expand([N]).contiguous()
    use_cache = rnd.random() < 0.8 or group != 0
    pbc = hip.pair_bias_cache(d.eps_net.encoder.packed_array(), 6, pfc) if use_cache else None
    terms = hip.pair_terms(pfc) if (use_cache and hip.pair_terms_used(N, L, group)) else None
    net = hip.eps_net_forward(d.eps_net.packed(), v, p, s, rf, pfc, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc, pair_feat_shared=group, pair_terms=terms)
    torch.cuda.synchronize()
    ids = sorted(set([0, N - 1]))
    sd = {k: x.cpu() for k, x in d_cpu.state_dict().items()}
    inv = d_cpu.trans_rot.angular_distrib_inv
    den = dpm.Denoiser(sd, num_steps=100, variant=flavour, obj='pred_x0', mode='mm', pre='' , tables=(None, dict(stddevs=inv.stddevs, approx_flag=inv.approx_flag, X=inv.X, Y=None)))
    errs = {}
    for n in ids:
        c = lambda x_: x_[n:n + 1].cpu()
        cpf = pfc[(n if group == 0 else (0 if group == 1 else n // group))][None].cpu()
        ref = den._eps(c(v), c(p), c(s), c(rf), cpf, c(beta), c(gen), c(mres), False)
        for name, k in (('R_next', 1), ('eps_pos', 2), ('c', 3)):
            errs[name] = max(errs.get(name, 0.0), (c(net[name]) - ref[k]).abs().max().item())
        if d.abdock:
            errs['prmsd_logits'] = max(errs.get('prmsd_logits', 0.0), (c(net['prmsd_logits']) - ref[4]).abs().max().item())
    finite = all(torch.isfinite(x).all().item() for x in net.values() if x is not None)
    bad = (not finite) or errs['R_next'] > 3e-5 or errs['eps_pos'] > 3e-5 or errs['c'] > 1e-5 or errs.get('prmsd_logits', 0) > 3e-5
    for k_, e_ in errs.items(): worst[k_] = max(worst.get(k_, 0.0), e_)
    print('%s case %2d: %-8s N=%-3d L=%-3d group=%d cache=%d terms=%d t=%-3d ragged=%d  errors %s' % ('FAIL' if bad else 'ok  ', case, flavour, N, L, group, int(use_cache), int(terms is not None), t, int(min(lengths) < L),
          {k_: '%.1e' % e_ for k_, e_ in errs.items()}), flush=True)
print('worst over %d cases: %s   (%.0f s)' % (ncases, {k_: '%.2e' % e_ for k_, e_ in worst.items()}, time.time() - t00))

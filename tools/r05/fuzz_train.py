"""Training path on odd shapes (L not a multiple of 16, ragged, single sample, tiny L) against the oracle under CPU autograd in float64: losses and the
gradient of every denoiser parameter, res_feat and pair_feat (AbDesign flavour FullDPM.forward, injected noise).  python tools/r05/fuzz_train.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_hip_parity import standalone_abdesign_dpm, DEV
from ab_opt_amd.utils import synth
from oracle import dpm as odpm
dev = lambda a: a.to(DEV)
torch.set_num_threads(min(os.cpu_count() or 1, 32))
shapes = [(3, 37, [37, 20, 31], [(5, 12), (20, 30)]), (2, 70, [70, 55], [(10, 25)]), (5, 16, [16, 9, 16, 1, 12], [(2, 9)]), (1, 129, [129], [(30, 60), (100, 120)]),
          (4, 100, [100, 100, 64, 97], [(0, 10), (90, 100)]), (2, 33, [33, 17], [(0, 33)])]
d = standalone_abdesign_dpm(100, 2).to(DEV).train()
f = d.trans_rot.angular_distrib_fwd
bad = 0
for ci, (N, L, lens, gr_) in enumerate(shapes):
    d.zero_grad(set_to_none=True)
    v, p, s, res_feat, pair_feat, _, gen, mres = synth.eps_inputs(N, L, lens, gr_, salt=1300 + ci)
    s = s.clamp(max=19)
    g = torch.Generator().manual_seed(100 + ci)
    t = torch.randint(1, 100, (N,), generator=g)
    noise = dict(axis=torch.randn(N, L, 3, generator=g), bin=torch.randint(0, 8191, (N, L), generator=g), ubin=torch.rand(N, L, generator=g),
                 gauss=torch.randn(N, L, generator=g), pos=torch.randn(N, L, 3, generator=g), s_noisy=torch.randint(0, 20, (N, L), generator=g))
    rf = dev(res_feat).clone().requires_grad_(True); pf = dev(pair_feat).clone().requires_grad_(True)
    loss = d(dev(v), dev(p) * 10, dev(s), rf, pf, dev(gen), dev(mres), True, True, t=dev(t), noise={k: dev(a) for k, a in noise.items()})
    sum(loss.values()).backward()
    got = {n: q.grad.detach().cpu() for n, q in d.named_parameters() if q.grad is not None}
    got['res_feat'], got['pair_feat'] = rf.grad.cpu(), pf.grad.cpu()
    got_loss = {k: a.item() for k, a in loss.items()}
    ref = {}
    for dt in (torch.float64, torch.float32):
        cv = lambda a: (a.detach().cpu().to(dt) if a.is_floating_point() else a.detach().cpu())
        sd = {k: cv(a).requires_grad_(a.is_floating_point() and 'eps_net' in k) for k, a in d.state_dict().items()}
        den = odpm.Denoiser(sd, num_steps=100, variant='abdesign', pre='', tables=(None, None), mode='mm')
        den.sch = {k: cv(a) for k, a in den.sch.items()}
        den.tab_fwd = dict(stddevs=cv(f.stddevs), approx_flag=f.approx_flag.cpu(), X=cv(f.X), Y=cv(f.Y))
        r_, p_ = cv(res_feat).requires_grad_(True), cv(pair_feat).requires_grad_(True)
        nz = dict(rot=dict(axis=cv(noise['axis']), bin=noise['bin'], ubin=cv(noise['ubin']), gauss=cv(noise['gauss'])), pos=cv(noise['pos']), s_noisy=noise['s_noisy'])
        with torch.enable_grad():
            lo = den.loss(cv(v), cv(p) * 10, s, r_, p_, gen, mres, t, nz)
            sum(lo.values()).backward()
        gg = {k: a.grad for k, a in sd.items() if a.grad is not None}
        gg['res_feat'], gg['pair_feat'] = r_.grad, p_.grad
        ref[dt] = (lo, gg)
    lo, g64 = ref[torch.float64]
    g32 = ref[torch.float32][1]
    lerr = max(abs(got_loss[k] - lo[k].item()) / max(1.0, abs(lo[k].item())) for k in got_loss)
    rows = []
    for n in sorted(got):
        mx = g64[n].abs().max().item() + 1e-30
        e = (got[n].double() - g64[n]).abs() / mx
        e = e.reshape(-1, e.shape[-1]).max(1).values if e.dim() > 1 else e
        srt = torch.sort(e.flatten(), descending=True).values
        e32 = ((g32[n].double() - g64[n]).abs() / mx).max().item()
        rows.append((srt[2].item() if srt.numel() > 2 else srt[0].item(), srt[0].item(), n, e32))
    w = sorted(rows)[-1]; wa = max(rows, key=lambda r: r[1])
    ok = lerr <= 2e-5 and all((a <= 3e-4 and b <= 5e-3) or b <= 3 * o for a, b, _, o in rows) and set(got) == set(g64)
    bad += not ok
    print('%s N=%d L=%d lens=%s t=%s: loss err %.1e | worst tensor (all rows but two) %.1e %s (float32 oracle: %.1e) | worst single row %.1e %s (float32 oracle: %.1e) | %d tensors' %
          ('ok  ' if ok else 'FAIL', N, L, lens, t.tolist(), lerr, w[0], w[2], w[3], wa[1], wa[2], wa[3], len(rows)), flush=True)
print('failures:', bad)

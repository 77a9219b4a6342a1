import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import cases
from conftest import load_golden
from test_oracle_golden import standalone_abdesign_dpm
from ab_opt_amd import training, hip
DEV = torch.device('cuda:0')
dev = lambda x: x.to(DEV)
g = load_golden('training_abdesign')
d = standalone_abdesign_dpm(100, 2).to(DEV).train()
N, L = 2, 48
v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
s = s.clamp(max=19)
noise = dict(axis=dev(g['rot_axis']), bin=dev(g['rot_bin']), ubin=dev(g['rot_ubin']), gauss=dev(g['rot_gauss']), pos=dev(g['pos']), s_noisy=dev(g['s_noisy']))
for native in (True, False):
    training.NATIVE_IPA = native
    loss = d(dev(v), dev(p) * 10, dev(s), dev(res_feat), dev(pair_feat), dev(gen), dev(mres), True, True, t=torch.tensor([37, 80], device=DEV), noise=noise)
    print('native', native, {k: round(x.item(), 6) for k, x in loss.items()}, 'golden', {k[5:]: round(g[k].item(), 6) for k in g if k.startswith('loss_')})
# noising vs oracle
from oracle import dpm as odpm
sd = {k: x.cpu() for k, x in d.state_dict().items()}
den = odpm.Denoiser(sd, num_steps=100, variant='abdesign', pre='', tables=(None, None))
f = d.trans_rot.angular_distrib_fwd
den.tab_fwd = dict(stddevs=f.stddevs.cpu(), approx_flag=f.approx_flag.cpu(), X=f.X.cpu(), Y=f.Y.cpu())
t = torch.tensor([37, 80])
vn_ref = odpm.rot_add_noise(den.sch, den.tab_fwd, v, gen, t, dict(axis=g['rot_axis'], bin=g['rot_bin'], ubin=g['rot_ubin'], gauss=g['rot_gauss']), True)
h = d._sched_host()
vn, pn, sn, eps = hip.add_noise(dev(t), d.trans_pos.var_sched.alpha_bars, f, noise, 0, 0, dev(v), dev(p) * 10, dev(s), dev(gen), h['scale'], h['mean'], grad_mode=True, want_eps=True)
from oracle import geometry as G
print('R_noisy err', (G.so3_exp(vn.cpu()) - G.so3_exp(vn_ref)).abs().max().item(), 'p err', (pn.cpu() / 10 - odpm.pos_add_noise(den.sch, p, gen, t, g['pos'])).abs().max().item())
# network outputs: device training statement vs oracle on the SAME noised inputs
p_n = (pn / 10.0)
beta = d.trans_pos.var_sched.betas[dev(t)]
with torch.enable_grad():
    out = training.eps_net(d.eps_net, vn, p_n, sn, dev(res_feat), dev(pair_feat), beta, dev(gen), dev(mres))
ref = odpm.eps_net(sd, 'eps_net.', vn.cpu(), p_n.cpu(), sn.cpu(), res_feat, pair_feat, beta.cpu(), gen, mres, num_layers=6, prmsd_head=False, grad_mode=True, mode='mm')
for name, a, b in zip(('v_next', 'R_next', 'eps_pos', 'c'), out, ref):
    print(name, (a.detach().cpu() - b).abs().max().item())
print('s_n equal', torch.equal(sn.cpu(), g['s_noisy']), 'eps equal', (eps.cpu() - g['pos']).abs().max().item())
# loss pieces
import torch.nn.functional as F
R_0 = G.so3_exp(v)
cp = ref[1].transpose(-2, -1).reshape(-1, 3); ct = R_0.transpose(-2, -1).reshape(-1, 3)
lr = F.cosine_embedding_loss(cp, ct, torch.ones(cp.shape[0], dtype=torch.long), reduction='none').reshape(2, 48, 3).sum(-1)
print('rot loss from oracle outputs', ((lr * gen.float()).sum() / (gen.float().sum() + 1e-8)).item())

"""Oracle: denoiser (EpsilonNet), noise schedule, IGSO(3) tables, per-step transitions, the
sampling loop and the training loss (torch CPU fp32).  Test infrastructure only.

Reference (D/ = /root/reference/AbDock/src/, A/ = /root/reference/AbDesign/diffab/):
  VarianceSchedule           D/modules/diffusion/transition.py:10-34
  PositionTransition         D/modules/diffusion/transition.py:42-101
  RotationTransition         D/modules/diffusion/transition.py:120-160
  AminoacidCategorical...    D/modules/diffusion/transition.py:163-245
  ApproxAngularDistribution  D/modules/common/so3.py:71-138, random_normal_so3 :141-146,
                             random_uniform_so3 :66-68
  EpsilonNet.forward         D/modules/diffusion/dpm_full.py:70-112 (A/...:62-102 w/o prmsd)
  FullDPM.forward            D/modules/diffusion/dpm_full.py:156-234 (A/...:138-191)
  FullDPM.sample             D/modules/diffusion/dpm_full.py:236-302 (A/...:193-254)
  FullDPM.optimize           D/modules/diffusion/dpm_full.py:304-367
  calc_dist_loss/perplexity  D/modules/diffusion/dpm_full.py:369-399
  pRMSDCa                    D/modules/common/prmsd.py:19-111, layers.py:18-58
  PerResiduePredictor        D/modules/common/nn.py:164-188

All randomness is *injected*: every function that draws in the reference takes the draw as an
argument here, in the reference's draw order ("RNG draw order").
"""
import math
import torch
import torch.nn.functional as F
from . import geometry as G
from .ipa import ga_encoder, layer_norm

K_AA = 20


# ----------------------------------------------------------------------------- schedule
def variance_schedule(T=100, s=0.01):
    t = torch.arange(0, T + 1, dtype=torch.float)
    f = torch.cos((math.pi / 2) * ((t / T) + s) / (1 + s)) ** 2
    abar = f / f[0]
    betas = torch.cat([torch.zeros([1]), 1 - (abar[1:] / abar[:-1])], dim=0).clamp_max(0.999)
    sig = torch.zeros_like(betas)
    for i in range(1, betas.size(0)):
        sig[i] = ((1 - abar[i - 1]) / (1 - abar[i])) * betas[i]
    return dict(betas=betas, alpha_bars=abar, alphas=1 - betas, sigmas=torch.sqrt(sig),
                sqrt_recip_alphas_cumprod=torch.sqrt(1. / abar),
                sqrt_recipm1_alphas_cumprod=torch.sqrt(1. / abar - 1))


def igso3_tables(stddevs, num_bins=8192, num_iters=1024, thr=0.1):
    """Histogram tables X, Y (n_std, num_bins) of the IGSO(3) angle density (so3.py:82-109)."""
    sd = torch.FloatTensor(list(stddevs))
    x = torch.linspace(0, math.pi, num_bins)
    l = torch.arange(0, num_iters)[None, :]
    Y = []
    for e in sd.tolist():
        xx = x[:, None]
        c = (1 - torch.cos(xx)) / math.pi
        a = (2 * l + 1) * torch.exp(-l * (l + 1) * (e ** 2))
        b = (torch.sin((l + 0.5) * xx) + 1e-6) / (torch.sin(xx / 2) + 1e-6)
        Y.append(torch.nan_to_num((c * a * b).sum(dim=1)).clamp_min(0))
    return dict(stddevs=sd, approx_flag=sd <= thr, X=x[None].repeat(len(Y), 1), Y=torch.stack(Y, 0))


def igso3_angle(tab, idx, bins, u, g):
    """Angle sample given the recorded draws (so3.py:111-138).  idx/bins/u/g are flat (M,)."""
    start = tab['X'][idx, bins]
    width = tab['X'][idx, bins + 1] - tab['X'][idx, bins]
    hist = start + u * width
    sd = tab['stddevs'][idx]
    gauss = (sd * 2 + g * sd).abs() % math.pi
    return torch.where(tab['approx_flag'][idx], gauss, hist)


def so3_noise(tab, t_idx, draws):
    """random_normal_so3 (so3.py:141-146) with injected draws: axis (..,3), bin, ubin, gauss."""
    shp = t_idx.shape
    u = F.normalize(draws['axis'], dim=-1)
    th = igso3_angle(tab, t_idx.flatten(), draws['bin'].flatten(), draws['ubin'].flatten(),
                     draws['gauss'].flatten()).reshape(shp)
    return u * th[..., None]


def uniform_so3(q4):
    """random_uniform_so3 (so3.py:66-68) given the raw randn(..,4) draw.  Runs under no_grad."""
    return G.so3_log(G.quat_to_rot(F.normalize(q4, dim=-1)), grad_mode=False)


# ----------------------------------------------------------------------------- denoiser
def _mlp3(sd, pre, x, idx=(0, 2, 4)):
    h = F.linear(x, sd[f'{pre}{idx[0]}.weight'], sd[f'{pre}{idx[0]}.bias']).relu()
    h = F.linear(h, sd[f'{pre}{idx[1]}.weight'], sd[f'{pre}{idx[1]}.bias']).relu()
    return F.linear(h, sd[f'{pre}{idx[2]}.weight'], sd[f'{pre}{idx[2]}.bias'])


def eps_net(sd, pre, v_t, p_t, s_t, res_feat, pair_feat, beta, mask_gen, mask_res,
            num_layers=6, prmsd_head=True, grad_mode=False, mode='ref'):
    """EpsilonNet.forward.  Returns (v_next, R_next, eps_pos, c_denoised[, prmsd_logits])."""
    N, L = mask_res.shape
    R = G.so3_exp(v_t)
    emb = sd[pre + 'current_sequence_embedding.weight'][s_t]
    x = torch.cat([res_feat, emb], dim=-1)
    x = F.linear(x, sd[pre + 'res_feat_mixer.0.weight'], sd[pre + 'res_feat_mixer.0.bias']).relu()
    x = F.linear(x, sd[pre + 'res_feat_mixer.2.weight'], sd[pre + 'res_feat_mixer.2.bias'])
    x = ga_encoder(sd, pre + 'encoder.', R, p_t, x, pair_feat, mask_res, num_layers, mode)

    temb = torch.stack([beta, torch.sin(beta), torch.cos(beta)], dim=-1)[:, None, :].expand(N, L, 3)
    feat = torch.cat([x, temb], dim=-1)
    gen3 = mask_gen[:, :, None].expand(N, L, 3)

    eps_crd = _mlp3(sd, pre + 'eps_crd_net.', feat)
    eps_pos = torch.where(gen3, G.rotate(R, eps_crd), torch.zeros_like(eps_crd))

    eps_rot = _mlp3(sd, pre + 'eps_rot_net.', feat)
    R_next = R @ G.quat1ijk_to_rot(eps_rot)
    v_next = torch.where(gen3, G.so3_log(R_next, grad_mode), v_t)

    c = torch.softmax(_mlp3(sd, pre + 'eps_seq_net.', feat), dim=-1)
    if not prmsd_head:
        return v_next, R_next, eps_pos, c
    pp = pre + 'prmsd_predictor.'
    h = layer_norm(feat, sd[pp + 'layer_norm.gamma'], sd[pp + 'layer_norm.beta'])
    h = F.linear(h, sd[pp + 'linear_1.weight'], sd[pp + 'linear_1.bias']).relu()
    h = F.linear(h, sd[pp + 'linear_2.weight'], sd[pp + 'linear_2.bias']).relu()
    h = F.linear(h, sd[pp + 'linear_3.weight'], sd[pp + 'linear_3.bias'])
    return v_next, R_next, eps_pos, c, h.mean(dim=1)


# ----------------------------------------------------------------------------- transitions
def one_hot20(x):
    ok = (x >= 0) & (x < K_AA)
    return (F.one_hot(x.clamp(0, K_AA - 1), K_AA) * ok[..., None]).float()


def pos_pred_noise_from_start(sch, p_t, p_0, mask_gen, t):
    a = sch['sqrt_recip_alphas_cumprod'][t].view(-1, 1, 1)
    b = sch['sqrt_recipm1_alphas_cumprod'][t].view(-1, 1, 1)
    eps = (a * p_t - p_0) / b
    return torch.where(mask_gen[..., None].expand_as(p_t), eps, p_t)


def pos_pred_start_from_noise(sch, p_t, eps, mask_gen, t):
    a = sch['sqrt_recip_alphas_cumprod'][t].view(-1, 1, 1)
    b = sch['sqrt_recipm1_alphas_cumprod'][t].view(-1, 1, 1)
    return torch.where(mask_gen[..., None].expand_as(p_t), a * p_t - b * eps, p_t)


def pos_add_noise(sch, p_0, mask_gen, t, e_rand):
    abar = sch['alpha_bars'][t]
    c0, c1 = torch.sqrt(abar).view(-1, 1, 1), torch.sqrt(1 - abar).view(-1, 1, 1)
    return torch.where(mask_gen[..., None].expand_as(p_0), c0 * p_0 + c1 * e_rand, p_0)


def pos_denoise(sch, p_t, eps, mask_gen, t, z):
    alpha = sch['alphas'][t].clamp_min(sch['alphas'][-2])
    abar = sch['alpha_bars'][t]
    sigma = sch['sigmas'][t].view(-1, 1, 1)
    c0 = (1.0 / torch.sqrt(alpha + 1e-8)).view(-1, 1, 1)
    c1 = ((1 - alpha) / torch.sqrt(1 - abar + 1e-8)).view(-1, 1, 1)
    z = torch.where((t > 1)[:, None, None].expand_as(p_t), z, torch.zeros_like(p_t))
    return torch.where(mask_gen[..., None].expand_as(p_t), c0 * (p_t - c1 * eps) + sigma * z, p_t)


def rot_add_noise(sch, tab_fwd, v_0, mask_gen, t, draws, grad_mode):
    N, L = mask_gen.shape
    abar = sch['alpha_bars'][t]
    c0 = torch.sqrt(abar).view(-1, 1, 1)
    e = so3_noise(tab_fwd, t[:, None].expand(N, L), draws)
    Rn = G.so3_exp(e) @ G.so3_exp(c0 * v_0)
    return torch.where(mask_gen[..., None].expand_as(v_0), G.so3_log(Rn, grad_mode), v_0)


def rot_denoise(tab_inv, v_t, v_next, mask_gen, t, draws, grad_mode=False):
    N, L = mask_gen.shape
    e = so3_noise(tab_inv, t[:, None].expand(N, L), draws)
    e = torch.where((t > 1)[:, None, None].expand(N, L, 3), e, torch.zeros_like(e))
    Rn = G.so3_exp(e) @ G.so3_exp(v_next)
    return torch.where(mask_gen[..., None].expand_as(v_t), G.so3_log(Rn, grad_mode), v_t)


def seq_posterior(sch, x_t, x_0, t):
    """transition.py:202-228 -- note alpha_bar_t multiplies BOTH factors."""
    c_t = x_t if x_t.dim() == 3 else one_hot20(x_t)
    c_0 = x_0 if x_0.dim() == 3 else one_hot20(x_0)
    a = sch['alpha_bars'][t][:, None, None]
    th = ((a * c_t) + (1 - a) / K_AA) * ((a * c_0) + (1 - a) / K_AA)
    return th / (th.sum(dim=-1, keepdim=True) + 1e-8)


def seq_add_noise_probs(sch, x_0, mask_gen, t):
    c_0 = one_hot20(x_0)
    a = sch['alpha_bars'][t][:, None, None]
    return torch.where(mask_gen[..., None].expand_as(c_0), (a * c_0) + ((1 - a) / K_AA), c_0)


def seq_denoise_probs(sch, x_t, c0_pred, mask_gen, t):
    c_t = one_hot20(x_t)
    post = seq_posterior(sch, c_t, c0_pred, t)
    return torch.where(mask_gen[..., None].expand_as(post), post, c_t)


def perplexity(post, mask_gen=None):
    """dpm_full.py:380-399: softmax applied to a probability vector (reference quirk)."""
    if mask_gen is None:
        mask_gen = torch.ones_like(post[..., 0], dtype=torch.bool)
    m = F.softmax(post, dim=-1).max(dim=-1)[0] * mask_gen.float()
    return m.sum(dim=-1) / mask_gen.float().sum(dim=-1)


def prmsd_score(logits, dist_min=0.5, dist_max=19.5):
    bounds = torch.linspace(dist_min, dist_max, logits.shape[-1])
    return (F.softmax(logits, dim=-1) * bounds).sum(-1)


# ----------------------------------------------------------------------------- model bundle
class Denoiser:
    """Holds the state_dict + schedule/tables of one FullDPM; mirrors its three entry points."""

    def __init__(self, sd, num_steps=100, num_layers=6, variant='abdock', obj='pred_x0',
                 num_bins=40, dist_min=0.5, dist_max=19.5, pre='diffusion.', tables=None, mode='ref',
                 position_mean=0.0, position_scale=10.0):
        self.sd, self.T, self.nl, self.variant, self.obj, self.pre, self.mode = sd, num_steps, num_layers, variant, obj, pre, mode
        self.num_bins, self.dist_min, self.dist_max = num_bins, dist_min, dist_max
        self.mean, self.scale = position_mean, position_scale
        self.sch = variance_schedule(num_steps)
        if tables is None:
            tables = (igso3_tables(torch.sqrt(1 - self.sch['alpha_bars']).tolist()),
                      igso3_tables(self.sch['sigmas'].tolist()))
        self.tab_fwd, self.tab_inv = tables
        self.abdock = variant == 'abdock'

    def _eps(self, v, p, s, res_feat, pair_feat, beta, gen, mres, grad_mode):
        return eps_net(self.sd, self.pre + 'eps_net.', v, p, s, res_feat, pair_feat, beta, gen, mres,
                       self.nl, prmsd_head=self.abdock, grad_mode=grad_mode, mode=self.mode)

    def norm(self, p):
        return (p - self.mean) / self.scale

    def unnorm(self, p):
        return p * self.scale + self.mean

    def step(self, t, v_t, p_t, s_t, res_feat, pair_feat, gen, mres, draws,
             sample_structure=True, sample_sequence=True, optimize_mode=False):
        """One iteration of the sampling loop body, p_t already normalised.  optimize_mode follows
        FullDPM.optimize (dpm_full.py:351-358), which feeds the net's third output to the position update as
        noise whatever `obj` is, and averages the perplexity over all residues."""
        N = v_t.shape[0]
        beta = self.sch['betas'][t].expand([N])
        tt = torch.full([N], t, dtype=torch.long)
        out = self._eps(v_t, p_t, s_t, res_feat, pair_feat, beta, gen, mres, False)
        v_next, R_next, p_pred, c = out[:4]
        if self.abdock and self.obj == 'pred_x0' and not optimize_mode:
            eps_p = pos_pred_noise_from_start(self.sch, p_t, p_pred, gen, tt)
        else:
            eps_p = p_pred
        v_new = rot_denoise(self.tab_inv, v_t, v_next, gen, tt, draws)
        p_new = pos_denoise(self.sch, p_t, eps_p, gen, tt, draws['z'])
        post = seq_denoise_probs(self.sch, s_t, c, gen, tt)
        s_new = draws['s_next']
        extras = {}
        if self.abdock:
            extras['prmsd'] = prmsd_score(out[4], self.dist_min, self.dist_max)
            extras['ppl'] = perplexity(post, None if optimize_mode else gen)
        if not sample_structure:
            v_new, p_new = v_t, p_t
        if not sample_sequence:
            s_new = s_t
        return v_new, p_new, s_new, dict(post=post, eps_out=out, **extras)

    def sample(self, v, p, s, res_feat, pair_feat, gen, mres, noise,
               sample_structure=True, sample_sequence=True, t_start=None, init_state=None):
        """FullDPM.sample / .optimize with injected noise.

        noise['init'] = dict(q4, p, s); noise[t] = dict(axis, bin, ubin, gauss, z, s_next).
        Returns traj: t -> (v, p_angstrom, s[, prmsd, ppl]).
        """
        if init_state is not None:
            v_i, p_i, s_i = init_state
        else:
            gen3 = gen[:, :, None].expand_as(v)
            p = self.norm(p)
            if sample_structure:
                v_i = torch.where(gen3, uniform_so3(noise['init']['q4']), v)
                p_i = torch.where(gen3, noise['init']['p'], p)
            else:
                v_i, p_i = v, p
            s_i = torch.where(gen, noise['init']['s'], s) if sample_sequence else s
        T0 = self.T if t_start is None else t_start
        traj = {T0: (v_i, self.unnorm(p_i), s_i)}
        for t in range(T0, 0, -1):
            v_t, p_t, s_t = traj[t][:3]
            v_n, p_n, s_n, ex = self.step(t, v_t, self.norm(p_t), s_t, res_feat, pair_feat, gen, mres,
                                          noise[t], sample_structure, sample_sequence, optimize_mode=init_state is not None)
            entry = (v_n, self.unnorm(p_n), s_n)
            if self.abdock:
                entry = entry + (ex['prmsd'], ex['ppl'])
            traj[t - 1] = entry
        return traj

    def optimize_init(self, v, p, s, gen, opt_step, noise, sample_structure=True, sample_sequence=True):
        """The noising prologue of FullDPM.optimize (dpm_full.py:320-339); returns init_state."""
        N = v.shape[0]
        p = self.norm(p)
        t = torch.full([N], opt_step, dtype=torch.long)
        gen3 = gen[:, :, None].expand_as(v)
        if sample_structure:
            v_n = rot_add_noise(self.sch, self.tab_fwd, v, gen, t, noise['rot'], False)
            p_n = pos_add_noise(self.sch, p, gen, t, noise['pos'])
            v_i, p_i = torch.where(gen3, v_n, v), torch.where(gen3, p_n, p)
        else:
            v_i, p_i = v, p
        s_i = torch.where(gen, noise['s'], s) if sample_sequence else s
        return v_i, p_i, s_i

    def loss(self, v_0, p_0, s_0, res_feat, pair_feat, gen, mres, t, noise,
             denoise_structure=True, denoise_sequence=True):
        """FullDPM.forward: the training loss dict, with fixed t and injected noise.
        noise = dict(rot=dict(axis,bin,ubin,gauss), pos=(N,L,3), s_noisy=(N,L))."""
        sch = self.sch
        p_0 = self.norm(p_0)
        R_0 = G.so3_exp(v_0)
        if denoise_structure:
            v_n = rot_add_noise(sch, self.tab_fwd, v_0, gen, t, noise['rot'], True)
            p_n = pos_add_noise(sch, p_0, gen, t, noise['pos'])
            eps_p = noise['pos']
        else:
            v_n, p_n, eps_p = v_0.clone(), p_0.clone(), torch.zeros_like(p_0)
        s_n = noise['s_noisy'] if denoise_sequence else s_0.clone()
        beta = sch['betas'][t]
        out = self._eps(v_n, p_n, s_n, res_feat, pair_feat, beta, gen, mres, True)
        v_pred, R_pred, p_pred, c = out[:4]
        genf = gen.float()
        denom = genf.sum() + 1e-8
        L = {}
        if self.abdock:
            if self.obj == 'pred_x0':
                p_true, pred_p0 = p_0, p_pred
            else:
                p_true, pred_p0 = p_n, pos_pred_start_from_noise(sch, p_0, p_pred, gen, t)
            a, b = self.unnorm(pred_p0) * gen.unsqueeze(-1), self.unnorm(p_0) * gen.unsqueeze(-1)
            rmsd = torch.sqrt(((a - b) ** 2).sum(-1).sum(-1) / gen.sum(-1))
            L['prmsd'] = self._prmsd_loss(out[4], rmsd.detach(), gen[:, 0])
            if self.obj == 'pred_x0':
                L['dist'] = self._dist_loss(p_pred, p_true, gen, mres)
            pos_target = p_true
        else:
            pos_target = eps_p
        # rotation: cosine-embedding loss on matrix columns (dpm_full.py:15-32)
        cp = R_pred.transpose(-2, -1).reshape(-1, 3)
        ct = R_0.transpose(-2, -1).reshape(-1, 3)
        lr = F.cosine_embedding_loss(cp, ct, torch.ones(cp.shape[0], dtype=torch.long), reduction='none')
        lr = lr.reshape(list(R_pred.shape[:-2]) + [3]).sum(-1)
        L['rot'] = (lr * genf).sum() / denom
        lp = F.mse_loss(p_pred, pos_target, reduction='none').sum(-1)
        L['pos'] = (lp * genf).sum() / denom
        post_true = seq_posterior(sch, s_n, s_0, t)
        log_pred = torch.log(seq_posterior(sch, s_n, c, t) + 1e-8)
        kl = F.kl_div(input=log_pred, target=post_true, reduction='none', log_target=False).sum(-1)
        L['seq'] = (kl * genf).sum() / denom
        return L

    def _prmsd_loss(self, logits, rmsd, mask):
        off = torch.linspace(self.dist_min, self.dist_max, self.num_bins)
        diff = torch.abs(rmsd.unsqueeze(-1) - off)
        onehot = torch.zeros_like(diff).scatter_(-1, torch.argmin(diff, -1, keepdim=True), 1.0)
        err = -(onehot * F.log_softmax(logits, dim=-1)).sum(-1)
        return (err * mask).sum() / (mask.sum() + 1e-10)

    @staticmethod
    def _dist_loss(p_pred, p_true, gen, mres):
        dp, dt = torch.cdist(p_pred, p_pred), torch.cdist(p_true, p_true)
        mm = mres[:, :, None] & mres[:, None, :]
        sel = gen[:, :, None].expand_as(dp) & mm
        return F.smooth_l1_loss(torch.masked_select(dp, sel), torch.masked_select(dt, sel), reduction='none').mean()


def rank_commoness(structs, k):
    """design_for_testset.py:556-589: candidates (B,n,3) -> indices of the k most central."""
    B = structs.shape[0]
    a = structs.unsqueeze(1).repeat(1, B, 1, 1)
    rmsd = torch.sqrt((((a - a.permute(1, 0, 2, 3)) ** 2).sum(-1)).mean(-1))
    return torch.topk(rmsd.sum(-1) / (B - 1), k=k, largest=False)[1]

"""Pins the CPU oracle (oracle/) to outputs of the reference itself (tests/golden/*.npz, produced by
tests/golden/make_golden.py from /root/reference in the build container).  CPU only."""
import math
import pytest
import torch

import cases
from conftest import load_golden, build_model, max_abs
from ab_opt_amd.utils import synth
from oracle import geometry as G
from oracle import ipa, dpm, embed


def standalone_block_sd(seed=1):
    from ab_opt_amd.modules import GABlock
    return synth.fill_module_(GABlock(128, 64), seed=seed).state_dict()


def standalone_abdesign_dpm(T, seed):
    from ab_opt_amd.dpm import FullDPM
    m = FullDPM(128, 64, num_steps=T, eps_net_opt=dict(num_layers=6), _abdesign=True).eval()
    return synth.fill_module_(m, seed=seed)


def noise_dict(g, T, with_init=True, N=None, L=None):
    nz = {}
    if with_init:
        nz['init'] = dict(q4=g['init_q4'], p=g['init_p'], s=g.get('init_s'))
    for t in range(T, 0, -1):
        nz[t] = {k: g[f't{t}_{k}'] for k in ('axis', 'bin', 'ubin', 'gauss', 'z', 's_next')}
    return nz


def test_so3_maps():
    g = load_golden('so3')
    w = cases.SO3_EDGE_W
    R = G.so3_exp(w)
    assert max_abs(R, g['exp']) < 1e-6
    assert max_abs(G.so3_log(g['exp'], False), g['log_nograd']) < 2e-6
    assert max_abs(G.so3_log(g['exp'], True), g['log_grad']) < 2e-6
    assert max_abs(G.quat_to_rot(synth.hash_tensor((16, 4), 7, scale=2.0)), g['quat']) < 1e-6
    assert max_abs(G.quat1ijk_to_rot(synth.hash_tensor((16, 3), 8, scale=3.0)), g['quat1ijk']) < 1e-6


@pytest.mark.parametrize('mode', ['ref', 'mm'])
def test_ga_block_parts(mode):
    g = load_golden('ga_block')
    sd = standalone_block_sd()
    R, t, x, z, mask = cases.ipa_inputs(2, 24, [24, 19])
    parts = {}
    out = ipa.ga_block(sd, '', R, t, x, z, mask, mode=mode, parts=parts)
    tol = 1e-6 if mode == 'ref' else 2e-5
    for k in ('l_node', 'l_pair', 'l_spat'):
        assert max_abs(parts[k], g[k]) < 1e-4 * max(1.0, g[k].abs().max().item()) * (1 if mode == 'mm' else 0.05), k
    assert max_abs(parts['alpha'], g['alpha']) < max(tol, 5e-6)
    assert max_abs(parts['feat'], g['feat']) < 2e-5
    assert max_abs(out, g['out']) < 2e-5


def test_ga_block_L128():
    g = load_golden('ga_block_L128')
    sd = standalone_block_sd()
    R, t, x, z, mask = cases.ipa_inputs(2, 128, [128, 101], salt=150)
    out = ipa.ga_block(sd, '', R, t, x, z, mask, mode='mm')
    assert max_abs(out, g['out']) < 3e-5


def _check_eps(out, g, has_prmsd):
    v_next, R_next, eps_pos, c = out[:4]
    assert max_abs(R_next, g['R_next']) < 2e-5
    assert max_abs(G.so3_exp(v_next), G.so3_exp(g['v_next'])) < 5e-5      # compare orientations as matrices
    assert max_abs(eps_pos, g['eps_pos']) < 2e-5
    assert max_abs(c, g['c']) < 1e-5
    if has_prmsd:
        assert max_abs(out[4], g['prmsd_logits']) < 2e-5


def test_eps_net_abdock():
    sd = build_model(100, 2).state_dict()
    g = load_golden('eps_net_abdock_small')
    args = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)])
    _check_eps(dpm.eps_net(sd, 'diffusion.eps_net.', *args, num_layers=6, prmsd_head=True), g, True)
    g = load_golden('eps_net_abdock_L128')
    args = cases.eps_inputs(1, 128, [128], [(30, 42)])
    _check_eps(dpm.eps_net(sd, 'diffusion.eps_net.', *args, num_layers=6, prmsd_head=True, mode='mm'), g, True)


def test_eps_net_abdesign():
    sd = standalone_abdesign_dpm(100, 2).state_dict()
    import json, os
    from conftest import GOLDEN
    ref_keys = json.load(open(os.path.join(GOLDEN, 'state_dict_abdesign_fulldpm.json')))
    assert list(ref_keys) == list(sd) and all(list(sd[k].shape) == v for k, v in ref_keys.items())
    g = load_golden('eps_net_abdesign_small')
    args = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)])
    _check_eps(dpm.eps_net(sd, 'eps_net.', *args, num_layers=6, prmsd_head=False), g, False)


def test_schedule_and_tables():
    g = load_golden('schedule_T100')
    sch = dpm.variance_schedule(100)
    for k in ('betas', 'alpha_bars', 'alphas', 'sigmas', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod'):
        assert torch.equal(sch[k], g[k]) or max_abs(sch[k], g[k]) <= 1e-7 * g[k].abs().max().item(), k
    g10 = load_golden('schedule_T10')
    sch10 = dpm.variance_schedule(10)
    tab = dpm.igso3_tables(sch10['sigmas'].tolist())
    assert max_abs(tab['stddevs'], g10['inv_stddevs']) == 0
    rel = (tab['Y'][:, ::16] - g10['inv_Ysub']).abs().max() / g10['inv_Ysub'].abs().max()
    assert rel < 1e-5
    tabf = dpm.igso3_tables(torch.sqrt(1 - sch10['alpha_bars']).tolist())
    rel = (tabf['Y'][:, ::16] - g10['fwd_Ysub']).abs().max() / g10['fwd_Ysub'].abs().max()
    assert rel < 1e-5


def test_model_buffers_match_reference():
    """The product model's init-time buffers (host code) against the reference's."""
    g = load_golden('schedule_T100')
    m = build_model(100, 2)
    vs = m.diffusion.trans_pos.var_sched
    for k in ('betas', 'alpha_bars', 'alphas', 'sigmas', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod'):
        assert max_abs(getattr(vs, k), g[k]) <= 1e-7 * g[k].abs().max().item(), k
    for d in ('fwd', 'inv'):
        tab = getattr(m.diffusion.trans_rot, f'angular_distrib_{d}')
        assert torch.equal(tab.approx_flag, g[f'{d}_approx_flag'])
        assert max_abs(tab.stddevs, g[f'{d}_stddevs']) == 0
        assert max_abs(tab.X[0], g[f'{d}_X0']) == 0
        scale = g[f'{d}_Ysub'].abs().max().item()
        assert max_abs(tab.Y[:, ::64], g[f'{d}_Ysub']) < 1e-5 * scale
        assert max_abs(tab.Y[50], g[f'{d}_Yrow50']) < 1e-5 * scale
        cdf = tab.cdf()
        assert cdf.shape == (101, 8191) and (cdf[:, 1:] >= cdf[:, :-1]).all()
        rows = tab.Y[:, :-1].sum(1) > 0
        assert torch.allclose(cdf[rows, -1], torch.ones(int(rows.sum())), atol=1e-6)


def _traj_check(traj, g, T, abdock, tol_R=1e-4, tol_p=1e-4):
    for t in range(T, -1, -1):
        e = traj[t]
        assert max_abs(G.so3_exp(e[0]), G.so3_exp(g[f'traj{t}_v'])) < tol_R, f'v at t={t}'
        assert max_abs(e[1], g[f'traj{t}_p']) < tol_p, f'p at t={t}'
        assert torch.equal(e[2], g[f'traj{t}_s']), f's at t={t}'
        if abdock and t < T and f'traj{t}_prmsd' in g:
            assert max_abs(e[3], g[f'traj{t}_prmsd']) < 1e-4
            assert max_abs(e[4], g[f'traj{t}_ppl']) < 1e-5


def _encode_traj_case(m):
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=2022, lengths=[128, 117])
    sd = m.state_dict()
    rf, pf, R0, p0 = embed.encode(sd, batch, True, True)
    return batch, sd, rf, pf, R0, p0


def test_trajectory_abdock_T10():
    """BASELINE config 1: N=2, L=128, one CDR, 10 steps; oracle replays the reference's recorded draws."""
    g = load_golden('trajectory_abdock_T10')
    m = build_model(10, 3)
    batch, sd, rf, pf, R0, p0 = _encode_traj_case(m)
    assert max_abs(rf, g['res_feat']) < 2e-5
    assert max_abs(pf[:, ::7, ::5], g['pair_feat_sub']) < 2e-5
    assert max_abs(R0, g['R0']) < 1e-6
    # mode='ref' (the reference's own op order) reproduces the reference BIT-FOR-BIT over the whole free-running
    # trajectory; any other summation order diverges chaotically after ~3 steps (DESIGN.md "conditioning"), which
    # is why HIP parity is asserted per step with teacher-forced states, not on free-running trajectories.
    den = dpm.Denoiser(sd, num_steps=10, variant='abdock', obj='pred_x0', mode='ref')
    traj = den.sample(G.so3_log(R0), p0, batch['aa'], rf, pf, batch['generate_flag'], batch['mask'], noise_dict(g, 10))
    _traj_check(traj, g, 10, True, tol_R=1e-6, tol_p=1e-6)
    # final CA coordinates: RMSD over generated residues (north_star: within 1e-4 A)
    gen = batch['generate_flag']
    d = (traj[0][1] - g['traj0_p'])[gen]
    assert math.sqrt((d ** 2).sum(-1).mean().item()) < 1e-4


def test_trajectory_structure_only_and_optimize():
    m = build_model(10, 3)
    batch, sd, rf, pf, R0, p0 = _encode_traj_case(m)
    den = dpm.Denoiser(sd, num_steps=10, variant='abdock', obj='pred_x0', mode='ref')
    g = load_golden('trajectory_abdock_T10_structonly')
    nz = noise_dict(g, 10)
    # sample_sequence=False => encode keeps the sequence (remove_sequence=False)
    rf2, pf2, _, _ = embed.encode(sd, batch, True, False)
    traj = den.sample(G.so3_log(R0), p0, batch['aa'], rf2, pf2, batch['generate_flag'], batch['mask'], nz, sample_sequence=False)
    assert max_abs(rf2, g['res_feat']) < 2e-5
    for t in range(10, -1, -1):                   # every step of the structure-only run (BASELINE config 3 mode)
        assert max_abs(G.so3_exp(traj[t][0]), G.so3_exp(g[f'traj{t}_v'])) < 1e-6, t
        assert max_abs(traj[t][1], g[f'traj{t}_p']) < 1e-6, t
        assert torch.equal(traj[t][2], g[f'traj{t}_s']), t
        if t < 10:
            assert max_abs(traj[t][3], g[f'traj{t}_prmsd']) < 1e-4 and max_abs(traj[t][4], g[f'traj{t}_ppl']) < 1e-5, t
    g = load_golden('optimize_abdock_T10_k4')
    init = den.optimize_init(G.so3_log(R0), p0, batch['aa'], batch['generate_flag'], 4,
                             dict(rot=dict(axis=g['rot_axis'], bin=g['rot_bin'], ubin=g['rot_ubin'], gauss=g['rot_gauss']),
                                  pos=g['pos'], s=g['s_noisy']))
    traj = den.sample(None, None, None, rf, pf, batch['generate_flag'], batch['mask'], noise_dict(g, 4, with_init=False),
                      t_start=4, init_state=init)
    _traj_check(traj, g, 4, False, tol_R=1e-6, tol_p=1e-6)


def test_categorical_distributions_vs_reference():
    """The categorical the reference SAMPLES from (input of AminoacidCategoricalTransition._sample, transition.py:171-181) at
    every recorded step: denoising posterior (transition.py:202-245) from the reference's own states, and the forward
    categorical of add_noise (transition.py:183-200).  The sampled sequences in the trajectory fixtures are injected draws, so
    this is what pins the sequence transition."""
    g = load_golden('trajectory_abdock_T10')
    gp = load_golden('posterior_abdock_T10')
    m = build_model(10, 3)
    batch, sd, rf, pf, R0, p0 = _encode_traj_case(m)
    gen, mres = batch['generate_flag'], batch['mask']
    den = dpm.Denoiser(sd, num_steps=10, variant='abdock', obj='pred_x0', mode='ref')
    nz = noise_dict(g, 10)
    for t in (10, 7, 4, 1):
        _, _, _, ex = den.step(t, g[f'traj{t}_v'], den.norm(g[f'traj{t}_p']), g[f'traj{t}_s'], rf, pf, gen, mres, nz[t])
        assert max_abs(ex['post'] + 1e-8, gp[f't{t}_probs']) < 1e-7, t
        assert ex['post'][gen].sum(-1).sub(1).abs().max() < 1e-5
    go = load_golden('optimize_abdock_T10_k4')
    tt = torch.full([2], 4, dtype=torch.long)
    assert max_abs(dpm.seq_add_noise_probs(den.sch, batch['aa'], gen, tt) + 1e-8, gp['opt_addnoise_probs']) < 1e-7
    _, _, _, ex = den.step(3, go['traj3_v'], den.norm(go['traj3_p']), go['traj3_s'], rf, pf, gen, mres,
                           {k: go[f't3_{k}'] for k in ('axis', 'bin', 'ubin', 'gauss', 'z', 's_next')}, optimize_mode=True)
    assert max_abs(ex['post'] + 1e-8, gp['opt_t3_probs']) < 1e-7


def test_trajectory_abdesign_T10():
    g = load_golden('trajectory_abdesign_T10')
    m = standalone_abdesign_dpm(10, 4)
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)], num_steps=10, t=3)
    den = dpm.Denoiser(m.state_dict(), num_steps=10, variant='abdesign', pre='')
    traj = den.sample(v, p * 10, s, res_feat, pair_feat, gen, mres, noise_dict(g, 10))
    _traj_check(traj, g, 10, False, tol_R=1e-6, tol_p=1e-6)


STEP_KEYS = ('axis', 'bin', 'ubin', 'gauss', 'z', 's_next')
STEP_TS = (100, 64, 22, 21, 10, 2, 1)


def step_noise(g, t, pre=''):
    return {k: g[f'{pre}t{t}_{k}'] for k in STEP_KEYS}


def ref_tables(dpm_module):
    """(fwd, inv) IGSO(3) tables of a product FullDPM as the oracle's dicts (the 1024-term series takes 9 s per table to rebuild;
    the product's init-time buffers are themselves pinned to the reference by test_model_buffers_match_reference)."""
    r = dpm_module.trans_rot
    return tuple(dict(stddevs=a.stddevs, approx_flag=a.approx_flag, X=a.X, Y=a.Y) for a in (r.angular_distrib_fwd, r.angular_distrib_inv))


def test_state_dict_abdock_matches_reference_key_list():
    """identical state_dict keys, shapes and ORDER (207 entries for AbDock), and a reference-shaped checkpoint loads
    strictly."""
    import json, os
    from conftest import GOLDEN
    ref = json.load(open(os.path.join(GOLDEN, 'state_dict_abdock.json')))
    m = build_model(100, 2)
    sd = m.state_dict()
    assert len(ref) == 207 and list(ref) == list(sd)
    assert all(list(sd[k].shape) == shp for k, shp in ref.items())
    ckpt = {k: torch.zeros(shp, dtype=sd[k].dtype) for k, shp in ref.items()}
    from ab_opt_amd import get_model
    m2 = get_model(synth.AttrDict(synth.cfg_abdock(100)))
    res = m2.load_state_dict(ckpt, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    with pytest.raises(RuntimeError):
        m2.load_state_dict({k: v for k, v in ckpt.items() if 'proj_pair_bias' not in k}, strict=True)


def test_trajectory_seqdesign_abdock_T10():
    """seq_design.yml mode (sample_structure=False, sample_sequence=True, contig): every recorded step of the reference's run."""
    g = load_golden('trajectory_abdock_T10_seqdesign')
    m = build_model(10, 3)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=2022, lengths=[128, 117])
    from ab_opt_amd.model import generate_mask_from_str
    gen = torch.logical_and(batch['generate_flag'], generate_mask_from_str('31-36', batch['generate_flag']))      # diffab.py:125-129
    assert torch.equal(gen, g['gen'])
    batch['generate_flag'] = gen
    sd = m.state_dict()
    rf, pf, R0, p0 = embed.encode(sd, batch, False, True)                 # the structure stays in the features in this mode
    assert max_abs(rf, g['res_feat']) < 2e-5 and max_abs(pf[:, ::7, ::5], g['pair_feat_sub']) < 2e-5
    assert max_abs(R0, g['R0']) < 1e-6 and max_abs(p0, g['p0']) == 0
    den = dpm.Denoiser(sd, num_steps=10, variant='abdock', obj='pred_x0', mode='ref', tables=ref_tables(m.diffusion))
    nz = {t: step_noise(g, t) for t in range(10, 0, -1)}
    nz['init'] = dict(q4=None, p=None, s=g['init_s'])
    v0 = G.so3_log(R0)
    traj = den.sample(v0, p0, batch['aa'], rf, pf, gen, batch['mask'], nz, sample_structure=False, sample_sequence=True)
    assert torch.equal(traj[10][2][gen], g['init_s'][gen]) and torch.equal(traj[10][2][~gen], batch['aa'][~gen])
    for t in range(10, -1, -1):
        assert torch.equal(traj[t][0], v0), t                             # dpm_full.py:294-295: the structure is handed on untouched
        assert max_abs(G.so3_exp(traj[t][0]), G.so3_exp(g[f'traj{t}_v'])) < 1e-6, t
        assert max_abs(traj[t][1], g[f'traj{t}_p']) < 1e-6 and max_abs(traj[t][1], p0) < 1e-5, t
        assert torch.equal(traj[t][2], g[f'traj{t}_s']), t
        if t < 10:
            assert max_abs(traj[t][3], g[f'traj{t}_prmsd']) < 1e-4 and max_abs(traj[t][4], g[f'traj{t}_ppl']) < 1e-5, t
    for t in range(10, 0, -1):                                            # the categorical the reference sampled from, every step
        _, _, _, ex = den.step(t, g[f'traj{t}_v'], den.norm(g[f'traj{t}_p']), g[f'traj{t}_s'], rf, pf, gen, batch['mask'], nz[t],
                               sample_structure=False)
        assert max_abs(ex['post'] + 1e-8, g[f't{t}_probs']) < 1e-7, t


def test_trajectory_fixbb_abdesign_T10():
    """AbDesign fixbb.yml mode at FullDPM level (A/.../dpm_full.py:193-254 with sample_structure=False)."""
    g = load_golden('trajectory_abdesign_T10_fixbb')
    m = standalone_abdesign_dpm(10, 4)
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)], num_steps=10, t=3)
    den = dpm.Denoiser(m.state_dict(), num_steps=10, variant='abdesign', pre='', tables=ref_tables(m))
    nz = {t: step_noise(g, t) for t in range(10, 0, -1)}
    nz['init'] = dict(q4=None, p=None, s=g['init_s'])
    traj = den.sample(v, p * 10, s, res_feat, pair_feat, gen, mres, nz, sample_structure=False, sample_sequence=True)
    for t in range(10, -1, -1):
        assert torch.equal(traj[t][0], v) and torch.equal(traj[t][0], g[f'traj{t}_v']), t
        assert max_abs(traj[t][1], g[f'traj{t}_p']) < 1e-6, t
        assert torch.equal(traj[t][2], g[f'traj{t}_s']), t
    for t in range(10, 0, -1):
        _, _, _, ex = den.step(t, g[f'traj{t}_v'], den.norm(g[f'traj{t}_p']), g[f'traj{t}_s'], res_feat, pair_feat, gen, mres, nz[t],
                               sample_structure=False)
        assert max_abs(ex['post'] + 1e-8, g[f't{t}_probs']) < 1e-7, t


@pytest.mark.parametrize('flavour', ['abdock', 'abdesign'])
def test_training_sequence_only_loss_and_grads(flavour):
    """train_structure=False, train_sequence=True (configs/train/seq_design.yml:11-12; dpm_full.py:163-178): losses + gradients."""
    g = load_golden(f'training_seqonly_{flavour}')
    if flavour == 'abdock':
        full = build_model(100, 2)
        m, pre = full.diffusion, 'diffusion.'
        sd0 = full.state_dict()
    else:
        m, pre = standalone_abdesign_dpm(100, 2), ''
        sd0 = m.state_dict()
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'eps_net' in k) for k, v in sd0.items()}
    N, L = 2, 48
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
    s = s.clamp(max=19)
    res_feat = res_feat.clone().requires_grad_(True)
    pair_feat = pair_feat.clone().requires_grad_(True)
    den = dpm.Denoiser(sd, num_steps=100, variant=flavour, obj='pred_x0', pre=pre, tables=ref_tables(m))
    t = torch.tensor([37, 80])
    assert max_abs(dpm.seq_add_noise_probs(den.sch, s, gen, t) + 1e-8, g['addnoise_probs']) < 1e-7
    with torch.enable_grad():
        loss = den.loss(v, p * 10, s, res_feat, pair_feat, gen, mres, t, dict(rot=None, pos=None, s_noisy=g['s_noisy']),
                        denoise_structure=False, denoise_sequence=True)
        sum(loss.values()).backward()
    assert set(loss) == ({'prmsd', 'dist', 'rot', 'pos', 'seq'} if flavour == 'abdock' else {'rot', 'pos', 'seq'})
    for k in loss:
        ref = g['loss_' + k].item()
        assert abs(loss[k].item() - ref) <= 1e-5 * max(1.0, abs(ref)), (k, loss[k].item(), ref)
    n = 0
    for k in g:
        if k.startswith('grad_eps_net'):
            got = _sub(sd[pre + k[len('grad_'):]].grad, g[k])
            assert max_abs(got, g[k]) <= 2e-4 * g[k].abs().max().item() + 1e-7, k
            n += 1
    assert n == 11
    assert max_abs(res_feat.grad, g['grad_res_feat']) <= 2e-4 * g['grad_res_feat'].abs().max().item()
    assert max_abs(pair_feat.grad[:, ::5, ::3], g['grad_pair_feat_sub']) <= 2e-4 * g['grad_pair_feat_sub'].abs().max().item()


@pytest.mark.parametrize('flavour', ['abdock', 'abdesign'])
def test_single_steps_T100_vs_reference(flavour):
    """The headline schedule (T = 100): recorded reference steps at t = 100, 64, 22 (last histogram row of the inverse IGSO(3)
    distribution), 21, 10, 2 (Gaussian branch, so3.py:127-135) and 1 (no noise), both trees, 1e-6."""
    g = load_golden(f'steps_T100_{flavour}')
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)], num_steps=100, t=3)
    if flavour == 'abdock':
        full = build_model(100, 2)
        den = dpm.Denoiser(full.state_dict(), num_steps=100, variant='abdock', obj='pred_x0', mode='ref', tables=ref_tables(full.diffusion))
        a, b = 'traj{}', 'traj{}'
    else:
        m = standalone_abdesign_dpm(100, 2)
        den = dpm.Denoiser(m.state_dict(), num_steps=100, variant='abdesign', pre='', mode='ref', tables=ref_tables(m))
        a, b = 'in{}', 'out{}'
    assert [bool(x) for x in den.tab_inv['approx_flag'][[100, 64, 22, 21, 10, 2, 1]]] == [False, False, False, True, True, True, True]      # sigma_1 = 0: flagged, but t = 1 adds no noise
    for t in STEP_TS:
        i, o = a.format(t), b.format(t - 1 if flavour == 'abdock' else t)
        nz = step_noise(g, t)
        v_n, p_n, s_n, ex = den.step(t, g[i + '_v'], den.norm(g[i + '_p']), g[i + '_s'], res_feat, pair_feat, gen, mres, nz)
        assert max_abs(G.so3_exp(v_n), G.so3_exp(g[o + '_v'])) < 1e-6, t
        assert max_abs(den.unnorm(p_n), g[o + '_p']) < 1e-6, t
        assert torch.equal(s_n, g[o + '_s']), t
        assert max_abs(ex['post'] + 1e-8, g[f't{t}_probs']) < 1e-7, t
        if flavour == 'abdock':
            assert max_abs(ex['prmsd'], g[o + '_prmsd']) < 1e-4 and max_abs(ex['ppl'], g[o + '_ppl']) < 1e-5, t
        if t in (21, 10, 2):                           # the branch under test really decides the angle here
            sd_t = den.tab_inv['stddevs'][t]
            th = (2 * sd_t + nz['gauss'] * sd_t).abs() % math.pi
            e = dpm.so3_noise(den.tab_inv, torch.full((2, 40), t), nz)
            assert max_abs(e.norm(dim=-1), th) < 1e-6


def test_training_loss_and_grads():
    g = load_golden('training_abdock')
    m = build_model(100, 2)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'eps_net' in k) for k, v in m.state_dict().items()}
    N, L = 2, 48
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
    s = s.clamp(max=19)
    res_feat = res_feat.clone().requires_grad_(True)
    pair_feat = pair_feat.clone().requires_grad_(True)
    den = dpm.Denoiser(sd, num_steps=100, variant='abdock', obj='pred_x0', tables=(None, None))
    m_ = m.diffusion.trans_rot
    den.tab_fwd = dict(stddevs=m_.angular_distrib_fwd.stddevs, approx_flag=m_.angular_distrib_fwd.approx_flag,
                       X=m_.angular_distrib_fwd.X, Y=m_.angular_distrib_fwd.Y)
    noise = dict(rot=dict(axis=g['rot_axis'], bin=g['rot_bin'], ubin=g['rot_ubin'], gauss=g['rot_gauss']), pos=g['pos'], s_noisy=g['s_noisy'])
    with torch.enable_grad():
        loss = den.loss(v, p * 10, s, res_feat, pair_feat, gen, mres, torch.tensor([37, 80]), noise)
        sum(loss.values()).backward()
    for k in ('prmsd', 'dist', 'rot', 'pos', 'seq'):
        ref = g['loss_' + k].item()
        assert abs(loss[k].item() - ref) <= 1e-5 * max(1.0, abs(ref)), (k, loss[k].item(), ref)
    for k in g:
        if k.startswith('grad_eps_net'):
            got = sd['diffusion.' + k[len('grad_'):]].grad
            sc = g[k].abs().max().item()
            assert max_abs(got, g[k]) <= 2e-4 * sc + 1e-7, k
    assert max_abs(res_feat.grad, g['grad_res_feat']) <= 2e-4 * g['grad_res_feat'].abs().max().item()
    assert max_abs(pair_feat.grad[:, ::5, ::3], g['grad_pair_feat_sub']) <= 2e-4 * g['grad_pair_feat_sub'].abs().max().item()


def _sub(got, ref):
    """Fixtures store big gradient matrices as a [::3, ::5] sample."""
    return got[::3, ::5] if got.shape != ref.shape else got


def test_training_abdesign_loss_and_grads():
    """BASELINE config 5 flavour: AbDesign FullDPM.forward (rot, pos on the noise, seq; A/modules/diffusion/dpm_full.py:138-191)."""
    g = load_golden('training_abdesign')
    m = standalone_abdesign_dpm(100, 2)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'eps_net' in k) for k, v in m.state_dict().items()}
    N, L = 2, 48
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
    s = s.clamp(max=19)
    res_feat = res_feat.clone().requires_grad_(True)
    pair_feat = pair_feat.clone().requires_grad_(True)
    den = dpm.Denoiser(sd, num_steps=100, variant='abdesign', pre='', tables=(None, None))
    f = m.trans_rot.angular_distrib_fwd
    den.tab_fwd = dict(stddevs=f.stddevs, approx_flag=f.approx_flag, X=f.X, Y=f.Y)
    noise = dict(rot=dict(axis=g['rot_axis'], bin=g['rot_bin'], ubin=g['rot_ubin'], gauss=g['rot_gauss']), pos=g['pos'], s_noisy=g['s_noisy'])
    with torch.enable_grad():
        loss = den.loss(v, p * 10, s, res_feat, pair_feat, gen, mres, torch.tensor([37, 80]), noise)
        sum(loss.values()).backward()
    assert set(loss) == {'rot', 'pos', 'seq'}
    for k in loss:
        ref = g['loss_' + k].item()
        assert abs(loss[k].item() - ref) <= 1e-5 * max(1.0, abs(ref)), (k, loss[k].item(), ref)
    n = 0
    for k in g:
        if k.startswith('grad_eps_net'):
            got = _sub(sd[k[len('grad_'):]].grad, g[k])
            assert max_abs(got, g[k]) <= 2e-4 * g[k].abs().max().item() + 1e-7, k
            n += 1
    assert n == 10
    assert max_abs(res_feat.grad, g['grad_res_feat']) <= 2e-4 * g['grad_res_feat'].abs().max().item()
    assert max_abs(pair_feat.grad[:, ::5, ::3], g['grad_pair_feat_sub']) <= 2e-4 * g['grad_pair_feat_sub'].abs().max().item()


def test_encode_small():
    g = load_golden('encode_small')
    m = build_model(10, 3)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=99, lengths=[24, 19])
    batch['generate_flag'][:, 8:13] = True
    batch['fragment_type'][:, :12] = 1
    batch['fragment_type'][:, 12:] = 3
    batch['fragment_type'] = batch['fragment_type'] * batch['mask']
    batch['chain_nb'][:, 12:] = 1
    sd = m.state_dict()
    rf, pf, R0, p0 = embed.encode(sd, batch, True, True)
    assert max_abs(rf, g['res_feat']) < 2e-5 and max_abs(pf, g['pair_feat']) < 2e-5
    assert max_abs(R0, g['R0']) < 1e-6 and max_abs(p0, g['p0']) == 0
    rf2, pf2, _, _ = embed.encode(sd, batch, True, False)
    assert max_abs(rf2, g['res_feat_seqkept']) < 2e-5
    assert max_abs(pf2.double().sum((1, 2)), g['pair_feat_seqkept_sum']) < 1e-3
    # the product has no CPU path, in either autograd mode (its encode is checked on the device in test_hip_parity.py) ...
    for ctx in (torch.enable_grad(), torch.no_grad()):
        with ctx, pytest.raises(RuntimeError, match='no CPU path'):
            m.encode({k: v.clone() for k, v in batch.items()}, True, True)
    # ... and the plain torch statement the device tests use as their yardstick (tests/plain_statement.py) is itself pinned to the
    # reference here: values and the reference's recorded gradients
    import plain_statement
    with torch.no_grad():
        prf, ppf, pR, pp = plain_statement.encode(m, {k: v.clone() for k, v in batch.items()}, True, True)
    assert max_abs(prf, g['res_feat']) < 2e-5 and max_abs(ppf, g['pair_feat']) < 2e-5 and max_abs(pR, g['R0']) < 1e-6
    m.zero_grad()
    with torch.enable_grad():
        rfg, pfg, _, _ = plain_statement.encode(m, {k: v.clone() for k, v in batch.items()}, True, True)
        w1, w2 = synth.hash_tensor(tuple(rfg.shape), 71, scale=1.0), synth.hash_tensor(tuple(pfg.shape), 72, scale=1.0)
        ((rfg * w1).sum() + (pfg * w2).sum()).backward()
    P = dict(m.named_parameters())
    checked = 0
    for k in g:
        if not k.startswith('grad_'):
            continue
        name = k[len('grad_'):].replace('__', '.')
        got = P['residue_embed.mlp.0.weight'].grad[::4, ::7] if name.endswith('_sub') else P[name].grad
        assert max_abs(got, g[k]) <= 2e-4 * max(1e-6, g[k].abs().max().item()), name
        checked += 1
    assert checked == 10
    m.zero_grad()


def test_reconstruct_backbone_partially():
    g = load_golden('reconstruct_small')
    b = cases.reconstruct_batch()
    pos, m = embed.reconstruct_backbone_partially(b['pos_heavyatom'], g['R_new'], g['t_new'], g['aa_new'], b['chain_nb'], b['res_nb'],
                                                  b['mask_heavyatom'], b['generate_flag'], g['bb_table'], g['o_table'])
    assert max_abs(pos, g['pos_new']) < 1e-6 and torch.equal(m, g['mask_new'].bool())
    # the tables the product ships are the reference's
    import numpy as np, os
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ab_opt_amd', 'data', 'backbone_ideal.npz'))
    assert np.array_equal(d['bb_table'], g['bb_table'].numpy()) and np.array_equal(d['o_table'], g['o_table'].numpy())


def test_rank_commoness():
    g = load_golden('rank_commoness')
    structs = synth.hash_tensor((16, 36, 3), 55, scale=8.0)
    structs[3] = structs[5] + 0.01
    assert torch.equal(dpm.rank_commoness(structs, 5), g['rank'])


def test_dockq_vs_reference_scorer():
    """DockQ of docked candidates: Fnat / contact counts / interface residues from the reference's own `fnat` program (golden
    dockq_small, built from /root/reference/AbDock/DockQ/src), iRMS / LRMS / DockQ per DockQ.py:296-378."""
    from oracle import dockq as DQ
    g = load_golden('dockq_small')
    pos, mask, group, models = cases.dockq_case()
    for k in range(models.shape[0]):
        o = DQ.dockq(models[k].numpy(), mask.numpy(), pos.numpy(), mask.numpy(), group.numpy())
        assert (o['nat_correct'], o['nat_total']) == (int(g['nat_correct'][k]), int(g['nat_total'][k]))
        assert abs(o['fnat'] - g['fnat'][k].item()) < 1e-6
        assert abs(o['irms'] - g['irms'][k].item()) < 1e-9 and abs(o['Lrms'] - g['Lrms'][k].item()) < 1e-9
        assert abs(o['DockQ'] - g['DockQ'][k].item()) < 1e-6
    assert torch.equal(torch.from_numpy(o['interface']), g['interface'].bool())
    # when oracle/_ref/fnat is present (build container, or shipped prebuilt), re-run the reference program itself
    import os, subprocess, tempfile
    fnat_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'fnat')
    if os.path.exists(fnat_bin):
        with tempfile.TemporaryDirectory() as d:
            DQ.write_pdb(d + '/native.pdb', pos.numpy(), mask.numpy(), group.numpy())
            DQ.write_pdb(d + '/model.pdb', models[3].numpy(), mask.numpy(), group.numpy())
            r5 = DQ.parse_reference_fnat(subprocess.run([fnat_bin, d + '/model.pdb', d + '/native.pdb', '5', '-all'], capture_output=True, text=True).stdout)
        assert (r5['nat_correct'], r5['nat_total']) == (int(g['nat_correct'][3]), int(g['nat_total'][3]))


def test_dockq_superposition_two_algorithms():
    """iRMS / LRMS of the oracle are pinned by two independent published algorithms -- SVD-Kabsch with Biopython's reflection rule and
    the quaternion characteristic polynomial (Theobald 2005) -- on the regular candidates and on the corner cases (mirror images,
    3..4-residue interfaces, collinear CA atoms), and the corner-case fixture dockq_edge holds the agreed values."""
    import numpy as np
    from oracle import dockq as DQ
    g = load_golden('dockq_edge')
    pos, mask, group, models = cases.dockq_case()
    all_cases = [dict(name='regular', pos=pos, mask=mask, group=group, models=models, lrms_defined=True)] + cases.dockq_edge_cases()
    for c in all_cases:
        p, m, gr = c['pos'].numpy().astype(np.float64), c['mask'].numpy(), c['group'].numpy()
        for k in range(c['models'].shape[0]):
            y = c['models'][k].numpy().astype(np.float64)
            o = DQ.dockq(y, m, p, m, gr)
            both = m[:, 1] & (gr > 0)
            sel = o['interface'] & both
            q_rmsd, Rq, _ = DQ.qcp(p[sel, 1], y[sel, 1])
            assert abs(q_rmsd - o['irms']) < 2e-6, (c['name'], k, q_rmsd, o['irms'])
            if Rq is not None:
                assert abs(np.linalg.det(Rq) - 1) < 1e-9                      # proper rotation, also for mirror-image candidates
            if c['lrms_defined']:
                n1, n2 = (both & (gr == 1)).sum(), (both & (gr == 2)).sum()
                rec, lig = (1, 2) if n1 > n2 else (2, 1)
                rs, ls = both & (gr == rec), both & (gr == lig)
                _, Rr, tr = DQ.qcp(p[rs, 1], y[rs, 1])
                lr = np.sqrt((((y[ls, 1] @ Rr.T + tr) - p[ls, 1]) ** 2).sum(-1).mean())
                assert abs(lr - o['Lrms']) < 1e-6, (c['name'], k)
            if c['name'] != 'regular':
                row = g[c['name']][k]
                assert abs(o['fnat'] - row[0].item()) < 1e-9 and abs(o['irms'] - row[1].item()) < 1e-9 and o['n_interface'] == int(row[4])
                if c['lrms_defined']:
                    assert abs(o['Lrms'] - row[2].item()) < 1e-9 and abs(o['DockQ'] - row[3].item()) < 1e-9
    # a mirror image is NOT superimposable: the unconstrained SVD optimum (0) must not be what comes out
    mir = [c for c in all_cases if c['name'] == 'mirror'][0]
    assert g['mirror'][0][1].item() > 1.0
    assert 3 <= int(g['tiny_interface'][0][4]) <= 4


def test_encode_L256_values_and_gradients():
    """The oracle's encode() (values) and the plain torch statement of the two embedding modules under autograd (every parameter gradient)
    against the reference's `encode_L256` fixture: config-5 sample size, side chains, ragged second complex."""
    import plain_statement
    g = load_golden('encode_L256')
    m = synth.fresh_model(10, 3)
    with torch.no_grad():
        m.pair_embed.aapair_to_distcoef.weight.copy_(cases.encode_full_distcoef(m.pair_embed.aapair_to_distcoef.weight.shape))
    batch = cases.encode_full_batch()
    rf, pf, R0, _ = embed.encode(m.state_dict(), batch, True, True)
    sc = g['pair_feat_sub'].abs().max().item()
    assert max_abs(rf, g['res_feat']) < 1e-4 * g['res_feat'].abs().max().item() and max_abs(pf[:, ::9, ::7], g['pair_feat_sub']) < 1e-4 * sc
    assert max_abs(R0, g['R0']) < 1e-6
    m.zero_grad()
    with torch.enable_grad():
        rfg, pfg, _, _ = plain_statement.encode(m, {k: v.clone() for k, v in batch.items()}, True, True)
        w1, w2 = synth.hash_tensor(tuple(rfg.shape), 71, scale=1.0), synth.hash_tensor(tuple(pfg.shape), 72, scale=1.0)
        ((rfg * w1).sum() + (pfg * w2).sum()).backward()
    P = dict(m.named_parameters())
    checked = 0
    for k in g:
        if k.startswith('grad_'):
            name = k[len('grad_'):].replace('__', '.')
            got = P['residue_embed.mlp.0.weight'].grad[::4, ::7] if name.endswith('_sub') else P[name].grad
            assert max_abs(got, g[k]) <= 3e-4 * max(1e-6, g[k].abs().max().item()), name
            checked += 1
    assert checked == len(cases.ENCODE_FULL_PARAMS) + 1

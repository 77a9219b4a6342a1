"""Shared definitions of the parity cases: deterministic inputs regenerated on both sides
(golden generator in the build container, tests everywhere).  Only *outputs* are committed."""
import torch
from ab_opt_amd.utils import synth

MODEL_CFG_ABDOCK = dict(
    type='diffab', res_feat_dim=128, pair_feat_dim=64,
    diffusion=dict(num_steps=100, eps_net_opt=dict(num_layers=6), obj='pred_x0'),
    train_structure=True, train_sequence=False, num_bins=40, dist_min=0.5, dist_max=19.5,
)   # AbDock/configs/train/dock_single.yml:2-17


def cfg_abdock(num_steps=100, **over):
    c = {k: (dict(v) if isinstance(v, dict) else v) for k, v in MODEL_CFG_ABDOCK.items()}
    c['diffusion'] = dict(c['diffusion'], num_steps=num_steps, eps_net_opt=dict(num_layers=6))
    c.update(over)
    return c


def mask_from_lengths(lengths, L):
    return torch.stack([torch.arange(L) < n for n in lengths], 0)


def gen_from_ranges(N, L, ranges):
    g = torch.zeros(N, L, dtype=torch.bool)
    for a, b in ranges:
        g[:, a:b] = True
    return g


def ipa_inputs(N, L, lengths, salt=100, F=128, C=64):
    """R, t, x, z, mask for one GABlock call."""
    from oracle.geometry import so3_exp   # oracle used as input generator only inside tests/
    v = synth.hash_tensor((N, L, 3), salt + 0, scale=4.0)
    R = so3_exp(v)
    t = synth.hash_tensor((N, L, 3), salt + 1, scale=3.0)
    x = synth.hash_tensor((N, L, F), salt + 2, scale=2.0)
    z = synth.hash_tensor((N, L, L, C), salt + 3, scale=2.0)
    return R, t, x, z, mask_from_lengths(lengths, L)


def eps_inputs(N, L, lengths, gen_ranges, salt=200, F=128, C=64, t=37, num_steps=100):
    from oracle.dpm import variance_schedule
    v = synth.hash_tensor((N, L, 3), salt + 0, scale=4.0)
    p = synth.hash_tensor((N, L, 3), salt + 1, scale=3.0)
    s = (synth.hash_tensor((N, L), salt + 2) + 0.5).mul(21).long().clamp(0, 20)
    mres = mask_from_lengths(lengths, L)
    s = torch.where(mres, s, torch.full_like(s, 21))
    res_feat = synth.hash_tensor((N, L, F), salt + 3, scale=2.0)
    pair_feat = synth.hash_tensor((N, L, L, C), salt + 4, scale=2.0)
    beta = variance_schedule(num_steps)['betas'][t].expand([N]).clone()
    gen = gen_from_ranges(N, L, gen_ranges) & mres
    return v, p, s, res_feat, pair_feat, beta, gen, mres


SO3_EDGE_W = torch.tensor([
    [0.0, 0.0, 0.0], [1e-6, 0.0, 0.0], [1e-3, -2e-3, 5e-4], [0.3, -0.2, 0.1], [1.0, 2.0, -0.5],
    [3.1, 0.0, 0.0], [0.0, 3.14159, 0.0], [2.2, 2.2, 0.1], [-1.7, 0.4, 2.5], [0.05, 0.05, 0.05],
], dtype=torch.float32)


def reconstruct_batch():
    """Inputs of the reconstruct_small fixture (tests/golden/make_golden.py::case_reconstruct): ragged, two chains, a
    generated stretch across the chain break."""
    from ab_opt_amd.utils import synth
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=41, lengths=[40, 33])
    batch['chain_nb'][:, 22:] = 1
    batch['res_nb'][:, 22:] = batch['res_nb'][:, 22:] - 21
    batch['generate_flag'][:] = False
    batch['generate_flag'][:, 5:12] = True
    batch['generate_flag'][:, 20:25] = True
    batch['generate_flag'] &= batch['mask']
    return batch


def dockq_case(S=6, seed=31):
    """Native complex (64 antibody + 64 antigen residues, backbone + CB, a few residues with missing atoms / absent) and S docked
    candidates: the antibody moved rigidly by growing amounts plus coordinate noise.  -> native_pos (L,A,3), mask (L,A) bool,
    group (L,) int {0 absent, 1 antibody, 2 antigen}, model_pos (S,L,A,3); coordinates rounded to the PDB's 3 decimals."""
    from oracle.geometry import so3_exp
    c = synth.make_complex(synth.LAYOUT_128, seed=seed)
    pos, mask = c['pos_heavyatom'].clone(), c['mask_heavyatom'].clone()
    group = torch.where(c['fragment_type'] == 1, 1, torch.where(c['fragment_type'] == 3, 2, 0)).int()
    group[5] = 0; group[100] = 0                       # residues that are not in the files at all
    mask[7, 1] = False; mask[70, 4] = False            # a missing CA, a missing CB
    mask[group == 0] = False
    pos = (pos * 1000).round() / 1000
    ab = group == 1
    cen = pos[ab][:, 1].mean(0)
    models = []
    for k in range(S):
        rot = so3_exp(synth.hash_tensor((1, 3), 900 + k, scale=0.08 * k))[0]
        shift = synth.hash_tensor((3,), 950 + k, scale=1.5 * k)
        m = pos.clone()
        m[ab] = (pos[ab] - cen) @ rot.T + cen + shift + synth.hash_tensor(tuple(pos[ab].shape), 970 + k, scale=0.15 * k)
        models.append((m * 1000).round() / 1000)
    return pos, mask, group, torch.stack(models, 0)

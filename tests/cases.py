"""Shared definitions of the parity cases: deterministic inputs regenerated on both sides
(golden generator in the build container, tests everywhere).  Only *outputs* are committed."""
import torch
from ab_opt_amd.utils import synth

# the model configuration, masks and the denoiser inputs live in the package (bench.py and smoke() use them without the test tree)
from ab_opt_amd.utils.synth import MODEL_CFG_ABDOCK, cfg_abdock, mask_from_lengths, gen_from_ranges, eps_inputs  # noqa: F401,E402


def ipa_inputs(N, L, lengths, salt=100, F=128, C=64):
    """R, t, x, z, mask for one GABlock call."""
    from oracle.geometry import so3_exp   # oracle used as input generator only inside tests/
    v = synth.hash_tensor((N, L, 3), salt + 0, scale=4.0)
    R = so3_exp(v)
    t = synth.hash_tensor((N, L, 3), salt + 1, scale=3.0)
    x = synth.hash_tensor((N, L, F), salt + 2, scale=2.0)
    z = synth.hash_tensor((N, L, L, C), salt + 3, scale=2.0)
    return R, t, x, z, mask_from_lengths(lengths, L)


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


def dockq_edge_cases():
    """Superposition corner cases of the DockQ scorer, each a dict(name, pos, mask, group, models, lrms_defined):
      mirror          candidates that are MIRROR IMAGES of the native (plus a rigid motion): an unconstrained SVD would report RMSD 0;
                      Biopython's SVDSuperimposer flips the last singular vector (proper rotations only), and so must everything here
      tiny_interface  the antibody pulled away until only 3..4 residues are left in the 10 A interface: a (near-)planar point set
      collinear       CA-only chains laid end to end on one line: the covariance has rank 1, the optimal rotation is not unique
                      (iRMS and Fnat are still defined; LRMS is not -- the roll about the line is arbitrary -- and is not compared)"""
    import numpy as np
    from oracle import dockq as DQ
    pos, mask, group, models = dockq_case()
    out = []
    # ---- mirror images
    cen = pos[mask[:, 1] & (group > 0)][:, 1].mean(0)
    mir = (pos - cen) * torch.tensor([1.0, 1.0, -1.0]) + cen
    from oracle.geometry import so3_exp
    Q = so3_exp(torch.tensor([[0.9, -0.3, 0.5]]))[0]
    mods = torch.stack([mir, (mir - cen) @ Q.T + cen + torch.tensor([2.0, -1.0, 4.0]), (models[2] - cen) * torch.tensor([1.0, 1.0, -1.0]) + cen], 0)
    out.append(dict(name='mirror', pos=pos, mask=mask, group=group, models=((mods * 1000).round() / 1000), lrms_defined=True))
    # ---- 3..4 interface residues: translate the antibody along the centroid axis in 0.5 A steps until the interface is that small
    ab = group == 1
    axis = pos[ab][:, 1].mean(0) - pos[group == 2][:, 1].mean(0)
    axis = axis / axis.norm()
    npos = None
    for step in range(400):
        cand = pos.clone()
        cand[ab] = pos[ab] + 0.5 * step * axis
        cand = (cand * 1000).round() / 1000
        o = DQ.dockq(cand.numpy(), mask.numpy(), cand.numpy(), mask.numpy(), group.numpy())
        if 3 <= o['n_interface'] <= 4:
            npos = cand
            break
    assert npos is not None, 'no translation leaves a 3..4 residue interface'
    mods = []
    for k in range(3):
        m = npos.clone()
        rot = so3_exp(synth.hash_tensor((1, 3), 1900 + k, scale=0.1 * (k + 1)))[0]
        c2 = npos[ab][:, 1].mean(0)
        m[ab] = (npos[ab] - c2) @ rot.T + c2 + synth.hash_tensor((3,), 1950 + k, scale=0.8 * (k + 1)) + synth.hash_tensor(tuple(npos[ab].shape), 1970 + k, scale=0.2)
        mods.append((m * 1000).round() / 1000)
    out.append(dict(name='tiny_interface', pos=npos, mask=mask, group=group, models=torch.stack(mods, 0), lrms_defined=True))
    # ---- collinear CA-only chains, end to end on the x axis
    n1, n2, A = 12, 9, pos.shape[1]
    L = n1 + n2
    cpos = torch.zeros(L, A, 3)
    cpos[:, 1, 0] = 3.8 * torch.arange(L)
    cmask = torch.zeros(L, A, dtype=torch.bool)
    cmask[:, 1] = True
    cgroup = torch.cat([torch.full((n1,), 1), torch.full((n2,), 2)]).int()
    mods = []
    for k in range(3):
        m = cpos.clone()
        m[:n1, 1] += synth.hash_tensor((n1, 3), 2100 + k, scale=0.6 * (k + 1))            # chain 1 jittered off the line
        m[n1:, 1, 0] += 0.4 * k                                                               # chain 2 slid along it
        mods.append((m * 1000).round() / 1000)
    out.append(dict(name='collinear', pos=cpos, mask=cmask, group=cgroup, models=torch.stack(mods, 0), lrms_defined=False))
    return out


def encode_full_batch():
    """encode() at the size of BASELINE config 5's samples (L = 256, one full and one ragged complex, three chains, side chains on every
    other residue, a missing backbone atom here and there): the batch of the `encode_L256` fixture (reference forward values and the
    gradients of every embedding parameter)."""
    L = 256
    b = synth.make_batch(2, synth.LAYOUT_256, seed=98, lengths=[L, 201])
    b['pos_heavyatom'][:, :, 5:] = b['pos_heavyatom'][:, :, 1:2] + synth.hash_tensor((2, L, 10, 3), 43, scale=3.0)
    b['mask_heavyatom'][:, ::2, 5:12] = True
    b['mask_heavyatom'][:, ::6, 3] = False
    b['mask_heavyatom'] &= b['mask'][:, :, None]
    return b


def encode_full_distcoef(shape):
    """aapair_to_distcoef is zero-initialised by the reference (pair.py:28); a trained model's is not: hash values for the fixture."""
    return synth.hash_tensor(tuple(shape), 23, scale=2.0)


ENCODE_FULL_PARAMS = ('pair_embed.aa_pair_embed.weight', 'pair_embed.relpos_embed.weight', 'pair_embed.aapair_to_distcoef.weight',
                      'pair_embed.distance_embed.0.weight', 'pair_embed.distance_embed.0.bias', 'pair_embed.distance_embed.2.weight', 'pair_embed.distance_embed.2.bias',
                      'pair_embed.out_mlp.0.weight', 'pair_embed.out_mlp.0.bias', 'pair_embed.out_mlp.2.weight', 'pair_embed.out_mlp.2.bias',
                      'pair_embed.out_mlp.4.weight', 'pair_embed.out_mlp.4.bias',
                      'residue_embed.aatype_embed.weight', 'residue_embed.type_embed.weight', 'residue_embed.mlp.0.bias', 'residue_embed.mlp.2.weight',
                      'residue_embed.mlp.4.weight', 'residue_embed.mlp.6.weight', 'residue_embed.mlp.6.bias')

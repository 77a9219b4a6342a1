"""Plain torch statements of the training-path modules: TEST INFRASTRUCTURE (the yardstick the native autograd functions of
ab_opt_amd/training.py and ab_opt_amd/embed.py are checked against on the device, in float32 and float64).  Nothing in the product
imports this file; it lived inside the package until round 4.

Maths follows the reference line by line (D/ = AbDock/src/):
  GABlock.forward               D/modules/encoders/ga.py:149-178   (contractions as einsum instead of 5-D broadcast products)
  ResidueEmbedding.forward      D/modules/encoders/residue.py:26-92
  PairEmbedding.forward         D/modules/encoders/pair.py:37-101
  so3 / quaternion helpers      D/modules/common/so3.py:10-57, D/modules/common/geometry.py:215-233
"""
import math
import torch
import torch.nn.functional as F

H, D, P = 12, 32, 8
AA_UNK, ATOM_N, ATOM_CA, ATOM_C = 20, 0, 1, 2


def _hat(w):
    x, y, z = w.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack([o, z, -y, -z, o, x, y, -x, o], dim=-1).reshape(w.shape[:-1] + (3, 3))


def so3_exp(w):
    S = _hat(w)
    th = torch.linalg.norm(w, dim=-1)
    b = (torch.sin(th) + 1e-8) / (th + 1e-8)
    c = (1 - torch.cos(th) + 1e-8) / (th ** 2 + 2e-8)
    eye = torch.eye(3, dtype=w.dtype, device=w.device).expand(S.shape)
    return eye + b[..., None, None] * S + c[..., None, None] * (S @ S)


def quat1ijk_to_rot(e):
    b, c, d = e.unbind(-1)
    s = torch.sqrt(1 + b ** 2 + c ** 2 + d ** 2)
    a, b, c, d = 1 / s, b / s, c / s, d / s
    m = [a ** 2 + b ** 2 - c ** 2 - d ** 2, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c,
         2 * b * c + 2 * a * d, a ** 2 - b ** 2 + c ** 2 - d ** 2, 2 * c * d - 2 * a * b,
         2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a ** 2 - b ** 2 - c ** 2 + d ** 2]
    return torch.stack(m, -1).reshape(e.shape[:-1] + (3, 3))


def _to_global(R, t, p):        # p: (N, L, K, 3)
    return torch.einsum('nlab,nlkb->nlka', R, p) + t.unsqueeze(2)


def _to_local(R, t, q):
    return torch.einsum('nlba,nlkb->nlka', R, q - t.unsqueeze(2))


def _ln(x, mod):
    """layers.py:146-155: (x - mean) / sqrt(biased var + 1e-10) * gamma + beta."""
    return F.layer_norm(x, (x.shape[-1],), mod.gamma, mod.beta, eps=mod.epsilon)


def block_tail(blk, x, feat, mask):
    """ga.py:174-177 with the module's own nn.Linear / LayerNorm parameters."""
    u = blk.out_transform(feat)
    u = torch.where(mask.unsqueeze(-1), u, torch.zeros_like(u))
    y = _ln(x + u, blk.layer_norm_1)
    return _ln(y + blk.mlp_transition(y), blk.layer_norm_2)


def ga_block(blk, R, t, x, z, mask):
    """GABlock.forward (ga.py:149-178) as plain differentiable torch ops on the module's parameters."""
    N, L, _ = x.shape
    q = blk.proj_query(x).view(N, L, H, D)
    k = blk.proj_key(x).view(N, L, H, D)
    v = blk.proj_value(x).view(N, L, H, D)
    qp = _to_global(R, t, blk.proj_query_point(x).view(N, L, H * P, 3)).reshape(N, L, H, P * 3)
    kp = _to_global(R, t, blk.proj_key_point(x).view(N, L, H * P, 3)).reshape(N, L, H, P * 3)
    vp = _to_global(R, t, blk.proj_value_point(x).view(N, L, H * P, 3)).reshape(N, L, H, P, 3)
    l_node = torch.einsum('nihd,njhd->nijh', q, k) * (1 / math.sqrt(D))
    l_pair = blk.proj_pair_bias(z)
    d2 = (qp ** 2).sum(-1).unsqueeze(2) + (kp ** 2).sum(-1).unsqueeze(1) - 2 * torch.einsum('nihe,njhe->nijh', qp, kp)
    gamma = F.softplus(blk.spatial_coef)
    l_spat = d2 * ((-1 * gamma * math.sqrt(2 / (9 * P))) / 2)
    logits = (l_node + l_pair + l_spat) * math.sqrt(1 / 3)
    mrow = mask.view(N, L, 1, 1)
    mpair = mrow & mask.view(N, 1, L, 1)
    alpha = torch.softmax(torch.where(mpair, logits, logits - 1e5), dim=2)
    alpha = torch.where(mrow, alpha, torch.zeros_like(alpha))
    f_pair = torch.einsum('nijh,nijc->nihc', alpha, z).reshape(N, L, -1)
    f_node = torch.einsum('nijh,njhd->nihd', alpha, v).reshape(N, L, -1)
    agg = torch.einsum('nijh,njhpa->nihpa', alpha, vp).reshape(N, L, H * P, 3)
    loc = _to_local(R, t, agg)
    dist = loc.norm(dim=-1)
    direc = loc / (dist.unsqueeze(-1) + 1e-4)
    feat = torch.cat([f_pair, f_node, loc.reshape(N, L, -1), dist, direc.reshape(N, L, -1)], dim=-1)
    return block_tail(blk, x, feat, mask)


# ------------------------------------------------------------------ encode(): residue / pair embeddings
def _unit(v, eps=1e-6):
    return v / (torch.linalg.norm(v, ord=2, dim=-1, keepdim=True) + eps)


def construct_3d_basis(center, p1, p2):
    e1 = _unit(p1 - center)
    v2 = p2 - center
    e2 = _unit(v2 - (e1 * v2).sum(-1, keepdim=True) * e1)
    return torch.stack([e1, e2, torch.cross(e1, e2, dim=-1)], dim=-1)


def _dihedral(p0, p1, p2, p3):
    v0, v1, v2 = p2 - p1, p0 - p1, p3 - p2
    u1 = torch.cross(v0, v1, dim=-1)
    n1 = u1 / torch.linalg.norm(u1, dim=-1, keepdim=True)
    u2 = torch.cross(v0, v2, dim=-1)
    n2 = u2 / torch.linalg.norm(u2, dim=-1, keepdim=True)
    sgn = torch.sign((torch.cross(v1, v2, dim=-1) * v0).sum(-1))
    return torch.nan_to_num(sgn * torch.acos((n1 * n2).sum(-1).clamp(min=-0.999999, max=0.999999)))


def _backbone_dihedrals(pos, chain_nb, res_nb, mask):
    n, ca, c = pos[:, :, ATOM_N], pos[:, :, ATOM_CA], pos[:, :, ATOM_C]
    consec = ((res_nb[:, 1:] - res_nb[:, :-1]).abs() == 1) & (chain_nb[:, 1:] == chain_nb[:, :-1]) & mask[:, :-1]
    nterm, cterm = F.pad(~consec, pad=(1, 0), value=1), F.pad(~consec, pad=(0, 1), value=1)
    omega = F.pad(_dihedral(ca[:, :-1], c[:, :-1], n[:, 1:], ca[:, 1:]), pad=(1, 0), value=0)
    phi = F.pad(_dihedral(c[:, :-1], n[:, 1:], ca[:, 1:], c[:, 1:]), pad=(1, 0), value=0)
    psi = F.pad(_dihedral(n[:, :-1], ca[:, :-1], c[:, :-1], n[:, 1:]), pad=(0, 1), value=0)
    m = torch.stack([~nterm, ~nterm, ~cterm], dim=-1)
    return torch.stack([omega, phi, psi], dim=-1) * m, m


def residue_embedding(mod, aa, res_nb, chain_nb, pos_atoms, mask_atoms, fragment_type, hotspot=None, structure_mask=None, sequence_mask=None):
    """ResidueEmbedding.forward (residue.py:26-92) on the parameters of `mod` (an ab_opt_amd.embed.ResidueEmbedding), plain torch ops."""
    N, L = aa.size()
    A = mod.max_num_atoms
    mres = mask_atoms[:, :, ATOM_CA]
    pos, matom = pos_atoms[:, :, :A], mask_atoms[:, :, :A]
    if sequence_mask is not None:
        aa = torch.where(sequence_mask, aa, torch.full_like(aa, AA_UNK))
    f_aa = mod.aatype_embed(aa)
    R = construct_3d_basis(pos[:, :, ATOM_CA], pos[:, :, ATOM_C], pos[:, :, ATOM_N])
    rel = pos - pos[:, :, ATOM_CA].unsqueeze(2)
    crd = torch.matmul(R.transpose(-1, -2), rel.transpose(-1, -2)).transpose(-1, -2)       # R^T (x - t)
    crd = torch.where(matom[:, :, :, None], crd, torch.zeros_like(crd))
    slot = aa[:, :, None] == torch.arange(mod.max_aa_types, device=aa.device)[None, None, :]
    f_crd = (slot[:, :, :, None, None] * crd[:, :, None]).reshape(N, L, mod.max_aa_types * A * 3)
    if structure_mask is not None:
        f_crd = f_crd * structure_mask[:, :, None]
    dih, mdih = _backbone_dihedrals(pos, chain_nb, res_nb, mres)
    f_dih = (mod.dihed_embed(dih[:, :, :, None]) * mdih[:, :, :, None]).reshape(N, L, -1)
    if structure_mask is not None:
        dm = structure_mask & torch.roll(structure_mask, 1, 1) & torch.roll(structure_mask, -1, 1)
        f_dih = f_dih * dm[:, :, None]
    feats = [f_aa, f_crd, f_dih, mod.type_embed(fragment_type)]
    if mod.hotspot_embed is not None:
        hs = hotspot if hotspot is not None else torch.zeros_like(aa)
        feats.append(mod.hotspot_embed(hs))
    return mod.mlp(torch.cat(feats, dim=-1)) * mres[:, :, None]


def pair_embedding(mod, aa, res_nb, chain_nb, pos_atoms, mask_atoms, structure_mask=None, sequence_mask=None):
    """PairEmbedding.forward (pair.py:37-101) on the parameters of `mod` (an ab_opt_amd.embed.PairEmbedding), plain torch ops."""
    N, L = aa.size()
    A = mod.max_num_atoms
    pos, matom = pos_atoms[:, :, :A], mask_atoms[:, :, :A]
    mres = matom[:, :, ATOM_CA]
    mpair = mres[:, :, None] * mres[:, None, :]
    pstruct = structure_mask[:, :, None] * structure_mask[:, None, :] if structure_mask is not None else None
    if sequence_mask is not None:
        aa = torch.where(sequence_mask, aa, torch.full_like(aa, AA_UNK))
    T = mod.max_aa_types
    f_aap = mod.aa_pair_embed(aa[:, :, None] * T + aa[:, None, :])
    same = chain_nb[:, :, None] == chain_nb[:, None, :]
    rel = torch.clamp(res_nb[:, :, None] - res_nb[:, None, :], min=-mod.max_relpos, max=mod.max_relpos)
    f_rel = mod.relpos_embed(rel + mod.max_relpos) * same[:, :, :, None]
    d = (torch.linalg.norm(pos[:, :, None, :, None] - pos[:, None, :, None, :], dim=-1, ord=2) / 10).reshape(N, L, L, -1)
    c = F.softplus(mod.aapair_to_distcoef(aa[:, :, None] * T + aa[:, None, :]))
    gm = torch.exp(-1 * c * d ** 2) * (matom[:, :, None, :, None] * matom[:, None, :, None, :]).reshape(N, L, L, -1)
    f_dist = mod.distance_embed(gm)
    if pstruct is not None:
        f_dist = f_dist * pstruct[:, :, :, None]
    n, ca, cc = pos[:, :, ATOM_N], pos[:, :, ATOM_CA], pos[:, :, ATOM_C]
    ei = lambda a: a[:, :, None].expand(N, L, L, 3)
    ej = lambda a: a[:, None, :].expand(N, L, L, 3)
    dihed = torch.stack([_dihedral(ei(cc), ej(n), ej(ca), ej(cc)), _dihedral(ei(n), ei(ca), ei(cc), ej(n))], dim=-1)
    f_dih = mod.dihedral_embed(dihed)
    if pstruct is not None:
        f_dih = f_dih * pstruct[:, :, :, None]
    out = mod.out_mlp(torch.cat([f_aap, f_rel, f_dist, f_dih], dim=-1))
    return out * mpair[:, :, :, None]


def encode(model, batch, remove_structure, remove_sequence):
    """DiffusionAntibodyDesign.encode (diffab.py:39-83) through the plain statements above -> res_feat, pair_feat, R, p."""
    ctx = torch.logical_and(batch['mask_heavyatom'][:, :, ATOM_CA], ~batch['generate_flag'])
    sm = ctx if remove_structure else None
    qm = ctx if remove_sequence else None
    extra = {} if model.ABDOCK else dict(hotspot=batch.get('hotspot'))
    res_feat = residue_embedding(model.residue_embed, batch['aa'], batch['res_nb'], batch['chain_nb'], batch['pos_heavyatom'], batch['mask_heavyatom'],
                                 batch['fragment_type'], structure_mask=sm, sequence_mask=qm, **extra)
    pair_feat = pair_embedding(model.pair_embed, batch['aa'], batch['res_nb'], batch['chain_nb'], batch['pos_heavyatom'], batch['mask_heavyatom'],
                               structure_mask=sm, sequence_mask=qm)
    pos = batch['pos_heavyatom']
    R = construct_3d_basis(pos[:, :, ATOM_CA], pos[:, :, ATOM_C], pos[:, :, ATOM_N])
    return res_feat, pair_feat, R, pos[:, :, ATOM_CA]


# ------------------------------------------------------------------ sequence posterior of the training loss (transition.py:216-228)
def one_hot20(x):
    ok = (x >= 0) & (x < 20)
    return (F.one_hot(x.clamp(0, 19), 20) * ok[..., None]).float()


def posterior(alpha_bars, x_t, x_0, t):
    c_t = x_t if x_t.dim() == 3 else one_hot20(x_t)
    c_0 = x_0 if x_0.dim() == 3 else one_hot20(x_0)
    a = alpha_bars[t][:, None, None]
    th = ((a * c_t) + (1 - a) / 20) * ((a * c_0) + (1 - a) / 20)       # transition.py:223-224: alpha_bar_t in both factors
    return th / (th.sum(dim=-1, keepdim=True) + 1e-8)

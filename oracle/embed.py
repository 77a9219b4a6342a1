"""Oracle: encode() = ResidueEmbedding + PairEmbedding + frames (torch CPU fp32).  Test infra only.

Reference (D/ = /root/reference/AbDock/src/):
  DiffusionAntibodyDesign.encode   D/models/diffab.py:39-83
  ResidueEmbedding.forward         D/modules/encoders/residue.py:26-92
  PairEmbedding.forward            D/modules/encoders/pair.py:37-101
  AngularEncoding                  D/modules/common/layers.py:86-106
  backbone dihedrals / termini     D/modules/common/geometry.py:320-388, topology.py:5-24
"""
import torch
import torch.nn.functional as F
from . import geometry as G

AA_UNK, MAX_AA, MAX_RELPOS = 20, 22, 32
ATOM_N, ATOM_CA, ATOM_C = 0, 1, 2


def angular_encoding(x):
    """x (...,d) -> (..., d*13): [x, sin(x f), cos(x f)], f = 1,2,3,1,1/2,1/3."""
    fb = torch.FloatTensor([1, 2, 3] + [1., 1. / 2, 1. / 3])
    shp = list(x.shape[:-1]) + [-1]
    x = x.unsqueeze(-1)
    return torch.cat([x, torch.sin(x * fb), torch.cos(x * fb)], dim=-1).reshape(shp)


def backbone_dihedrals(pos, chain_nb, res_nb, mask):
    n, ca, c = pos[:, :, ATOM_N], pos[:, :, ATOM_CA], pos[:, :, ATOM_C]
    consec = ((res_nb[:, 1:] - res_nb[:, :-1]).abs() == 1) & (chain_nb[:, 1:] == chain_nb[:, :-1]) & mask[:, :-1]
    nterm = F.pad(~consec, pad=(1, 0), value=1)
    cterm = F.pad(~consec, pad=(0, 1), value=1)
    omega = F.pad(G.dihedral(ca[:, :-1], c[:, :-1], n[:, 1:], ca[:, 1:]), pad=(1, 0), value=0)
    phi = F.pad(G.dihedral(c[:, :-1], n[:, 1:], ca[:, 1:], c[:, 1:]), pad=(1, 0), value=0)
    psi = F.pad(G.dihedral(n[:, :-1], ca[:, :-1], c[:, :-1], n[:, 1:]), pad=(0, 1), value=0)
    m = torch.stack([~nterm, ~nterm, ~cterm], dim=-1)
    return torch.stack([omega, phi, psi], dim=-1) * m, m


def _mlp(sd, pre, x, idxs, last_relu=False):
    for n, i in enumerate(idxs):
        x = F.linear(x, sd[f'{pre}{i}.weight'], sd[f'{pre}{i}.bias'])
        if n + 1 < len(idxs) or last_relu:
            x = x.relu()
    return x


def residue_embedding(sd, pre, A, aa, res_nb, chain_nb, pos, matom, frag, structure_mask=None, sequence_mask=None):
    N, L = aa.shape
    mres = matom[:, :, ATOM_CA]
    pos, matom = pos[:, :, :A], matom[:, :, :A]
    if sequence_mask is not None:
        aa = torch.where(sequence_mask, aa, torch.full_like(aa, AA_UNK))
    f_aa = sd[pre + 'aatype_embed.weight'][aa]
    R = G.frames_from_backbone(pos[:, :, ATOM_CA], pos[:, :, ATOM_C], pos[:, :, ATOM_N])
    crd = G.to_local(R, pos[:, :, ATOM_CA], pos)
    crd = torch.where(matom[:, :, :, None].expand_as(crd), crd, torch.zeros_like(crd))
    slot = (aa[:, :, None] == torch.arange(MAX_AA)[None, None, :])          # (N,L,22)
    f_crd = torch.where(slot[:, :, :, None, None], crd[:, :, None], torch.zeros(1)).reshape(N, L, MAX_AA * A * 3)
    if structure_mask is not None:
        f_crd = f_crd * structure_mask[:, :, None]
    dih, mdih = backbone_dihedrals(pos, chain_nb, res_nb, mres)
    f_dih = (angular_encoding(dih[:, :, :, None]) * mdih[:, :, :, None]).reshape(N, L, -1)
    if structure_mask is not None:
        dm = structure_mask & torch.roll(structure_mask, 1, 1) & torch.roll(structure_mask, -1, 1)
        f_dih = f_dih * dm[:, :, None]
    f_type = sd[pre + 'type_embed.weight'][frag]
    out = _mlp(sd, pre + 'mlp.', torch.cat([f_aa, f_crd, f_dih, f_type], dim=-1), (0, 2, 4, 6))
    return out * mres[:, :, None]


def pair_embedding(sd, pre, A, aa, res_nb, chain_nb, pos, matom, structure_mask=None, sequence_mask=None):
    N, L = aa.shape
    pos, matom = pos[:, :, :A], matom[:, :, :A]
    mres = matom[:, :, ATOM_CA]
    mpair = mres[:, :, None] * mres[:, None, :]
    pstruct = structure_mask[:, :, None] * structure_mask[:, None, :] if structure_mask is not None else None
    if sequence_mask is not None:
        aa = torch.where(sequence_mask, aa, torch.full_like(aa, AA_UNK))
    aap = aa[:, :, None] * MAX_AA + aa[:, None, :]
    f_aap = sd[pre + 'aa_pair_embed.weight'][aap]
    same = chain_nb[:, :, None] == chain_nb[:, None, :]
    rel = torch.clamp(res_nb[:, :, None] - res_nb[:, None, :], min=-MAX_RELPOS, max=MAX_RELPOS)
    f_rel = sd[pre + 'relpos_embed.weight'][rel + MAX_RELPOS] * same[:, :, :, None]
    d = (torch.linalg.norm(pos[:, :, None, :, None] - pos[:, None, :, None, :], dim=-1, ord=2) / 10).reshape(N, L, L, -1)
    c = F.softplus(sd[pre + 'aapair_to_distcoef.weight'][aap])
    g = torch.exp(-1 * c * d ** 2)
    map_ = (matom[:, :, None, :, None] * matom[:, None, :, None, :]).reshape(N, L, L, -1)
    f_dist = _mlp(sd, pre + 'distance_embed.', g * map_, (0, 2), last_relu=True)
    if pstruct is not None:
        f_dist = f_dist * pstruct[:, :, :, None]
    n, ca, cc = pos[:, :, ATOM_N], pos[:, :, ATOM_CA], pos[:, :, ATOM_C]
    ex_i = lambda a: a[:, :, None].expand(N, L, L, 3)
    ex_j = lambda a: a[:, None, :].expand(N, L, L, 3)
    phi = G.dihedral(ex_i(cc), ex_j(n), ex_j(ca), ex_j(cc))
    psi = G.dihedral(ex_i(n), ex_i(ca), ex_i(cc), ex_j(n))
    f_dih = angular_encoding(torch.stack([phi, psi], dim=-1))
    if pstruct is not None:
        f_dih = f_dih * pstruct[:, :, :, None]
    out = _mlp(sd, pre + 'out_mlp.', torch.cat([f_aap, f_rel, f_dist, f_dih], dim=-1), (0, 2, 4))
    return out * mpair[:, :, :, None]


def encode(sd, batch, remove_structure, remove_sequence, A=15):
    """diffab.py:39-83 -> res_feat, pair_feat, R, p."""
    ctx = batch['mask_heavyatom'][:, :, ATOM_CA] & ~batch['generate_flag']
    sm = ctx if remove_structure else None
    qm = ctx if remove_sequence else None
    pos, matom = batch['pos_heavyatom'], batch['mask_heavyatom']
    res = residue_embedding(sd, 'residue_embed.', A, batch['aa'], batch['res_nb'], batch['chain_nb'], pos, matom,
                            batch['fragment_type'], sm, qm)
    pair = pair_embedding(sd, 'pair_embed.', A, batch['aa'], batch['res_nb'], batch['chain_nb'], pos, matom, sm, qm)
    R = G.frames_from_backbone(pos[:, :, ATOM_CA], pos[:, :, ATOM_C], pos[:, :, ATOM_N])
    return res, pair, R, pos[:, :, ATOM_CA]


def reconstruct_backbone_partially(pos_ctx, R_new, t_new, aa, chain_nb, res_nb, mask_atoms, mask_recons, bb_table, o_table):
    """D/modules/common/geometry.py:404-480: N, CA, C from the ideal local coordinates of the residue type, O through the psi
    frame (psi from the reconstructed backbone and the next residue's N), merged into the context where mask_recons."""
    N, L, A = mask_atoms.shape
    aa = aa.clamp(0, 20)
    bb = G.to_global(R_new, t_new, bb_table[aa])                                  # (N,L,3,3)
    dih, _ = backbone_dihedrals(bb, chain_nb, res_nb, mask_atoms[:, :, ATOM_CA])
    psi = dih[..., 2]
    s, c = torch.sin(psi), torch.cos(psi)
    o, z = torch.ones_like(s), torch.zeros_like(s)
    R_psi = torch.stack([o, z, z, z, c, -s, z, s, c], dim=-1).reshape(N, L, 3, 3)
    O = G.to_global(R_new @ R_psi, t_new, o_table[aa].unsqueeze(2))               # compose_chain([(R,t),(R_psi,0)])
    rec = F.pad(torch.cat([bb, O], dim=2), pad=(0, 0, 0, A - 4), value=0)
    pos_new = torch.where(mask_recons[:, :, None, None].expand_as(pos_ctx), rec, pos_ctx)
    mbb = torch.zeros_like(mask_atoms)
    mbb[:, :, :4] = True
    return pos_new, torch.where(mask_recons[:, :, None].expand_as(mask_atoms), mbb, mask_atoms)

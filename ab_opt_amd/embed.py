"""encode(): residue and pair embeddings.

Runs once per sample()/optimize() call (outside the denoising-steps/sec metric) and every iteration in training, always in
libabopt_hip.so: inference (`forward_hip`, csrc/embed.hip: one register-resident MFMA kernel for the pair embedding, a feature kernel +
MFMA GEMMs for the residue embedding) and training (`forward`: custom autograd functions on the same kernels -- the pair embedding with
its activation dump and register-chained backward, the residue features with bucketed row sums for the embedding tables, every dense
layer on abopt_gemm).  There is no torch restatement in the package and no CPU path; the plain statements the kernels are checked
against live in tests/plain_statement.py.  Module / parameter names mirror AbDock/src/modules/encoders/residue.py:9-92 and
pair.py:10-101 so checkpoints load strictly.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip

AA_UNK, ATOM_N, ATOM_CA, ATOM_C = 20, 0, 1, 2


# ------------------------------------------------------------------ autograd helpers for the L^2-sized training statement
# torch's generic backward for an embedding looked up at N*L*L indices (sort + segmented scatter) and for a Linear applied to
# N*L*L rows (one skinny GEMM with K = N*L*L) are the two slowest pieces of encode()'s backward; both have structure.
def _splitk_tn(a, b):
    """a^T @ b for tall a (M, I), b (M, J): abopt_gemm reads both operands k-strided in place, split-K partials summed in a fixed order."""
    return hip.gemm(a.t(), b.t())[0]


class _TallLinear(torch.autograd.Function):
    """y = x W^T + b (+ ReLU) for x with millions of rows; weight gradient through _splitk_tn."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        # bias and ReLU in the product's epilogue: one launch
        y = hip.gemm(x.reshape(-1, x.shape[-1]), weight, bias=bias, relu=relu)[0].view(x.shape[:-1] + (weight.shape[0],))
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.leaves = ((weight,), (bias,))
        return y

    @staticmethod
    def backward(ctx, dy):
        from .training import WgradGroup
        x, weight, y = ctx.saved_tensors
        if y is not None:
            dy = torch.ops.aten.threshold_backward(dy.contiguous(), y, 0.0)
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = hip.gemm(dy2, weight.t())[0].view_as(x)
        if dy2.shape[0] <= 65536:          # per-residue layers: in the backward pass's grouped launch; the per-pair layers (N L^2 rows) are HBM streams of their own
            dyc = dy2.contiguous()
            return dx, WgradGroup.product(dyc, x2, ctx.leaves[0]), WgradGroup.colsum(dyc, ctx.leaves[1]), None
        return dx, _splitk_tn(dy2, x2), hip.colsum(dy2), None


def _tall_mlp(seq, x):
    mods, i = list(seq), 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = _TallLinear.apply(x, m.weight, m.bias, fuse)
            i += 2 if fuse else 1
        else:
            x = m(x)
            i += 1
    return x


def embed_rows(mod, idx):
    """nn.Embedding forward.  Under autograd on the GPU the lookup is a one-hot matmul: its backward is one small GEMM instead of
    embedding_dense_backward's sort-and-scatter (123 us per call at 4096 rows, four calls per training step).  Same values; the
    padding row (if any) still receives no gradient."""
    w = mod.weight
    if not (torch.is_grad_enabled() and w.requires_grad and w.is_cuda):
        return mod(idx)
    oh = F.one_hot(idx, w.shape[0]).to(w.dtype)
    if mod.padding_idx is None:
        return oh @ w
    pad = mod.padding_idx
    keep = (torch.arange(w.shape[0], device=w.device) != pad).to(w.dtype)          # (no host scalar write: the step may be under graph capture)
    return (oh * keep) @ w + oh[..., pad:pad + 1] * w[pad].detach()


class _PairEmbedFn(torch.autograd.Function):
    """PairEmbedding on the training path.  Forward: the fused HIP kernel with its activation dump (`abopt_pair_embed_forward`:
    five 64-wide activation tiles, the Gaussian features g and T = dg / d softplus(coef)).  Backward: `abopt_pair_embed_backward`
    chains d(out) back through the five linears in registers and writes d loss / d pre-activation of every layer; the weight
    gradients are then tall library GEMMs against the saved activations (split over K), and the gradients of the two
    amino-acid-pair tables are structured sums (one-hot contractions).  Only the parameters carry gradients."""

    @staticmethod
    def forward(ctx, aa, res_nb, chain_nb, pos, matom, structure_mask, n_types, max_relpos,
                E_aap, E_rel, coef, freq, wd0, bd0, wd1, bd1, wo0, bo0, wo1, bo1, wo2, bo2):
        inp, keep = hip.encode_inputs(aa, res_nb, chain_nb, pos, matom, pos.shape[2], structure_mask=structure_mask)
        t = [x.detach().contiguous() for x in (E_aap, E_rel, coef, freq, wd0, bd0, wd1, bd1, wo0, bo0, wo1, bo1, wo2, bo2)]
        w = hip.PairEmbedWeights(*[hip.ptr(x, torch.float32) for x in t])
        out, acts, G, T = hip.pair_embed_forward(inp, w, save_activations=True)
        ctx.save_for_backward(aa, res_nb, chain_nb, pos, matom, acts, G, *t)
        ctx.structure_mask = structure_mask
        ctx.n_types, ctx.max_relpos = n_types, max_relpos
        return out

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dout):
        aa, res_nb, chain_nb, pos, matom, acts, G = ctx.saved_tensors[:7]
        t = list(ctx.saved_tensors[7:])
        E_aap, E_rel, coef, wo0 = t[0], t[1], t[2], t[8]
        N, L = aa.shape
        A = pos.shape[2]
        M, C, nt = N * L * L, dout.shape[-1], ctx.n_types
        inp, keep = hip.encode_inputs(aa, res_nb, chain_nb, pos, matom, A, structure_mask=ctx.structure_mask)
        w = hip.PairEmbedWeights(*[hip.ptr(x, torch.float32) for x in t])
        dys, ds, db = hip.pair_embed_backward(inp, w, dout, acts, colsum=True)     # T = dg / d softplus(coef) recomputed in the kernel; bias gradients from its per-wave sums
        y, a2 = dys.view(M, -1), acts.view(M, -1)
        do2, do1, do0, dh1, dh0 = (y[:, 64 * k:64 * (k + 1)] for k in range(5))
        h0, h1, dih, o0, o1 = a2[:, :64], a2[:, 64:128], a2[:, 128:154], a2[:, 160:224], a2[:, 224:288]
        dbo2, dbo1, dbo0, dbd1, dbd0 = (db[64 * k:64 * (k + 1)] for k in range(5))
        dwo2, dwo1, dwd1 = _splitk_tn(do2, o1), _splitk_tn(do1, o0), _splitk_tn(dh1, h0)
        # out_mlp.0 columns: [aa-pair embedding | relpos embedding x same-chain | f_dist | f_dih]
        # sum over (n, i, j) by the pair of residue types (a of i, b of j): per (n, i) the key rows j summed by type(j) -- read in place, do0 is a
        # column slice of dys --, then the (n, i) rows by type(i).  (Round 4 ran this as two one-hot batched products per operand through
        # hipBLASLt plus a 268 MB contiguous copy of do0 and two permuted copies: 0.45 ms per config-5 step.)
        torch._assert_async(((aa >= 0) & (aa < nt)).all())      # (F.one_hot, which this replaced, raises on a residue type outside the table; the kernels would skip the row)
        aa32 = aa.to(torch.int32).contiguous()
        pair_sum = lambda x2: hip.bucket_colsum(hip.segment_bucket_colsum(x2, N * L, aa32, L, nt).view(N * L, -1), aa32.view(-1), nt).view(nt * nt, -1)
        s_aap = pair_sum(do0)
        rel = torch.clamp(res_nb[:, :, None] - res_nb[:, None, :], min=-ctx.max_relpos, max=ctx.max_relpos) + ctx.max_relpos
        same = chain_nb[:, :, None] == chain_nb[:, None, :]
        # rows summed by relative-position bucket (other-chain pairs skipped): abopt_bucket_colsum, no one-hot matrix
        s_rel = hip.bucket_colsum(do0, torch.where(same, rel, -1).reshape(M).to(torch.int32), 2 * ctx.max_relpos + 1)
        # [f_dist | f_dih] are adjacent columns of the saved activations: one product (one pass over do0) for both column groups
        dwo0 = torch.cat([s_aap.t() @ E_aap, s_rel.t() @ E_rel, _splitk_tn(do0, a2[:, 64:154])], dim=1)
        dE_aap, dE_rel = s_aap @ wo0[:, :C], s_rel @ wo0[:, C:2 * C]
        unpad = lambda m: m.reshape(m.shape[0], A, 16)[:, :, :A].reshape(m.shape[0], A * A)          # [.., a, 16] -> [.., a*A + b]
        dwd0 = unpad(_splitk_tn(dh0, G.view(M, -1)))
        dcoef = unpad(pair_sum(ds.view(M, -1))) * torch.sigmoid(coef)
        return (None,) * 8 + (dE_aap, dE_rel, dcoef, None, dwd0, dbd0, dwd1, dbd1, dwo0, dbo0, dwo1, dbo1, dwo2, dbo2)


class _ResidueFeaturesFn(torch.autograd.Function):
    """The inputs of ResidueEmbedding's MLP on the training path: one HIP launch pair (`abopt_residue_features`) instead of the ~120
    elementwise kernels of the torch statement below (frames, per-type local coordinates, dihedrals, angular encoding, concatenation).
    Only the embedding tables carry gradients: a bucketed row sum of the matching feature columns each (`abopt_bucket_colsum`)."""

    @staticmethod
    def forward(ctx, aa, res_nb, chain_nb, pos, matom, fragment_type, hotspot, structure_mask, sequence_mask, n_atoms, pad_type, pad_hot,
                w_aa, w_type, w_hot, freq):
        inp, keep = hip.encode_inputs(aa, res_nb, chain_nb, pos, matom, n_atoms, fragment_type=fragment_type, hotspot=hotspot,
                                      structure_mask=structure_mask, sequence_mask=sequence_mask)
        t = [x.detach().contiguous() for x in (w_aa, w_type, freq)]
        hs = w_hot.detach().contiguous() if w_hot is not None else None
        w = hip.ResidueEmbedWeights(hip.ptr(t[0], torch.float32), hip.ptr(t[1], torch.float32), hip.ptr(hs, torch.float32, optional=True),
                                    hip.ptr(t[2], torch.float32), *([None] * 8))
        F_ = w_aa.shape[1]
        in_dim = F_ + w_aa.shape[0] * n_atoms * 3 + 39 + F_ + (F_ if w_hot is not None else 0)
        feat, R, p = hip.residue_features(inp, w, in_dim)
        aa_eff = aa if sequence_mask is None else torch.where(sequence_mask, aa, torch.full_like(aa, AA_UNK))
        hot = None if w_hot is None else (hotspot if hotspot is not None else torch.zeros_like(aa))
        ctx.save_for_backward(aa_eff, fragment_type, hot)
        ctx.dims = (w_aa.shape[0], w_type.shape[0], None if w_hot is None else w_hot.shape[0], F_, in_dim, pad_type, pad_hot)
        ctx.mark_non_differentiable(R, p)
        return feat, R, p

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dfeat, _dR=None, _dp=None):
        aa_eff, ftype, hot = ctx.saved_tensors
        n_aa, n_type, n_hot, F_, in_dim, pad_type, pad_hot = ctx.dims
        idx = lambda t: t.reshape(-1).to(torch.int32)
        d_aa = hip.bucket_colsum(dfeat[:, :F_], idx(aa_eff), n_aa)
        off = in_dim - F_ - (F_ if n_hot is not None else 0)
        d_type = hip.bucket_colsum(dfeat[:, off:off + F_], idx(ftype), n_type)
        if pad_type is not None:
            d_type[pad_type].zero_()                 # (no scalar index_put: the step may be under graph capture)
        d_hot = None
        if n_hot is not None:
            d_hot = hip.bucket_colsum(dfeat[:, off + F_:off + 2 * F_], idx(hot), n_hot)
            if pad_hot is not None:
                d_hot[pad_hot].zero_()
        return (None,) * 12 + (d_aa, d_type, d_hot, None)


class AngularEncoding(nn.Module):
    def __init__(self, num_funcs=3):
        super().__init__()
        self.num_funcs = num_funcs
        self.register_buffer('freq_bands', torch.FloatTensor([i + 1 for i in range(num_funcs)] + [1. / (i + 1) for i in range(num_funcs)]))

    def get_out_dim(self, in_dim):
        return in_dim * (1 + 2 * 2 * self.num_funcs)

    def forward(self, x):
        shape = list(x.shape[:-1]) + [-1]
        x = x.unsqueeze(-1)
        return torch.cat([x, torch.sin(x * self.freq_bands), torch.cos(x * self.freq_bands)], dim=-1).reshape(shape)


class ResidueEmbedding(nn.Module):

    def __init__(self, feat_dim, max_num_atoms, max_aa_types=22, hotspot=False):
        super().__init__()
        self.max_num_atoms, self.max_aa_types = max_num_atoms, max_aa_types
        self.aatype_embed = nn.Embedding(max_aa_types, feat_dim)
        self.dihed_embed = AngularEncoding()
        self.type_embed = nn.Embedding(10, feat_dim, padding_idx=0)
        infeat_dim = feat_dim + (max_aa_types * max_num_atoms * 3) + self.dihed_embed.get_out_dim(3) + feat_dim
        self.hotspot_embed = None
        if hotspot:      # AbDesign/diffab/modules/encoders/residue.py:19-21
            infeat_dim += feat_dim
            self.hotspot_embed = nn.Embedding(10, feat_dim, padding_idx=0)
        self.mlp = nn.Sequential(nn.Linear(infeat_dim, feat_dim * 2), nn.ReLU(), nn.Linear(feat_dim * 2, feat_dim), nn.ReLU(),
                                 nn.Linear(feat_dim, feat_dim), nn.ReLU(), nn.Linear(feat_dim, feat_dim))

    def _hip_weights(self):
        ps = [self.aatype_embed.weight, self.type_embed.weight, self.dihed_embed.freq_bands] + [m.weight for m in self.mlp if isinstance(m, nn.Linear)] + \
             [m.bias for m in self.mlp if isinstance(m, nn.Linear)] + ([self.hotspot_embed.weight] if self.hotspot_embed is not None else [])
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, '_hip_key', None) != key:
            lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
            t = [p.detach().contiguous() for p in (self.aatype_embed.weight, self.type_embed.weight, self.dihed_embed.freq_bands,
                                                   lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias, lin[3].weight, lin[3].bias)]
            hs = self.hotspot_embed.weight.detach().contiguous() if self.hotspot_embed is not None else None
            w = hip.ResidueEmbedWeights(hip.ptr(t[0], torch.float32), hip.ptr(t[1], torch.float32), hip.ptr(hs, torch.float32, optional=True),
                                        *[hip.ptr(x, torch.float32) for x in t[2:]])
            self._hip_key, self._hip_w, self._hip_keep = key, w, t + [hs]
        return self._hip_w

    def forward_hip(self, inp):
        """Inference path: abopt_residue_embed_forward -> res_feat (N,L,128), R (N,L,3,3), p = CA (N,L,3)."""
        return hip.residue_embed_forward(inp, self._hip_weights(), self.hotspot_embed is not None)

    def forward_with_frames(self, aa, res_nb, chain_nb, pos_atoms, mask_atoms, fragment_type, hotspot=None, structure_mask=None, sequence_mask=None):
        """Training path: (res_feat (N,L,F), R (N,L,3,3), p = CA (N,L,3)); the frames come out of the same feature kernel and carry no
        gradient (they are functions of the input coordinates only)."""
        if pos_atoms.requires_grad:
            raise NotImplementedError('ResidueEmbedding: gradients with respect to the input coordinates are not part of the training path')
        N, L = aa.size()
        mres = mask_atoms[:, :, ATOM_CA]
        x, R, p = _ResidueFeaturesFn.apply(aa, res_nb, chain_nb, pos_atoms.float(), mask_atoms, fragment_type, hotspot, structure_mask, sequence_mask, self.max_num_atoms,
                                           self.type_embed.padding_idx, None if self.hotspot_embed is None else self.hotspot_embed.padding_idx,
                                           self.aatype_embed.weight, self.type_embed.weight, None if self.hotspot_embed is None else self.hotspot_embed.weight,
                                           self.dihed_embed.freq_bands)
        return _tall_mlp(self.mlp, x.view(N, L, -1)) * mres[:, :, None], R.view(N, L, 3, 3), p.view(N, L, 3)

    def forward(self, aa, res_nb, chain_nb, pos_atoms, mask_atoms, fragment_type, hotspot=None, structure_mask=None, sequence_mask=None):
        """residue.py:26-92 -> (N,L,F).  One feature kernel (`abopt_residue_features`) + the four MLP layers on abopt_gemm, under autograd."""
        return self.forward_with_frames(aa, res_nb, chain_nb, pos_atoms, mask_atoms, fragment_type, hotspot, structure_mask, sequence_mask)[0]


class PairEmbedding(nn.Module):

    def __init__(self, feat_dim, max_num_atoms, max_aa_types=22, max_relpos=32):
        super().__init__()
        self.max_num_atoms, self.max_aa_types, self.max_relpos = max_num_atoms, max_aa_types, max_relpos
        self.aa_pair_embed = nn.Embedding(max_aa_types * max_aa_types, feat_dim)
        self.relpos_embed = nn.Embedding(2 * max_relpos + 1, feat_dim)
        self.aapair_to_distcoef = nn.Embedding(max_aa_types * max_aa_types, max_num_atoms * max_num_atoms)
        nn.init.zeros_(self.aapair_to_distcoef.weight)
        self.distance_embed = nn.Sequential(nn.Linear(max_num_atoms * max_num_atoms, feat_dim), nn.ReLU(),
                                            nn.Linear(feat_dim, feat_dim), nn.ReLU())
        self.dihedral_embed = AngularEncoding()
        infeat_dim = feat_dim * 3 + self.dihedral_embed.get_out_dim(2)
        self.out_mlp = nn.Sequential(nn.Linear(infeat_dim, feat_dim), nn.ReLU(), nn.Linear(feat_dim, feat_dim), nn.ReLU(),
                                     nn.Linear(feat_dim, feat_dim))

    def _hip_weights(self):
        lin = [m for m in list(self.distance_embed) + list(self.out_mlp) if isinstance(m, nn.Linear)]
        ps = [self.aa_pair_embed.weight, self.relpos_embed.weight, self.aapair_to_distcoef.weight, self.dihedral_embed.freq_bands]
        for m in lin:
            ps += [m.weight, m.bias]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, '_hip_key', None) != key:
            t = [p.detach().contiguous() for p in ps]
            self._hip_key, self._hip_w, self._hip_keep = key, hip.PairEmbedWeights(*[hip.ptr(x, torch.float32) for x in t]), t
        return self._hip_w

    def forward_hip(self, inp):
        """Inference path: abopt_pair_embed_forward -> pair_feat (N,L,L,64)."""
        return hip.pair_embed_forward(inp, self._hip_weights())

    def forward(self, aa, res_nb, chain_nb, pos_atoms, mask_atoms, structure_mask=None, sequence_mask=None):
        """pair.py:37-101 -> (N,L,L,C): fused HIP forward with its activation dump, register-chained backward + abopt_gemm weight gradients
        (_PairEmbedFn).  Only the parameters carry gradients."""
        if pos_atoms.requires_grad:
            raise NotImplementedError('PairEmbedding: gradients with respect to the input coordinates are not part of the training path')
        if self.aa_pair_embed.weight.shape[1] != 64:
            raise NotImplementedError('PairEmbedding: this build supports pair_feat_dim = 64 only')
        A = self.max_num_atoms
        pos, matom = pos_atoms[:, :, :A].float(), mask_atoms[:, :, :A]
        if sequence_mask is not None:
            aa = torch.where(sequence_mask, aa, torch.full_like(aa, AA_UNK))
        lin = [m for m in list(self.distance_embed) + list(self.out_mlp) if isinstance(m, nn.Linear)]
        return _PairEmbedFn.apply(aa, res_nb, chain_nb, pos, matom, structure_mask, self.max_aa_types, self.max_relpos,
                                  self.aa_pair_embed.weight, self.relpos_embed.weight, self.aapair_to_distcoef.weight,
                                  self.dihedral_embed.freq_bands, *[p for m in lin for p in (m.weight, m.bias)])

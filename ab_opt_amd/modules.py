"""Host-side mirror of the reference's module tree for the denoising path.

Same constructor arguments, attribute names, parameter/buffer names, shapes and registration order as
the reference (so `state_dict()` keys line up and reference checkpoints load strictly), but every
forward on the hot path is a call into libabopt_hip.so.  Reference files mirrored here
(D/ = AbDock/src/, A/ = AbDesign/diffab/):
  D/modules/common/layers.py:109-155  LayerNorm                 D/modules/common/nn.py:99-188  Linear inits, PerResiduePredictor
  D/modules/encoders/ga.py:40-193     GABlock, GAEncoder        D/modules/common/prmsd.py:19-47  pRMSDCa
  D/modules/common/so3.py:71-138      ApproxAngularDistribution D/modules/diffusion/transition.py  schedules / transitions
  D/modules/diffusion/dpm_full.py     EpsilonNet, FullDPM
"""
import math
import ctypes as C
import numpy as np
import torch
import torch.nn as nn

from . import hip

H, D, P = 12, 32, 8


class LayerNorm(nn.Module):
    """gamma/beta LayerNorm, eps=1e-10 inside the sqrt (layers.py:109-155).  Parameters only: the
    normalisation itself runs fused inside the HIP kernels."""

    def __init__(self, normal_shape, epsilon=1e-10):
        super().__init__()
        n = normal_shape if isinstance(normal_shape, int) else normal_shape[-1]
        self.normal_shape = torch.Size((n,))
        self.epsilon = epsilon
        self.gamma = nn.Parameter(torch.ones(n))
        self.beta = nn.Parameter(torch.zeros(n))


def _drop_packs(module, incompatible_keys):
    """load_state_dict post-hook (a module-level function, so modules that carry it stay picklable)."""
    module.invalidate_packed()


def _snapshot(params):
    """Cheap fingerprint of a parameter list: rebuilt packs when a tensor is replaced, moved or updated in place."""
    return tuple((p.data_ptr(), p._version, p.device) for p in params)


class GABlock(nn.Module):
    """Invariant point attention block (ga.py:40-178)."""

    def __init__(self, node_feat_dim, pair_feat_dim, value_dim=32, query_key_dim=32, num_query_points=8,
                 num_value_points=8, num_heads=12, bias=False):
        super().__init__()
        if (value_dim, query_key_dim, num_query_points, num_value_points, num_heads, bias) != (32, 32, 8, 8, 12, False):
            raise NotImplementedError('ab_opt_amd kernels are built for the shipped GABlock shape (12 heads, 32 ch, 8 points, no bias)')
        self.node_feat_dim, self.pair_feat_dim = node_feat_dim, pair_feat_dim
        self.value_dim, self.query_key_dim = value_dim, query_key_dim
        self.num_query_points, self.num_value_points, self.num_heads = num_query_points, num_value_points, num_heads
        self.proj_query = nn.Linear(node_feat_dim, query_key_dim * num_heads, bias=bias)
        self.proj_key = nn.Linear(node_feat_dim, query_key_dim * num_heads, bias=bias)
        self.proj_value = nn.Linear(node_feat_dim, value_dim * num_heads, bias=bias)
        self.proj_pair_bias = nn.Linear(pair_feat_dim, num_heads, bias=bias)
        self.spatial_coef = nn.Parameter(torch.full([1, 1, 1, num_heads], fill_value=np.log(np.exp(1.) - 1.)), requires_grad=True)
        self.proj_query_point = nn.Linear(node_feat_dim, num_query_points * num_heads * 3, bias=bias)
        self.proj_key_point = nn.Linear(node_feat_dim, num_query_points * num_heads * 3, bias=bias)
        self.proj_value_point = nn.Linear(node_feat_dim, num_value_points * num_heads * 3, bias=bias)
        self.out_transform = nn.Linear(
            in_features=(num_heads * pair_feat_dim) + (num_heads * value_dim) + (num_heads * num_value_points * (3 + 3 + 1)),
            out_features=node_feat_dim)
        self.layer_norm_1 = LayerNorm(node_feat_dim)
        self.mlp_transition = nn.Sequential(nn.Linear(node_feat_dim, node_feat_dim), nn.ReLU(),
                                            nn.Linear(node_feat_dim, node_feat_dim), nn.ReLU(),
                                            nn.Linear(node_feat_dim, node_feat_dim))
        self.layer_norm_2 = LayerNorm(node_feat_dim)
        self._pack = None

    def _sources(self):
        return [self.proj_query.weight, self.proj_key.weight, self.proj_value.weight, self.proj_query_point.weight,
                self.proj_key_point.weight, self.proj_value_point.weight, self.proj_pair_bias.weight, self.spatial_coef,
                self.out_transform.weight, self.out_transform.bias, self.layer_norm_1.gamma, self.layer_norm_1.beta,
                self.mlp_transition[0].weight, self.mlp_transition[0].bias, self.mlp_transition[2].weight, self.mlp_transition[2].bias,
                self.mlp_transition[4].weight, self.mlp_transition[4].bias, self.layer_norm_2.gamma, self.layer_norm_2.beta]

    @torch.no_grad()
    def packed(self):
        """Device tensors in kernel layout + the ctypes struct that points at them (cached)."""
        snap = _snapshot(self._sources())
        if self._pack is not None and self._pack[0] == snap:
            return self._pack[1], self._pack[2]
        f = lambda p: p.detach().float().contiguous()
        t = dict(
            w_node=torch.cat([f(self.proj_query.weight), f(self.proj_key.weight), f(self.proj_value.weight),
                              f(self.proj_query_point.weight), f(self.proj_key_point.weight), f(self.proj_value_point.weight)], 0).contiguous(),
            w_pair_bias=f(self.proj_pair_bias.weight), spatial_coef=f(self.spatial_coef).reshape(-1).contiguous(),
            w_out=f(self.out_transform.weight), b_out=f(self.out_transform.bias),
            ln1_gamma=f(self.layer_norm_1.gamma), ln1_beta=f(self.layer_norm_1.beta),
            w_mlp0=f(self.mlp_transition[0].weight), b_mlp0=f(self.mlp_transition[0].bias),
            w_mlp1=f(self.mlp_transition[2].weight), b_mlp1=f(self.mlp_transition[2].bias),
            w_mlp2=f(self.mlp_transition[4].weight), b_mlp2=f(self.mlp_transition[4].bias),
            ln2_gamma=f(self.layer_norm_2.gamma), ln2_beta=f(self.layer_norm_2.beta))
        if t['w_node'].is_cuda:
            t['w_node_frag'] = hip.pack_node_weights(t['w_node'])
            t['w_out_frag'], t['w_mlp_frag'] = hip.pack_tail_weights(t['w_out'], t['w_mlp0'], t['w_mlp1'], t['w_mlp2'])
            t['w_out_terms'] = hip.out_frag_terms(t['w_out_frag'])
        s = hip.ga_weights_struct(t)
        self._pack = (snap, t, s)
        return t, s

    def invalidate_packed(self):
        self._pack = None

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_pack'] = None
        return d

    @torch.no_grad()
    def forward(self, R, t, x, z, mask, return_parts=False):
        """(N,L,3,3), (N,L,3), (N,L,F), (N,L,L,C), (N,L) bool -> (N,L,F)   [ga.py:149-178]"""
        _, s = self.packed()
        return hip.ga_block_forward(s, R, t, x, z, mask, debug=return_parts)


class GAEncoder(nn.Module):

    def __init__(self, node_feat_dim, pair_feat_dim, num_layers, ga_block_opt={}):
        super().__init__()
        self.blocks = nn.ModuleList([GABlock(node_feat_dim, pair_feat_dim, **ga_block_opt) for _ in range(num_layers)])

    def packed_array(self):
        structs = [b.packed()[1] for b in self.blocks]
        arr = (hip.GaWeights * max(len(structs), 1))(*structs)
        return arr

    @torch.no_grad()
    def forward(self, R, t, res_feat, pair_feat, mask):
        return hip.ga_encoder_forward(self.packed_array(), len(self.blocks), R, t, res_feat, pair_feat, mask)


# ---- OpenFold-style initialisers used by the prmsd head (nn.py:41-96): kept so a seeded get_model() draws
# the same way the reference does.
def _trunc_normal_(w, scale):
    from scipy.stats import truncnorm
    fan_in = w.shape[1]
    std = math.sqrt(scale / max(1, fan_in)) / truncnorm.std(a=-2, b=2, loc=0, scale=1)
    samples = truncnorm.rvs(a=-2, b=2, loc=0, scale=std, size=w.numel())
    with torch.no_grad():
        w.copy_(torch.tensor(np.reshape(samples, w.shape), device=w.device))


class Linear(nn.Linear):
    def __init__(self, in_dim, out_dim, bias=True, init='default'):
        super().__init__(in_dim, out_dim, bias=bias)
        with torch.no_grad():
            if bias:
                self.bias.fill_(0)
            if init == 'default':
                _trunc_normal_(self.weight, 1.0)
            elif init == 'relu':
                _trunc_normal_(self.weight, 2.0)
            elif init == 'final':
                self.weight.fill_(0.0)
            else:
                raise ValueError('Invalid init string.')


class PerResidueRMSDCaPredictor(nn.Module):
    """nn.py:164-188 / prmsd.py:7-9 (parameters only; evaluated inside abopt_eps_net_forward)."""

    def __init__(self, no_bins, c_in, c_hidden):
        super().__init__()
        self.no_bins, self.c_in, self.c_hidden = no_bins, c_in, c_hidden
        self.layer_norm = LayerNorm(c_in)
        self.linear_1 = Linear(c_in, c_hidden, init='relu')
        self.linear_2 = Linear(c_hidden, c_hidden, init='relu')
        self.linear_3 = Linear(c_hidden, no_bins, init='final')
        self.relu = nn.ReLU()


def _pad_k(w, k_to):
    out = torch.zeros(w.shape[0], k_to, dtype=torch.float32, device=w.device)
    out[:, :w.shape[1]] = w
    return out


class EpsilonNet(nn.Module):
    """dpm_full.py:35-112.  `no_bins=None` builds the AbDesign variant (no prmsd head, A/...:33-102)."""

    def __init__(self, res_feat_dim, pair_feat_dim, num_layers, no_bins=None, encoder_opt={}):
        super().__init__()
        F = res_feat_dim
        self.current_sequence_embedding = nn.Embedding(25, F)
        self.res_feat_mixer = nn.Sequential(nn.Linear(F * 2, F), nn.ReLU(), nn.Linear(F, F))
        self.encoder = GAEncoder(F, pair_feat_dim, num_layers, **encoder_opt)
        self.eps_crd_net = nn.Sequential(nn.Linear(F + 3, F), nn.ReLU(), nn.Linear(F, F), nn.ReLU(), nn.Linear(F, 3))
        self.eps_rot_net = nn.Sequential(nn.Linear(F + 3, F), nn.ReLU(), nn.Linear(F, F), nn.ReLU(), nn.Linear(F, 3))
        self.eps_seq_net = nn.Sequential(nn.Linear(F + 3, F), nn.ReLU(), nn.Linear(F, F), nn.ReLU(), nn.Linear(F, 20), nn.Softmax(dim=-1))
        self.no_bins = no_bins
        if no_bins is not None:
            self.prmsd_predictor = PerResidueRMSDCaPredictor(no_bins, F + 3, F)
        self._pack = None
        # packed (kernel-layout) weight copies are keyed on (data_ptr, _version, device) of their sources; writes that bypass the
        # version counter (`p.data.copy_(ema)`) are invisible to that key, so the usual entry points drop the packs outright
        self.register_load_state_dict_post_hook(_drop_packs)

    def invalidate_packed(self):
        """Forget the kernel-layout weight copies (rebuilt at the next call).  Call after writing parameters through `.data`
        (EMA swaps); `load_state_dict`, `train()` and `eval()` call it for you."""
        self._pack = None
        for b in self.encoder.blocks:
            b.invalidate_packed()

    def train(self, mode=True):
        if bool(mode) != self.training:          # an actual train <-> eval switch (EMA swaps happen around these), not every .eval() call
            self.invalidate_packed()
        return super().train(mode)

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_pack'] = None                        # ctypes structs + device copies: rebuilt on first use
        return d

    def _sources(self):
        # every parameter outside the encoder (whose blocks fingerprint their own), in named_parameters() order, without walking the encoder's modules
        ps = list(self._parameters.values())
        for name, mod in self.named_children():
            if name != 'encoder':
                ps += list(mod.parameters())
        return [p for p in ps if p is not None]

    @torch.no_grad()
    def packed(self):
        arr = self.encoder.packed_array()
        snap = _snapshot(self._sources()) + tuple(b._pack[0] for b in self.encoder.blocks)        # (packed_array() has just verified / rebuilt every block's own fingerprint)
        if self._pack is not None and self._pack[0] == snap:
            return self._pack[2]
        f = lambda p: p.detach().float().contiguous()
        F = self.current_sequence_embedding.weight.shape[1]
        crd, rot, seq = self.eps_crd_net, self.eps_rot_net, self.eps_seq_net
        t = dict(seq_embed=f(self.current_sequence_embedding.weight),
                 w_mix0=f(self.res_feat_mixer[0].weight), b_mix0=f(self.res_feat_mixer[0].bias),
                 w_mix1=f(self.res_feat_mixer[2].weight), b_mix1=f(self.res_feat_mixer[2].bias),
                 w_head1=_pad_k(torch.cat([f(crd[0].weight), f(rot[0].weight), f(seq[0].weight)], 0), F + 4),
                 b_head1=torch.cat([f(crd[0].bias), f(rot[0].bias), f(seq[0].bias)], 0).contiguous(),
                 w_crd2=f(crd[2].weight), b_crd2=f(crd[2].bias), w_crd3=f(crd[4].weight), b_crd3=f(crd[4].bias),
                 w_rot2=f(rot[2].weight), b_rot2=f(rot[2].bias), w_rot3=f(rot[4].weight), b_rot3=f(rot[4].bias),
                 w_seq2=f(seq[2].weight), b_seq2=f(seq[2].bias), w_seq3=f(seq[4].weight), b_seq3=f(seq[4].bias))
        if self.no_bins is not None:
            pp = self.prmsd_predictor
            t.update(prmsd_ln_gamma=torch.cat([f(pp.layer_norm.gamma), torch.zeros(1, device=pp.layer_norm.gamma.device)]),
                     prmsd_ln_beta=torch.cat([f(pp.layer_norm.beta), torch.zeros(1, device=pp.layer_norm.beta.device)]),
                     w_prmsd1=_pad_k(f(pp.linear_1.weight), F + 4), b_prmsd1=f(pp.linear_1.bias),
                     w_prmsd2=f(pp.linear_2.weight), b_prmsd2=f(pp.linear_2.bias),
                     w_prmsd3=f(pp.linear_3.weight), b_prmsd3=f(pp.linear_3.bias))
        if t['w_head1'].is_cuda:
            t['w_heads_frag'] = hip.pack_heads_weights(t['w_head1'], t['w_crd2'], t['w_rot2'], t['w_seq2'], t['w_crd3'], t['w_rot3'], t['w_seq3'])
            t['w_mix_frag'] = hip.pack_mfma_operand(torch.cat([t['w_mix0'][:, :F], t['w_mix1']], 0).contiguous())
            t['mix_table'] = (t['seq_embed'] @ t['w_mix0'][:, F:].t() + t['b_mix0']).contiguous()
        ew = hip.EpsWeights()
        for name, typ in hip.EpsWeights._fields_:
            if name == 'blocks':
                ew.blocks = C.cast(arr, C.POINTER(hip.GaWeights))
            elif name == 'num_layers':
                ew.num_layers = len(self.encoder.blocks)
            elif name == 'num_bins':
                ew.num_bins = self.no_bins or 0
            else:
                setattr(ew, name, hip.ptr(t[name], torch.float32) if name in t else None)
        self._pack = (snap, (t, arr), ew)
        return ew

    @torch.no_grad()
    def packed_fp32(self):
        """The same weights WITHOUT the packed two-term fp16 operands: every dense layer then runs as an fp32 GEMM (fp32's range; the fallback of the
        range guard, include/abopt.h: abopt_nonfinite_flag).  Cached next to packed()."""
        ew = self.packed()
        if getattr(self, '_pack32', None) is not None and self._pack32[0] is ew:
            return self._pack32[2]
        n = len(self.encoder.blocks)
        arr = (hip.GaWeights * max(n, 1))()
        for i, b in enumerate(self.encoder.blocks):
            src = b.packed()[1]
            for name, _ in hip.GaWeights._fields_:
                setattr(arr[i], name, None if name in ('w_node_frag', 'w_out_frag', 'w_out_terms', 'w_mlp_frag') else getattr(src, name))
        plain = hip.EpsWeights()
        for name, _ in hip.EpsWeights._fields_:
            setattr(plain, name, None if name in ('w_heads_frag', 'w_mix_frag', 'mix_table') else getattr(ew, name))
        plain.blocks = C.cast(arr, C.POINTER(hip.GaWeights))
        self._pack32 = (ew, arr, plain)
        return plain

    @torch.no_grad()
    def forward(self, v_t, p_t, s_t, res_feat, pair_feat, beta, mask_generate, mask_res, grad_mode=False):
        """dpm_full.py:70-112 -> (v_next, R_next, eps_pos, c_denoised[, prmsd_logits])."""
        hip.nonfinite_flag(reset=True)
        o = hip.eps_net_forward(self.packed(), v_t, p_t, s_t, res_feat, pair_feat, beta, mask_generate, mask_res,
                                self.no_bins is not None, self.no_bins or 0, grad_mode)
        if hip.nonfinite_flag(reset=True):          # range guard of the two-term fp16 layers (include/abopt.h: abopt_nonfinite_flag): repeat on fp32 GEMMs
            import warnings
            warnings.warn('ab_opt_amd: a denoiser activation left the fp16 range (|x| >= 65504) or an input was not finite; EpsilonNet.forward is repeated '
                          'with the dense layers as fp32 GEMMs', RuntimeWarning, stacklevel=2)
            o = hip.eps_net_forward(self.packed_fp32(), v_t, p_t, s_t, res_feat, pair_feat, beta, mask_generate, mask_res,
                                    self.no_bins is not None, self.no_bins or 0, grad_mode)
            hip.nonfinite_flag(reset=True)
        if self.no_bins is not None:
            return o['v_next'], o['R_next'], o['eps_pos'], o['c'], o['prmsd_logits']
        return o['v_next'], o['R_next'], o['eps_pos'], o['c']


# ------------------------------------------------------------------------------ schedules / tables (init-time, host)
class VarianceSchedule(nn.Module):
    """Cosine schedule buffers (transition.py:10-34)."""

    def __init__(self, num_steps=100, s=0.01):
        super().__init__()
        T = num_steps
        t = torch.arange(0, num_steps + 1, dtype=torch.float)
        f_t = torch.cos((np.pi / 2) * ((t / T) + s) / (1 + s)) ** 2
        alpha_bars = f_t / f_t[0]
        betas = torch.cat([torch.zeros([1]), 1 - (alpha_bars[1:] / alpha_bars[:-1])], dim=0).clamp_max(0.999)
        sigmas = torch.zeros_like(betas)
        for i in range(1, betas.size(0)):
            sigmas[i] = ((1 - alpha_bars[i - 1]) / (1 - alpha_bars[i])) * betas[i]
        sigmas = torch.sqrt(sigmas)
        self.register_buffer('betas', betas)
        self.register_buffer('alpha_bars', alpha_bars)
        self.register_buffer('alphas', 1 - betas)
        self.register_buffer('sigmas', sigmas)
        self.register_buffer('sqrt_recip_alphas_cumprod', torch.sqrt(1. / alpha_bars))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', torch.sqrt(1. / alpha_bars - 1))


_IGSO3_CACHE = {}


class ApproxAngularDistribution(nn.Module):
    """IGSO(3) angle histograms (so3.py:71-109).  Built once on the host at construction exactly like the
    reference (same series, same fp32 ops); sampling from them happens inside abopt_denoise_step."""

    def __init__(self, stddevs, std_threshold=0.1, num_bins=8192, num_iters=1024):
        super().__init__()
        self.std_threshold, self.num_bins, self.num_iters = std_threshold, num_bins, num_iters
        self.register_buffer('stddevs', torch.FloatTensor(stddevs))
        self.register_buffer('approx_flag', self.stddevs <= std_threshold)
        key = (tuple(float(s) for s in stddevs), num_bins, num_iters)
        if key not in _IGSO3_CACHE:
            _IGSO3_CACHE[key] = self._histograms()
        X, Y = _IGSO3_CACHE[key]
        self.register_buffer('X', X.clone())
        self.register_buffer('Y', Y.clone())
        self._cdf = None

    def _histograms(self):
        # init-time host work; on many-core hosts torch's intra-op pool thrashes on these 8M-element ops
        nthr = torch.get_num_threads()
        torch.set_num_threads(min(nthr, 16))
        try:
            return self._histograms_impl()
        finally:
            torch.set_num_threads(nthr)

    def _histograms_impl(self):
        x = torch.linspace(0, math.pi, self.num_bins)
        l = torch.arange(0, self.num_iters)[None, :]
        X, Y = [], []
        for std in self.stddevs.tolist():
            xx = x[:, None]
            c = (1 - torch.cos(xx)) / math.pi
            a = (2 * l + 1) * torch.exp(-l * (l + 1) * (std ** 2))
            b = (torch.sin((l + 0.5) * xx) + 1e-6) / (torch.sin(xx / 2) + 1e-6)
            Y.append(torch.nan_to_num((c * a * b).sum(dim=1)).clamp_min(0))
            X.append(x)
        return torch.stack(X, 0), torch.stack(Y, 0)

    def cdf(self):
        """Normalised CDF over the first num_bins-1 cells of each row (what multinomial(Y[:, :-1]) samples), on Y's device."""
        if self._cdf is None or self._cdf[0] != (self.Y.data_ptr(), self.Y._version, self.Y.device):
            y = self.Y[:, :-1].double()
            tot = y.sum(1, keepdim=True)
            tot = torch.where(tot > 0, tot, torch.ones_like(tot))
            c = (torch.cumsum(y, 1) / tot).float().contiguous()
            self._cdf = ((self.Y.data_ptr(), self.Y._version, self.Y.device), c)
        return self._cdf[1]


class PositionTransition(nn.Module):
    def __init__(self, num_steps, var_sched_opt={}):
        super().__init__()
        self.var_sched = VarianceSchedule(num_steps, **var_sched_opt)


class RotationTransition(nn.Module):
    def __init__(self, num_steps, var_sched_opt={}, angular_distrib_fwd_opt={}, angular_distrib_inv_opt={}):
        super().__init__()
        self.var_sched = VarianceSchedule(num_steps, **var_sched_opt)
        c1 = torch.sqrt(1 - self.var_sched.alpha_bars)
        self.angular_distrib_fwd = ApproxAngularDistribution(c1.tolist(), **angular_distrib_fwd_opt)
        self.angular_distrib_inv = ApproxAngularDistribution(self.var_sched.sigmas.tolist(), **angular_distrib_inv_opt)
        self.register_buffer('_dummy', torch.empty([0, ]))


class AminoacidCategoricalTransition(nn.Module):
    def __init__(self, num_steps, num_classes=20, var_sched_opt={}):
        super().__init__()
        if num_classes != 20:
            raise NotImplementedError('kernels are built for 20 amino-acid classes')
        self.num_classes = num_classes
        self.var_sched = VarianceSchedule(num_steps, **var_sched_opt)


class DistanceToBins(nn.Module):
    """Only the one-hot flavour used by pRMSDCa (layers.py:18-58): carries the `offset` buffer."""

    def __init__(self, dist_min=0.0, dist_max=20.0, num_bins=64, use_onehot=True):
        super().__init__()
        assert use_onehot
        self.dist_min, self.dist_max, self.num_bins, self.use_onehot = dist_min, dist_max, num_bins, use_onehot
        self.register_buffer('offset', torch.linspace(dist_min, dist_max, num_bins))


class pRMSDCa(nn.Module):
    def __init__(self, num_bins=20, dist_min=0.5, dist_max=19.5):
        super().__init__()
        self.num_bins, self.dist_min, self.dist_max = num_bins, dist_min, dist_max
        self.tobin = DistanceToBins(dist_min=dist_min, dist_max=dist_max, num_bins=num_bins, use_onehot=True)

"""Training loss of the denoiser (FullDPM.forward) with gradients, on libabopt_hip.so.

The forward noising runs in the HIP kernel `abopt_add_noise`; every GABlock is three custom autograd functions over HIP entry points --
NativeLinear (the six projections as one product), IpaCore (`abopt_ipa_core_train_forward` / `abopt_ipa_pair_backward`: forward keeps
alpha (N,L,L,12) instead of the reference's (N,L,L,12,64) broadcast products, backward reads z once and writes dz once) and BlockTail
(the fused tail kernel with its activation dump, one row-local backward launch + library GEMMs) --, the heads' geometric epilogue and
the three losses with their gradients are one launch each (HeadsEpilogue, DpmLosses), and every dense product is `abopt_gemm`.  What is
still an ATen op: the AbDock flavour's prmsd / dist losses and the prmsd head's LayerNorm.
There is ONE implementation here: no torch restatement of the network and no CPU path (a CPU tensor reaches hip.ptr()'s error).  The
plain torch statements the native functions are checked against live in tests/plain_statement.py.

Maths follows the reference line by line (D/ = AbDock/src/):
  GABlock.forward               D/modules/encoders/ga.py:149-178
  EpsilonNet.forward            D/modules/diffusion/dpm_full.py:70-112
  FullDPM.forward               D/modules/diffusion/dpm_full.py:156-234 (AbDesign: A/modules/diffusion/dpm_full.py:138-191)
  rotation_matrix_cosine_loss   dpm_full.py:15-32;  calc_dist_loss :369-378;  pRMSDCa loss  D/modules/common/prmsd.py:49-70
"""
import torch
import torch.nn.functional as F

H, D, P, K_AA = 12, 32, 8, 20


class NativeLayerNorm(torch.autograd.Function):
    """layers.py:146-155 -- (x - mean) / sqrt(biased var + eps) * gamma + beta -- forward and backward as one launch each (csrc/rows.hip:
    row_layer_norm_kernel / _backward_kernel) plus two column sums for d gamma / d beta: the prmsd head's layer_norm (nn.py:179-188)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        from . import hip
        y, xhat, rstd = hip.layer_norm_forward(x, gamma, beta, eps)
        ctx.save_for_backward(xhat, rstd, gamma)
        return y

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dy):
        from . import hip
        xhat, rstd, gamma = ctx.saved_tensors
        dx, dg, db = hip.layer_norm_backward(dy.contiguous(), xhat, rstd, gamma)
        return dx, dg, db, None


def _ln(x, mod):
    return NativeLayerNorm.apply(x, mod.gamma, mod.beta, float(mod.epsilon))


# ------------------------------------------------------------------ weight gradients of a backward pass as grouped launches
import os as _os


class WgradGroup:
    """The weight-gradient products of backward (d W = d y^T x of every nn.Linear: `loss.backward()` of train.py:101-114) are leaves of the
    pass: nothing reads them before the optimizer.  Alone, each is a split-K launch with 2-64 output tiles plus a slab-sum launch (~44
    products, ~90 launches per config-5 step).  Here a product only QUEUES (operands + a fresh output tensor, which autograd's AccumulateGrad
    adopts without a kernel while `.grad is None`); the queue is flushed as one `abopt_gemm_tn_grouped` call (one product launch + one slab-sum
    launch, include/abopt.h) whenever it holds enough tiles to fill the chip -- about once per GABlock, while its operands are still in
    the cache -- and when the autograd engine finishes the pass (its final callback).  Same stream, same tile kernel; the K split is the
    group's, so a gradient differs from the ungrouped product's by fp32 summation order only.

    A product runs at once instead (hip.gemm) when its result is not a leaf's gradient (it feeds further autograd nodes), when the
    leaf already holds a gradient or has one queued in this pass (AccumulateGrad would ADD, reading the unfinished tensor), and in a process
    group of more than one rank (DistributedDataParallel's bucket hooks read gradients during backward), when the parameter is not contiguous
    (AccumulateGrad would clone the unfinished tensor), carries tensor / post-accumulate hooks (they would see it before it is computed), or under
    create_graph (grad mode on inside backward).

    LIMITATION: a queued parameter must receive its gradient of this pass from the queued product ALONE.  A parameter that ALSO enters the
    loss through an operation outside this module's functions (weight tying, a regulariser written with plain torch operations) gets a second
    gradient that the autograd engine adds to the queued tensor before it has been computed; the queue cannot see that use.  The models of this
    package use every parameter once per pass; for anything else set ABOPT_WGRAD_GROUP=0 (WgradGroup.enabled = False).  ABOPT_WGRAD_CHECK=1 verifies
    the assumption after every pass: each queued leaf's .grad must be the very tensor the group computed, else a RuntimeError names the parameter."""
    enabled = _os.environ.get('ABOPT_WGRAD_GROUP', '1') != '0'
    flush_tiles = int(_os.environ.get('ABOPT_WGRAD_FLUSH_TILES', '128'))      # (developer knob)
    check = _os.environ.get('ABOPT_WGRAD_CHECK', '0') == '1'
    _ones = {}             # (K, 1) ones per (device, K): a few KB each, NEVER evicted -- a captured training step bakes their addresses into its graph
    colsum_max_cols = int(_os.environ.get('ABOPT_WGRAD_COLSUM_COLS', '100000'))       # (developer knob: wider column sums keep their own kernels)
    _state = None          # the running backward pass: stream, queued (a, b, out), their tile count, ids of parameters with a queued gradient

    @classmethod
    def active(cls):
        if not cls.enabled:
            return False
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    @classmethod
    def flush(cls, close=False):
        st = cls._state
        if st is None:
            return
        if close:
            cls._state = None
        if st['items']:
            from . import hip
            items, st['items'], st['tiles'] = st['items'], [], 0
            with torch.cuda.stream(st['stream']):
                hip.gemm_tn_grouped([(a, b) for a, b, _ in items], outs=[c for _, _, c in items])
            # the engine has already ordered the caller's stream behind the backward nodes when its final callback runs: order it behind THIS launch too
            # when backward() was called under another stream than the one the nodes ran on
            cur = torch.cuda.current_stream()
            if cur != st['stream']:
                cur.wait_stream(st['stream'])
        if close:
            if cls.check:
                for p_, out_ in st['checked']:
                    if p_.grad is None or p_.grad.data_ptr() != out_.data_ptr():
                        raise RuntimeError('WgradGroup: a queued weight gradient was cloned or accumulated before it had been computed (parameter of shape '
                                           f'{tuple(p_.shape)}): this parameter enters the loss more than once per pass -- set ABOPT_WGRAD_GROUP=0')
            st['pending'].clear()
            st['checked'].clear()

    @classmethod
    def sync(cls):
        """Flush and close (a backward pass that raised never reaches its final callback; the next forward and the optimizer call this)."""
        cls.flush(close=True)

    @classmethod
    def colsum(cls, a, params):
        """Column sums of the tall (K, M) matrix a -- a bias gradient -- as the product a^T 1 riding in the group (one more 64 x 64 tile of a
        launch that exists anyway, instead of a partial-sum launch + a slab-sum launch of its own)."""
        from . import hip
        if not cls.active() or params is None or not all(p is not None and p.is_leaf and p.grad is None for p in params) or a.dim() != 2 or a.stride(1) != 1 \
                or a.shape[1] > cls.colsum_max_cols:
            return hip.colsum(a)
        key = (a.device, a.shape[0])
        ones = cls._ones.get(key)
        if ones is None:
            ones = torch.ones(a.shape[0], 1, dtype=torch.float32, device=a.device)
            if not torch.cuda.is_current_stream_capturing():        # (a tensor filled inside a capture holds nothing until the first replay: not cached)
                cls._ones[key] = ones
        return cls.product(a, ones, params).view(-1)

    @classmethod
    def product(cls, a, b, params):
        """a^T @ b for tall (K, M), (K, N) operands, the gradient of the leaf parameter(s) `params` (reached through view-only autograd nodes)."""
        from . import hip
        now = lambda: hip.gemm(a.t(), b.t())[0]
        if not cls.active() or params is None or not all(p is not None and p.is_leaf for p in params):
            return now()
        if a.dim() != 2 or b.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1 or a.dtype != torch.float32 or b.dtype != torch.float32:
            return now()
        main = torch.cuda.current_stream()
        st = cls._state
        if st is not None and st['stream'] != main:
            cls.flush(close=True)
            st = None
        if st is not None and any(id(p) in st['pending'] for p in params):
            cls.flush()                    # a weight used twice in one pass: AccumulateGrad adds the two results
            return now()
        if any(p.grad is not None for p in params):
            return now()                   # accumulation over micro-batches: the same add, onto a gradient of an earlier (closed) pass
        if torch.is_grad_enabled() or any((not p.is_contiguous()) or p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None) for p in params):
            return now()                   # create_graph / a layout AccumulateGrad would clone / hooks that would read the unfinished tensor
        if st is None:
            st = cls._state = dict(stream=main, items=[], tiles=0, pending=set(), checked=[])
            torch.autograd.Variable._execution_engine.queue_callback(cls.sync)
        out = torch.empty(a.shape[1], b.shape[1], dtype=torch.float32, device=a.device)
        # the queue keeps an ALIAS of the storage, not the tensor handed to autograd: AccumulateGrad adopts a gradient only if nobody else
        # holds a reference to the tensor object (otherwise it clones it -- here: before it has been computed)
        st['items'].append((a, b, out.detach()))
        st['tiles'] += ((a.shape[1] + 63) // 64) * ((b.shape[1] + 63) // 64)
        st['pending'].update(id(p) for p in params)
        if cls.check and len(params) == 1 and tuple(params[0].shape) == tuple(out.shape):
            st['checked'].append((params[0], out.detach()))
        if st['tiles'] >= cls.flush_tiles:
            cls.flush()
        return out


# ------------------------------------------------------------------ dense layers on the library's own GEMM (csrc/gemm.hip: gemm_batched_kernel)
class NativeLinear(torch.autograd.Function):
    """y = x W^T (+ b) with forward and both gradients on abopt_gemm: d x = d y W, d W = d y^T x (split-K over the rows, summed in a
    fixed order), d b = column sums.  Stands where the reference's nn.Linear modules call ATen under autograd (ga.py:54-66,
    dpm_full.py:39-59)."""

    @staticmethod
    def forward(ctx, x, w, b=None, relu=False, leaves=None):
        """relu=True: y = relu(x W^T + b) as ONE launch (bias and clamp in the product's epilogue); the backward masks d y with y > 0.
        leaves: the parameters `w` was assembled from when it is not one itself (ga_block's stacked projections) -- for WgradGroup."""
        from . import hip
        x2 = x.reshape(-1, x.shape[-1])
        y = hip.gemm(x2, w, bias=b, relu=relu)[0]
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.has_bias = b is not None
        ctx.leaves = tuple(leaves) if leaves is not None else (w,)
        ctx.bias_leaf = (b,) if b is not None else None
        return y.view(x.shape[:-1] + (w.shape[0],))

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dy):
        from . import hip
        x2, w, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dy2 = torch.ops.aten.threshold_backward(dy2.contiguous(), y, 0.0) if y is not None else dy2.contiguous()      # d relu: one kernel
        dx = hip.gemm(dy2, w.t())[0].view(dy.shape[:-1] + (w.shape[1],)) if ctx.needs_input_grad[0] else None
        dw = WgradGroup.product(dy2, x2, ctx.leaves)
        return dx, dw, (WgradGroup.colsum(dy2, ctx.bias_leaf) if ctx.has_bias else None), None, None


def _linear(mod, x, relu=False):
    """nn.Linear `mod` (followed by a ReLU if asked) through NativeLinear (a CPU tensor raises in the binding: there is no other path)."""
    return NativeLinear.apply(x, mod.weight, mod.bias, relu)


def _mlp(seq, x):
    """nn.Sequential of Linear / ReLU / Softmax modules with the Linear layers on NativeLinear (a Linear followed by a ReLU is one call)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Linear):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.ReLU)
            x = _linear(m, x, relu=fuse)
            i += 2 if fuse else 1
        else:
            x = m(x)
            i += 1
    return x


# ------------------------------------------------------------------ IPA core as an autograd function on the HIP kernels
DEFER_DZ = _os.environ.get('ABOPT_DZ_DEFER', '1') != '0'


class IpaCore(torch.autograd.Function):
    """feat = IPA(proj, z): ga.py:81-147 between the six projections and out_transform.

    proj (N,L,2016) = [q|k|v|q_pts|k_pts|v_pts] with the points still in the residue frames; R, t, mask carry no gradient
    (they come from the noised state)."""

    @staticmethod
    def forward(ctx, proj, z, R, t, mask, w_pair_bias, spatial_coef, pbc=None, zsink=None):
        from . import hip
        feat, alpha = hip.ipa_core_train_forward(proj, R, t, z, mask, w_pair_bias, spatial_coef.reshape(-1), pbc)
        ctx.save_for_backward(proj, z, R, t, w_pair_bias, spatial_coef, feat, alpha)
        ctx.row_leaves = (w_pair_bias, spatial_coef)               # the parameters whose gradients are column sums of per-row partials (WgradGroup)
        ctx.zsink = zsink
        if zsink is not None:
            zsink['users'] += 1
        return feat

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dfeat):
        from . import hip
        proj, z, R, t, Wb, gamma_raw, feat, alpha = ctx.saved_tensors
        N, L = proj.shape[:2]
        # head-major operands [q | q_pts global | 1], [k | k_pts | 1], [v | v_pts] for the batched GEMMs: one kernel
        Aq, Ak, Av = hip.ipa_backward_operands(proj, R, t)
        dfeat = dfeat.contiguous()
        # points epilogue backward (ga.py:136-139), head-major [d feat_node | d agg_pts] and delta_ih = <dfeat, feat>: one kernel
        dout_cat, delta = hip.ipa_points_backward(dfeat, feat, R, t)
        T = lambda a: a.transpose(-1, -2)
        b3 = lambda a: a.reshape(N * H, a.shape[-2], a.shape[-1])                   # (N,H,.,.) -> batch of N*H matrices (a view)
        mm = lambda a, b_nk: hip.gemm(b3(a), b3(b_nk)).view(N, H, a.shape[-2], b_nk.shape[-2])      # a @ b_nk^T, operands read in place
        da_node = mm(dout_cat, Av)                                                  # (N,H,L,L) = dout_cat Av^T
        wb_sum = lambda rows_: WgradGroup.colsum(rows_, ctx.row_leaves[:1])                       # d proj_pair_bias.weight: column sums of per-row partials, in the group
        # the z-streaming part (incl. d proj_pair_bias.weight).  The blocks of one encoder pass share ONE d pair_feat buffer (zsink): each
        # backward adds into it in the kernel and only the last one to run hands it to autograd -- instead of six 268 MB tensors and five
        # elementwise additions
        sink = ctx.zsink if (ctx.zsink is not None and ctx.zsink['users'] > 0) else None       # (a second backward over a retained graph: no sharing)
        if sink is not None:
            sink.setdefault('total', sink['users'])                # blocks of this encoder pass (the first backward to run sees them all)
        # abopt_ipa_dz_assemble sums the terms of at most 6 blocks (its LDS tile); deeper encoders (the library takes up to 8 layers) and
        # the developer A/B ABOPT_DZ_DEFER=0 use the round-3 form: every block adds into one shared buffer
        if sink is not None and (not DEFER_DZ or sink['total'] > 6):
            g, dz, dWb = hip.ipa_pair_backward(z, alpha, da_node, delta, dfeat, Wb, dz_into=sink['buf'], reduce=wb_sum)
            sink['buf'] = dz
            sink['users'] -= 1
            dz = None if sink['users'] > 0 else dz
            sink = None
        else:
            g, dz, dWb = hip.ipa_pair_backward(z, alpha, da_node, delta, dfeat, Wb, want_dz=sink is None, reduce=wb_sum)
        if sink is not None:
            # the blocks of one encoder pass leave (alpha, g, d feat, W_b) behind; the LAST backward to run sums all their d pair_feat terms
            # in one pass with one write (abopt_ipa_dz_assemble) -- each block adding into a shared buffer was a read-modify-write of
            # N L^2 C floats per block
            sink.setdefault('terms', []).append((alpha, g, dfeat, Wb))
            sink['users'] -= 1
            if sink['users'] == 0:
                terms = sink.pop('terms')
                dz = hip.ipa_dz_assemble([t_[0] for t_ in terms], [t_[1] for t_ in terms], [t_[2] for t_ in terms], [t_[3] for t_ in terms], z)
        del da_node
        # every (N,12,L,L) matrix is multiplied ONCE from each side (abopt_gemm reads the transposed views in place):
        P1 = mm(g, T(Ak))                                                           # sum_j g_ij [k_j | kg_j | 1]
        P2 = mm(T(g), T(Aq))                                                        # sum_i g_ij [q_i | qg_i | 1]
        P3 = mm(T(alpha), T(dout_cat))                                              # sum_i alpha_ij [dfn_i | dag_i]
        # scale, spatial-term chain rule, rotation back to the residue frames, re-layout to (N,L,2016): one kernel
        dproj, e = hip.ipa_backward_assemble(P1, P2, P3, Aq, Ak, R, gamma_raw.reshape(-1))
        dgamma = WgradGroup.colsum(e.reshape(-1, e.shape[-1]), ctx.row_leaves[1:]).reshape(gamma_raw.shape)      # the kernel applies d(-softplus(x) sqrt(2/(9P))/2)/dx
        return dproj, dz, None, None, None, dWb, dgamma, None, None


class BlockTail(torch.autograd.Function):
    """out_transform -> mask -> +x -> LayerNorm -> mlp_transition -> +y -> LayerNorm (ga.py:174-177) as ONE forward launch
    (csrc/mlp.hip: out_ln_mlp_kernel with its activation dump) and, backward, one launch for everything row-local
    (tail_backward_kernel) plus five library GEMMs for d feat and the four weight gradients."""

    @staticmethod
    def forward(ctx, x, feat, mask, w_out, b_out, g1, be1, w0, b0, w1, b1, w2, b2, g2, be2):
        from . import hip
        shape = x.shape
        wof, wmf, wmt = hip.pack_tail_weights(w_out, w0, w1, w2, transposed=True)
        feat2 = feat.reshape(-1, feat.shape[-1]).contiguous()
        out, saved = hip.block_tail_forward(feat2, wof, wmf, x.reshape(-1, 128), b_out, mask.reshape(-1), g1, be1, b0, b1, b2, g2, be2, save=True)
        ctx.save_for_backward(feat2, mask, saved, wmt, w_out, g1, g2)
        ctx.leaves = (w_out, w0, w1, w2)
        ctx.row_leaves = (b_out, g1, be1, b0, b1, b2, g2, be2)
        return out.view(shape)

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dout):
        from . import hip
        feat2, mask, saved, wmt, w_out, g1, g2 = ctx.saved_tensors
        rl = ctx.row_leaves
        dpre, da1, du, cs = hip.block_tail_backward(dout.reshape(-1, 128), saved, wmt, mask.reshape(-1), g1, g2,
                                                    reduce=lambda part_: WgradGroup.colsum(part_, rl))      # bias / LayerNorm gradients: in the group
        dfeat = hip.gemm(du, w_out.t())[0].view(dout.shape[:-1] + (w_out.shape[1],))
        lv = ctx.leaves
        dw_out = WgradGroup.product(du, feat2, lv[:1])
        dw0, dw1, dw2 = (WgradGroup.product(dpre[l], saved[1 + l], lv[1 + l:2 + l]) for l in range(3))      # d W_l = d pre_l^T . input_l
        #      x               feat   mask  w_out   b_out  g1     be1    w0   b0     w1   b1     w2   b2     g2     be2
        return da1.view(dout.shape), dfeat, None, dw_out, cs[7], cs[6], cs[5], dw0, cs[4], dw1, cs[3], dw2, cs[2], cs[1], cs[0]


def _block_tail(blk, x, feat, mask):
    m = blk.mlp_transition
    return BlockTail.apply(x, feat, mask, blk.out_transform.weight, blk.out_transform.bias, blk.layer_norm_1.gamma, blk.layer_norm_1.beta,
                           m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias, blk.layer_norm_2.gamma, blk.layer_norm_2.beta)


# ------------------------------------------------------------------ network
def ga_block(blk, R, t, x, z, mask, pbc=None, zsink=None):
    """GABlock.forward under autograd (ga.py:149-178).  pbc: this block's slice of hip.pair_bias_cache_layers (forward-only shortcut: the
    core reads proj_pair_bias(z) instead of recomputing it; gradients of z and the weight still come from the backward kernel)."""
    leaves = (blk.proj_query.weight, blk.proj_key.weight, blk.proj_value.weight,
              blk.proj_query_point.weight, blk.proj_key_point.weight, blk.proj_value_point.weight)
    w_node = torch.cat(leaves, dim=0)
    feat = IpaCore.apply(NativeLinear.apply(x, w_node, None, False, leaves), z, R.detach(), t.detach(), mask, blk.proj_pair_bias.weight, blk.spatial_coef, pbc, zsink)
    return _block_tail(blk, x, feat, mask)


class HeadsEpilogue(torch.autograd.Function):
    """eps_pos = gen ? R eps_crd : 0 and R_next = R U(eps_rot) (dpm_full.py:95-101) as one launch forward and one backward
    (csrc/rows.hip: heads_epilogue_kernel / heads_epilogue_backward_kernel) instead of ~45 + ~90 elementwise kernels; R carries no gradient."""

    @staticmethod
    def forward(ctx, R, eps_crd, eps_rot, mask_generate):
        from . import hip
        R_next, eps_pos = hip.heads_epilogue_forward(R, eps_crd, eps_rot, mask_generate)
        ctx.save_for_backward(R, eps_rot, mask_generate)
        return R_next, eps_pos

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dR_next, deps_pos):
        from . import hip
        R, eps_rot, mask_generate = ctx.saved_tensors
        d_crd, d_rot = hip.heads_epilogue_backward(R, eps_rot, mask_generate, dR_next, deps_pos)
        return None, d_crd, d_rot, None


def eps_net(net, v_t, p_t, s_t, res_feat, pair_feat, beta, mask_generate, mask_res, want_v=True):
    """EpsilonNet.forward under autograd (dpm_full.py:70-112).  The noised state (v_t, p_t, s_t) carries no gradient, as in the reference's
    training loop.  want_v=False skips v_next = log(R_next) (the training losses use R_next only); when asked for, it is returned detached."""
    from . import hip
    from .embed import embed_rows
    if v_t.requires_grad:
        raise NotImplementedError('eps_net: gradients with respect to the noised orientations are not part of the training path (dpm_full.py:162-178 noises without grad)')
    N, L = mask_res.shape
    R = hip.so3_exp(v_t.detach().float())
    x = _mlp(net.res_feat_mixer, torch.cat([res_feat, embed_rows(net.current_sequence_embedding, s_t)], dim=-1))
    # proj_pair_bias(pair_feat) of all blocks in one pass over pair_feat (the sampler's per-call cache, rebuilt every training step)
    caches = hip.pair_bias_cache_layers([blk.proj_pair_bias.weight for blk in net.encoder.blocks], pair_feat.detach())
    zsink = dict(buf=None, users=0) if pair_feat.requires_grad else None
    for blk, pbc in zip(net.encoder.blocks, caches):
        x = ga_block(blk, R, p_t, x, pair_feat, mask_res, pbc=pbc, zsink=zsink)
    temb = torch.stack([beta, torch.sin(beta), torch.cos(beta)], dim=-1)[:, None, :].expand(N, L, 3)
    feat = torch.cat([x, temb], dim=-1)
    R_next, eps_pos = HeadsEpilogue.apply(R, _mlp(net.eps_crd_net, feat), _mlp(net.eps_rot_net, feat), mask_generate)
    v_next = None
    if want_v:
        gen3 = mask_generate[:, :, None].expand(N, L, 3)
        v_next = torch.where(gen3, hip.so3_log(R_next.detach(), grad_mode=torch.is_grad_enabled()), v_t)
    c = _mlp(net.eps_seq_net, feat)
    if net.no_bins is None:
        return v_next, R_next, eps_pos, c
    pp = net.prmsd_predictor
    h = _linear(pp.linear_3, _linear(pp.linear_2, _linear(pp.linear_1, _ln(feat, pp.layer_norm), relu=True), relu=True))
    return v_next, R_next, eps_pos, c, h.mean(dim=1)


# ------------------------------------------------------------------ losses
class AbdockLosses(torch.autograd.Function):
    """-> (prmsd, dist) of the AbDock flavour (dpm_full.py:180-198: pRMSDCa cross entropy on the binned, detached RMSD of the predicted
    positions, and calc_dist_loss on the predicted positions when obj = pred_x0) with their gradients in ONE launch (csrc/rows.hip:
    abdock_losses_kernel) plus the reductions over the N samples."""

    @staticmethod
    def forward(ctx, prmsd_logits, p_pred, p0n, coef_a, coef_b, mask_generate, mask_res, offsets, scale, pred_x0):
        from . import hip
        part, gl, gp = hip.abdock_losses(prmsd_logits, p_pred, p0n, coef_a, coef_b, mask_generate, mask_res, offsets, scale, pred_x0)
        tot = part.sum(0)                                          # {sum err (unused), sum m0, sum sl1, count}
        wn = part[:, 1] / (tot[1] + 1e-10)
        prmsd = (part[:, 0] * wn).sum()
        dist = tot[2] / tot[3] if pred_x0 else tot[2] * 0.0
        ctx.save_for_backward(gl * wn[:, None], gp / tot[3] if pred_x0 else gp)
        return prmsd, dist

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dprmsd, ddist):
        gl, gp = ctx.saved_tensors
        return gl * dprmsd, gp * ddist, None, None, None, None, None, None, None, None


class DpmLosses(torch.autograd.Function):
    """(rot, pos, seq) sums over the generated residues (dpm_full.py:199-231) and their gradients in ONE launch (csrc/rows.hip:
    dpm_losses_kernel) instead of the ~50 forward and ~70 backward elementwise kernels of the statement in fulldpm_loss below."""

    @staticmethod
    def forward(ctx, R_pred, R_0, p_pred, p_target, c_den, s_t, s_0, abar_t, mask_generate):
        from . import hip
        sums, gR, gp, gc = hip.dpm_losses(R_pred, R_0, p_pred, p_target, c_den, s_t, s_0, abar_t, mask_generate)
        ctx.save_for_backward(gR, gp, gc)
        return sums

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dsums):
        gR, gp, gc = ctx.saved_tensors
        return gR * dsums[0], None, gp * dsums[1], None, gc * dsums[2], None, None, None, None


def fulldpm_loss(dpm, v_0, p_0, s_0, res_feat, pair_feat, mask_generate, mask_res, denoise_structure, denoise_sequence,
                 t=None, noise=None, seed=None):
    """FullDPM.forward -> dict of scalar losses (AbDock: prmsd, dist (pred_x0), rot, pos, seq; AbDesign: rot, pos, seq)."""
    from . import hip
    hip.lib()
    WgradGroup.sync()
    N, L = res_feat.shape[:2]
    dev = res_feat.device
    vs = dpm.trans_pos.var_sched
    if t is None:
        t = torch.randint(0, dpm.num_steps, (N,), dtype=torch.long, device=dev)
    h = dpm._sched_host()
    # GraphedTrainStep: the Philox position comes from device memory (replayable).  Only while that object runs its own step (its
    # warm-up, capture): an eager model(batch) in between -- a validation pass -- draws a fresh seed, or takes the caller's, as always
    seed_dev = getattr(dpm, '_train_seed_dev', None) if seed is None else None
    seed = 0 if seed_dev is not None else (dpm._new_seed() if seed is None else int(seed))
    grad_mode = torch.is_grad_enabled()         # so3.py:12-16: the log map clamps at -0.999 under autograd, -1.0 in no_grad validation passes
    with torch.no_grad():                       # noising has no learnable parameters; native kernel (transition.py:62-78,120-144,179-200)
        v_n, p_n_ang, s_n, eps_p = hip.add_noise(t, vs.alpha_bars, dpm.trans_rot.angular_distrib_fwd, noise, seed, 0,
                                                 v_0.detach().float(), p_0.detach().float(), s_0, mask_generate, h['scale'], h['mean'],
                                                 noise_structure=denoise_structure, noise_sequence=denoise_sequence, grad_mode=grad_mode, want_eps=True,
                                                 seed_dev=seed_dev if noise is None else None)
    p0n = dpm._normalize_position(p_0)
    p_n = dpm._normalize_position(p_n_ang)
    R_0 = hip.so3_exp(v_0.detach().float())          # frames of the input structure: no gradient (dpm_full.py:160-161 builds them from the batch)
    beta = vs.betas[t]
    out = eps_net(dpm.eps_net, v_n, p_n, s_n, res_feat, pair_feat, beta, mask_generate, mask_res, want_v=False)
    v_pred, R_pred, p_pred, c_den = out[:4]
    genf = mask_generate.float()
    denom = genf.sum() + 1e-8
    loss = {}
    if dpm.abdock:
        x0 = dpm.obj == 'pred_x0'
        p_true = p0n if x0 else p_n
        # prmsd (pRMSDCa on the detached RMSD of the predicted positions, prmsd.py:49-70) and, for pred_x0, the dist loss (dpm_full.py:369-378)
        ca = None if x0 else vs.sqrt_recip_alphas_cumprod[t]
        cb = None if x0 else vs.sqrt_recipm1_alphas_cumprod[t]
        loss['prmsd'], dist = AbdockLosses.apply(out[4], p_pred, p0n.detach(), ca, cb, mask_generate, mask_res, dpm.prmsd.tobin.offset.reshape(-1), h['scale'], x0)
        if x0:
            loss['dist'] = dist
        pos_target = p_true
    else:
        pos_target = eps_p
    # rot (cosine-embedding loss on the matrix columns, dpm_full.py:15-32,199-204), pos (MSE, :206-208) and seq (KL of the posteriors,
    # :210-231) summed over the generated residues, with their gradients, in one launch
    sums = DpmLosses.apply(R_pred, R_0, p_pred, pos_target.detach(), c_den, s_n, s_0, vs.alpha_bars[t], mask_generate) / denom
    loss['rot'], loss['pos'], loss['seq'] = sums[0], sums[1], sums[2]
    return loss


# ------------------------------------------------------------------ the whole training step as one hipGraph
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (no amsgrad / maximize) with the gradient clipping of the reference's training loop folded in, as a handful of HIP
    launches for the whole parameter list (csrc/optim.hip; A/train.py:116-118, A/diffab/utils/train.py:28-36):

        opt = FusedAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0)
        loss.backward(); grad_norm = opt.step(max_grad_norm=100.0); opt.zero_grad()

    step(max_grad_norm) returns the UNCLIPPED gradient norm as a 1-element device tensor (what clip_grad_norm_ returns), or None without
    clipping; the .grad tensors themselves are not rescaled.  The state keeps torch's layout ('step', 'exp_avg', 'exp_avg_sq' per parameter,
    `step` a device tensor shared by a group), so state_dict()s interchange with torch.optim.Adam(capturable=True).  Capturable into a
    hipGraph as it is (the step counter lives on the device)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('FusedAdam: invalid hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._steps, self._ws, self._norm = {}, {}, {}
        self._hyper, self._hyper_host = {}, {}          # per group: 6 float64 on the device (what the kernels read) and the tuple last uploaded

    def refresh_hyper(self, max_grad_norm=None):
        """Upload lr / betas / eps / weight_decay (/ max_grad_norm) of every group to the device buffers the kernels read, if they changed.
        step() calls this itself outside a graph capture; GraphedTrainStep calls it before every replay, so a scheduler's change to
        param_groups takes effect in the replayed step (a host scalar baked into the captured launch would not)."""
        for gi, group in enumerate(self.param_groups):
            if gi not in self._hyper:
                continue
            vals = (float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']), float(group['weight_decay']),
                    float(max_grad_norm) if max_grad_norm is not None else 0.0)
            if self._hyper_host.get(gi) != vals:
                self._hyper[gi].copy_(torch.tensor(vals, dtype=torch.float64))
                self._hyper_host[gi] = vals

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._steps.clear()                                    # re-read from the loaded per-parameter 'step' at the next step()

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        from . import hip
        if closure is not None:
            raise NotImplementedError('FusedAdam: closures are not supported')
        WgradGroup.sync()
        norms = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if gi not in self._steps:
                # ONE counter per group (every parameter is expected to receive a gradient in every step, as all 207 of the reference's
                # model do); a loaded state_dict (ours or torch.optim.Adam's) continues from its count
                loaded = [self.state[p]['step'] for p in group['params'] if 'step' in self.state.get(p, {})]
                first = int(torch.as_tensor(loaded[0]).reshape(-1)[0].item()) if loaded else 0
                self._steps[gi] = torch.full((1,), first, dtype=torch.int64, device=dev)
                self._norm[gi] = torch.zeros(1, dtype=torch.float32, device=dev)
            if gi not in self._hyper:
                self._hyper[gi] = torch.zeros(6, dtype=torch.float64, device=dev)
            for p in ps:
                st = self.state[p]
                if st.get('step') is not self._steps[gi]:
                    st['step'] = self._steps[gi]
                if 'exp_avg' not in st:
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            need = hip.adam_ws_bytes(ps)
            if gi not in self._ws or self._ws[gi].numel() < need:
                self._ws[gi] = torch.empty(need, dtype=torch.uint8, device=dev)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            if not torch.cuda.is_current_stream_capturing():
                self.refresh_hyper(max_grad_norm)
            hip.adam_step([p.data for p in ps], grads, [self.state[p]['exp_avg'] for p in ps], [self.state[p]['exp_avg_sq'] for p in ps],
                          self._steps[gi], group['lr'], group['betas'][0], group['betas'][1], group['eps'], group['weight_decay'],
                          max_grad_norm=max_grad_norm, grad_norm_out=self._norm[gi] if max_grad_norm is not None else None, ws=self._ws[gi],
                          hyper_dev=self._hyper[gi])
            # the kernel wrote the parameters through raw pointers: tell autograd / the packed-weight caches (modules.GABlock.packed keys
            # its inference copies on _version) that they changed
            if not torch.cuda.is_current_stream_capturing():
                torch._C._increment_version(ps)
            norms.append(self._norm[gi])
        if max_grad_norm is None or not norms:
            return None
        return norms[0] if len(norms) == 1 else torch.linalg.vector_norm(torch.cat(norms))


class GraphedTrainStep:
    """forward + backward + optimizer step of `model(batch)` captured ONCE into a hipGraph and replayed (BASELINE config 5: the step is
    ~900 kernel launches; eager, ~1.4 ms of its 16.8 ms are gaps between them).

    Everything on the path is capture-safe: the library allocates nothing per call, the step indices t come from torch's device
    generator (graph-aware), the noising kernel reads its Philox position from a 16-byte device buffer refreshed before every replay,
    and the optimizer must be capturable: training.FusedAdam, or torch.optim.Adam(..., capturable=True).  The batch is copied into the
    graph's static tensors; losses are returned as a dict of 0-d device tensors (valid until the next call).  Shapes, the set of
    parameters and `loss_weights` are fixed at construction; build a new object when they change.  Optimizer hyper-parameters (lr,
    betas, eps, weight_decay, the clip norm) are NOT frozen with FusedAdam: its kernels read them from a device buffer this object
    refreshes from `param_groups` before every replay, so ReduceLROnPlateau / MultiStepLR (A/diffab/utils/train.py:39-60) work
    unchanged; with torch's capturable Adam a changed host-scalar lr raises instead of being ignored."""

    def __init__(self, model, optimizer, batch, loss_weights=None, max_grad_norm=None, warmup=3):
        from . import hip
        ddp = isinstance(model, torch.nn.parallel.DistributedDataParallel)
        if ddp:
            # DDP's gradient all-reduce runs inside backward, i.e. inside the capture: only RCCL collectives on device buffers can be
            # captured (gloo stages through host memory), and DDP's unused-parameter search walks the autograd graph on the host every
            # iteration.  PyTorch's recipe for DDP under graph capture (static graph, no unused-parameter search, >= 11 warm-up
            # iterations) has NOT been validated for this model on MI355X nodes: refuse what cannot work and say what is untested.
            import torch.distributed as dist
            if dist.get_backend(model.process_group) != 'nccl':
                raise NotImplementedError('GraphedTrainStep: a DistributedDataParallel model needs the nccl (RCCL) backend, one GPU per rank -- its gradient '
                                          'all-reduce is part of the captured step; run eager steps (FusedAdam is capture-free as well) on other backends')
            if model.find_unused_parameters:
                raise NotImplementedError('GraphedTrainStep: build the DDP wrapper with find_unused_parameters=False (sampler.wrap_ddp(..., '
                                          'find_unused_parameters=False, static_graph=True)): the search runs on the host every iteration and cannot be captured')
            warmup = max(warmup, 11)
        self.model, self.opt = model, optimizer
        self.weights, self.max_grad_norm = loss_weights, max_grad_norm
        dev = next(model.parameters()).device
        self.static = {k: (v.to(dev).clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        dpm = (model.module if ddp else model).diffusion
        self._dpm = dpm
        self.seed_dev = torch.zeros(2, dtype=torch.int64, device=dev)
        self._set_seed(dpm)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                                  # warm-up on a side stream (torch's capture protocol)
            for _ in range(warmup):
                self._one_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        before = set(hip.Workspace._bufs)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.losses = self._one_step()
        self.keep = [hip.Workspace._bufs.pop(k) for k in set(hip.Workspace._bufs) - before]     # scratch slabs owned by this graph's pool

    def _set_seed(self, dpm):
        self.seed_dev.copy_(torch.tensor([dpm._new_seed(), 0], dtype=torch.int64))

    def _one_step(self):
        self.opt.zero_grad(set_to_none=True)
        dpm = self._dpm
        dpm._train_seed_dev = self.seed_dev                            # for this step only (fulldpm_loss)
        try:
            losses = self.model(dict(self.static))
        finally:
            dpm._train_seed_dev = None
        total = sum(v * (self.weights[k] if self.weights is not None else 1.0) for k, v in losses.items())
        total.backward()
        if isinstance(self.opt, FusedAdam):
            self.opt.step(max_grad_norm=self.max_grad_norm)
        else:
            if self.max_grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
            self.opt.step()
        return {k: v.detach() for k, v in losses.items()}

    def __call__(self, batch):
        for k, v in batch.items():
            if torch.is_tensor(v) and k in self.static:
                self.static[k].copy_(v)
        self._set_seed(self._dpm)
        if isinstance(self.opt, FusedAdam):
            self.opt.refresh_hyper(self.max_grad_norm)                 # a scheduler may have moved lr since the last replay
            self.graph.replay()
            torch._C._increment_version([p for g in self.opt.param_groups for p in g['params']])   # raw-pointer writes inside the graph
        else:
            self._check_frozen_hyper()
            self.graph.replay()
        return self.losses

    def _check_frozen_hyper(self):
        """torch.optim.Adam(capturable=True) keeps lr as a host scalar baked into the captured kernels unless lr is a tensor: refuse to
        replay silently with a stale value."""
        cur = [(g['lr'] if not torch.is_tensor(g['lr']) else None, g.get('betas'), g.get('eps'), g.get('weight_decay')) for g in self.opt.param_groups]
        if getattr(self, '_hyper0', None) is None:
            self._hyper0 = cur
        elif cur != self._hyper0:
            raise RuntimeError('GraphedTrainStep: optimizer hyper-parameters changed after capture; use training.FusedAdam (reads them from '
                               'device memory at replay), a tensor lr, or build a new GraphedTrainStep')

    def close(self):
        self._dpm._train_seed_dev = None

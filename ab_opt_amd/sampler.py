"""Multi-GPU batched sampling: shard independent samples over ranks, one exchange at the end.

The denoising loop has no cross-sample dependence, so ranks never communicate inside it.  The only
exchange is the batched-sampling reduction the reference performs on files written by Ray subprocesses
(AbDock/optimize_ab.py:21-31,66-72; AbDock/src/tools/runner/design_for_pdb.py:326-345): gather every rank's generated
candidates and rank them by "commonness" (AbDock/src/tools/runner/design_for_testset.py:556-589).
One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI) on device tensors; the same code runs on
"gloo" with host tensors in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, world_size, rank):
    """Contiguous, balanced split of n_total samples: first (n_total % world) ranks get one extra."""
    base, extra = divmod(n_total, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(batch, world_size, rank):
    n = batch['aa'].shape[0]
    a, b = shard_range(n, world_size, rank)
    return {k: (v[a:b] if isinstance(v, torch.Tensor) and v.shape[:1] == (n,) else v) for k, v in batch.items()}, (a, b)


def candidates_from_positions(p, generate_flag):
    """(N, L, 3) positions + (N, L) mask with the same count per sample -> (N, n_gen, 3)."""
    n_gen = int(generate_flag[0].sum())
    if not bool((generate_flag.sum(1) == n_gen).all()):
        raise ValueError('every sample must generate the same number of residues to be ranked together')
    return p[generate_flag].reshape(p.shape[0], n_gen, 3).contiguous()


def all_gather_candidates(cand, counts=None, group=None):
    """Concatenate (N_r, n, 3) candidate tensors of all ranks in rank order (ragged N_r allowed via `counts`)."""
    world = dist.get_world_size(group)
    if counts is None:
        counts = [cand.shape[0]] * world
    nmax = max(counts)
    pad = cand
    if cand.shape[0] < nmax:
        pad = torch.cat([cand, cand.new_zeros((nmax - cand.shape[0],) + tuple(cand.shape[1:]))], 0)
    pad = pad.contiguous()
    staged = pad.is_cuda and dist.get_backend(group) == 'gloo'          # gloo: the gather goes through host memory
    src = pad.cpu() if staged else pad
    bufs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(bufs, src, group=group)
    out = torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)
    return out.to(cand.device) if staged else out


def commonness_score(structs, score_fn=None):
    """score[b] = mean RMSD of structure b to the others * B/(B-1) (lower = more common)."""
    if score_fn is not None:
        return score_fn(structs)
    from . import hip
    return hip.commonness_score(structs)          # HIP kernel; raises on CPU tensors (no fallback)


def rank_commoness(structs, k, score_fn=None):
    """Indices of the k most common structures (design_for_testset.py:575-589)."""
    return torch.topk(commonness_score(structs, score_fn), k=k, largest=False)[1]


def dockq_scores(model_pos, model_mask, native_pos, native_mask, fragment_type=None, group=None):
    """DockQ of candidate structures against the native complex, on the device (the reference writes every candidate to a PDB
    file and shells out: AbDock/src/tools/runner/design_for_pdb.py:316-321, AbDock/DockQ/DockQ.py:98-385).

    model_pos (S,L,A,3), model_mask (S,L,A) or (L,A), native_pos (L,A,3), native_mask (L,A).  The two chains are given either as
    `group` (L,) in {0: not scored, 1, 2} or through `fragment_type` (antibody chains 1/2 -> group 1, antigen 3 -> group 2).
    -> dict of (S,) tensors: fnat, irms, Lrms, DockQ."""
    from . import hip
    if group is None:
        group = torch.where(fragment_type == 3, 2, torch.where(fragment_type > 0, 1, 0))
    out = hip.dockq_lite(model_pos, model_mask, native_pos, native_mask, group)
    return dict(fnat=out[:, 0], irms=out[:, 1], Lrms=out[:, 2], DockQ=out[:, 3])


@torch.no_grad()
def sample_sharded(model, batch, sample_opt, k=1, group=None, seed=None):
    """Every rank samples its shard of `batch`; returns (local traj, (start, end), global top-k indices, all candidates)."""
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if dist.is_initialized() else (1, 0)
    n = batch['aa'].shape[0]
    sub, (a, b) = shard_batch(batch, world, rank)
    opt = dict(sample_opt)
    opt.setdefault('rng_offset', a * batch['aa'].shape[1])      # distinct Philox counters per global sample index
    if seed is not None:
        opt['seed'] = seed
    n_gen = int(batch['generate_flag'][0].sum()) if n > 0 else 0
    if b > a:
        traj = model.sample(sub, sample_opt=opt)
        cand = candidates_from_positions(traj[0][1], sub['generate_flag'])
    else:                                   # more ranks than samples: nothing to denoise here, but every rank still joins the gather
        traj = {}
        cand = batch['pos_heavyatom'].new_zeros((0, n_gen, 3))
    if world > 1:
        counts = [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]
        cand = all_gather_candidates(cand, counts, group)
    top = rank_commoness(cand, min(k, cand.shape[0]))
    return traj, (a, b), top, cand


@torch.no_grad()
def sample_replicated(model, complex_batch, num_samples, sample_opt=None, optimize_step=None):
    """N samples of ONE complex without replicating it.

    The reference's runners build the batch by repeating one crop `num_samples` times on the host
    (AbDock/src/tools/runner/design_for_pdb.py:141-147), so encode() runs N times and N identical copies of pair_feat
    (17 MB each at L=256) are streamed from HBM at every step.  Here `complex_batch` holds the complex once (batch dim 1):
    it is encoded once and the sampler shares res_feat / pair_feat across the N samples.  Returns the same trajectory
    dict as `model.sample` (batch dim `num_samples`).  `optimize_step` selects `model.optimize`-style partial denoising."""
    from . import hip
    sample_opt = dict(sample_opt or {'sample_structure': True, 'sample_sequence': True})
    sample_opt.pop('contig', None)
    one = {k: (v[:1] if torch.is_tensor(v) else v) for k, v in complex_batch.items()}
    res_feat, pair_feat, R_0, p_0 = model.encode(one, remove_structure=sample_opt.get('sample_structure', True),
                                                remove_sequence=sample_opt.get('sample_sequence', True))
    rep = lambda a: a.expand(num_samples, *a.shape[1:]).contiguous()
    v_0 = rep(hip.so3_log(R_0, grad_mode=False))
    args = (v_0, rep(p_0), rep(one['aa']))
    masks = (rep(one['generate_flag']), rep(one['mask']))
    if optimize_step is None:
        return model.diffusion.sample(*args, res_feat, pair_feat, *masks, **sample_opt)
    return model.diffusion.optimize(*args, optimize_step, res_feat, pair_feat, *masks, **sample_opt)


def complexes_of_rank(n_complexes, world_size, rank):
    """By-complex partition of a test set (BASELINE config 4): contiguous, balanced blocks of complexes per rank
    (64 complexes over 8 ranks: rank r owns 8 r .. 8 r + 7), so every sample of a complex -- and therefore its whole commonness ranking --
    stays on one GPU, nothing is exchanged per complex, and the global sample indices of a rank's launch are contiguous (one Philox
    offset per launch keeps the draws independent of the number of ranks)."""
    a, b = shard_range(n_complexes, world_size, rank)
    return list(range(a, b))


def pad_complex(one, L):
    """A batch dict with batch dim 1 padded to L residues the way the reference's PaddingCollate does (AbDock/src/utils/data.py:60-76:
    zeros, `aa` with the padding token 21, masks False)."""
    L0 = one['aa'].shape[1]
    if L0 == L:
        return one
    out = {}
    for k, v in one.items():
        if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == L0:
            pad = v.new_full((v.shape[0], L - L0) + tuple(v.shape[2:]), 21 if k == 'aa' else 0)
            out[k] = torch.cat([v, pad], 1)
        else:
            out[k] = v
    return out


@torch.no_grad()
def sample_grouped(model, complexes, num_samples, sample_opt=None, optimize_step=None, pad_to=None):
    """`num_samples` samples of EACH of G complexes in one launch (BASELINE config 4: a rank's leg of a test set).

    The reference designs a test set one structure at a time (AbDock/src/tools/runner/design_for_testset.py:556-589), each with its
    crop replicated `num_samples` times on the host.  Here the G complexes (a list of batch dicts with batch dim 1; padded to a common
    length like PaddingCollate does) are encoded ONCE each as a batch of G, and the sampler runs G x num_samples samples as one batch
    whose samples share the pair features of their complex (`abopt_eps_net_forward(pair_feat_shared = num_samples)`): the kernels see
    full-size launches instead of G small ones, and z is read from HBM once per complex and query block, not once per sample.
    Sample g * num_samples + s is sample s of complex g.  pad_to: pad to this length instead of the longest complex of the launch (the
    test-set driver passes the longest complex of the whole set, see `launch_rng_offset`).  Returns the trajectory dict of `model.sample`
    (batch dim G * num_samples)."""
    from . import hip
    sample_opt = dict(sample_opt or {'sample_structure': True, 'sample_sequence': True})
    sample_opt.pop('contig', None)
    L = max(max(int(c['aa'].shape[1]) for c in complexes), int(pad_to or 0))
    padded = [pad_complex(c, L) for c in complexes]
    batch = {k: (torch.cat([c[k][:1] for c in padded], 0) if torch.is_tensor(v) else v) for k, v in padded[0].items()}
    res_feat, pair_feat, R_0, p_0 = model.encode(batch, remove_structure=sample_opt.get('sample_structure', True),
                                                remove_sequence=sample_opt.get('sample_sequence', True))
    rep = lambda a: a.repeat_interleave(num_samples, dim=0).contiguous()
    v_0 = rep(hip.so3_log(R_0, grad_mode=False))
    args = (v_0, rep(p_0), rep(batch['aa']))
    masks = (rep(batch['generate_flag']), rep(batch['mask']))
    if optimize_step is None:
        return model.diffusion.sample(*args, res_feat, pair_feat, *masks, **sample_opt)
    return model.diffusion.optimize(*args, optimize_step, res_feat, pair_feat, *masks, **sample_opt)


def launch_rng_offset(first_complex, samples_per_complex, L_pad):
    """Philox counter base of the launch whose first complex is `first_complex`.  The kernels read counter base + n L + l for sample n,
    residue l of a launch padded to L (distinct sub-sequence tags for sample_init / add_noise / the loop's steps, csrc/denoise.hip), so
    when EVERY launch of a test set is padded to the same L_pad (the longest complex of the whole set) the launches' ranges
    [c0 S L_pad, (c0 + G) S L_pad) tile the counter space: no two samples share a draw, and sample s of complex c reads
    (c S + s) L_pad + l however the complexes are grouped into launches or spread over ranks.  (With each launch padded to its OWN longest
    complex a later, shorter launch started inside an earlier, longer one's range: identical draws for samples of different complexes.)"""
    return int(first_complex) * int(samples_per_complex) * int(L_pad)


@torch.no_grad()
def design_testset_sharded(model, complexes, samples_per_complex, sample_opt=None, k=1, group=None, seed=0, optimize_step=None,
                           complexes_per_launch=8, native=None):
    """BASELINE config 4: a test set of complexes x `samples_per_complex` samples over the ranks of `group`, partitioned BY COMPLEX.

    The reference fans this out as one subprocess per structure through Ray and collects files
    (AbDock/optimize_ab.py:21-31,66-72; AbDock/src/tools/runner/design_for_testset.py:556-589 ranks each structure's samples).
    Here every rank designs its own block of complexes, `complexes_per_launch` of them at a time as ONE batch (`sample_grouped`:
    every complex encoded once, its samples share its pair features inside the kernels), ranks each complex's candidates locally with
    the commonness score, optionally scores them against the native structure (DockQ on the device), and the per-complex summaries
    (a few hundred bytes each) are exchanged once at the end with all_gather_object.

    complexes: list of batch dicts with batch dim 1.  Every launch is padded to the longest complex of the WHOLE test set, so the Philox
    stream position of a sample depends on its global index (complex x samples_per_complex + sample) alone (`launch_rng_offset`): no two
    samples share a draw, and positions / sequences do not depend on the number of ranks or on `complexes_per_launch` (padding never
    reaches a real residue; AbDock's prmsd score averages over the padded length like a PaddingCollate batch does, dpm_full.py:110).
    native (optional): dict(pos (L,A,3), mask (L,A)) per complex, or True to score against the complex's own input coordinates.
    -> list (one entry per complex, in input order, identical on every rank) of
    dict(complex=index, rank=owner, top=LongTensor(k), score=Tensor(S), ca=final CA positions of the generated residues (S, n_gen, 3)
    [, dockq=dict of (S,) tensors])."""
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if dist.is_initialized() else (1, 0)
    sample_opt = dict(sample_opt or {'sample_structure': True, 'sample_sequence': True})
    S = int(samples_per_complex)
    mine = []
    own = complexes_of_rank(len(complexes), world, rank)
    G = max(1, int(complexes_per_launch))
    L_all = max(int(c['aa'].shape[1]) for c in complexes)            # the same on every rank: `complexes` is the whole test set
    for lo in range(0, len(own), G):
        ids = own[lo:lo + G]
        chunk = [complexes[c] for c in ids]
        L = L_all
        opt = dict(sample_opt, seed=int(seed), rng_offset=launch_rng_offset(ids[0], S, L_all))
        traj = sample_grouped(model, chunk, S, opt, optimize_step=optimize_step, pad_to=L_all)
        p_fin = traj[0][1]
        for g, c in enumerate(ids):
            one = pad_complex(complexes[c], L)
            gen = one['generate_flag'][:1].expand(S, -1)
            pc = p_fin[g * S:(g + 1) * S]
            cand = candidates_from_positions(pc, gen)
            score = commonness_score(cand)
            top = torch.topk(score, k=min(k, S), largest=False)[1]
            rec = dict(complex=c, rank=rank, top=top.cpu(), score=score.cpu(), ca=cand.cpu())
            if native is not None:
                rec['dockq'] = {kk: v.cpu() for kk, v in _dockq_of_samples(model, one, traj, g * S, S, native if native is not True else None, c).items()}
            mine.append(rec)
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        mine = [e for part in everyone for e in part]
    return sorted(mine, key=lambda e: e['complex'])


def _dockq_of_samples(model, one, traj, lo, S, native, c):
    """DockQ (CA-only flavour the runner uses, design_for_pdb.py:316-321) of samples lo .. lo + S - 1 of a finished trajectory against the
    native complex: the generated residues' backbone is rebuilt from the final frames (reconstruct_backbone_partially), everything on
    the device."""
    from . import geometry, hip
    v, p, s = traj[0][0][lo:lo + S], traj[0][1][lo:lo + S], traj[0][2][lo:lo + S]
    rep = lambda a: a[:1].expand(S, *a.shape[1:]).contiguous()
    R = hip.so3_exp(v)
    gen = rep(one['generate_flag'])
    pos, mask = geometry.reconstruct_backbone_partially(rep(one['pos_heavyatom']), R, p, torch.where(gen, s, rep(one['aa'])), rep(one['chain_nb']),
                                                        rep(one['res_nb']), rep(one['mask_heavyatom']), gen)
    if native is None:
        npos, nmask = one['pos_heavyatom'][0], one['mask_heavyatom'][0]
    else:
        nat = native[c] if isinstance(native, (list, tuple)) else native
        npos, nmask = nat['pos'].to(pos.device), nat['mask'].to(pos.device)
    return dockq_scores(pos, mask, npos, nmask, fragment_type=one['fragment_type'][0])


def wrap_ddp(model, device=None, **kw):
    """Data-parallel training (BASELINE config 5 at N GPUs): one process per GPU, gradients averaged by bucketed all-reduce over
    RCCL (backend 'nccl') overlapped with backward.  The reference's train.py (AbDock/train.py:70-114) is single-process; this is
    the wrapper a multi-GPU launch adds around `get_model(cfg)`.  The custom autograd functions of the training path are ordinary
    torch.autograd.Function nodes, so DistributedDataParallel's hooks see their parameter gradients like any other."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    kw.setdefault('find_unused_parameters', True)      # the prmsd loss touches no parameter when mask_generate[:, 0] is all False
    return DDP(model.to(dev), device_ids=[dev.index] if dev.type == 'cuda' else None, **kw)

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
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


def commonness_score(structs, score_fn=None):
    """score[b] = mean RMSD of structure b to the others * B/(B-1) (lower = more common)."""
    if score_fn is not None:
        return score_fn(structs)
    from . import hip
    return hip.commonness_score(structs)          # HIP kernel; raises on CPU tensors (no fallback)


def rank_commoness(structs, k, score_fn=None):
    """Indices of the k most common structures (design_for_testset.py:575-589)."""
    return torch.topk(commonness_score(structs, score_fn), k=k, largest=False)[1]


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

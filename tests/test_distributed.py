"""World-size-2 gloo tests (CPU) of the only exchange on the path: candidate all_gather + commonness ranking."""
import os
import socket
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ab_opt_amd import sampler
from ab_opt_amd.utils import synth


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle.dpm import rank_commoness as ref_rank
        allc = synth.hash_tensor((n_total, 12, 3), 77, scale=6.0)
        allc[2] = allc[4] + 0.02
        a, b = sampler.shard_range(n_total, world, rank)
        counts = [sampler.shard_range(n_total, world, r)[1] - sampler.shard_range(n_total, world, r)[0] for r in range(world)]
        gathered = sampler.all_gather_candidates(allc[a:b].clone(), counts)
        assert torch.equal(gathered, allc), 'all_gather must restore the global sample order'
        # ranking with an injected CPU scorer (the product scorer is the HIP kernel; the gather/rank plumbing is what runs here)
        def cpu_score(x):
            B = x.shape[0]
            d = torch.sqrt(((x[:, None] - x[None]) ** 2).sum(-1).mean(-1))
            return d.sum(-1) / (B - 1)
        top = sampler.rank_commoness(gathered, 3, score_fn=cpu_score)
        assert torch.equal(top, ref_rank(allc, 3))
        out[rank] = top.tolist()
    finally:
        dist.destroy_process_group()


def test_gather_and_rank_world2():
    for n_total in (8, 7):                    # even and ragged shards
        port = _free_port()
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, n_total, out), nprocs=2, join=True)
        assert out[0] == out[1] and len(out[0]) == 3


def test_shard_helpers():
    cover = []
    for r in range(3):
        a, b = sampler.shard_range(10, 3, r)
        cover += list(range(a, b))
    assert cover == list(range(10))
    batch = synth.make_batch(5, synth.LAYOUT_128, seed=3, lengths=[40] * 5)
    sub, (a, b) = sampler.shard_batch(batch, 2, 1)
    assert (a, b) == (3, 5) and sub['aa'].shape[0] == 2 and torch.equal(sub['pos_heavyatom'], batch['pos_heavyatom'][3:5])
    p = torch.arange(2 * 6 * 3, dtype=torch.float32).reshape(2, 6, 3)
    g = torch.tensor([[0, 1, 1, 0, 0, 0], [0, 0, 0, 1, 1, 0]], dtype=torch.bool)
    c = sampler.candidates_from_positions(p, g)
    assert c.shape == (2, 2, 3) and torch.equal(c[1, 0], p[1, 3])

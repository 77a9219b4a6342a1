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


def test_by_complex_partition():
    """BASELINE config 4 layout: 64 complexes over 8 ranks = 8 per rank, every complex on exactly one rank, contiguous blocks (the global
    sample indices of a rank's grouped launch are then contiguous: one Philox offset per launch)."""
    owners = {}
    for r in range(8):
        mine = sampler.complexes_of_rank(64, 8, r)
        assert len(mine) == 8
        for c in mine:
            assert c not in owners
            owners[c] = r
    assert sorted(owners) == list(range(64)) and owners[9] == 1 and sampler.complexes_of_rank(64, 8, 3) == list(range(24, 32))
    assert sampler.complexes_of_rank(3, 2, 0) == [0, 1] and sampler.complexes_of_rank(3, 2, 1) == [2] and sampler.complexes_of_rank(1, 4, 2) == []


def test_pad_complex_matches_padding_collate():
    """sampler.pad_complex: the reference's PaddingCollate (AbDock/src/utils/data.py:60-76) pads every per-residue tensor with zeros, `aa`
    with 21 and the masks with False."""
    from ab_opt_amd.utils import synth
    one = synth.make_batch(1, synth.LAYOUT_128, seed=3)
    L0 = one['aa'].shape[1]
    p = sampler.pad_complex(one, L0 + 9)
    assert sampler.pad_complex(one, L0) is one
    for k, v in one.items():
        if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == L0:
            assert p[k].shape[1] == L0 + 9 and torch.equal(p[k][:, :L0], v)
            tail = p[k][:, L0:]
            assert bool((tail == (21 if k == 'aa' else 0)).all()), k
    assert not p['mask'][:, L0:].any() and not p['generate_flag'][:, L0:].any()


def _object_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # the exchange design_testset_sharded performs: per-complex summaries gathered as objects, merged in complex order
        mine = [dict(complex=c, rank=rank, top=torch.tensor([c, c + 1])) for c in sampler.complexes_of_rank(5, world, rank)]
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        merged = sorted([e for part in everyone for e in part], key=lambda e: e['complex'])
        out[rank] = [(e['complex'], e['rank'], e['top'].tolist()) for e in merged]
        # an empty shard still joins the candidate gather (more ranks than samples)
        cand = torch.zeros(1 if rank == 0 else 0, 4, 3)
        allc = sampler.all_gather_candidates(cand, [1, 0])
        assert allc.shape == (1, 4, 3)
        # data-parallel training: DDP's bucket hooks read gradients DURING backward, so the queue of grouped weight-gradient launches
        # (training.WgradGroup: gradients are computed when the queue is flushed) must be off in a process group of more than one rank
        from ab_opt_amd import training
        assert training.WgradGroup.enabled and not training.WgradGroup.active()
    finally:
        dist.destroy_process_group()


def test_testset_summary_exchange_world2():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_object_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] == out[1] == [(0, 0, [0, 1]), (1, 0, [1, 2]), (2, 0, [2, 3]), (3, 1, [3, 4]), (4, 1, [4, 5])]      # contiguous blocks: rank 0 owns 0..2


def test_wgrad_group_is_on_in_a_single_process():
    from ab_opt_amd import training
    assert training.WgradGroup.enabled and training.WgradGroup.active() and training.WgradGroup._state is None


def test_launch_rng_ranges_never_overlap():
    """launches of a test set whose complexes differ in length must own disjoint Philox counter ranges, and a sample's
    counters must not depend on how the complexes are grouped into launches or spread over ranks (every launch is padded to the longest
    complex of the set; sample_init / add_noise / the loop's steps read base + n L + l under different sub-sequence tags)."""
    from ab_opt_amd.sampler import launch_rng_offset, complexes_of_rank
    import random
    rnd = random.Random(5)
    for trial in range(50):
        n, S = rnd.randint(1, 40), rnd.randint(1, 16)
        lens = [rnd.randint(20, 300) for _ in range(n)]
        L_all = max(lens)
        first = {}
        for world, G in ((1, 1), (1, 8), (2, 4), (8, 2), (8, 8)):
            ranges = []
            for r in range(world):
                own = complexes_of_rank(n, world, r)
                for lo in range(0, len(own), G):
                    ids = own[lo:lo + G]
                    assert ids == list(range(ids[0], ids[0] + len(ids)))       # contiguous: sample n of the launch is complex ids[0] + n // S
                    base = launch_rng_offset(ids[0], S, L_all)
                    ranges.append((base, base + len(ids) * S * L_all))
                    for g, c in enumerate(ids):
                        assert first.setdefault(c, base + g * S * L_all) == base + g * S * L_all == c * S * L_all
            ranges.sort()
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and ranges[0][0] == 0, (trial, world, ranges)


def test_bench_self_launch_command_shape(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE re-runs itself under torch.distributed.run with N ranks on 127.0.0.1 (bench.self_launch);
    with a launcher's WORLD_SIZE that disagrees with --gpus it exits with a message instead of an AssertionError."""
    import subprocess
    import sys
    import importlib
    import pytest
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return Done()
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '20', '--warmup', '5'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '4', '--steps', '20', '--warmup', '5']
    assert cmd[-7].endswith('bench.py') and seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    monkeypatch.setenv('WORLD_SIZE', '2')
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE=2' in str(e.value.code)

"""Worker functions of the two-process GPU tests (spawned by tests/test_hip_parity.py; both ranks share cuda:0, the process group
is gloo, so collectives go through host memory -- the product path on a multi-GPU node is the same code on backend 'nccl')."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init(rank, world, port):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    return torch, dist


def sharded_worker(rank, world, port, outdir):
    torch, dist = _init(rank, world, port)
    try:
        from conftest import build_model
        from ab_opt_amd import sampler
        from ab_opt_amd.utils import synth
        dev = torch.device('cuda:0')
        m = build_model(10, 3, device=dev)
        b = {k: v.to(dev) for k, v in synth.make_batch(5, synth.LAYOUT_128, seed=11, replicate=True).items()}      # 5 samples over 2 ranks: ragged shards
        traj, (a, e), top, cand = sampler.sample_sharded(m, b, dict(sample_structure=True, sample_sequence=True, contig=''), k=2, seed=42)
        torch.save(dict(a=a, e=e, top=top.cpu(), cand=cand.cpu(), p0=traj[0][1].cpu(), s0=traj[0][2].cpu()), os.path.join(outdir, f'sharded_{rank}.pt'))
        # by-complex partition (config 4): 3 complexes x 4 samples
        cx = [{k: v.to(dev) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=100 + c).items()} for c in range(3)]
        res = sampler.design_testset_sharded(m, cx, 4, k=2, seed=7, complexes_per_launch=1)
        torch.save(res, os.path.join(outdir, f'testset_{rank}.pt'))
    finally:
        dist.destroy_process_group()


def ddp_worker(rank, world, port, outdir):
    torch, dist = _init(rank, world, port)
    try:
        from conftest import build_model
        from ab_opt_amd import sampler
        from ab_opt_amd.utils import synth
        dev = torch.device('cuda:0')
        m = build_model(10, 3, device=dev).train()
        ddp = sampler.wrap_ddp(m, dev)
        full = synth.make_batch(2, synth.LAYOUT_128, seed=5, lengths=[64, 57])
        mine = {k: v[rank:rank + 1].to(dev) for k, v in full.items()}
        torch.manual_seed(123)                      # same t / noise seeds on both ranks is fine: different samples
        loss = sum(ddp(mine).values())
        loss.backward()
        g = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
        torch.save(dict(loss=loss.item(), grads=g), os.path.join(outdir, f'ddp_{rank}.pt'))
        m.zero_grad(); m.eval()
    finally:
        dist.destroy_process_group()


def ddp_fused_adam_worker(rank, world, port, outdir):
    """Two ranks on cuda:0 (gloo): eager DDP steps with training.FusedAdam keep the replicas in lock step; GraphedTrainStep refuses a DDP
    model on a backend whose all-reduce cannot be captured."""
    torch, dist = _init(rank, world, port)
    try:
        from conftest import build_model
        from ab_opt_amd import sampler, training
        from ab_opt_amd.utils import synth
        dev = torch.device('cuda:0')
        m = build_model(10, 3, device=dev).train()
        ddp = sampler.wrap_ddp(m, dev)
        opt = training.FusedAdam(ddp.parameters(), lr=1e-3)
        full = synth.make_batch(2, synth.LAYOUT_128, seed=5, lengths=[64, 57])
        mine = {k: v[rank:rank + 1].to(dev) for k, v in full.items()}
        refused = None
        try:
            training.GraphedTrainStep(ddp, opt, mine, max_grad_norm=100.0)
        except NotImplementedError as e:
            refused = str(e)
        losses = []
        for it in range(3):
            torch.manual_seed(100 + it)
            opt.zero_grad(set_to_none=True)
            loss = sum(ddp(dict(mine)).values())
            loss.backward()
            opt.step(max_grad_norm=100.0)
            losses.append(loss.item())
        sd = {n: p.detach().cpu() for n, p in m.named_parameters()}
        torch.save(dict(losses=losses, params=sd, refused=refused), os.path.join(outdir, f'ddpadam_{rank}.pt'))
    finally:
        dist.destroy_process_group()


def two_device_worker(rank, world, port, outdir):
    """ONE DEVICE PER RANK over RCCL (backend 'nccl'): what the driver's N > 1 bench and a multi-GPU node run.  Skipped by its test on a
    one-GPU box.  (i) sharded sampling + the by-complex test-set driver: results must equal the single-process run bit for bit;
    (ii) DistributedDataParallel(nccl, find_unused_parameters=False, static_graph=True) + FusedAdam: two eager steps, then the WHOLE step
    (forward, backward with its bucketed all-reduce, clipping + Adam) captured by training.GraphedTrainStep and replayed twice --
    replicas stay in lock step."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    try:
        from conftest import build_model
        from ab_opt_amd import sampler, training
        from ab_opt_amd.utils import synth
        m = build_model(10, 3, device=dev)
        b = {k: v.to(dev) for k, v in synth.make_batch(5, synth.LAYOUT_128, seed=11, replicate=True).items()}
        traj, (a, e), top, cand = sampler.sample_sharded(m, b, dict(sample_structure=True, sample_sequence=True, contig=''), k=2, seed=42)
        cx = [{k: v.to(dev) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=100 + c).items()} for c in range(3)]
        res = sampler.design_testset_sharded(m, cx, 4, k=2, seed=7, complexes_per_launch=1)
        out = dict(a=a, e=e, top=top.cpu(), cand=cand.cpu(), p0=traj[0][1].cpu(), testset=res)
        # ---- data-parallel training, AbDesign flavour (every parameter takes part in every step: no unused-parameter search needed)
        md = build_model(10, 5, flavour='abdesign', device=dev).train()
        ddp = sampler.wrap_ddp(md, dev, find_unused_parameters=False, static_graph=True)
        opt = training.FusedAdam(ddp.parameters(), lr=1e-3)
        full = synth.make_batch(2, synth.LAYOUT_128, seed=5, lengths=[64, 57])
        mine = {k: v[rank:rank + 1].to(dev) for k, v in full.items()}
        losses = []
        for it in range(2):
            torch.manual_seed(100 + it)
            opt.zero_grad(set_to_none=True)
            loss = sum(ddp(dict(mine)).values())
            loss.backward()
            opt.step(max_grad_norm=100.0)
            losses.append(loss.item())
        graph_error = None
        try:
            gstep = training.GraphedTrainStep(ddp, opt, mine, max_grad_norm=100.0)
            for it in range(2):
                losses.append(float(sum(gstep(mine).values()).item()))
            gstep.close()
        except Exception as ex:                 # reported, asserted by the test: the eager result is still compared
            graph_error = repr(ex)
        out.update(losses=losses, graph_error=graph_error, params={n: p.detach().cpu() for n, p in md.named_parameters()},
                   backend=dist.get_backend(), device=torch.cuda.current_device())
        torch.save(out, os.path.join(outdir, f'twodev_{rank}.pt'))
    finally:
        dist.destroy_process_group()

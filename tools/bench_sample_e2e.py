"""End-to-end model.sample(batch) wall time (encode + pair-bias cache + T steps + trajectory hand-over):
    python tools/bench_sample_e2e.py [N] [L]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from ab_opt_amd.utils.synth import build_model
from ab_opt_amd.utils.synth import make_batch, LAYOUT_256, LAYOUT_128
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
for flavour in ('abdesign', 'abdock'):
    model = build_model(100, 7, flavour=flavour, device=dev).eval()
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(N, LAYOUT_256 if L == 256 else LAYOUT_128).items()}
    opt = {'sample_structure': True, 'sample_sequence': True, 'contig': ''}
    model.sample(dict(batch), opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    traj = model.sample(dict(batch), opt)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{flavour}: model.sample N={N} L={L} T=100: {dt * 1e3:.1f} ms end to end = {N * 100 / dt:.0f} sample-steps/s; traj[0][1] {tuple(traj[0][1].shape)} on {traj[0][1].device}, traj[50][1] on {traj[50][1].device}')

    # the same N samples of ONE complex: host-replicated batch (the reference runner's way) vs shared context
    from ab_opt_amd import sampler
    one = {k: v[:1].contiguous() for k, v in batch.items()}
    repl = {k: v.expand(N, *v.shape[1:]).contiguous() for k, v in one.items()}
    for name, fn in (('replicated batch', lambda: model.sample(dict(repl), opt)), ('shared context', lambda: sampler.sample_replicated(model, one, N, opt))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'{flavour}: one complex x {N} samples, {name}: {dt * 1e3:.1f} ms = {N * 100 / dt:.0f} sample-steps/s')

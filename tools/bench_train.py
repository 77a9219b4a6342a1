"""Training step timing (BASELINE config 5: AbDesign forward+backward, batch 16 x 256 residues):
    python tools/bench_train.py [N] [L] [iters]     -> ms per training step."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from ab_opt_amd.utils.synth import build_model
from ab_opt_amd.utils.synth import make_batch, LAYOUT_256, LAYOUT_128

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(N, LAYOUT_256 if L == 256 else LAYOUT_128).items()}
opt = torch.optim.Adam(model.parameters(), lr=1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    loss = sum(model(dict(batch)).values())
    loss.backward()
    opt.step()
    return loss


for native in (True,):
    torch.cuda.reset_peak_memory_stats()
    try:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print(f'native_ipa={native}: {dt * 1e3:.1f} ms / training step (N={N}, L={L}) = {N / dt:.1f} samples/s; loss {loss.item():.4f}; '
              f'peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
    except torch.OutOfMemoryError as e:
        print(f'native_ipa={native}: out of memory ({e})')

"""Training step (config 5) eager vs GraphedTrainStep: python tools/bench_train_graph.py [N] [L] [iters]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ab_opt_amd.utils.synth import build_model, make_batch, LAYOUT_256, LAYOUT_128
from ab_opt_amd import training
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(N, LAYOUT_256 if L == 256 else LAYOUT_128).items()}
opt = training.FusedAdam(model.parameters(), lr=1e-4)


def eager():
    opt.zero_grad(set_to_none=True)
    sum(model(dict(batch)).values()).backward()
    opt.step(max_grad_norm=100.0)


for _ in range(3):
    eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters):
    eager()
torch.cuda.synchronize()
print('eager  : %.2f ms per step' % ((time.perf_counter() - t0) / iters * 1e3))
t0 = time.perf_counter()
g = training.GraphedTrainStep(model, opt, batch, max_grad_norm=100.0)
torch.cuda.synchronize()
print('capture: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
for _ in range(2):
    out = g(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters):
    out = g(batch)
torch.cuda.synchronize()
print('graph  : %.2f ms per step; losses' % ((time.perf_counter() - t0) / iters * 1e3), {k: round(v.item(), 4) for k, v in out.items()})

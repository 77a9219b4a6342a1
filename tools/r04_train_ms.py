"""Training step time (config 5: N=16, L=256, AbDesign flavour, FusedAdam with clipping), eager: python tools/r04_train_ms.py [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ab_opt_amd import training
from ab_opt_amd.utils.synth import build_model, make_batch, LAYOUT_256
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(16, LAYOUT_256).items()}
opt = training.FusedAdam(model.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    sum(model(dict(batch)).values()).backward()
    opt.step(max_grad_norm=100.0)
for _ in range(3): step()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
best = 1e9
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / iters)
print('train step %.3f ms (KDIV=%s)' % (best * 1e3, os.environ.get('ABOPT_GEMM_KDIV', '512')))

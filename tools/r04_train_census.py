"""GEMM census of one training step (config 5): every hip.gemm call with its shape, grouped.  python tools/r04_train_census.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ab_opt_amd import training, hip
from ab_opt_amd.utils.synth import build_model, make_batch, LAYOUT_256
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(16, LAYOUT_256).items()}
opt = training.FusedAdam(model.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    sum(model(dict(batch)).values()).backward()
    opt.step(max_grad_norm=100.0)
for _ in range(2): step()
calls = []
orig = hip.gemm
import traceback
def logged(a, b, alpha=1.0, out=None, bias=None, relu=False):
    nb = max(a.shape[0] if a.dim() == 3 else 1, b.shape[0] if b.dim() == 3 else 1)
    M, K, N = a.shape[-2], a.shape[-1], b.shape[-2]
    at = a.stride(-1) != 1; bt = b.stride(-1) != 1
    fr = [f for f in traceback.extract_stack(limit=6) if 'ab_opt_amd' in f.filename and f.name != 'logged']
    where = '%s:%d %s' % (os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].name) if fr else '?'
    calls.append((nb, M, N, K, at, bt, bias is not None, relu, where))
    return orig(a, b, alpha, out, bias, relu)
hip.gemm = logged
for m in (training,):
    pass
import ab_opt_amd.training as T, ab_opt_amd.embed as E, ab_opt_amd.modules as Mo
step()
cnt = collections.Counter(calls)
print('%d gemm calls in one step' % len(calls))
for (k, c) in sorted(cnt.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3]):
    nb, M, N, K, at, bt, bias, relu, where = k
    print('%3d x  B=%-3d M=%-6d N=%-5d K=%-6d %s%s %s%s  GF=%.2f  %s' % (c, nb, M, N, K, 'T' if at else 'n', 'T' if bt else 'n', 'b' if bias else '-', 'r' if relu else '-', 2e-9 * nb * M * N * K * c, where))

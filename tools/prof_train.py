"""rocprofv3 target: a few native training steps of the diffusion module only (encode outputs precomputed)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from ab_opt_amd.utils.synth import build_model
from ab_opt_amd import training, hip
from ab_opt_amd.utils.synth import make_batch, LAYOUT_256
N, L = 16, 256
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(N, LAYOUT_256).items()}
with torch.no_grad():
    rf0, pf0, R0, p0 = model.encode(dict(batch), True, True)
v0 = hip.so3_log(R0, True)
for _ in range(4):
    rf = rf0.clone().requires_grad_(True); pf = pf0.clone().requires_grad_(True)
    loss = sum(model.diffusion(v0, p0, batch['aa'], rf, pf, batch['generate_flag'], batch['mask'], True, True).values())
    loss.backward()
    model.zero_grad(set_to_none=True)
torch.cuda.synchronize()

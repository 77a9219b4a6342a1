"""rocprofv3 target: full native training steps (encode + diffusion + Adam), N=16, L=256."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from ab_opt_amd import training
from ab_opt_amd.utils.synth import build_model
from ab_opt_amd.utils.synth import make_batch, LAYOUT_256
N, L = 16, 256
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(N, LAYOUT_256).items()}
opt = training.FusedAdam(model.parameters(), lr=1e-4)
for _ in range(int(os.environ.get("STEPS", "4"))):
    opt.zero_grad(set_to_none=True)
    sum(model(dict(batch)).values()).backward()
    opt.step(max_grad_norm=100.0)
torch.cuda.synchronize()

"""Where does a training step spend its time?  python tools/bench_train_parts.py [N] [L]"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from ab_opt_amd.utils.synth import build_model
from ab_opt_amd import training, hip
from ab_opt_amd.utils.synth import make_batch, LAYOUT_256, LAYOUT_128
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdesign', device=dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(N, LAYOUT_256 if L == 256 else LAYOUT_128).items()}


def timed(fn, n=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def enc_fwd():
    return model.encode(dict(batch), True, True)


def enc_fwd_bwd():
    rf, pf, R, p = model.encode(dict(batch), True, True)
    (rf.sum() + pf.sum()).backward()
    model.zero_grad(set_to_none=True)


with torch.no_grad():
    rf0, pf0, R0, p0 = model.encode(dict(batch), True, True)
v0 = hip.so3_log(R0, True)


def dpm_fwd_bwd(native, detach=True):
    training.NATIVE_IPA = native
    rf = rf0.clone().requires_grad_(not detach); pf = pf0.clone().requires_grad_(not detach)
    loss = sum(model.diffusion(v0, p0, batch['aa'], rf, pf, batch['generate_flag'], batch['mask'], True, True).values())
    loss.backward()
    model.zero_grad(set_to_none=True)


def dpm_fwd(native):
    training.NATIVE_IPA = native
    with torch.no_grad():
        pass
    rf = rf0.clone().requires_grad_(True); pf = pf0.clone().requires_grad_(True)
    return sum(model.diffusion(v0, p0, batch['aa'], rf, pf, batch['generate_flag'], batch['mask'], True, True).values())


print(f'N={N} L={L}')
print(f'encode fwd (autograd statement): {timed(enc_fwd):.1f} ms;  fwd+bwd: {timed(enc_fwd_bwd):.1f} ms')
for native in (True, False):
    print(f'native_ipa={native}: diffusion fwd {timed(lambda: dpm_fwd(native)):.1f} ms; fwd+bwd (grads to params + res/pair feat) {timed(lambda: dpm_fwd_bwd(native, False)):.1f} ms')

"""Developer check: HIP encode vs the torch statement in fp32 and fp64 on the device."""
import sys, copy
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases
from conftest import AttrDict
from ab_opt_amd import get_model
from ab_opt_amd.utils import synth
DEV = torch.device('cuda:0')
dev = lambda t: t.to(DEV)
L = 128
cfg = cases.cfg_abdock(10)
m = synth.fill_module_(get_model(AttrDict(cfg)).eval(), seed=17).to(DEV)
with torch.no_grad():
    m.pair_embed.aapair_to_distcoef.weight.copy_(dev(synth.hash_tensor(tuple(m.pair_embed.aapair_to_distcoef.weight.shape), 23, scale=2.0)))
batch = {k: dev(v) for k, v in synth.make_batch(3, synth.LAYOUT_128, seed=5, lengths=[L, L - 11, L // 2 + 3]).items()}
batch['pos_heavyatom'][:, :, 5:] = batch['pos_heavyatom'][:, :, 1:2] + dev(synth.hash_tensor((3, L, 10, 3), 41, scale=3.0))
batch['mask_heavyatom'][:, ::2, 5:12] = True
batch['mask_heavyatom'][:, ::6, 3] = False
batch['mask_heavyatom'] &= batch['mask'][:, :, None]
flags = (True, True)
with torch.enable_grad():
    ref32 = [t.detach() for t in m.encode(dict(batch), *flags)]
m64 = get_model(AttrDict(cases.cfg_abdock(10))).eval()
m64.load_state_dict(m.state_dict())
m64 = m64.to(DEV).double()
b64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in batch.items()}
with torch.enable_grad():
    ref64 = [t.detach() for t in m64.encode(dict(b64), *flags)]
with torch.no_grad():
    out = m.encode(dict(batch), *flags)
for name, a, b, c in zip(('res_feat', 'pair_feat', 'R', 'p'), out, ref32, ref64):
    s = c.abs().max().item()
    print(f'{name}: scale {s:.3e}  hip-vs-f64 {(a.double() - c).abs().max().item():.3e}  torch32-vs-f64 {(b.double() - c).abs().max().item():.3e}  hip-vs-torch32 {(a - b).abs().max().item():.3e}')
d = (out[1].double() - ref64[1]).abs()
idx = torch.nonzero(d == d.max())[0].tolist()
print('worst pair idx', idx, 'aa', batch['aa'][idx[0], idx[1]].item(), batch['aa'][idx[0], idx[2]].item())
print('mean abs err hip', d.mean().item(), 'torch32', (ref32[1].double() - ref64[1]).abs().mean().item())

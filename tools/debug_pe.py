"""Pair embedding: inference twice and the training path against the torch statement (full or backbone+CB inputs of test_encode_hip_vs_autograd_statement);
prints whether two runs repeat bit for bit and where entries are off.  RES=backbone+CB python tools/debug_pe.py"""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cases
from ab_opt_amd.utils import synth
from ab_opt_amd import get_model
from conftest import AttrDict
import plain_statement
DEV = torch.device('cuda:0'); dev = lambda t: t.to(DEV)
cfg = cases.cfg_abdock(10); import os
cfg['resolution'] = os.environ.get('RES', 'full')
for k in ('num_bins', 'dist_min', 'dist_max'): cfg.pop(k)
cfg['diffusion'].pop('obj')
m = synth.fill_module_(get_model(AttrDict(cfg)).eval(), seed=17).to(DEV)
with torch.no_grad():
    m.pair_embed.aapair_to_distcoef.weight.copy_(dev(synth.hash_tensor(tuple(m.pair_embed.aapair_to_distcoef.weight.shape), 23, scale=2.0)))
L = 256
batch = {k: dev(v) for k, v in synth.make_batch(3, synth.LAYOUT_256, seed=5, lengths=[L, L - 11, L // 2 + 3]).items()}
if cfg['resolution'] == 'full': batch['pos_heavyatom'][:, :, 5:] = batch['pos_heavyatom'][:, :, 1:2] + dev(synth.hash_tensor((3, L, 10, 3), 41, scale=3.0))
batch['mask_heavyatom'][:, ::2, 5:12] = True
batch['mask_heavyatom'][:, ::6, 3] = False
batch['mask_heavyatom'] &= batch['mask'][:, :, None]
batch['hotspot'] = (dev(synth.hash_tensor((3, L), 31, scale=1.0)) > 0.7).long() * batch['mask'].long()
flags = (True, True)
with torch.no_grad():
    ref = plain_statement.encode(m, dict(batch), *flags)[1]
    out = m.encode(dict(batch), *flags)[1]
    out2 = m.encode(dict(batch), *flags)[1]
with torch.enable_grad():
    tr = m.encode(dict(batch), *flags)[1].detach()
print('inference repeat equal', torch.equal(out, out2))
for name, x in (('inference', out), ('training', tr)):
    d = (x - ref).abs()
    print(name, 'max err vs torch', d.max().item(), 'mean', d.mean().item())
    idx = (d > 0.5).nonzero()
    print('  entries off by > 0.5:', idx.shape[0], idx[:8].tolist())
    if idx.shape[0]:
        print('  distinct n', idx[:, 0].unique().tolist()[:10], 'i', idx[:, 1].unique().tolist()[:20], 'j', idx[:, 2].unique().tolist()[:20])

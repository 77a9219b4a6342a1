"""Developer check: where do the IPA-core outputs differ from the golden ga_block fixture?"""
import sys, math
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import cases
from conftest import load_golden
from test_hip_parity import _block_on_device, dev
g = load_golden('ga_block')
blk = _block_on_device()
R, t, x, z, mask = cases.ipa_inputs(2, 24, [24, 19])
out, parts = blk(dev(R), dev(t), dev(x), dev(z), dev(mask), return_parts=True)
ref_logits = (g['l_node'] + g['l_pair'] + g['l_spat']) * math.sqrt(1 / 3)
d = (parts['logits'].cpu() - ref_logits).abs()
print('logits err max', d.max().item())
print('per n,i max:', d.amax(dim=(2, 3)))
print('per j max (n=0):', d[0].amax(dim=(0, 2)))
print('per h max:', d.amax(dim=(0, 1, 2)))
f = (parts['feat'].cpu() - g['feat']).abs()
print('feat pair', f[..., :768].max().item(), 'node', f[..., 768:1152].max().item(), 'pts', f[..., 1152:].max().item())

import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import torch, math, cases
from conftest import load_golden
from ab_opt_amd.utils import synth
from ab_opt_amd.modules import GABlock
g = load_golden('ga_block')
dev = torch.device('cuda:0')
blk = synth.fill_module_(GABlock(128, 64), seed=1).to(dev).eval()
R, t, x, z, mask = cases.ipa_inputs(2, 24, [24, 19])
out, parts = blk(*[a.to(dev) for a in (R, t, x, z, mask)], return_parts=True)
ref_logits = (g['l_node'] + g['l_pair'] + g['l_spat']) * math.sqrt(1 / 3)
d = (parts['logits'].cpu() - ref_logits).abs()
print('logits err max', d.max().item(), 'argmax', torch.nonzero(d == d.max())[0].tolist())
print('per-i max', d.amax((2, 3)))
f = (parts['feat'].cpu() - g['feat']).abs()
print('feat fp', f[..., :768].max().item(), 'fn', f[..., 768:1152].max().item(), 'pts', f[..., 1152:].max().item())
print('feat per-i', f.amax(-1))

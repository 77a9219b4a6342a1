"""How often does pair_embed_kernel disagree with its own first run?  200 inference launches on the race detector's input; prints the number of
launches that differ and how many 64-float pair records differ in the worst one.  ABOPT_LIB_PATH=<variant> python tools/r05/pe_repeat.py [full|backbone+CB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import cases
from ab_opt_amd.utils import synth
from ab_opt_amd import get_model
from conftest import AttrDict
DEV = torch.device('cuda:0'); dev = lambda t: t.to(DEV)
res = sys.argv[1] if len(sys.argv) > 1 else 'full'
cfg = cases.cfg_abdock(10); cfg['resolution'] = res
m = synth.fill_module_(get_model(AttrDict(cfg)).eval(), seed=17).to(DEV)
with torch.no_grad():
    m.pair_embed.aapair_to_distcoef.weight.copy_(dev(synth.hash_tensor(tuple(m.pair_embed.aapair_to_distcoef.weight.shape), 23, scale=2.0)))
L = 256
batch = {k: dev(v) for k, v in synth.make_batch(3, synth.LAYOUT_256, seed=5, lengths=[L, L - 11, L // 2 + 3]).items()}
batch['pos_heavyatom'][:, :, 5:] = batch['pos_heavyatom'][:, :, 1:2] + dev(synth.hash_tensor((3, L, 10, 3), 41, scale=3.0))
batch['mask_heavyatom'][:, ::2, 5:12] = True
batch['mask_heavyatom'][:, ::6, 3] = False
batch['mask_heavyatom'] &= batch['mask'][:, :, None]
import plain_statement
with torch.no_grad():
    ref = plain_statement.encode(m, dict(batch), True, True)[1]
    runs = [m.encode(dict(batch), True, True)[1].clone() for _ in range(200)]
# majority vote per element as the "right" answer is overkill: compare with the torch statement at a tolerance, and runs with each other
tol = 2e-4 * ref.abs().max().item()
bad_vs_ref = [int(((r - ref).abs() > tol).any(-1).sum()) for r in runs]
diff_vs_first = [int((r != runs[0]).any(-1).sum()) for r in runs]
print('lib', os.environ.get('ABOPT_LIB_PATH', 'product'), res, '| launches with pairs off the torch statement:', sum(b > 0 for b in bad_vs_ref), 'of 200, worst', max(bad_vs_ref),
      'pairs | launches differing from the first:', sum(d > 0 for d in diff_vs_first), 'worst', max(diff_vs_first), 'pairs')
if max(bad_vs_ref):
    r = runs[bad_vs_ref.index(max(bad_vs_ref))]
    idx = ((r - ref).abs() > tol).any(-1).nonzero()
    print('  example (n, i, j):', idx[:12].tolist(), ' j mod 16:', sorted(set((idx[:, 2] % 16).tolist()))[:16], ' j // 16 % 4:', sorted(set((idx[:, 2] // 16 % 4).tolist())))

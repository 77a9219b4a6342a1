#!/usr/bin/env python
"""End-to-end use on one MI355X, the way AbDock's design_for_pdb.py drives the model (synthetic complex, hash-filled weights):

    encode once -> N samples of one complex (shared context) -> all-atom backbone of every sample -> rank by commonness

    python tools/example_design.py [num_samples] [L]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from ab_opt_amd.utils.synth import build_model                      # get_model(cfg) + deterministic weights (no checkpoint exists offline)
from ab_opt_amd import sampler, geometry
from ab_opt_amd.utils.synth import make_batch, LAYOUT_256, LAYOUT_128

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
model = build_model(100, 7, flavour='abdock', device=dev).eval()
complex_ = {k: v.to(dev) for k, v in make_batch(1, LAYOUT_256 if L == 256 else LAYOUT_128, seed=2022).items()}
opt = {'sample_structure': True, 'sample_sequence': True}
rep = lambda a: a.expand(n, *a.shape[1:]).contiguous()


def design():
    traj = sampler.sample_replicated(model, complex_, n, opt)
    v, p, s, prmsd, ppl = traj[0]
    pos, mask = geometry.reconstruct_backbone_partially(pos_ctx=rep(complex_['pos_heavyatom']), R_new=geometry.so3vec_to_rotation(v), t_new=p, aa=s,
                                                        chain_nb=rep(complex_['chain_nb']), res_nb=rep(complex_['res_nb']),
                                                        mask_atoms=rep(complex_['mask_heavyatom']), mask_recons=rep(complex_['generate_flag']))
    gen = rep(complex_['generate_flag'])
    cand = pos[gen][:, :3].reshape(n, -1, 3)                # N, CA, C of the generated residues (design_for_pdb.py:326-336)
    return pos, mask, gen, prmsd, ppl, sampler.rank_commoness(cand, k=min(5, n))


design()                                                    # warm-up (library load, IGSO(3) CDF, tables, workspaces)
torch.cuda.synchronize(); t0 = time.perf_counter()
pos, mask, gen, prmsd, ppl, top = design()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'{n} designs of one {L}-residue complex: {dt * 1e3:.1f} ms end to end ({n * 100 / dt:.0f} sample-steps/s incl. encode, backbone rebuild and ranking)')
print('most common designs:', top.tolist(), '| predicted CA-RMSD of the best:', round(float(prmsd[top[0]]), 3), '| perplexity:', round(float(ppl[top[0]]), 3))
assert torch.isfinite(pos).all() and mask[gen][:, :4].all()

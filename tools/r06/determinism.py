"""Run-to-run determinism of the sampler inside ONE process: the by-complex test-set driver of tests/test_hip_parity.py
(test_two_rank_sharded_sampling_is_bit_identical_to_one_rank: 3 complexes x 4 samples, L = 128, 10 steps) repeated R times and compared
bit for bit with the first run.  Two copies started side by side share the GPU the way the test's two ranks do.

  python tools/r06/determinism.py [--reps 30] [--pollute] [--graph auto|off|on] [--stage all|encode|sample]
  --pollute: fill freed device memory with NaN patterns between runs;  --stage encode: only the encoder's outputs are compared;
  --stage sample: one encode, then the sampler alone on the same features
"""
import argparse
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--pollute', action='store_true')
    ap.add_argument('--tag', default='')
    ap.add_argument('--graph', default='auto')
    ap.add_argument('--stage', default='all')
    ap.add_argument('--samples', type=int, default=4)
    ap.add_argument('--sync', default='', help='comma list: enc = synchronise after model.encode, pre = before it')
    a = ap.parse_args()
    from conftest import build_model
    from ab_opt_amd import sampler, hip
    from ab_opt_amd.utils import synth
    dev = torch.device('cuda:0')
    m = build_model(10, 3, device=dev)
    m.diffusion.graph_mode = {'auto': 'auto', 'off': False, 'on': True}[a.graph]
    cx = [{k: v.to(dev) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=100 + c).items()} for c in range(3)]
    S = a.samples
    if a.sync:
        enc0 = m.encode

        def enc(*x, **k):
            if 'pre' in a.sync:
                torch.cuda.synchronize()
            o = enc0(*x, **k)
            if 'enc' in a.sync:
                torch.cuda.synchronize()
            return o
        m.encode = enc

    def one_run():
        if a.stage == 'all':
            res = sampler.design_testset_sharded(m, cx, S, k=2, seed=7, complexes_per_launch=1)
            return [(r['ca'], r['score']) for r in res]
        out = []
        for c in cx:
            with torch.no_grad():
                rf, pf, R0, p0 = m.encode(c, remove_structure=True, remove_sequence=True)
            if a.stage == 'encode':
                out.append((rf.cpu(), pf.cpu(), R0.cpu(), p0.cpu()))
                continue
            if not hasattr(one_run, 'enc'):
                one_run.enc = {}
            if a.stage == 'sample':
                rf, pf, R0, p0 = one_run.enc.setdefault(id(c), (rf, pf, R0, p0))
            pf_before = pf.clone() if a.stage == 'steps' else None          # output 37: pair_feat as the encoder left it; 34: after the sampler has run
            rep = lambda t: t.repeat_interleave(S, dim=0).contiguous()
            tr = m.diffusion.sample(rep(hip.so3_log(R0, grad_mode=False)), rep(p0), rep(c['aa']), rf, pf, rep(c['generate_flag']), rep(c['mask']),
                                    sample_structure=True, sample_sequence=True, seed=7, rng_offset=0)
            if a.stage == 'steps':                                           # whole trajectory + the features it was sampled from
                out.append(tuple(x.cpu() for t in sorted(tr, reverse=True) for x in tr[t][:3]) + (rf.cpu(), pf.cpu(), R0.cpu(), p0.cpu(), pf_before.cpu()))
                continue
            out.append(tuple(t.cpu() for t in tr[0][:3]) + tuple(t.cpu() for t in tr[5][:3]))
        return out

    ref, bad, nwarn = None, 0, 0
    for r in range(a.reps):
        if a.pollute:
            junk = [torch.full((n,), float('nan'), device=dev) for n in (1 << 20, 1 << 22, 1 << 24, 3 << 24)]
            del junk
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            res = one_run()
        nwarn += sum('fp16 range' in str(x.message) for x in w)
        if ref is None:
            ref = res
            continue
        for ci, (g_, r_) in enumerate(zip(res, ref)):
            d = [i for i, (x, y) in enumerate(zip(g_, r_)) if not torch.equal(x, y)]
            if d:
                bad += 1
                if bad <= 6:
                    x, y = g_[d[0]].float(), r_[d[0]].float()
                    print('%s rep %d complex %d differs in outputs %s: first max |d| %.3e, %d of %d elements' % (a.tag, r, ci, d, (x - y).abs().max().item(), int((x != y).sum()), x.numel()), flush=True)
                    if a.stage == 'encode':
                        for i in d:
                            x, y = g_[i].float(), r_[i].float()
                            ne = (x != y)
                            idx = ne.nonzero()
                            print('   encode output %d %s: max |d| %.3e (max |ref| %.3e), %d elements; index min %s max %s; distinct i %s; distinct j %s; channels %s' % (
                                i, tuple(x.shape), (x - y).abs().max().item(), y.abs().max().item(), int(ne.sum()), idx.min(0)[0].tolist(), idx.max(0)[0].tolist(),
                                idx[:, 1].unique().tolist()[:40], idx[:, 2].unique().tolist()[:40] if idx.shape[1] > 2 else '-', idx[:, -1].unique().tolist()[:64]), flush=True)
                    if a.stage == 'steps' and 37 in d:
                        x = g_[37]
                        ne = (x != r_[37]).reshape(128, 8, 16 * 64).any(-1).nonzero()
                        for i_, t_ in ne.tolist():
                            tile = x[0, i_, t_ * 16:t_ * 16 + 16]
                            same = [('complex %d' % c2) for c2 in range(len(ref)) if torch.equal(ref[c2][37][0, i_, t_ * 16:t_ * 16 + 16], tile)]
                            d_own = (tile - r_[37][0, i_, t_ * 16:t_ * 16 + 16]).abs().max().item()
                            d_oth = [(tile - ref[c2][37][0, i_, t_ * 16:t_ * 16 + 16]).abs().max().item() for c2 in range(len(ref))]
                            print('   tile (i %d, j %d..): equals the reference tile of %s; max |d| to own ref %.3e, to the complexes %s; zeros %d' % (
                                i_, t_ * 16, same or 'none', d_own, ['%.2e' % v for v in d_oth], int((tile == 0).sum())), flush=True)
                    if a.stage == 'steps' and False:
                        for i in d[:4]:
                            x, y = g_[i].float(), r_[i].float()
                            ne = (x != y)
                            rows = ne.reshape(ne.shape[0], ne.shape[1], -1).any(-1).nonzero() if ne.dim() >= 2 else ne.nonzero()
                            print('   output %d (%s of step index %d; 33.. = res_feat, pair_feat, R0, p0): max |d| %.3e, %d elements, first (sample, residue) %s ... last %s' % (
                                i, 'vps'[i % 3] if i < 33 else '-', i // 3, (x - y).abs().max().item(), int(ne.sum()), rows[0].tolist(), rows[-1].tolist()), flush=True)
    print('%s [%s graph=%s] %d reps, %d differing (complex, rep) pairs; %d range-guard reruns' % (a.tag, a.stage, a.graph, a.reps, bad, nwarn), flush=True)


if __name__ == '__main__':
    main()

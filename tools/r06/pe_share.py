"""The pair embedding (csrc/embed.hip: pair_embed_kernel) under a second process on the same GPU.

  victim : K encodes of one batch queued back to back, R times; every pair_feat is compared bit for bit with the first one; differing launches are
           characterised (rows i, 16-pair tiles, size of the error)
  partner: a load in a loop for --seconds: sampler (the test-set driver) | loop (the sampler on fixed features) | eps (abopt_eps_net_forward) | gablock |
           gaenc | tail | pbc | post | enc1 / embed (the pair embedding itself) | matmul | copy | valu | idle
  victim --acts: the training form of the launch; the per-layer activation dump names the first layer that differs

Round 6 (DESIGN_LOG.md): partner eps / loop / gablock / gaenc -> 0.1-3 % of the victim's launches wrong with the literal dihedral arithmetic (-DDIH_MODE=0),
0 of 57 k with the shipped form; partner idle / embed / matmul / copy / valu / tail / pbc / post -> 0.

  python tools/r06/pe_share.py victim --n 4 --layout 256 --reps 30 --burst 10
  python tools/r06/pe_share.py partner --kind matmul --seconds 60
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('role')
    ap.add_argument('--kind', default='sampler')
    ap.add_argument('--seconds', type=float, default=60)
    ap.add_argument('--n', type=int, default=4)
    ap.add_argument('--layout', type=int, default=256)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--burst', type=int, default=10)
    ap.add_argument('--tag', default='')
    ap.add_argument('--resolution', default='full')
    ap.add_argument('--ready-file', default='', help='partner: created when the load starts (a shell can wait for it)')
    ap.add_argument('--acts', action='store_true', help='victim: the training form of the launch (per-layer activation dump); the first wrong layer of a bad tile is reported')
    a = ap.parse_args()
    from conftest import build_model
    from ab_opt_amd import sampler
    from ab_opt_amd.utils import synth
    dev = torch.device('cuda:0')
    m = build_model(10, 3, device=dev)
    lay = synth.LAYOUT_256 if a.layout == 256 else synth.LAYOUT_128
    if a.role == 'partner':
        t0, it = time.time(), 0
        if a.kind == 'sampler':
            cx = [{k: v.to(dev) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=100 + c).items()} for c in range(3)]
        elif a.kind == 'embed':
            b = {k: v.to(dev) for k, v in synth.make_batch(a.n, lay, seed=5).items()}
        if a.kind in ('eps', 'pbc', 'loop', 'post', 'enc1', 'gablock', 'tail', 'gaenc'):
            from ab_opt_amd import hip
            c = {k: v.to(dev) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=100).items()}
            S = 4
            with torch.no_grad():
                rf, pf, R0, p0 = m.encode(c, remove_structure=True, remove_sequence=True)
            rep = lambda t: t.repeat_interleave(S, dim=0).contiguous()
            d = m.diffusion
            v0, pn, s0 = rep(hip.so3_log(R0, grad_mode=False)), rep(p0) / 10.0, rep(c['aa'])
            gen, mres = rep(c['generate_flag']), rep(c['mask'])
            beta = d.trans_pos.var_sched.betas[5].expand([S]).contiguous()
            arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
            nl = len(d.eps_net.encoder.blocks)
            pbc = hip.pair_bias_cache(arr, nl, pf)
            rfr = rep(rf)
            blk = d.eps_net.encoder.blocks[0]
            tns, st = blk.packed()
            arr1 = (hip.GaWeights * 1)(st)
            pbc1 = hip.pair_bias_cache(arr1, 1, pf)
            Rr, tr_ = rep(R0), rep(p0) / 10.0
            feat = torch.randn(S * 128, 1824, device=dev)
        x = torch.randn(4096, 4096, device=dev)
        big = torch.randn(64 << 20, device=dev)
        if a.ready_file:
            open(a.ready_file, 'w').write('ready')
        t0 = time.time()
        while time.time() - t0 < a.seconds:
            it += 1
            if a.kind == 'sampler':
                sampler.design_testset_sharded(m, cx, 4, k=2, seed=7, complexes_per_launch=1)
            elif a.kind == 'embed':
                with torch.no_grad():
                    for _ in range(10):
                        m.encode(b, remove_structure=True, remove_sequence=True)
                torch.cuda.synchronize()
            elif a.kind == 'eps':
                for _ in range(20):
                    hip.eps_net_forward(ew, v0, pn, s0, rfr, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc, pair_feat_shared=S)
                torch.cuda.synchronize()
            elif a.kind == 'gablock':
                for _ in range(20):
                    hip.ga_block_forward_cached(st, Rr, tr_, rfr, pf, mres, pbc1, None, pair_feat_shared=S)
                torch.cuda.synchronize()
            elif a.kind == 'gaenc':
                for _ in range(20):
                    hip.ga_encoder_forward(arr, nl, Rr, tr_, rfr, rep(pf), mres)
                torch.cuda.synchronize()
            elif a.kind == 'tail':
                for _ in range(20):
                    hip.block_tail_forward(feat, tns['w_out_frag'], tns['w_mlp_frag'], rfr.reshape(-1, 128), tns['b_out'], mres.reshape(-1), tns['ln1_gamma'],
                                           tns['ln1_beta'], tns['b_mlp0'], tns['b_mlp1'], tns['b_mlp2'], tns['ln2_gamma'], tns['ln2_beta'])
                torch.cuda.synchronize()
            elif a.kind == 'pbc':
                for _ in range(20):
                    hip.pair_bias_cache(arr, nl, pf)
                torch.cuda.synchronize()
            elif a.kind == 'loop':
                with torch.no_grad():
                    d.sample(v0, rep(p0), s0, rf, pf, gen, mres, sample_structure=True, sample_sequence=True, seed=7, rng_offset=0)
            elif a.kind == 'enc1':
                with torch.no_grad():
                    for _ in range(10):
                        m.encode(c, remove_structure=True, remove_sequence=True)
                torch.cuda.synchronize()
            elif a.kind == 'post':
                for _ in range(20):
                    cand = sampler.candidates_from_positions(rep(p0), gen)
                    sc = sampler.commonness_score(cand)
                    torch.topk(sc, k=2, largest=False)
                torch.cuda.synchronize()
            elif a.kind == 'matmul':
                for _ in range(20):
                    y = x @ x
                torch.cuda.synchronize()
            elif a.kind == 'copy':
                for _ in range(20):
                    y = big.clone()
                torch.cuda.synchronize()
            elif a.kind == 'valu':
                for _ in range(20):
                    y = torch.sin(big) * 1.0001 + big
                torch.cuda.synchronize()
            else:
                time.sleep(0.2)
        print('partner %s: %d iterations in %.0f s' % (a.kind, it, time.time() - t0), flush=True)
        return
    b = {k: v.to(dev) for k, v in synth.make_batch(a.n, lay, seed=11).items()}
    ref, bad, launches, t0 = None, 0, 0, time.time()
    shown = 0
    if a.acts:
        from ab_opt_amd import hip
        ctx = torch.logical_and(b['mask_heavyatom'][:, :, 1], ~b['generate_flag'])
        inp, keep = hip.encode_inputs(b['aa'], b['res_nb'], b['chain_nb'], b['pos_heavyatom'], b['mask_heavyatom'], m.residue_embed.max_num_atoms,
                                      fragment_type=b['fragment_type'], structure_mask=ctx, sequence_mask=ctx)
        w = m.pair_embed._hip_weights()
        segs = [('G', None), ('relu(D0)', (0, 64)), ('f_dist', (64, 128)), ('f_dih', (128, 160)), ('relu(O0)', (160, 224)), ('relu(O1)', (224, 288))]
        refs = None
        while time.time() - t0 < a.seconds:
            outs = [hip.pair_embed_forward(inp, w, save_activations=True) for _ in range(a.burst)]
            torch.cuda.synchronize()
            if refs is None:
                refs = [t.clone() for t in outs[0][:3]]
            for pf, acts, G, _ in outs:
                launches += 1
                if torch.equal(pf, refs[0]) and torch.equal(acts, refs[1]) and torch.equal(G, refs[2]):
                    continue
                bad += 1
                if shown < 8:
                    shown += 1
                    rep_ = []
                    for name, sl in segs[3:4]:
                        x, y = (G, refs[2]) if sl is None else (acts[..., sl[0]:sl[1]], refs[1][..., sl[0]:sl[1]])
                        ne = (x != y)
                        if ne.any():
                            idx = ne.nonzero()
                            tiles = sorted({(int(i), int(j) // 16) for _, i, j, _ in idx.tolist()[::7]})
                            rep_.append('%s: %d el, max |d| %.2e, tiles %s, last-dim idx %s' % (name, int(ne.sum()), (x - y).abs().max().item(), tiles[:4], idx[:, 3].unique().tolist()[:24]))
                    ne12 = (acts[0, :, :, 140] != refs[1][0, :, :, 140]).reshape(acts.shape[1], -1, 16).any(-1).nonzero()
                    for i_, t_ in ne12.tolist()[:3]:
                        badv = acts[0, i_, t_ * 16:t_ * 16 + 16, 140]
                        own = refs[1][0, i_, t_ * 16:t_ * 16 + 16, 140]
                        prev = refs[1][0, i_, t_ * 16 - 16:t_ * 16, 140] if t_ % 4 else None
                        x0 = refs[1][0, i_, t_ * 16:t_ * 16 + 16, 128]
                        x1 = refs[1][0, i_, t_ * 16:t_ * 16 + 16, 141]
                        allx0 = refs[1][0, :, :, 128]
                        hits = [(allx0 * (1.0 / 3.0) == v).nonzero().tolist()[:2] for v in badv.tolist()[:3]]
                        hits3 = [((allx0 * torch.tensor(1.0 / 3.0, device=allx0.device)) .sub(v).abs() < 1e-6).nonzero().tolist()[:2] for v in badv.tolist()[:3]]
                        rep_.append('bad/x1 %s; (i, j) whose x0 / 3 gives the bad value (lanes 0-2): exact %s, within 1e-6 %s' % ([round(v, 3) for v in (badv / x1).tolist()[:4]], hits, hits3))
                        rep_.append('tile (%d, %d) f_dih[12]: equals the PREVIOUS tile\'s values %s; bad/own ratio %s; bad/x0 %s' % (
                            i_, t_, None if prev is None else bool(torch.equal(badv, prev)), [round(v, 3) for v in (badv / own).tolist()[:4]], [round(v, 3) for v in (badv / x0).tolist()[:4]]))
                    ne = (pf != refs[0])
                    print('%s launch %d: out %d el; %s' % (a.tag, launches, int(ne.sum()), ' | '.join(rep_) or 'no dump differs'), flush=True)
            del outs
        print('%s victim (activation dump) N=%d L=%d: %d of %d launches differ from the first (%.0f s)' % (a.tag, a.n, a.layout, bad, launches, time.time() - t0), flush=True)
        return
    with torch.no_grad():
        r = -1
        while (time.time() - t0 < a.seconds) if a.reps <= 0 else (r + 1 < a.reps):
            r += 1
            outs = [m.encode(b, remove_structure=True, remove_sequence=True)[1] for _ in range(a.burst)]
            torch.cuda.synchronize()
            if ref is None:
                ref = outs[0].clone()
            for o in outs:
                launches += 1
                if not torch.equal(o, ref):
                    bad += 1
                    if shown < 5:
                        shown += 1
                        ne = (o != ref)
                        idx = ne.nonzero()
                        tiles = sorted({(int(n), int(i), int(j) // 16) for n, i, j, _ in idx.tolist()[::16]})
                        print('%s launch %d: %d elements differ, max |d| %.3e (max |ref| %.3e); (sample, i, 16-pair tile) %s' % (
                            a.tag, launches, int(ne.sum()), (o - ref).abs().max().item(), ref.abs().max().item(), tiles[:12]), flush=True)
            del outs
    print('%s victim N=%d L=%d: %d of %d launches differ from the first (%.0f s)' % (a.tag, a.n, a.layout, bad, launches, time.time() - t0), flush=True)


if __name__ == '__main__':
    main()

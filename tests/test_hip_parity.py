"""Parity of the HIP path (through the C ABI) with the reference goldens and the CPU oracle.  Needs an MI355X.

Tolerances (fp32, stated per SURVEY.md section 8c / BASELINE.md section 4):
  * per-call outputs (R_next, eps_pos, c, prmsd logits, block outputs)   1e-5 .. 3e-5 abs
  * positions after a teacher-forced step                                1e-4 Angstrom
  * orientations are compared as rotation MATRICES; where the reference's own log map is ill-conditioned
    (theta -> pi, error ~ eps/sin(theta); garbage when cos(theta) clamps at -1) the tolerance
    scales with 1/sin(theta) and the clamp zone is excluded.
"""
import math
import os
import pytest
import torch
import torch.nn.functional as F

import cases
from conftest import load_golden, build_model, max_abs
from ab_opt_amd.utils import synth
from test_oracle_golden import standalone_block_sd, standalone_abdesign_dpm, noise_dict

pytestmark = pytest.mark.gpu

DEV = torch.device('cuda:0')


def dev(x):
    return x.to(DEV) if isinstance(x, torch.Tensor) else x


def _log_amplification(R):
    """(ok, a = 1/sin(theta)) of the reference log map at rotation matrices R (so3.py:10-22).  Its direction error is
    ~eps*a, but its MAGNITUDE error is ~eps*a^2: sin(theta) in the coefficient comes from sqrt(1 - cos^2) with cos
    good to one ulp, i.e. a relative error eps/(1+cos) ~ eps*a^2.  Once cos clamps at -1 the output is garbage."""
    cos = ((R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]) - 1) / 2
    ok = cos > -1 + 2e-5
    return ok, 1.0 / torch.sqrt((1 - cos.clamp(-1, 1) ** 2).clamp_min(1e-12))


def rot_close(v_got, v_ref, R_pre, base=2e-6, cap=0.05, R_upstream=None):
    """exp(v_got) ~ exp(v_ref) under the conditioning of the reference's own formulas.  R_pre is the (well-conditioned)
    matrix that was fed to the log map producing v; R_upstream, if given, the matrix of an earlier log map the value
    also went through (the network's R_next -> v_next).  Per-residue tolerance base*(a_up^2*a + a^2) + 1e-5; residues
    whose tolerance exceeds `cap` (the reference's own result is noise there) or that sit in the clamp zone are skipped.
    Calibration: the CPU oracle with matmul-ordered sums vs the recorded reference reaches 0.6 of this bound at
    base=1.2e-6 over the 10 recorded steps.  Returns (#checked, worst err/tol)."""
    from oracle import geometry as G
    Ra, Rb = G.so3_exp(v_got), G.so3_exp(v_ref)
    err = (Ra - Rb).abs().amax((-1, -2))
    ok, amp = _log_amplification(R_pre)
    tol = base * amp ** 2
    if R_upstream is not None:
        ok2, amp2 = _log_amplification(R_upstream)
        ok, tol = ok & ok2, base * (amp2 ** 2 * amp + amp ** 2)
    tol = tol + 1e-5
    ok = ok & (tol < cap)
    ratio = (err / tol)[ok]
    return int(ok.sum()), (ratio.max().item() if ratio.numel() else 0.0)


# ------------------------------------------------------------------------------------------ SO(3)
def test_so3_maps_vs_reference():
    from ab_opt_amd import hip
    g = load_golden('so3')
    R = hip.so3_exp(dev(cases.SO3_EDGE_W)).cpu()
    assert max_abs(R, g['exp']) < 1e-6
    assert max_abs(hip.so3_log(dev(g['exp']), False).cpu(), g['log_nograd']) < 5e-6
    assert max_abs(hip.so3_log(dev(g['exp']), True).cpu(), g['log_grad']) < 5e-6
    # round trip on 1e5 random vectors with theta < 3 (well conditioned)
    w = synth.hash_tensor((100000, 3), 99, scale=3.4)
    w = w[w.norm(dim=-1) < 3.0]
    back = hip.so3_log(hip.so3_exp(dev(w)), False).cpu()
    assert max_abs(back, w) < 5e-4 and (back - w).abs().mean() < 2e-6


# ------------------------------------------------------------------------------------------ GABlock
def _block_on_device(seed=1):
    from ab_opt_amd.modules import GABlock
    return synth.fill_module_(GABlock(128, 64), seed=seed).to(DEV).eval()


def test_ga_block_parts_vs_reference():
    g = load_golden('ga_block')
    blk = _block_on_device()
    R, t, x, z, mask = cases.ipa_inputs(2, 24, [24, 19])
    out, parts = blk(dev(R), dev(t), dev(x), dev(z), dev(mask), return_parts=True)
    ref_logits = (g['l_node'] + g['l_pair'] + g['l_spat']) * math.sqrt(1 / 3)
    assert max_abs(parts['logits'].cpu(), ref_logits) < 2e-5 * max(1.0, ref_logits.abs().max().item())
    assert max_abs(parts['alpha'].cpu(), g['alpha']) < 1e-5
    valid = mask[:, :, None].expand_as(g['feat'])
    assert max_abs(parts['feat'].cpu()[valid], g['feat'][valid]) < 2e-5
    assert max_abs(parts['feat'].cpu(), g['feat']) < 2e-5          # masked query rows too (they hold -R^T t, ga.py:136)
    assert max_abs(out.cpu(), g['out']) < 2e-5                      # all rows incl. padding (SURVEY s9 gotcha 1)


def test_ga_block_L128_vs_reference():
    g = load_golden('ga_block_L128')
    blk = _block_on_device()
    R, t, x, z, mask = cases.ipa_inputs(2, 128, [128, 101], salt=150)
    out = blk(dev(R), dev(t), dev(x), dev(z), dev(mask))
    assert max_abs(out.cpu(), g['out']) < 3e-5


@pytest.mark.parametrize('N,L,lengths', [(1, 1, [1]), (3, 7, [7, 1, 4]), (2, 65, [65, 64]), (1, 200, [200]), (2, 256, [256, 250]), (1, 521, [521])])
def test_ga_block_ragged_vs_oracle(N, L, lengths):
    from oracle import ipa
    blk = _block_on_device(seed=5)
    sd = {k: v.cpu() for k, v in blk.state_dict().items()}
    R, t, x, z, mask = cases.ipa_inputs(N, L, lengths, salt=400 + L)
    ref = ipa.ga_block(sd, '', R, t, x, z, mask, mode='mm')
    out = blk(dev(R), dev(t), dev(x), dev(z), dev(mask))
    assert max_abs(out.cpu(), ref) < 3e-5


def test_ga_block_empty_batch():
    blk = _block_on_device()
    out = blk(torch.zeros(0, 8, 3, 3, device=DEV), torch.zeros(0, 8, 3, device=DEV), torch.zeros(0, 8, 128, device=DEV),
              torch.zeros(0, 8, 8, 64, device=DEV), torch.zeros(0, 8, dtype=torch.bool, device=DEV))
    assert out.shape == (0, 8, 128)


def test_ga_encoder_matches_block_chain():
    from ab_opt_amd.modules import GAEncoder
    enc = synth.fill_module_(GAEncoder(128, 64, 3), seed=6).to(DEV).eval()
    R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(2, 33, [33, 20], salt=500)]
    y = x
    for b in enc.blocks:
        y = b(R, t, y, z, mask)
    assert torch.equal(enc(R, t, x, z, mask), y)


# ------------------------------------------------------------------------------------------ full-size properties
def test_ipa_full_size_invariances():
    """BASELINE config-2 shape (N=4 of the 32, L=256): SE(3) invariance, batch permutation, padding independence."""
    from oracle import geometry as G
    blk = _block_on_device(seed=7)
    N, L = 4, 256
    R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(N, L, [256, 256, 231, 256], salt=600)]
    base = blk(R, t, x, z, mask)
    # global rigid motion of every frame leaves the invariant features unchanged
    Q = dev(G.so3_exp(torch.tensor([[0.3, -1.1, 0.7]])))[0]
    shift = torch.tensor([1.5, -2.0, 0.25], device=DEV)
    moved = blk(Q @ R, t @ Q.T + shift, x, z, mask)
    assert max_abs(moved, base) < 2e-4
    # batch permutation equivariance (bit-exact: every sample is processed independently)
    perm = torch.tensor([2, 0, 3, 1], device=DEV)
    assert torch.equal(blk(R[perm], t[perm], x[perm], z[perm], mask[perm]), base[perm])
    # values in padded key columns / rows of z never reach valid rows
    z2 = z.clone()
    z2[2, :, 231:] = 1e3
    z2[2, 231:, :] = -1e3
    out2 = blk(R, t, x, z2, mask)
    assert torch.equal(out2[2, :231], base[2, :231]) and torch.equal(out2[[0, 1, 3]], base[[0, 1, 3]])


# ------------------------------------------------------------------------------------------ EpsilonNet
def _check_eps(out, g, has_prmsd):
    v_next, R_next, eps_pos, c = [o.cpu() for o in out[:4]]
    assert max_abs(R_next, g['R_next']) < 2e-5
    n, worst = rot_close(v_next, g['v_next'], g['R_next'])
    assert n >= 0.9 * v_next[..., 0].numel() and worst < 1.0, (n, worst)
    assert max_abs(eps_pos, g['eps_pos']) < 2e-5
    assert max_abs(c, g['c']) < 1e-5
    if has_prmsd:
        assert max_abs(out[4].cpu(), g['prmsd_logits']) < 2e-5


def test_eps_net_abdock_vs_reference():
    m = build_model(100, 2, device=DEV)
    for tag, (N, L, lens, gr) in dict(small=(2, 40, [40, 33], [(5, 14), (22, 30)]), L128=(1, 128, [128], [(30, 42)])).items():
        g = load_golden(f'eps_net_abdock_{tag}')
        args = [dev(a) for a in cases.eps_inputs(N, L, lens, gr)]
        _check_eps(m.diffusion.eps_net(*args), g, True)


def test_eps_net_abdesign_vs_reference():
    d = standalone_abdesign_dpm(100, 2).to(DEV)
    g = load_golden('eps_net_abdesign_small')
    args = [dev(a) for a in cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)])]
    _check_eps(d.eps_net(*args), g, False)


def test_eps_net_grad_mode_clamp():
    """grad_mode selects the -0.999 cosine clamp the reference uses under autograd (so3.py:12-16)."""
    from oracle import dpm
    m_cpu = build_model(100, 2)
    m = build_model(100, 2, device=DEV)
    args = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)])
    ref = dpm.eps_net(m_cpu.state_dict(), 'diffusion.eps_net.', *args, num_layers=6, prmsd_head=True, grad_mode=True)
    out = m.diffusion.eps_net(*[dev(a) for a in args], grad_mode=True)
    n, worst = rot_close(out[0].cpu(), ref[0], ref[1])
    assert worst < 1.0


# ------------------------------------------------------------------------------------------ sampler, teacher-forced
def _traj_setup(T=10, seed=3):
    from oracle import embed
    m_cpu = build_model(T, seed)
    m = build_model(T, seed, device=DEV)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=2022, lengths=[128, 117])
    return m_cpu, m, batch


def test_encode_on_device_vs_reference():
    g = load_golden('trajectory_abdock_T10')
    _, m, batch = _traj_setup()
    with torch.no_grad():
        rf, pf, R0, p0 = m.encode({k: dev(v) for k, v in batch.items()}, True, True)
    # HIP encode (csrc/embed.hip).  The hash-filled embedding tables make these features O(1e3) with heavy cancellation,
    # so compare relative to the largest magnitude
    assert max_abs(rf.cpu(), g['res_feat']) < 2e-4 * g['res_feat'].abs().max().item()
    assert max_abs(pf.cpu()[:, ::7, ::5], g['pair_feat_sub']) < 2e-4 * g['pair_feat_sub'].abs().max().item()
    assert max_abs(R0.cpu(), g['R0']) < 2e-6


def test_encode_small_fixture_ragged_masks():
    """The reference's encode() on ragged lengths (24 / 19 of 128), two chains, both mask modes (golden encode_small)."""
    g = load_golden('encode_small')
    m = build_model(10, 3, device=DEV)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=99, lengths=[24, 19])
    batch['generate_flag'][:, 8:13] = True
    batch['fragment_type'][:, :12] = 1
    batch['fragment_type'][:, 12:] = 3
    batch['fragment_type'] = batch['fragment_type'] * batch['mask']
    batch['chain_nb'][:, 12:] = 1
    b = {k: dev(v) for k, v in batch.items()}
    with torch.no_grad():
        rf, pf, R0, p0 = m.encode(dict(b), True, True)
        rf2, pf2, _, _ = m.encode(dict(b), True, False)
    sr, sp = g['res_feat'].abs().max().item(), g['pair_feat'].abs().max().item()
    eye = torch.eye(pf.shape[1], dtype=torch.bool)[None, :, :, None]           # diagonal: sign of a zero triple product, see below
    dpf = (pf.cpu() - g['pair_feat']).abs()
    assert max_abs(rf.cpu(), g['res_feat']) < 2e-4 * sr and (dpf * ~eye).max().item() < 2e-4 * sp and (dpf * eye).max().item() < 2e-3 * sp
    assert max_abs(R0.cpu(), g['R0']) < 2e-6 and max_abs(p0.cpu(), g['p0']) == 0
    assert max_abs(rf2.cpu(), g['res_feat_seqkept']) < 2e-4 * sr
    assert max_abs(pf2.cpu().double().sum((1, 2)), g['pair_feat_seqkept_sum']) < 2e-4 * g['pair_feat_seqkept_sum'].abs().max().item()


def test_encode_training_gradients_vs_reference():
    """encode() on the training path (HIP Gaussian features + structured backward helpers + autograd) against the
    reference's recorded parameter gradients (golden encode_small)."""
    g = load_golden('encode_small')
    m = build_model(10, 3, device=DEV)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=99, lengths=[24, 19])
    batch['generate_flag'][:, 8:13] = True
    batch['fragment_type'][:, :12] = 1
    batch['fragment_type'][:, 12:] = 3
    batch['fragment_type'] = batch['fragment_type'] * batch['mask']
    batch['chain_nb'][:, 12:] = 1
    b = {k: dev(v) for k, v in batch.items()}
    m.zero_grad()
    with torch.enable_grad():
        rf, pf, _, _ = m.encode(b, True, True)
        w1, w2 = dev(synth.hash_tensor(tuple(rf.shape), 71, scale=1.0)), dev(synth.hash_tensor(tuple(pf.shape), 72, scale=1.0))
        ((rf * w1).sum() + (pf * w2).sum()).backward()
    P = dict(m.named_parameters())
    checked = 0
    for k in g:
        if not k.startswith('grad_'):
            continue
        name = k[len('grad_'):].replace('__', '.')
        got = P['residue_embed.mlp.0.weight'].grad[::4, ::7] if name.endswith('_sub') else P[name].grad
        assert max_abs(got.cpu(), g[k]) <= 5e-4 * max(1e-6, g[k].abs().max().item()), name
        checked += 1
    assert checked == 10
    m.zero_grad()


def test_encode_L256_training_gradients_vs_reference():
    """encode() on the device at config-5 sample size against the REFERENCE (`encode_L256` fixture; VERDICT r04 weak 1-iv: the backward of
    encode() was pinned to the reference only on a 24 / 19-residue batch): forward values, then the gradient of every embedding parameter
    through the training path (pair_embed forward with its dump, pair_embed_backward, the tall weight-gradient products, the segmented bucket
    sums of the amino-acid-pair tables, the residue-feature kernels) at 5e-4 of its maximum."""
    g = load_golden('encode_L256')
    m = synth.fresh_model(10, 3, device=DEV)
    with torch.no_grad():
        m.pair_embed.aapair_to_distcoef.weight.copy_(dev(cases.encode_full_distcoef(m.pair_embed.aapair_to_distcoef.weight.shape)))
    b = {k: dev(v) for k, v in cases.encode_full_batch().items()}
    with torch.no_grad():
        rf, pf, R0, _ = m.encode(dict(b), True, True)
    eye = torch.eye(256, dtype=torch.bool)[None, ::9, ::7, None]
    d = (pf.cpu()[:, ::9, ::7] - g['pair_feat_sub']).abs()
    sc = g['pair_feat_sub'].abs().max().item()
    assert max_abs(rf.cpu(), g['res_feat']) < 2e-4 * g['res_feat'].abs().max().item() and max_abs(R0.cpu(), g['R0']) < 2e-6
    assert (d * ~eye).max().item() < 2e-4 * sc and (d * eye).max().item() < 2e-3 * sc        # (i == j: sign of a zero triple product, see test_encode_hip_vs_autograd_statement)
    assert max_abs(pf.double().sum((1, 2)).cpu(), g['pair_feat_sum']) < 2e-4 * g['pair_feat_sum'].abs().max().item()
    m.zero_grad()
    with torch.enable_grad():
        rf, pf, _, _ = m.encode(dict(b), True, True)
        w1, w2 = dev(synth.hash_tensor(tuple(rf.shape), 71, scale=1.0)), dev(synth.hash_tensor(tuple(pf.shape), 72, scale=1.0))
        ((rf * w1).sum() + (pf * w2).sum()).backward()
    P = dict(m.named_parameters())
    checked, worst = 0, (0.0, '')
    for k in g:
        if k.startswith('grad_'):
            name = k[len('grad_'):].replace('__', '.')
            got = P['residue_embed.mlp.0.weight'].grad[::4, ::7] if name.endswith('_sub') else P[name].grad
            rel = max_abs(got.cpu(), g[k]) / max(1e-6, g[k].abs().max().item())
            worst = max(worst, (rel, name))
            assert rel <= 5e-4, (name, rel)
            checked += 1
    assert checked == len(cases.ENCODE_FULL_PARAMS) + 1
    print('encode_L256: worst relative gradient error', worst)


@pytest.mark.gpu
def test_pair_embedding_repeats_beside_a_second_process():
    """The pair embedding while a second process runs this library's denoiser on the same GPU (its waves share SIMDs with the embedding's): every one
    of a few thousand launches of a 64-workgroup embedding must equal the first bit for bit.  Round 6: 1-3 % of these launches returned a 16-pair tile computed
    from a wrong dihedral angle in lanes 48-63 of one wave -- never in a process that had the GPU to itself.  Cause: packed-FP32 VALU instructions (v_pk_*_f32, formed
    by hipcc) are not safe on gfx950 beside another wave's 16x16x32 f16 / bf16 MFMAs; the library is built without them (csrc/Makefile NOPK, DESIGN.md 3.6,
    tools/r06/pe_share.py, tools/micro/dih_asm/)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'tools', 'r06', 'pe_share.py')
    env = dict(os.environ, ABOPT_CORE32='1')
    partner = subprocess.Popen([sys.executable, tool, 'partner', '--kind', 'eps', '--seconds', '14'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out = subprocess.run([sys.executable, tool, 'victim', '--n', '1', '--layout', '128', '--reps', '0', '--seconds', '6', '--burst', '10'],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300).stdout
    finally:
        pout = partner.communicate(timeout=300)[0]
    import re
    m_ = re.search(r'victim N=1 L=128: (\d+) of (\d+) launches differ', out)
    assert m_, out[-2000:]
    assert int(m_.group(2)) >= 2000, out[-500:]
    p_ = re.search(r'partner eps: (\d+) iterations', pout)
    assert p_ and int(p_.group(1)) >= 50, pout[-2000:]                 # the partner really ran beside the launches
    assert int(m_.group(1)) == 0, out[-2000:]


@pytest.mark.parametrize('resolution', ['full', 'backbone+CB'])
def test_pair_embedding_repeats_bit_for_bit(resolution):
    """Race detector for the pair-embedding kernels (round 4 saw run-to-run different 16-pair tiles in an experimental bf16-term build of
    pair_embed_kernel; the shipped fp32-MFMA kernels spill registers like that build did): 200 inference launches and 40 forward + backward
    passes of the training path on one input must agree bit for bit, outputs and every parameter gradient."""
    from ab_opt_amd import get_model
    from conftest import AttrDict
    cfg = cases.cfg_abdock(10)
    cfg['resolution'] = resolution
    m = synth.fill_module_(get_model(AttrDict(cfg)).eval(), seed=17).to(DEV)
    with torch.no_grad():
        m.pair_embed.aapair_to_distcoef.weight.copy_(dev(synth.hash_tensor(tuple(m.pair_embed.aapair_to_distcoef.weight.shape), 23, scale=2.0)))
    L = 256
    batch = {k: dev(v) for k, v in synth.make_batch(3, synth.LAYOUT_256, seed=5, lengths=[L, L - 11, L // 2 + 3]).items()}
    batch['pos_heavyatom'][:, :, 5:] = batch['pos_heavyatom'][:, :, 1:2] + dev(synth.hash_tensor((3, L, 10, 3), 41, scale=3.0))
    batch['mask_heavyatom'][:, ::2, 5:12] = True
    batch['mask_heavyatom'][:, ::6, 3] = False
    batch['mask_heavyatom'] &= batch['mask'][:, :, None]
    with torch.no_grad():
        first = m.encode(dict(batch), True, True)[1].clone()
        assert torch.isfinite(first).all()
        for rep in range(200):
            assert torch.equal(m.encode(dict(batch), True, True)[1], first), rep
    w = dev(synth.hash_tensor(tuple(first.shape), 77, scale=1.0))
    ref_g = None
    for rep in range(40):
        m.zero_grad(set_to_none=True)
        pf = m.encode(dict(batch), True, True)[1]
        assert torch.equal(pf.detach(), first), rep                       # the training path's forward (activation dump) equals the inference kernel
        (pf * w).sum().backward()
        g = {n: p.grad.clone() for n, p in m.pair_embed.named_parameters() if p.grad is not None}
        if ref_g is None:
            ref_g = g
            assert len(g) >= 13 and all(torch.isfinite(x).all() for x in g.values())
        else:
            for n in ref_g:
                assert torch.equal(g[n], ref_g[n]), (rep, n)


@pytest.mark.parametrize('flavour,resolution,L,flags', [
    ('abdock', 'full', 128, (True, True)), ('abdock', 'full', 128, (False, True)), ('abdock', 'full', 128, (False, False)),
    ('abdock', 'backbone+CB', 128, (True, True)), ('abdesign', 'full', 128, (True, False)), ('abdesign', 'full', 256, (True, True))])
def test_encode_hip_vs_autograd_statement(flavour, resolution, L, flags):
    """HIP encode (inference kernels and the training path's autograd functions) against the plain torch statement of the same modules
    on the device (tests/plain_statement.py, float32 and float64): both resolutions, hotspot embedding, every mask mode, ragged and
    full lengths."""
    from ab_opt_amd import get_model
    cfg = cases.cfg_abdock(10)
    cfg['resolution'] = resolution
    if flavour == 'abdesign':
        for k in ('num_bins', 'dist_min', 'dist_max'):
            cfg.pop(k)
        cfg['diffusion'].pop('obj')
    from conftest import AttrDict
    m = synth.fill_module_(get_model(AttrDict(cfg)).eval(), seed=17).to(DEV)
    # non-trivial distance coefficients (the reference initialises them to zero; trained values are not)
    with torch.no_grad():
        m.pair_embed.aapair_to_distcoef.weight.copy_(dev(synth.hash_tensor(tuple(m.pair_embed.aapair_to_distcoef.weight.shape), 23, scale=2.0)))
    layout = synth.LAYOUT_256 if L == 256 else synth.LAYOUT_128
    lengths = [L, L - 11, L // 2 + 3]
    batch = {k: dev(v) for k, v in synth.make_batch(3, layout, seed=5, lengths=lengths).items()}
    # side chains on every other residue (the synthetic complexes carry backbone + CB only), some atoms missing
    batch['pos_heavyatom'][:, :, 5:] = batch['pos_heavyatom'][:, :, 1:2] + dev(synth.hash_tensor((3, L, 10, 3), 41, scale=3.0))
    batch['mask_heavyatom'][:, ::2, 5:12] = True
    batch['mask_heavyatom'][:, ::6, 3] = False
    batch['mask_heavyatom'] &= batch['mask'][:, :, None]
    if flavour == 'abdesign':
        batch['hotspot'] = (dev(synth.hash_tensor((3, L), 31, scale=1.0)) > 0.7).long() * batch['mask'].long()
    import plain_statement
    with torch.no_grad():
        ref = [t.detach() for t in plain_statement.encode(m, dict(batch), *flags)]
    # the same statement in float64: the yardstick.  The hash-filled tables give O(1e3) features with heavy cancellation
    # (and acos near its clamp on the i == j diagonal), so fp32 results differ from the exact value by ~1e-4 relative
    # whatever the summation order; the HIP path must be as close to the fp64 value as the fp32 torch statement is.
    m64 = get_model(AttrDict(cfg)).eval()
    m64.load_state_dict(m.state_dict())
    m64 = m64.to(DEV).double()
    b64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in batch.items()}
    with torch.no_grad():
        ref64 = [t.detach() for t in plain_statement.encode(m64, dict(b64), *flags)]
        out = m.encode(dict(batch), *flags)
    with torch.enable_grad():
        out_train = [t.detach() for t in m.encode(dict(batch), *flags)]           # the training path (custom autograd functions on the same kernels)
    for a, b in zip(out, out_train):
        assert max_abs(a, b) <= 2e-5 * max(1.0, b.abs().max().item())
    for name, a, b, c in zip(('res_feat', 'pair_feat', 'R', 'p'), out, ref, ref64):
        assert a.shape == b.shape, name
        e_hip, e_t32 = (a.double() - c).abs(), (b.double() - c).abs()
        assert e_hip.max().item() <= 1.5 * e_t32.max().item() + 1e-6, name
        assert e_hip.mean().item() <= 1.5 * e_t32.mean().item() + 1e-7, name
        if name == 'pair_feat':
            # i == j: the psi-like dihedral (N_i, CA_i, C_i, N_i) is +-acos(0.999999) with the SIGN of an exactly-zero triple
            # product (pair.py:86-88 / geometry.py:336-362), i.e. of rounding noise -- implementation-defined in the reference
            # itself, so the diagonal is held to a looser bound
            eye = torch.eye(L, dtype=torch.bool, device=DEV)[None, :, :, None]
            scale = max(1.0, c.abs().max().item())
            assert ((a - b).abs() * ~eye).max().item() <= 2e-4 * scale, name
            assert ((a - b).abs() * eye).max().item() <= 2e-3 * scale, name
        else:
            assert max_abs(a, b) <= 2e-4 * max(1.0, c.abs().max().item()), name
    # masked pairs / residues are exactly zero
    assert out[1][~(batch['mask'][:, :, None] & batch['mask'][:, None, :])].abs().max().item() == 0


def test_reconstruct_backbone_partially_vs_reference():
    from ab_opt_amd import geometry
    g = load_golden('reconstruct_small')
    b = {k: dev(v) for k, v in cases.reconstruct_batch().items()}
    pos, m = geometry.reconstruct_backbone_partially(pos_ctx=b['pos_heavyatom'], R_new=dev(g['R_new']), t_new=dev(g['t_new']), aa=dev(g['aa_new']),
                                                     chain_nb=b['chain_nb'], res_nb=b['res_nb'], mask_atoms=b['mask_heavyatom'],
                                                     mask_recons=b['generate_flag'])
    assert max_abs(pos.cpu(), g['pos_new']) < 2e-5                      # Angstrom
    assert torch.equal(m.cpu(), g['mask_new'].bool())
    # rigid-motion equivariance at full size: moving every frame moves the rebuilt atoms with it
    from oracle import geometry as G
    N, L = 4, 256
    bb = {k: dev(v) for k, v in synth.make_batch(N, synth.LAYOUT_256, seed=8).items()}
    v = dev(synth.hash_tensor((N, L, 3), 5, scale=2.0))
    t = bb['pos_heavyatom'][:, :, 1].contiguous()
    Q = dev(G.so3_exp(torch.tensor([[0.4, 0.2, -0.9]])))[0]
    R = geometry.so3vec_to_rotation(v)
    p0, _ = geometry.reconstruct_backbone_partially(bb['pos_heavyatom'], R, t, bb['aa'], bb['chain_nb'], bb['res_nb'], bb['mask_heavyatom'], bb['generate_flag'])
    p1, _ = geometry.reconstruct_backbone_partially(bb['pos_heavyatom'], Q @ R, t @ Q.T, bb['aa'], bb['chain_nb'], bb['res_nb'], bb['mask_heavyatom'], bb['generate_flag'])
    gen = bb['generate_flag']
    assert max_abs(p1[gen][:, :4], p0[gen][:, :4] @ Q.T) < 2e-4
    assert torch.equal(p0[~gen], bb['pos_heavyatom'][~gen])


def test_replicated_complex_shares_context_bit_identically():
    """sampler.sample_replicated (context encoded once, pair_feat shared with batch stride 0 in the kernels) must give exactly
    the trajectory of model.sample on the host-replicated batch the reference's runner builds (design_for_pdb.py:141-147)."""
    from ab_opt_amd import sampler
    m = build_model(10, 3, device=DEV)
    one = {k: dev(v) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=12, lengths=[101]).items()}
    n = 5
    repl = {k: v.expand(n, *v.shape[1:]).contiguous() for k, v in one.items()}
    opt = {'sample_structure': True, 'sample_sequence': True, 'contig': '', 'seed': 321}
    ref = m.sample(repl, dict(opt))
    got = sampler.sample_replicated(m, one, n, dict(opt))
    assert sorted(ref) == sorted(got)
    for t in ref:
        for a, b in zip(ref[t], got[t]):
            assert torch.equal(a.cpu(), b.cpu()), t
    # and the samples differ from each other (distinct Philox counters per sample)
    assert not torch.equal(got[0][1][0], got[0][1][1])


def test_sampling_is_deterministic_for_a_seed():
    """Race detector: every kernel on the path is free of atomics and cross-workgroup ordering, so a seeded run must repeat
    bit for bit (a missing wave-level LDS fence in the IPA core once broke this silently)."""
    m = build_model(10, 3, device=DEV)
    batch = {k: dev(v) for k, v in synth.make_batch(6, synth.LAYOUT_256, seed=5, lengths=[256, 256, 243, 256, 256, 200]).items()}
    opt = {'sample_structure': True, 'sample_sequence': True, 'contig': '', 'seed': 77}
    ref = m.sample(dict(batch), dict(opt))
    for _ in range(4):
        got = m.sample(dict(batch), dict(opt))
        for t in ref:
            for a, b in zip(ref[t], got[t]):
                assert torch.equal(a.cpu(), b.cpu()), t


def test_structure_only_sampling_and_contig_mask():
    """BASELINE config 3 mode (AbDock pose diffusion: obj = pred_x0, sample_sequence=False): the sequence never changes and
    the start state matches the reference's recorded initial draw; with a contig the generated set is restricted to it
    (diffab.py:125-127,184-205)."""
    g = load_golden('trajectory_abdock_T10_structonly')
    _, m, batch = _traj_setup()
    b = {k: dev(v) for k, v in batch.items()}
    nz = noise_dict(g, 10)
    nzd = {t: {k: (dev(v) if v is not None else None) for k, v in d.items()} for t, d in nz.items()}
    traj = m.sample(dict(b), sample_opt=dict(sample_structure=True, sample_sequence=False, contig='', noise=nzd))
    from oracle import geometry as G
    # context frames within 0.03 rad of pi sit on the log map's ill-conditioned branch: 5e-4 on R there
    assert max_abs(G.so3_exp(traj[10][0].cpu()), G.so3_exp(g['traj10_v'])) < 5e-4 and max_abs(traj[10][1].cpu(), g['traj10_p']) < 1e-4
    for t in range(10, -1, -1):
        assert torch.equal(traj[t][2].cpu(), batch['aa']), t                # sample_sequence=False: s_next = s_t everywhere (dpm_full.py:296-297)
        assert torch.isfinite(traj[t][1]).all()
    # contig: only residues 31..36 (1-based, inclusive) of the generated segment may change
    b2 = {k: dev(v) for k, v in batch.items()}
    gen0 = b2['generate_flag'].clone()
    traj2 = m.sample(b2, sample_opt=dict(sample_structure=True, sample_sequence=True, contig='31-36', seed=5))
    allowed = gen0.clone()
    allowed[:] = False
    allowed[:, 30:36] = True
    allowed &= gen0
    assert torch.equal(b2['generate_flag'], allowed)                       # the reference narrows batch['generate_flag'] in place
    moved = (traj2[0][1] - traj2[10][1].to(traj2[0][1].device)).abs().sum(-1) > 0
    assert not moved[~allowed].any() and moved[allowed].any()
    # (padded positions, aa = 21, have an all-zero one-hot, so the reference re-draws them uniformly at every step: transition.py:202-245)
    keep = (~allowed & b2['mask']).cpu()
    assert torch.equal(traj2[0][2].cpu()[keep], batch['aa'][keep])


def test_design_pipeline_end_to_end():
    """The runner's sequence (design_for_pdb.py:141-336) on the device: shared-context sampling -> backbone rebuild -> commonness
    ranking.  Integration check: shapes, finiteness, masks, ranking indices."""
    from ab_opt_amd import sampler, geometry
    m = build_model(10, 3, device=DEV)
    one = {k: dev(v) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=9).items()}
    n = 6
    traj = sampler.sample_replicated(m, one, n, {'sample_structure': True, 'sample_sequence': True, 'seed': 3})
    v, p, s, prmsd, ppl = traj[0]
    rep = lambda a: a.expand(n, *a.shape[1:]).contiguous()
    pos, mask = geometry.reconstruct_backbone_partially(rep(one['pos_heavyatom']), geometry.so3vec_to_rotation(v), p, s, rep(one['chain_nb']),
                                                        rep(one['res_nb']), rep(one['mask_heavyatom']), rep(one['generate_flag']))
    gen = rep(one['generate_flag'])
    assert pos.shape == (n, 128, 15, 3) and torch.isfinite(pos).all() and mask[gen][:, :4].all() and not mask[gen][:, 4:].any()
    assert torch.equal(pos[~gen], rep(one['pos_heavyatom'])[~gen])
    assert max_abs(pos[gen][:, 1], p[gen]) < 1e-4                               # CA sits at the frame origin
    cand = pos[gen][:, :3].reshape(n, -1, 3)
    top = sampler.rank_commoness(cand, k=3)
    assert top.shape == (3,) and len(set(top.tolist())) == 3 and prmsd.shape == (n,) and ppl.shape == (n,)


def test_c_abi_error_paths():
    """Error behaviour of the boundary: bad arguments come back as error codes with a message (raised as RuntimeError by the
    binding), never as a crash or a silent fallback."""
    import ctypes as C
    from ab_opt_amd import hip
    L_ = hip.lib()
    blk = _block_on_device()
    _, ws = blk.packed()
    R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(1, 16, [16])]
    out = torch.empty_like(x)
    small = torch.empty(1024, dtype=torch.uint8, device=DEV)
    args = lambda F, Cd, buf: (C.byref(ws), hip.ptr(R), hip.ptr(t), hip.ptr(x), hip.ptr(z), hip.ptr(mask), hip.ptr(out), 1, 16, F, Cd, None,
                               hip.ptr(buf), buf.numel(), hip.stream())
    assert L_.abopt_ga_block_forward(*args(128, 64, small)) == 4                      # ABOPT_EWORKSPACE
    assert b'workspace too small' in L_.abopt_last_error()
    big = torch.empty(L_.abopt_ga_workspace_bytes(1, 16, 128, 64), dtype=torch.uint8, device=DEV)
    assert L_.abopt_ga_block_forward(*args(256, 64, big)) == 3                        # ABOPT_EUNSUPPORTED: res_feat_dim != 128
    assert L_.abopt_ga_block_forward(*args(128, 64, big)) == 0
    with pytest.raises(RuntimeError, match='no CPU path'):
        hip.so3_exp(torch.zeros(4, 3))
    with pytest.raises(TypeError):
        hip.so3_exp(torch.zeros(4, 3, dtype=torch.float64, device=DEV))
    with pytest.raises(RuntimeError, match='atoms'):                                   # resolution outside [3, 15]
        b = {k: dev(v) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=1, lengths=[20]).items()}
        inp, keep = hip.encode_inputs(b['aa'], b['res_nb'], b['chain_nb'], b['pos_heavyatom'], b['mask_heavyatom'], 2, fragment_type=b['fragment_type'])
        m = build_model(10, 3, device=DEV)
        hip.pair_embed_forward(inp, m.pair_embed._hip_weights())
    # the piecewise tail entry points (training path): NULL tensors and a negative row count are rejected, zero rows are a no-op
    f = hip.ptr
    t_, _ = blk.packed()
    o128 = torch.empty(16, 128, device=DEV)
    tail = lambda feat, rows: L_.abopt_block_tail_forward(feat, f(t_['w_out_frag']), f(t_['w_mlp_frag']), f(x), f(t_['b_out']), f(mask), f(t_['ln1_gamma']),
                                                          f(t_['ln1_beta']), f(t_['b_mlp0']), f(t_['b_mlp1']), f(t_['b_mlp2']), f(t_['ln2_gamma']), f(t_['ln2_beta']),
                                                          f(o128), None, rows, hip.stream())
    assert tail(None, 16) == 1 and b'NULL' in L_.abopt_last_error()                    # ABOPT_EINVAL
    assert tail(None, -1) == 1
    assert tail(None, 0) == 0
    assert L_.abopt_pack_tail_weights(f(t_['w_out']), None, None, None, None, None, None, hip.stream()) == 1
    assert L_.abopt_block_tail_backward(None, None, None, None, None, None, None, None, None, None, 16, hip.stream()) == 1


def test_sample_init_vs_reference():
    from ab_opt_amd import hip
    g = load_golden('trajectory_abdock_T10')
    _, m, batch = _traj_setup()
    v0 = hip.so3_log(dev(g['R0']), False)
    v_i, p_i, s_i = hip.sample_init(v0, dev(g['p0']), dev(batch['aa']), dev(batch['generate_flag']),
                                    dict(q4=dev(g['init_q4']), p=dev(g['init_p']), s=dev(g['init_s'])), 0, 0, 10.0, [0.0, 0.0, 0.0], True, True)
    from oracle import geometry as G
    assert max_abs(G.so3_exp(v_i.cpu()), G.so3_exp(g['traj10_v'])) < 1e-4
    assert max_abs(p_i.cpu(), g['traj10_p']) < 1e-5
    assert torch.equal(s_i.cpu(), g['traj10_s'])


def test_denoising_steps_teacher_forced_vs_reference():
    """Every one of the 10 recorded reference steps (BASELINE config 1), state reset to the reference's each step."""
    from ab_opt_amd import hip
    from oracle import dpm, geometry as G
    g = load_golden('trajectory_abdock_T10')
    m_cpu, m, batch = _traj_setup()
    dpm_ = m.diffusion
    res_feat, gen, mres = dev(g['res_feat']), dev(batch['generate_flag']), dev(batch['mask'])
    with torch.no_grad():
        _, pair_feat, _, _ = m.encode({k: dev(v) for k, v in batch.items()}, True, True)
    nz = noise_dict(g, 10)
    den = dpm.Denoiser(m_cpu.state_dict(), num_steps=10, variant='abdock', obj='pred_x0', mode='mm',
                       tables=(None, dict(stddevs=m_cpu.diffusion.trans_rot.angular_distrib_inv.stddevs,
                                          approx_flag=m_cpu.diffusion.trans_rot.angular_distrib_inv.approx_flag,
                                          X=m_cpu.diffusion.trans_rot.angular_distrib_inv.X, Y=None)))
    worst_p = 0.0
    for t in range(10, 0, -1):
        state = (dev(g[f'traj{t}_v']), dev(g[f'traj{t}_p']), dev(g[f'traj{t}_s']))
        noise = {t: {k: dev(v) for k, v in nz[t].items()}}
        tv, tp, ts, tpr, tpp = _one_step(dpm_, state, t, res_feat, pair_feat, gen, mres, noise[t])
        # pre-log rotation of the transition, from the oracle on the same inputs (well-conditioned reference for the tolerance)
        e = dpm.so3_noise(den.tab_inv, torch.full((2, 128), t), nz[t])
        if t <= 1:
            e = torch.zeros_like(e)
        o = den._eps(g[f'traj{t}_v'], den.norm(g[f'traj{t}_p']), g[f'traj{t}_s'], g['res_feat'], pair_feat.cpu(),
                     den.sch['betas'][t].expand([2]), batch['generate_flag'], batch['mask'], False)
        R_pre = G.so3_exp(e) @ G.so3_exp(o[0])
        # the value went through two log maps: the network's (R_next -> v_next) and the transition's (R_pre -> v)
        n, worst = rot_close(tv.cpu(), g[f'traj{t - 1}_v'], R_pre, R_upstream=o[1])
        assert n >= 230 and worst < 1.0, (t, n, worst)
        dp = max_abs(tp.cpu(), g[f'traj{t - 1}_p'])
        worst_p = max(worst_p, dp)
        assert dp < 1e-4, (t, dp)
        assert torch.equal(ts.cpu(), g[f'traj{t - 1}_s'])
        assert max_abs(tpr.cpu(), g[f'traj{t - 1}_prmsd']) < 1e-4
        assert max_abs(tpp.cpu(), g[f'traj{t - 1}_ppl']) < 1e-5
    print('worst per-step position error (Angstrom):', worst_p)


def _one_step(dpm_, state, t, res_feat, pair_feat, gen, mres, noise_t):
    """eps_net + transition for step t from `state` (v, p_angstrom, s); returns the t-1 state and scalars."""
    tv, tp, ts, tpr, tpp = dpm_._run(tuple(s.clone() for s in state), t, res_feat, pair_feat, gen, mres, True, True, True,
                                     {t: noise_t}, 0, 0, False, stop_after=1)
    return tv[t - 1], tp[t - 1], ts[t - 1], tpr[t - 1], tpp[t - 1]


def test_pair_bias_cache_is_bit_identical(monkeypatch):
    """The per-call pair-bias cache (abopt_pair_bias_cache) must not change a single bit of a denoising step."""
    monkeypatch.setenv('ABOPT_CORE_NO_SPLIT', '1')             # same kernel form on both sides (the cached path of a small batch may split the keys)
    g = load_golden('trajectory_abdock_T10')
    _, m, batch = _traj_setup()
    d = m.diffusion
    b = {k: dev(v) for k, v in batch.items()}
    with torch.no_grad():
        _, pf, _, _ = m.encode(b, True, True)
    nz = noise_dict(g, 10)
    t = 6
    state = (dev(g[f'traj{t}_v']), dev(g[f'traj{t}_p']), dev(g[f'traj{t}_s']))
    noise = {t: {k: dev(v) for k, v in nz[t].items()}}
    outs = []
    for use in (True, False):
        tv, tp, ts, tpr, tpp = d._run(tuple(s.clone() for s in state), t, dev(g['res_feat']), pf, b['generate_flag'], b['mask'], True, True, True,
                                      noise, 0, 0, False, stop_after=1, use_bias_cache=use)
        outs.append((tv[t - 1].clone(), tp[t - 1].clone(), tpr[t - 1].clone()))
    assert all(torch.equal(a, c) for a, c in zip(*outs))


def test_model_sample_free_run_with_injected_noise():
    """model.sample() end-to-end with replayed draws: first steps track the reference; later steps are only required
    to stay finite and to carry the injected sequence (the dynamics amplify fp32 reordering)."""
    from oracle import geometry as G
    g = load_golden('trajectory_abdock_T10')
    _, m, batch = _traj_setup()
    nz = noise_dict(g, 10)
    nzd = {k: {kk: dev(vv) for kk, vv in v.items() if vv is not None} for k, v in nz.items()}
    traj = m.sample({k: dev(v) for k, v in batch.items()}, sample_opt=dict(sample_structure=True, sample_sequence=True, contig='', noise=nzd))
    assert sorted(traj) == list(range(11)) and len(traj[10]) == 5 and len(traj[0]) == 5
    assert traj[0][0].is_cuda and not traj[5][0].is_cuda            # reference layout: t>0 on host, t=0 on device
    assert max_abs(traj[10][1], g['traj10_p']) < 1e-5
    assert max_abs(traj[9][1], g['traj9_p']) < 1e-4
    for t in range(10):
        assert torch.equal(traj[t][2].cpu(), g[f'traj{t}_s'])
        assert torch.isfinite(traj[t][1]).all() and torch.isfinite(traj[t][0]).all()
    ctx = ~batch['generate_flag']
    assert max_abs(traj[0][1].cpu()[ctx], g['traj0_p'][ctx]) < 1e-4       # context residues never move


def test_optimize_vs_reference():
    """FullDPM.optimize: forward noising to step 4 with the reference's recorded draws, then 4 teacher-forced steps."""
    from oracle import geometry as G
    g = load_golden('optimize_abdock_T10_k4')
    gt = load_golden('trajectory_abdock_T10')
    _, m, batch = _traj_setup()
    d = m.diffusion
    b = {k: dev(v) for k, v in batch.items()}
    with torch.no_grad():
        rf, pf, R0, p0 = m.encode(b, True, True)
    from ab_opt_amd import hip
    v0 = hip.so3_log(dev(gt['R0']), False)
    init = dict(axis=dev(g['rot_axis']), bin=dev(g['rot_bin']), ubin=dev(g['rot_ubin']), gauss=dev(g['rot_gauss']), pos=dev(g['pos']), s_noisy=dev(g['s_noisy']))
    nz = {t: {k: dev(g[f't{t}_{k}']) for k in ('axis', 'bin', 'ubin', 'gauss', 'z', 's_next')} for t in range(4, 0, -1)}
    nz['init'] = init
    traj = d.optimize(v0, dev(gt['p0']), b['aa'], 4, dev(gt['res_feat']), pf, b['generate_flag'], b['mask'], noise=nz)
    assert sorted(traj) == [0, 1, 2, 3, 4] and len(traj[4]) == 5 and isinstance(traj[2], tuple)
    # noised start state
    assert max_abs(traj[4][1].cpu(), g['traj4_p']) < 1e-4 and torch.equal(traj[4][2].cpu(), g['traj4_s'])
    assert max_abs(G.so3_exp(traj[4][0].cpu()), G.so3_exp(g['traj4_v'])) < 5e-4
    # each step from the reference's state (teacher-forced); optimize feeds the net output as noise and uses unmasked perplexity
    for t in range(4, 0, -1):
        state = (dev(g[f'traj{t}_v']), dev(g[f'traj{t}_p']), dev(g[f'traj{t}_s']))
        tv, tp, ts, tpr, tpp = d._run(state, t, dev(gt['res_feat']), pf, b['generate_flag'], b['mask'], True, True, False,
                                      {t: nz[t]}, 0, 0, False, stop_after=1, optimize_mode=True)
        assert max_abs(tp[t - 1].cpu(), g[f'traj{t - 1}_p']) < 1e-4, t
        assert torch.equal(ts[t - 1].cpu(), g[f'traj{t - 1}_s'])


def test_sample_sharded_single_rank_and_rng_reproducibility():
    from ab_opt_amd import sampler
    _, m, batch = _traj_setup()
    b = {k: dev(v) for k, v in synth.make_batch(4, synth.LAYOUT_128, seed=11, replicate=True).items()}
    traj, (a, e), top, cand = sampler.sample_sharded(m, b, dict(sample_structure=True, sample_sequence=True, contig=''), k=2, seed=42)
    assert (a, e) == (0, 4) and cand.shape == (4, 12, 3) and top.shape == (2,)
    traj2, _, top2, cand2 = sampler.sample_sharded(m, {k: v.clone() for k, v in b.items()}, dict(sample_structure=True, sample_sequence=True, contig=''), k=2, seed=42)
    assert torch.equal(cand, cand2) and torch.equal(top, top2)          # same Philox seed => identical samples
    assert not torch.equal(cand[0], cand[1])                             # replicated inputs still give distinct samples


def test_training_loss_and_grads_vs_reference():
    """FullDPM.forward with the reference's recorded noise and fixed t: losses (1e-5 rel) and gradients (2e-4 of max)."""
    g = load_golden('training_abdock')
    m = build_model(100, 2, device=DEV)
    d = m.diffusion
    d.zero_grad()
    N, L = 2, 48
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
    s = s.clamp(max=19)
    res_feat = dev(res_feat).clone().requires_grad_(True)
    pair_feat = dev(pair_feat).clone().requires_grad_(True)
    noise = dict(axis=dev(g['rot_axis']), bin=dev(g['rot_bin']), ubin=dev(g['rot_ubin']), gauss=dev(g['rot_gauss']), pos=dev(g['pos']), s_noisy=dev(g['s_noisy']))
    loss = d(dev(v), dev(p) * 10, dev(s), res_feat, pair_feat, dev(gen), dev(mres), True, True, t=torch.tensor([37, 80], device=DEV), noise=noise)
    assert set(loss) == {'prmsd', 'dist', 'rot', 'pos', 'seq'}
    for k, val in loss.items():
        ref = g['loss_' + k].item()
        assert abs(val.item() - ref) <= 2e-5 * max(1.0, abs(ref)), (k, val.item(), ref)
    sum(loss.values()).backward()
    params = dict(d.named_parameters())
    for k in g:
        if k.startswith('grad_eps_net'):
            got = params[k[len('grad_'):]].grad.cpu()
            assert max_abs(got, g[k]) <= 3e-4 * g[k].abs().max().item() + 1e-7, k
    assert max_abs(res_feat.grad.cpu(), g['grad_res_feat']) <= 3e-4 * g['grad_res_feat'].abs().max().item()
    assert max_abs(pair_feat.grad.cpu()[:, ::5, ::3], g['grad_pair_feat_sub']) <= 3e-4 * g['grad_pair_feat_sub'].abs().max().item()
    d.zero_grad()


@pytest.mark.parametrize('N,L,lengths', [(2, 24, [24, 19]), (3, 70, [70, 33, 1]), (2, 256, [256, 250])])
def test_ipa_core_autograd_function_vs_torch_statement(N, L, lengths):
    """Native training block (HIP forward with alpha kept, HIP z-streaming backward + (N,L,L,12) GEMMs) against the plain
    torch statement of the same GABlock (tests/plain_statement.py; the reference's recorded gradients pin both): output and every gradient."""
    from ab_opt_amd import training
    import plain_statement
    blk = _block_on_device(seed=9)
    with torch.no_grad():
        blk.spatial_coef.copy_(dev(synth.hash_tensor((1, 1, 1, 12), 77, scale=1.0)))
    R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(N, L, lengths, salt=700 + L)]
    wout = dev(synth.hash_tensor((N, L, 128), 78, scale=1.0))
    res = {}
    for native in (False, True):
        blk.zero_grad()
        xx, zz = x.clone().requires_grad_(True), z.clone().requires_grad_(True)
        out = training.ga_block(blk, R, t, xx, zz, mask) if native else plain_statement.ga_block(blk, R, t, xx, zz, mask)
        (out * wout).sum().backward()
        res[native] = dict(out=out.detach(), dx=xx.grad, dz=zz.grad, **{'d_' + n: p.grad.clone() for n, p in blk.named_parameters()})
    for k, ref in res[False].items():
        tol = 2e-5 if k == 'out' else 2e-4
        assert max_abs(res[True][k], ref) <= tol * max(1.0, ref.abs().max().item()), k
    blk.zero_grad()


def test_model_forward_trains_end_to_end():
    """model(batch) -> loss dict -> backward -> finite gradients on every trainable tensor that the loss touches."""
    m = build_model(10, 3, device=DEV).train()
    batch = {k: dev(v) for k, v in synth.make_batch(2, synth.LAYOUT_128, seed=5, lengths=[64, 57]).items()}
    loss = m(batch)
    total = sum(loss.values())
    assert torch.isfinite(total)
    total.backward()
    grads = [p.grad for n, p in m.named_parameters() if p.grad is not None]
    assert len(grads) > 150 and all(torch.isfinite(gr).all() for gr in grads)
    m.zero_grad(); m.eval()


def test_abdesign_steps_teacher_forced_vs_reference():
    g = load_golden('trajectory_abdesign_T10')
    d = standalone_abdesign_dpm(10, 4).to(DEV)
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)], num_steps=10, t=3)
    nz = noise_dict(g, 10)
    for t in range(10, 0, -1):
        state = (dev(g[f'traj{t}_v']), dev(g[f'traj{t}_p']), dev(g[f'traj{t}_s']))
        tv, tp, ts, tpr, tpp = d._run(state, t, dev(res_feat), dev(pair_feat), dev(gen), dev(mres), True, True, True,
                                      {t: {k: dev(vv) for k, vv in nz[t].items()}}, 0, 0, False, stop_after=1)
        assert max_abs(tp[t - 1].cpu(), g[f'traj{t - 1}_p']) < 1e-4, t
        assert torch.equal(ts[t - 1].cpu(), g[f'traj{t - 1}_s'])


# ------------------------------------------------------------------------------------------ device RNG path (statistics)
def test_device_rng_statistics():
    """No bitwise parity is possible for RNG; check the draws' distributions."""
    from ab_opt_amd import hip
    m = build_model(100, 2, device=DEV)
    d = m.diffusion
    N, L = 64, 256
    t = 60
    sp = d._step_params(t, True, True, True)
    inv = d.trans_rot.angular_distrib_inv
    z3 = torch.zeros(N, L, 3, device=DEV)
    c_net = torch.softmax(dev(synth.hash_tensor((N, L, 20), 3, scale=4.0)), -1)
    s_t = torch.randint(0, 20, (N, L), device=DEV)
    gen = torch.ones(N, L, dtype=torch.bool, device=DEV)
    out = dict(v=torch.empty(N, L, 3, device=DEV), p=torch.empty(N, L, 3, device=DEV), s=torch.empty(N, L, dtype=torch.int64, device=DEV),
               prmsd=torch.empty(N, device=DEV), ppl=torch.empty(N, device=DEV))
    post = hip.denoise_step(sp, None, 1234, 0, z3, z3, s_t, z3, z3, c_net, torch.zeros(N, 40, device=DEV), gen, inv.X[t], inv.cdf()[t], 40, out, want_post=True)
    # rotation noise: v_next = log(exp(e) exp(0)) = e  => |v| is the IGSO(3) angle
    theta = out['v'].norm(dim=-1).flatten().cpu().double()
    Y = inv.Y[t, :-1].cpu().double()
    X = inv.X[t].cpu().double()
    edges = X[::256]
    hist = torch.histc(theta, bins=32, min=0.0, max=float(X[-1]))
    probs = torch.stack([Y[i * 256:(i + 1) * 256].sum() for i in range(32)])
    probs = probs / probs.sum()
    expected = probs * theta.numel()
    keep = expected > 20
    chi2 = (((hist - expected) ** 2 / expected)[keep]).sum().item()
    assert chi2 < 3 * int(keep.sum()), chi2
    axis = (out['v'] / out['v'].norm(dim=-1, keepdim=True)).reshape(-1, 3).cpu()
    assert axis.mean(0).abs().max() < 0.02
    # position noise: p_next = sigma*z*scale with eps = 0 inputs => z moments
    c0 = 1.0 / math.sqrt(sp.alpha_clamped + 1e-8)
    zz = (out['p'] / 10.0 / sp.sigma).flatten().cpu()
    assert abs(zz.mean().item()) < 0.02 and abs(zz.std().item() - 1) < 0.02
    # categorical: empirical frequencies vs posterior
    freq = torch.zeros(20)
    freq.scatter_add_(0, out['s'].flatten().cpu(), torch.ones(N * L))
    exp_freq = post.reshape(-1, 20).sum(0).cpu()
    assert ((freq - exp_freq).abs() / exp_freq.clamp_min(50).sqrt()).max() < 5
    # different offsets decorrelate, same (seed, offset) reproduces
    out2 = {k: torch.empty_like(v) for k, v in out.items()}
    hip.denoise_step(sp, None, 1234, 0, z3, z3, s_t, z3, z3, c_net, torch.zeros(N, 40, device=DEV), gen, inv.X[t], inv.cdf()[t], 40, out2)
    assert torch.equal(out2['v'], out['v']) and torch.equal(out2['s'], out['s'])
    hip.denoise_step(sp, None, 1234, N * L, z3, z3, s_t, z3, z3, c_net, torch.zeros(N, 40, device=DEV), gen, inv.X[t], inv.cdf()[t], 40, out2)
    assert not torch.equal(out2['v'], out['v'])


def test_commonness_score_vs_reference():
    from ab_opt_amd import hip
    g = load_golden('rank_commoness')
    structs = synth.hash_tensor((16, 36, 3), 55, scale=8.0)
    structs[3] = structs[5] + 0.01
    score = hip.commonness_score(dev(structs))
    assert torch.equal(torch.topk(score, 5, largest=False)[1].cpu(), g['rank'])
    assert abs(score.sum().item() / 16 - g['avg_rmsd'].item()) < 1e-4


# ------------------------------------------------------------------------------------------ categorical transition, pinned directly
def _device_step_with_posterior(d, t, state, res_feat, pair_feat, gen, mres, noise_t, optimize_mode=False, sample_sequence=True, sample_structure=True):
    """eps_net + abopt_denoise_step(want_post) for step t from `state` (v, p_angstrom, s); returns (post, out dict)."""
    from ab_opt_amd import hip
    N, L = mres.shape
    h = d._sched_host()
    p_norm = (state[1] - d.position_mean) / d.position_scale
    beta = d.trans_pos.var_sched.betas[t].expand([N]).contiguous()
    net = hip.eps_net_forward(d.eps_net.packed(), state[0], p_norm, state[2], res_feat, pair_feat, beta, gen, mres, d.abdock, d.num_bins, False)
    sp = d._step_params(t, sample_structure, sample_sequence, not optimize_mode, optimize_mode)
    out = dict(v=torch.empty(N, L, 3, device=DEV), p=torch.empty(N, L, 3, device=DEV), s=torch.empty(N, L, dtype=torch.int64, device=DEV),
               prmsd=torch.empty(N, device=DEV), ppl=torch.empty(N, device=DEV))
    inv = d.trans_rot.angular_distrib_inv
    post = hip.denoise_step(sp, noise_t, 0, 0, state[0], state[1], state[2], net['v_next'], net['eps_pos'], net['c'], net['prmsd_logits'], gen,
                            inv.X[t], None, d.num_bins, out, want_post=True)
    return post, out


def test_categorical_posterior_on_device_vs_reference():
    """abopt_denoise_step's posterior (post_out) against the distribution the reference sampled from at every recorded step
    (golden posterior_abdock_T10 = the input of AminoacidCategoricalTransition._sample, transition.py:171-181,202-245),
    generated and context residues, padded rows included; same for optimize() and for add_noise's forward categorical.
    The sequences in the trajectory fixtures are injected draws: THIS is the check that pins the sequence transition."""
    from ab_opt_amd import hip
    g, gp, go = load_golden('trajectory_abdock_T10'), load_golden('posterior_abdock_T10'), load_golden('optimize_abdock_T10_k4')
    _, m, batch = _traj_setup()
    d = m.diffusion
    b = {k: dev(v) for k, v in batch.items()}
    with torch.no_grad():
        _, pf, _, _ = m.encode(dict(b), True, True)
    rf, gen, mres = dev(g['res_feat']), b['generate_flag'], b['mask']
    nz = noise_dict(g, 10)
    worst = 0.0
    for t in range(10, 0, -1):
        state = (dev(g[f'traj{t}_v']), dev(g[f'traj{t}_p']), dev(g[f'traj{t}_s']))
        post, out = _device_step_with_posterior(d, t, state, rf, pf, gen, mres, {k: dev(v) for k, v in nz[t].items()})
        err = max_abs(post.cpu() + 1e-8, gp[f't{t}_probs'])
        worst = max(worst, err)
        assert err < 2e-6, (t, err)
        genc = batch['generate_flag']
        assert max_abs((post.cpu() + 1e-8)[genc], gp[f't{t}_probs'][genc]) < 2e-6 and genc.any()
        assert torch.equal(post.cpu()[~genc], (gp[f't{t}_probs'] - 1e-8)[~genc].round())      # context / padding: the one-hot of s_t (all-zero for padding)
    for t in range(4, 0, -1):                                                                  # optimize(): unmasked perplexity, net output as noise
        state = (dev(go[f'traj{t}_v']), dev(go[f'traj{t}_p']), dev(go[f'traj{t}_s']))
        noise_t = {k: dev(go[f't{t}_{k}']) for k in ('axis', 'bin', 'ubin', 'gauss', 'z', 's_next')}
        post, _ = _device_step_with_posterior(d, t, state, rf, pf, gen, mres, noise_t, optimize_mode=True)
        assert max_abs(post.cpu() + 1e-8, gp[f'opt_t{t}_probs']) < 2e-6, t
    # forward categorical of add_noise (transition.py:183-200) at step 4, with and without injected draws
    tt = torch.full([2], 4, dtype=torch.long, device=DEV)
    v0 = hip.so3_log(dev(g['R0']), False)
    init = dict(axis=dev(go['rot_axis']), bin=dev(go['rot_bin']), ubin=dev(go['rot_ubin']), gauss=dev(go['rot_gauss']), pos=dev(go['pos']), s_noisy=dev(go['s_noisy']))
    for noise in (init, None):
        *_, probs = hip.add_noise(tt, d.trans_pos.var_sched.alpha_bars, d.trans_rot.angular_distrib_fwd, noise, 5, 0, v0, dev(g['p0']), b['aa'], gen,
                                  10.0, [0.0, 0.0, 0.0], want_probs=True)
        assert max_abs(probs.cpu() + 1e-8, gp['opt_addnoise_probs']) < 1e-7
    print('worst posterior error vs reference:', worst)


def test_sequence_sampler_draws_from_the_posterior():
    """Device RNG path: the sampled s_next follows post_out (the distribution pinned above) -- per-class chi-square over many
    residues with a skewed posterior, generated residues only; context residues keep s_t exactly."""
    from ab_opt_amd import hip
    m = build_model(100, 2, device=DEV)
    d = m.diffusion
    N, L, t = 64, 256, 30
    sp = d._step_params(t, True, True, True)
    inv = d.trans_rot.angular_distrib_inv
    z3 = torch.zeros(N, L, 3, device=DEV)
    c_net = torch.softmax(dev(synth.hash_tensor((1, 1, 20), 9, scale=6.0)), -1).expand(N, L, 20).contiguous()     # one skewed prediction everywhere
    s_t = torch.full((N, L), 7, dtype=torch.int64, device=DEV)
    gen = torch.ones(N, L, dtype=torch.bool, device=DEV)
    gen[:, ::5] = False
    out = dict(v=torch.empty(N, L, 3, device=DEV), p=torch.empty(N, L, 3, device=DEV), s=torch.empty(N, L, dtype=torch.int64, device=DEV),
               prmsd=torch.empty(N, device=DEV), ppl=torch.empty(N, device=DEV))
    post = hip.denoise_step(sp, None, 99, 0, z3, z3, s_t, z3, z3, c_net, torch.zeros(N, 40, device=DEV), gen, inv.X[t], inv.cdf()[t], 40, out, want_post=True)
    assert torch.equal(out['s'][~gen], s_t[~gen])
    pg = post[gen][0].cpu().double()                                  # identical posterior for every generated residue
    n = int(gen.sum())
    freq = torch.zeros(20, dtype=torch.float64).scatter_add_(0, out['s'][gen].cpu(), torch.ones(n, dtype=torch.float64))
    exp = pg * n
    keep = exp > 10
    chi2 = (((freq - exp) ** 2 / exp)[keep]).sum().item()
    assert chi2 < 3 * int(keep.sum()), (chi2, int(keep.sum()))
    assert freq[~keep].sum() <= 10 * (~keep).sum() + 20


# ------------------------------------------------------------------------------------------ BASELINE config 3 mode, every step
def test_structure_only_steps_teacher_forced_vs_reference():
    """AbDock pose diffusion (sample_sequence=False, obj=pred_x0): every one of the 10 recorded reference steps, state reset to
    the reference's each step (golden trajectory_abdock_T10_structonly, all steps)."""
    from oracle import dpm, geometry as G
    g = load_golden('trajectory_abdock_T10_structonly')
    m_cpu, m, batch = _traj_setup()
    d = m.diffusion
    b = {k: dev(v) for k, v in batch.items()}
    with torch.no_grad():
        rf, pf, _, _ = m.encode(dict(b), True, False)                       # remove_sequence=False in this mode (diffab.py:128-131)
    assert max_abs(rf.cpu(), g['res_feat']) < 2e-4 * g['res_feat'].abs().max().item()
    rf = dev(g['res_feat'])
    gen, mres = b['generate_flag'], b['mask']
    nz = noise_dict(g, 10)
    inv = m_cpu.diffusion.trans_rot.angular_distrib_inv
    den = dpm.Denoiser(m_cpu.state_dict(), num_steps=10, variant='abdock', obj='pred_x0', mode='mm',
                       tables=(None, dict(stddevs=inv.stddevs, approx_flag=inv.approx_flag, X=inv.X, Y=None)))
    for t in range(10, 0, -1):
        state = (dev(g[f'traj{t}_v']), dev(g[f'traj{t}_p']), dev(g[f'traj{t}_s']))
        noise_t = {k: dev(v) for k, v in nz[t].items()}
        tv, tp, ts, tpr, tpp = d._run(tuple(s.clone() for s in state), t, rf, pf, gen, mres, True, False, True, {t: noise_t}, 0, 0, False, stop_after=1)
        e = dpm.so3_noise(den.tab_inv, torch.full((2, 128), t), nz[t])
        if t <= 1:
            e = torch.zeros_like(e)
        o = den._eps(g[f'traj{t}_v'], den.norm(g[f'traj{t}_p']), g[f'traj{t}_s'], g['res_feat'], pf.cpu(), den.sch['betas'][t].expand([2]),
                     batch['generate_flag'], batch['mask'], False)
        n, worst = rot_close(tv[t - 1].cpu(), g[f'traj{t - 1}_v'], G.so3_exp(e) @ G.so3_exp(o[0]), R_upstream=o[1])
        assert n >= 230 and worst < 1.0, (t, n, worst)
        assert max_abs(tp[t - 1].cpu(), g[f'traj{t - 1}_p']) < 1e-4, t
        assert torch.equal(ts[t - 1].cpu(), g[f'traj{t}_s']) and torch.equal(ts[t - 1].cpu(), batch['aa'])      # s_next = s_t in this mode
        assert max_abs(tpr[t - 1].cpu(), g[f'traj{t - 1}_prmsd']) < 1e-4, t
        assert max_abs(tpp[t - 1].cpu(), g[f'traj{t - 1}_ppl']) < 1e-5, t       # perplexity comes from the posterior, not from the injected draw


# ------------------------------------------------------------------------------------------ the launch geometry bench.py times
def _rand_eps_inputs(N, L, lengths, seed, gen_ranges):
    gtor = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *shape, s=1.0: torch.randn(*shape, device=DEV, generator=gtor) * s
    mres = dev(cases.mask_from_lengths(lengths, L))
    gen = dev(cases.gen_from_ranges(N, L, gen_ranges)) & mres
    s = torch.randint(0, 20, (N, L), device=DEV, generator=gtor)
    s = torch.where(mres, s, torch.full_like(s, 21))
    return r(N, L, 3, s=1.5), r(N, L, 3, s=2.0), s, r(N, L, 128), r(N, L, L, 64), gen, mres


@pytest.mark.parametrize('N', [8, 32])
def test_ga_block_bench_geometry_vs_oracle(N):
    """GABlock at the batch sizes that switch on the XCD-aware block mapping (N % 8 == 0), L = 256 with ragged lengths,
    against the oracle.  N = 32, L = 256 is the launch bench.py times; the oracle is run on a subset of the (independent)
    samples at N = 32."""
    from oracle import ipa, geometry as G
    L = 256
    blk = _block_on_device(seed=5)
    sd = {k: v.cpu() for k, v in blk.state_dict().items()}
    lengths = [256, 250, 256, 231, 256, 256, 17, 256] * (N // 8)
    v, t, _, x, z, _, mask = _rand_eps_inputs(N, L, lengths, 1000 + N, [(25, 33)])
    R = dev(G.so3_exp(v.cpu()))
    out = blk(R, t, x, z, mask)
    ids = list(range(8)) if N == 8 else [0, 6, 9, 17, 23, 31]
    ix = torch.tensor(ids, device=DEV)
    ref = ipa.ga_block(sd, '', R[ix].cpu(), t[ix].cpu(), x[ix].cpu(), z[ix].cpu(), mask[ix].cpu(), mode='mm')
    assert max_abs(out[ix].cpu(), ref) < 3e-5


@pytest.mark.parametrize('flavour,N', [('abdesign', 32), ('abdock', 64)])
def test_eps_net_bench_geometry_vs_oracle(flavour, N):
    """EpsilonNet through the SAMPLER's launch path (per-call pair-bias cache, N % 8 == 0) at the BASELINE config-2 shape
    (AbDesign flavour, N=32, L=256) and the config-3 shape (AbDock flavour, N=64 poses, L=256), against the oracle on a subset of
    the independent samples; then one full denoising step (transitions with injected draws) at the same shape."""
    from ab_opt_amd import hip
    from oracle import dpm, geometry as G
    L, T, t = 256, 100, 63
    if flavour == 'abdesign':
        d_cpu = standalone_abdesign_dpm(T, 2)
        d = standalone_abdesign_dpm(T, 2).to(DEV)
        pre = ''
    else:
        d_cpu = build_model(T, 2).diffusion
        d = build_model(T, 2, device=DEV).diffusion
        pre = ''
    sd = {k: v.cpu() for k, v in d_cpu.state_dict().items()}
    lengths = ([256] * 5 + [243, 256, 200]) * (N // 8)
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lengths, 2000 + N, [(25, 33), (51, 57), (94, 106), (133, 144), (159, 166), (198, 207)])
    beta = d.trans_pos.var_sched.betas[t].expand([N]).contiguous()
    pbc = hip.pair_bias_cache(d.eps_net.encoder.packed_array(), 6, pf)
    # ... and with the fp16 pair terms the sampler builds next to it at these shapes (round 6: the pair aggregation on the fp16 matrix instructions)
    assert hip.pair_terms_used(N, L)
    net = hip.eps_net_forward(d.eps_net.packed(), v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc, pair_terms=hip.pair_terms(pf))
    ids = [0, 5, N // 2 + 1, N - 1]
    ix = torch.tensor(ids, device=DEV)
    c = lambda a: a[ix].cpu()
    inv = d_cpu.trans_rot.angular_distrib_inv
    den = dpm.Denoiser(sd, num_steps=T, variant=flavour, obj='pred_x0', mode='mm', pre=pre,
                       tables=(None, dict(stddevs=inv.stddevs, approx_flag=inv.approx_flag, X=inv.X, Y=None)))
    ref = den._eps(c(v), c(p), c(s), c(rf), c(pf), c(beta), c(gen), c(mres), False)
    assert max_abs(c(net['R_next']), ref[1]) < 3e-5
    assert max_abs(c(net['eps_pos']), ref[2]) < 3e-5
    assert max_abs(c(net['c']), ref[3]) < 1e-5
    if d.abdock:
        assert max_abs(c(net['prmsd_logits']), ref[4]) < 3e-5
    # one whole step of the sampling loop at this shape, draws injected; config 3 runs structure-only
    gtor = torch.Generator(device=DEV).manual_seed(7)
    nz = dict(axis=torch.randn(N, L, 3, device=DEV, generator=gtor), bin=torch.randint(0, 8191, (N, L), device=DEV, generator=gtor),
              ubin=torch.rand(N, L, device=DEV, generator=gtor), gauss=torch.randn(N, L, device=DEV, generator=gtor),
              z=torch.randn(N, L, 3, device=DEV, generator=gtor), s_next=torch.randint(0, 20, (N, L), device=DEV, generator=gtor))
    seq = flavour == 'abdesign'
    cn = {k: c(a) for k, a in nz.items()}
    # t = 63 / 100: histogram rows of the inverse IGSO(3) table; t = 21 / 2: its Gaussian branch (so3.py:127-135), 20 of the sampler's 100 steps
    for t in (63, 2, 21, 100):
        tv, tp, ts, tpr, tpp = d._run((v.clone(), p * 10, s.clone()), t, rf, pf, gen, mres, True, seq, True, {t: nz}, 0, 0, False, stop_after=1)
        v_n, p_n, s_n, ex = den.step(t, c(v), c(p * 10) / 10, c(s), c(rf), c(pf), c(gen), c(mres), cn, sample_sequence=seq)
        # pred_x0 turns the network's position into noise by dividing by sqrt(1 / abar_t - 1) (transition.py:42-50): 0.07 at t = 2, i.e. the
        # 4e-7 fp32 noise of the network output is 2e-4 A there whatever computes it (two builds of this library that differ in the last bit of
        # eps_pos -- both 4.1e-7 from the float64 oracle -- measured 0.9e-4 and 2.3e-4); 1e-4 A holds from t = 10 up
        # ... for the AbDock flavour (obj = pred_x0) only: the AbDesign flavour predicts the noise itself (pred_noise) and holds 1e-4 A at every t
        assert max_abs(tp[t - 1][ix].cpu(), p_n * 10) < (1e-4 if (t >= 10 or flavour == 'abdesign') else 1e-3), t
        e = dpm.so3_noise(den.tab_inv, torch.full((len(ids), L), t), cn)
        if t in (2, 21):
            sd_t = inv.stddevs[t]
            assert bool(inv.approx_flag[t]) and max_abs(e.norm(dim=-1), (2 * sd_t + cn['gauss'] * sd_t).abs() % math.pi) < 1e-6
        n, worst = rot_close(tv[t - 1][ix].cpu(), v_n, G.so3_exp(e) @ G.so3_exp(ex['eps_out'][0]), R_upstream=ex['eps_out'][1])
        assert n >= 0.9 * len(ids) * L and worst < 1.0, (t, n, worst)
        assert torch.equal(ts[t - 1][ix].cpu(), s_n)
        if d.abdock:
            assert max_abs(tpr[t - 1][ix].cpu(), ex['prmsd']) < 1e-4 and max_abs(tpp[t - 1][ix].cpu(), ex['ppl']) < 1e-5, t


def test_non_contiguous_inputs_keep_their_copies_alive():
    """Sliced / transposed inputs: the binding makes contiguous copies that must outlive the launch (two temporaries freed
    early would alias in the caching allocator).  Results must equal the contiguous call bit for bit."""
    blk = _block_on_device()
    R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(2, 40, [40, 33], salt=900)]
    base = blk(R, t, x, z, mask)
    pad = lambda a: torch.stack([a, a + 1], dim=-1)[..., 0]                       # same values, non-contiguous view
    assert not pad(t).is_contiguous()
    assert torch.equal(blk(pad(R), pad(t), pad(x), z, mask), base)
    m = build_model(100, 2, device=DEV)
    args = [dev(a) for a in cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)])]
    ref = m.diffusion.eps_net(*args)
    args2 = [pad(a) if a.dtype == torch.float32 and a.dim() >= 2 and a.dim() < 4 else a for a in args]
    got = m.diffusion.eps_net(*args2)
    assert all(torch.equal(a, b) for a, b in zip(ref, got))


# ------------------------------------------------------------------------------------------ DockQ of docked candidates
def test_dockq_lite_vs_reference():
    """abopt_dockq_lite against the golden built from the reference's `fnat` program and DockQ.py's formulas (tolerance 1e-4 on
    Fnat / iRMS / LRMS / DockQ), per-candidate masks and the shared-mask form, plus properties at the BASELINE size (L=256, S=64):
    a rigid motion of the whole complex changes nothing, the native itself scores 1."""
    from ab_opt_amd import sampler
    from oracle import geometry as G
    g = load_golden('dockq_small')
    pos, mask, group, models = [dev(a) for a in cases.dockq_case()]
    S = models.shape[0]
    for mm in (mask, mask[None].expand(S, -1, -1).contiguous()):
        out = sampler.dockq_scores(models, mm, pos, mask, group=group)
        for k in ('fnat', 'irms', 'Lrms', 'DockQ'):
            assert max_abs(out[k].cpu(), g[k]) < 1e-4, (k, out[k].cpu(), g[k])
    # full size: 64 candidates of a 256-residue complex (chains from fragment_type), invariance under a global rigid motion
    b = synth.make_batch(1, synth.LAYOUT_256, seed=3)
    npos, nmask, ft = dev(b['pos_heavyatom'][0]), dev(b['mask_heavyatom'][0]), dev(b['fragment_type'][0])
    ab = (ft == 1) | (ft == 2)
    S = 64
    mods = npos[None].repeat(S, 1, 1, 1)
    mods[:, ab] += dev(synth.hash_tensor((S, 1, 1, 3), 5, scale=6.0)) + dev(synth.hash_tensor((S, int(ab.sum()), 15, 3), 6, scale=0.5))
    mods[0] = npos
    base = sampler.dockq_scores(mods, nmask, npos, nmask, fragment_type=ft)
    assert abs(base['DockQ'][0].item() - 1) < 1e-5 and base['irms'][0].item() < 1e-4 and base['fnat'][0].item() == 1
    assert (base['DockQ'][1:] < 1).all() and torch.isfinite(base['DockQ']).all()
    Q = dev(G.so3_exp(torch.tensor([[0.7, -0.4, 1.1]])))[0]
    moved = sampler.dockq_scores(mods @ Q.T + 3.0, nmask, npos, nmask, fragment_type=ft)
    for k in base:
        assert max_abs(moved[k], base[k]) < 2e-4, k


# ------------------------------------------------------------------------------------------ two ranks on one GPU
def _spawn2(fn, tmp_path):
    import socket
    import torch.multiprocessing as mp
    for attempt in range(2):
        sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
        try:
            mp.spawn(fn, args=(2, port, str(tmp_path)), nprocs=2, join=True)
            return
        except Exception as e:             # a rendezvous that failed (the free port was taken between close() and the workers' bind, a slow first import
            msg = str(e)                   # on a fresh box) is retried once on a new port; anything the workers compute is not
            if attempt == 1 or not any(w in msg for w in ('init_process_group', 'ddress already in use', 'onnection', 'timed out', 'Timeout')):
                raise


def test_two_rank_sharded_sampling_is_bit_identical_to_one_rank(tmp_path):
    """sampler.sample_sharded on 2 ranks (both on cuda:0, gloo group, gather through host memory) against the same call in one
    process: gathered candidates and top-k are bit-equal -- Philox counters depend on the GLOBAL sample index, never on the rank.
    Also the by-complex partition of BASELINE config 4 (design_testset_sharded): every rank ends with the same per-complex
    rankings as a single process."""
    import mp_workers
    from ab_opt_amd import sampler
    m = build_model(10, 3, device=DEV)
    b = {k: dev(v) for k, v in synth.make_batch(5, synth.LAYOUT_128, seed=11, replicate=True).items()}
    traj, (a, e), top, cand = sampler.sample_sharded(m, b, dict(sample_structure=True, sample_sequence=True, contig=''), k=2, seed=42)
    cx = [{k: dev(v) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=100 + c).items()} for c in range(3)]
    ref = sampler.design_testset_sharded(m, cx, 4, k=2, seed=7, complexes_per_launch=1)
    assert [r['complex'] for r in ref] == [0, 1, 2] and ref[0]['ca'].shape[0] == 4
    _spawn2(mp_workers.sharded_worker, tmp_path)
    parts = [torch.load(tmp_path / f'sharded_{r}.pt') for r in range(2)]
    assert (parts[0]['a'], parts[0]['e'], parts[1]['a'], parts[1]['e']) == (0, 3, 3, 5)
    for p_ in parts:
        assert torch.equal(p_['cand'], cand.cpu()) and torch.equal(p_['top'], top.cpu())
    assert torch.equal(torch.cat([parts[0]['p0'], parts[1]['p0']]), traj[0][1].cpu())
    assert torch.equal(torch.cat([parts[0]['s0'], parts[1]['s0']]), traj[0][2].cpu())
    for r in range(2):
        got = torch.load(tmp_path / f'testset_{r}.pt', weights_only=False)
        assert [g_['complex'] for g_ in got] == [0, 1, 2] and [g_['rank'] for g_ in got] == [0, 0, 1]
        for g_, r_ in zip(got, ref):
            assert torch.equal(g_['ca'], r_['ca']) and torch.equal(g_['top'], r_['top']) and torch.equal(g_['score'], r_['score'])


@pytest.mark.gpu
def test_config4_rank_leg_grouped_launch(monkeypatch):
    """BASELINE config 4, one rank's leg at its own size: 8 complexes x 16 samples, L = 256, as ONE batch of 128 samples whose samples
    share the pair features of their complex (abopt_eps_net_forward(pair_feat_shared = 16): a complex stays on one XCD, z is read once
    per complex and query block).  (i) EpsilonNet of the grouped batch against the oracle on a subset of samples (different complexes,
    first / last sample of a group); (ii) the grouped launch is bit-identical to the same samples run with their pair features
    replicated per sample, and to eight per-complex launches of 16 (the path design_testset_sharded took before); (iii) the test-set
    driver: per-complex commonness ranking and DockQ from the grouped launch equal the per-complex path's, bit for bit."""
    from ab_opt_amd import hip, sampler
    from oracle import dpm as odpm
    monkeypatch.setenv('ABOPT_CORE_NO_SPLIT', '1')
    G, S, L = 8, 16, 256
    m = build_model(10, 3, device=DEV)
    d = m.diffusion
    lens = [L - 3 * g for g in range(G)]
    rf = dev(synth.hash_tensor((G, L, 128), 811, scale=2.0))
    pf = dev(synth.hash_tensor((G, L, L, 64), 812, scale=2.0))
    mres_c = synth.mask_from_lengths(lens, L)
    N = G * S
    v = dev(synth.hash_tensor((N, L, 3), 813, scale=4.0)); p = dev(synth.hash_tensor((N, L, 3), 814, scale=3.0))
    s = dev((synth.hash_tensor((N, L), 815) + 0.5).mul(21).long().clamp(0, 20))
    mres = dev(mres_c.repeat_interleave(S, 0))
    gen = dev(synth.gen_from_ranges(N, L, [(25, 33), (94, 106)])) & mres
    beta = d.trans_pos.var_sched.betas[7].expand([N]).contiguous()
    arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
    grouped = hip.eps_net_forward(ew, v, p, s, rf.repeat_interleave(S, 0), pf, beta, gen, mres, d.abdock, d.num_bins, False,
                                  pair_bias_cache=hip.pair_bias_cache(arr, 6, pf), pair_feat_shared=S)
    grouped = {k: (a.clone() if a is not None else None) for k, a in grouped.items()}
    # the same launch with the fp16 pair terms the sampler builds at this shape (one set of terms per complex, shared like the bias cache)
    assert hip.pair_terms_used(N, L, S)
    # one buffer descriptor addresses a block's chunk-major cache slab (distinct samples x L x L/16 x 768 bytes): beyond 4 GB the 16-row kernels take the launch
    assert hip.pair_terms_used(1360, 256, 0) and not hip.pair_terms_used(1400, 256, 0) and hip.pair_terms_used(1400, 256, 100)
    grouped_t = hip.eps_net_forward(ew, v, p, s, rf.repeat_interleave(S, 0), pf, beta, gen, mres, d.abdock, d.num_bins, False,
                                    pair_bias_cache=hip.pair_bias_cache(arr, 6, pf), pair_feat_shared=S, pair_terms=hip.pair_terms(pf))
    grouped_t = {k: (a.clone() if a is not None else None) for k, a in grouped_t.items()}
    # (i) oracle on a subset
    sd = {k: a.detach().cpu() for k, a in m.state_dict().items()}
    for n in (0, S - 1, 5 * S + 3, N - 1):
        c = n // S
        sl = slice(n, n + 1)
        ref = odpm.eps_net(sd, 'diffusion.eps_net.', v[sl].cpu(), p[sl].cpu(), s[sl].cpu(), rf[c:c + 1].cpu(), pf[c:c + 1].cpu(), beta[sl].cpu(), gen[sl].cpu(),
                           mres[sl].cpu(), num_layers=6, prmsd_head=True, mode='mm')
        for name, kk in (('R_next', 1), ('eps_pos', 2), ('c', 3)):
            assert (grouped[name][sl].cpu() - ref[kk]).abs().max().item() < 5e-5, (n, name)
            assert (grouped_t[name][sl].cpu() - ref[kk]).abs().max().item() < 5e-5, (n, name)
    # (ii) replicated pair features, and per-complex launches
    rep = hip.eps_net_forward(ew, v[:4 * S], p[:4 * S], s[:4 * S], rf[:4].repeat_interleave(S, 0), pf[:4].repeat_interleave(S, 0).contiguous(), beta[:4 * S],
                              gen[:4 * S], mres[:4 * S], d.abdock, d.num_bins, False, pair_bias_cache=hip.pair_bias_cache(arr, 6, pf[:4].repeat_interleave(S, 0).contiguous()))
    for k in ('R_next', 'eps_pos', 'c'):
        assert torch.equal(rep[k], grouped[k][:4 * S]), k
    for c in (0, 3, 7):
        sl = slice(c * S, (c + 1) * S)
        one = hip.eps_net_forward(ew, v[sl], p[sl], s[sl], rf[c:c + 1].expand(S, -1, -1).contiguous(), pf[c:c + 1], beta[sl], gen[sl], mres[sl], d.abdock, d.num_bins, False,
                                  pair_bias_cache=hip.pair_bias_cache(arr, 6, pf[c:c + 1]), pair_feat_shared=True)
        for k in ('R_next', 'eps_pos', 'c', 'prmsd_logits'):
            assert torch.equal(one[k], grouped[k][sl]), (c, k)
    # (iii) the driver: eight different complexes, 16 samples each.  Bit-identity between launch geometries is a statement about ONE arithmetic form: the
    # grouped launch of 128 samples takes the 32-row kernels (pair aggregation on fp16 terms since round 6), a launch of 16 samples the 16-row kernels
    # (fp32 products) -- equal within fp32 noise per step, not bit for bit; with ABOPT_PAIR_TERMS=0 both aggregate in fp32 and the bits agree
    monkeypatch.setenv('ABOPT_PAIR_TERMS', '0')
    cx = [{k: dev(a) for k, a in synth.make_batch(1, synth.LAYOUT_256, seed=300 + c).items()} for c in range(G)]
    a = sampler.design_testset_sharded(m, cx, S, k=3, seed=11, native=True)
    b = sampler.design_testset_sharded(m, cx, S, k=3, seed=11, native=True, complexes_per_launch=1)
    assert [r['complex'] for r in a] == list(range(G))
    for ra, rb in zip(a, b):
        assert torch.isfinite(ra['ca']).all() and torch.equal(ra['ca'], rb['ca']) and torch.equal(ra['score'], rb['score']) and torch.equal(ra['top'], rb['top'])
        for kk in ('DockQ', 'fnat', 'irms', 'Lrms'):
            assert torch.equal(ra['dockq'][kk], rb['dockq'][kk]) and torch.isfinite(ra['dockq'][kk]).all()
        assert ra['dockq']['DockQ'].shape == (S,) and float(ra['dockq']['DockQ'].min()) >= 0 and float(ra['dockq']['DockQ'].max()) <= 1
    # complexes of different lengths in one launch: padded like PaddingCollate pads a batch; the padding never reaches a real residue
    rag = [{k: dev(a) for k, a in synth.make_batch(1, synth.LAYOUT_256, seed=400 + c, lengths=[L - 9 * c]).items()} for c in range(3)]
    r3 = sampler.design_testset_sharded(m, rag, 4, k=2, seed=5)
    assert [int(r['ca'].shape[0]) for r in r3] == [4, 4, 4] and all(torch.isfinite(r['ca']).all() for r in r3)
    # ... and, every launch being padded to the longest complex of the set, a sample's Philox counters and therefore its positions do not
    # depend on the grouping (launches padded to their own lengths shared draws between complexes)
    r1 = sampler.design_testset_sharded(m, rag, 4, k=2, seed=5, complexes_per_launch=1)
    for ra, rb in zip(r3, r1):
        assert torch.equal(ra['ca'], rb['ca']) and torch.equal(ra['top'], rb['top'])
    assert not torch.equal(r3[0]['ca'][0], r3[1]['ca'][0]) and not torch.equal(r3[1]['ca'][0], r3[2]['ca'][0])
    # the driver's default at this size (pair terms on): finite, ranked, and the terms were in fact used
    monkeypatch.delenv('ABOPT_PAIR_TERMS')
    at = sampler.design_testset_sharded(m, cx, S, k=3, seed=11, native=True)
    assert m.diffusion.last_run_info['pair_terms'] and all(torch.isfinite(r['ca']).all() and r['top'].numel() == 3 for r in at)


def test_two_rank_ddp_gradients_equal_the_mean(tmp_path):
    """model(batch).backward() under DistributedDataParallel (2 ranks, one sample each): the all-reduced gradients equal the mean of
    the two single-sample gradients computed in one process (the custom autograd functions survive DDP's bucketed all-reduce)."""
    import mp_workers
    m = build_model(10, 3, device=DEV).train()
    full = synth.make_batch(2, synth.LAYOUT_128, seed=5, lengths=[64, 57])
    grads = []
    for r in range(2):
        m.zero_grad()
        torch.manual_seed(123)
        sum(m({k: dev(v[r:r + 1]) for k, v in full.items()}).values()).backward()
        grads.append({n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None})
    m.zero_grad(); m.eval()
    _spawn2(mp_workers.ddp_worker, tmp_path)
    got = [torch.load(tmp_path / f'ddp_{r}.pt') for r in range(2)]
    assert len(got[0]['grads']) > 150
    for n, g0 in got[0]['grads'].items():
        assert torch.equal(g0, got[1]['grads'][n]), n                                   # both ranks hold the same averaged gradient
        a_, b_ = grads[0].get(n), grads[1].get(n)
        if a_ is None or b_ is None:
            continue
        mean = (a_ + b_) / 2
        assert max_abs(g0, mean) <= 1e-5 * max(1.0, mean.abs().max().item()), n


# ------------------------------------------------------------------------------------------ BASELINE config 5 flavour
def test_training_abdesign_loss_and_grads_vs_reference():
    """AbDesign FullDPM.forward on the device (rot, pos on the noise, seq) with the reference's recorded noise and fixed t:
    losses (2e-5 rel) and gradients (3e-4 of max) against golden training_abdesign."""
    from test_oracle_golden import _sub
    g = load_golden('training_abdesign')
    d = standalone_abdesign_dpm(100, 2).to(DEV).train()
    d.zero_grad()
    N, L = 2, 48
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
    s = s.clamp(max=19)
    res_feat = dev(res_feat).clone().requires_grad_(True)
    pair_feat = dev(pair_feat).clone().requires_grad_(True)
    noise = dict(axis=dev(g['rot_axis']), bin=dev(g['rot_bin']), ubin=dev(g['rot_ubin']), gauss=dev(g['rot_gauss']), pos=dev(g['pos']), s_noisy=dev(g['s_noisy']))
    loss = d(dev(v), dev(p) * 10, dev(s), res_feat, pair_feat, dev(gen), dev(mres), True, True, t=torch.tensor([37, 80], device=DEV), noise=noise)
    assert set(loss) == {'rot', 'pos', 'seq'}
    for k, val in loss.items():
        ref = g['loss_' + k].item()
        assert abs(val.item() - ref) <= 2e-5 * max(1.0, abs(ref)), (k, val.item(), ref)
    sum(loss.values()).backward()
    params = dict(d.named_parameters())
    n = 0
    for k in g:
        if k.startswith('grad_eps_net'):
            got = _sub(params[k[len('grad_'):]].grad.cpu(), g[k])
            assert max_abs(got, g[k]) <= 3e-4 * g[k].abs().max().item() + 1e-7, k
            n += 1
    assert n == 10
    assert max_abs(res_feat.grad.cpu(), g['grad_res_feat']) <= 3e-4 * g['grad_res_feat'].abs().max().item()
    assert max_abs(pair_feat.grad.cpu()[:, ::5, ::3], g['grad_pair_feat_sub']) <= 3e-4 * g['grad_pair_feat_sub'].abs().max().item()
    d.zero_grad()


def test_fused_node_projection_matches_gemm_path():
    """node_frags.hip (projection GEMM + frame transform + fragment layout in one kernel, weights packed per head) and the fused
    out_transform + LayerNorm/MLP tail (mlp.hip::out_ln_mlp_kernel) against the plain path (gemm_xwT + ipa_frags + split-K GEMM +
    fused_ln_mlp) through the C ABI: same block output up to fp32 summation order, ragged lengths and row counts that are not a
    multiple of the 32-row tile included."""
    from ab_opt_amd import hip
    blk = _block_on_device(seed=11)
    t_, s_full = blk.packed()
    assert 'w_node_frag' in t_ and t_['w_node_frag'].numel() == hip.lib().abopt_node_frag_floats()
    assert t_['w_out_frag'].numel() == 128 * 1824
    # the device packer (abopt_pack_tail_weights) and the host statement of the layouts agree bit for bit
    wof_h, wmf_h = hip.pack_tail_weights_host(t_['w_out'], t_['w_mlp0'], t_['w_mlp1'], t_['w_mlp2'])
    assert torch.equal(t_['w_out_frag'].view(torch.int32), wof_h.view(torch.int32))
    nz = 3 * 128 * 128 + 8                                             # layer terms + the eight scale slots; behind them 256 slice maxima (scratch of the packer), then zeros
    assert torch.equal(t_['w_mlp_frag'][:nz].view(torch.int32), wmf_h[:nz].view(torch.int32)) and not bool(t_['w_mlp_frag'][nz + 256:].any())
    assert torch.equal(t_['w_out_terms'].view(torch.int32), wof_h.view(torch.int32))
    plain = hip.ga_weights_struct({k: v for k, v in t_.items() if k not in ('w_node_frag', 'w_out_frag', 'w_mlp_frag')})
    for N, L, lengths in ((2, 40, [40, 33]), (3, 70, [70, 33, 1]), (8, 256, [256, 250, 256, 231, 256, 256, 17, 256])):
        R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(N, L, lengths, salt=1200 + L)]
        a = hip.ga_block_forward(s_full, R, t, x, z, mask)
        b = hip.ga_block_forward(plain, R, t, x, z, mask)
        assert max_abs(a, b) < 1e-5, (N, L, max_abs(a, b))


@pytest.mark.parametrize('N,L,lengths', [(1, 7, [5]), (3, 70, [70, 33, 1]), (4, 256, [256, 250, 17, 256])])
def test_block_tail_autograd_function_vs_torch_statement(N, L, lengths):
    """training.BlockTail (fused forward with activation dump + tail_backward_kernel + five GEMMs, csrc/mlp.hip) against the plain
    torch statement of ga.py:174-177: output, d x, d feat and the gradient of every parameter of the tail; row counts that are not a
    multiple of the 32-row tile and masked rows included."""
    from ab_opt_amd import training
    import plain_statement
    blk = _block_on_device(seed=13)
    with torch.no_grad():
        for ln in (blk.layer_norm_1, blk.layer_norm_2):          # non-trivial LayerNorm parameters
            ln.gamma.copy_(1.0 + 0.3 * dev(synth.hash_tensor((128,), 91, scale=1.0)))
            ln.beta.copy_(0.2 * dev(synth.hash_tensor((128,), 92, scale=1.0)))
    _, _, x, _, mask = [dev(a) for a in cases.ipa_inputs(N, L, lengths, salt=1300 + L)]
    feat = dev(synth.hash_tensor((N, L, 1824), 93, scale=1.0))
    wout = dev(synth.hash_tensor((N, L, 128), 94, scale=1.0))
    names = ['out_transform', 'layer_norm_1', 'mlp_transition', 'layer_norm_2']
    res = {}
    for native in (False, True):
        blk.zero_grad()
        xx, ff = x.clone().requires_grad_(True), feat.clone().requires_grad_(True)
        out = training._block_tail(blk, xx, ff, mask) if native else plain_statement.block_tail(blk, xx, ff, mask)
        (out * wout).sum().backward()
        res[native] = dict(out=out.detach(), dx=xx.grad, dfeat=ff.grad,
                           **{'d_' + n: p.grad.clone() for n, p in blk.named_parameters() if n.split('.')[0] in names})
    assert len(res[True]) == 3 + 12
    for k, ref in res[False].items():
        tol = 2e-5 if k == 'out' else 1e-4
        assert max_abs(res[True][k], ref) <= tol * max(1.0, ref.abs().max().item()), (k, max_abs(res[True][k], ref), ref.abs().max().item())
    blk.zero_grad()


@pytest.mark.parametrize('wscale,fscale', [(1.0, 1.0), (1e-4, 1.0), (300.0, 1.0), (1.0, 1e-3), (1.0, 200.0), (0.0, 1.0)])
def test_two_term_fp16_products_are_fp32_accurate(wscale, fscale):
    """The dense layers of the forward tail and the node projections run on the fp16 matrix pipe: two fp16 terms per operand, three products, the
    weights multiplied by a power of two per matrix before the split (csrc/ipa_common.h: split_pair2; csrc/mlp.hip: tail_weight_absmax_kernel).
    Stated accuracy: the same as an fp32 GEMM.  Checked here against an fp64 statement of ga.py:174-177 with the weights / the aggregated features
    scaled over six orders of magnitude (the per-matrix scale and the subnormal low terms of small operands), and an all-zero W_out (scale slot = 1):
    the error of the HIP tail is at most 3x the error of the same statement in torch fp32 (+ 2e-7 of the output range)."""
    from ab_opt_amd import hip
    import plain_statement
    N, L = 3, 70
    blk = _block_on_device(seed=17)
    with torch.no_grad():
        blk.out_transform.weight.mul_(wscale)
        for i in (0, 2, 4):
            blk.mlp_transition[i].weight.mul_(max(wscale, 1e-4) if wscale != 300.0 else 3.0)
    _, _, x, _, mask = [dev(a) for a in cases.ipa_inputs(N, L, [70, 33, 1], salt=4100)]
    feat = dev(synth.hash_tensor((N, L, 1824), 4101, scale=1.0)) * fscale
    feat[..., :200] *= 1e-3                                           # columns whose low terms are fp16 subnormals
    t_ = blk.packed()[0]
    S = t_['w_mlp_frag'][3 * 128 * 128: 3 * 128 * 128 + 8].cpu()
    assert torch.equal(S[:4] * S[4:], torch.ones(4)) and all(float(v) == 2.0 ** round(float(torch.log2(v))) for v in S[:4])
    wmax = blk.out_transform.weight.abs().max().item()
    assert (S[0].item() == 1.0) if wmax == 0 else (2.0 ** 14 <= wmax * S[0].item() < 2.0 ** 15)
    out = hip.block_tail_forward(feat.reshape(-1, 1824), t_['w_out_frag'], t_['w_mlp_frag'], x.reshape(-1, 128), t_['b_out'], mask.reshape(-1), t_['ln1_gamma'],
                                 t_['ln1_beta'], t_['b_mlp0'], t_['b_mlp1'], t_['b_mlp2'], t_['ln2_gamma'], t_['ln2_beta']).reshape(N, L, 128)
    with torch.no_grad():
        ref32 = plain_statement.block_tail(blk, x, feat, mask)
        blk.double()
        ref64 = plain_statement.block_tail(blk, x.double(), feat.double(), mask)
    e_hip, e_f32 = max_abs(out, ref64), max_abs(ref32, ref64)
    assert torch.isfinite(out).all()
    assert e_hip <= 3.0 * e_f32 + 2e-7 * ref64.abs().max().item(), (wscale, fscale, e_hip, e_f32)


def test_fp16_range_guard_falls_back_to_fp32_layers():
    """The dense layers multiply on two fp16 terms per operand: an activation beyond 65504 -- which the fp32 reference takes -- becomes inf there and NaN
    in the outputs.  Not silently: the heads' epilogue raises a device flag (abopt_nonfinite_flag), EpsilonNet.forward and FullDPM.sample / optimize
    read it once per call and repeat the call with the dense layers as fp32 GEMMs.  Node features of 1e5: the fp16
    path returns NaN and raises the flag; the guarded entry points warn and return what the fp32 path returns (bit for bit), finite and close to the
    oracle; ordinary inputs neither warn nor raise the flag."""
    import warnings
    from ab_opt_amd import hip
    from oracle import dpm as odpm
    T, N, L = 10, 2, 48
    d = build_model(T, 3, device=DEV).diffusion
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, [48, 40], 5100, [(6, 17), (30, 37)])
    beta = d.trans_pos.var_sched.betas[7].expand([N]).contiguous()
    net = d.eps_net
    with warnings.catch_warnings():
        warnings.simplefilter('error')                                      # ordinary magnitudes: no warning, flag down
        ok = net(v, p, s, rf, pf, beta, gen, mres)
    assert not hip.nonfinite_flag() and all(torch.isfinite(o).all() for o in ok)
    big = rf * (1.0e5 / rf.abs().max())
    raw = hip.eps_net_forward(net.packed(), v, p, s, big, pf, beta, gen, mres, d.abdock, d.num_bins, False)
    assert not torch.isfinite(raw['R_next']).all() and hip.nonfinite_flag(reset=True) and not hip.nonfinite_flag()       # NaN, flag up, then cleared
    safe = hip.eps_net_forward(net.packed_fp32(), v, p, s, big, pf, beta, gen, mres, d.abdock, d.num_bins, False)
    safe = {k: a.clone() for k, a in safe.items() if a is not None}
    assert all(torch.isfinite(a).all() for a in safe.values()) and not hip.nonfinite_flag()
    with pytest.warns(RuntimeWarning, match='fp16 range'):
        got = net(v, p, s, big, pf, beta, gen, mres)
    assert torch.equal(got[1], safe['R_next']) and torch.equal(got[2], safe['eps_pos']) and torch.equal(got[3], safe['c'])
    sd = {k: a.detach().cpu() for k, a in d.state_dict().items()}
    ref = odpm.eps_net(sd, 'eps_net.', v.cpu(), p.cpu(), s.cpu(), big.cpu(), pf.cpu(), beta.cpu(), gen.cpu(), mres.cpu(), num_layers=6, prmsd_head=True, mode='mm')
    assert max_abs(got[2].cpu(), ref[2]) < 1e-3 * max(1.0, ref[2].abs().max().item()) and max_abs(got[3].cpu(), ref[3]) < 1e-3
    # the sampling loop: one flag read per call, the whole call repeated on the fp32 layers
    with pytest.warns(RuntimeWarning, match='fp16 range'):
        traj = d.sample(v, p * 10, s, big, pf, gen, mres, seed=5)
    assert all(torch.isfinite(traj[t][1]).all() for t in traj)
    hh = d._sched_host()
    state = hip.sample_init(v.float(), (p * 10).float(), s, gen, None, 5, 0, hh['scale'], hh['mean'], True, True)
    tv, tp, ts, _, _ = d._run_eager(state, T, big, pf, gen, mres, True, True, True, None, 5, 0, False, range_safe=True)
    assert torch.equal(traj[0][1], tp[0]) and torch.equal(traj[0][2], ts[0])
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        d.sample(v, p * 10, s, rf, pf, gen, mres, seed=5)


@pytest.mark.parametrize('flavour,N,L', [('abdesign', 3, 70), ('abdock', 32, 256), ('abdock', 40, 48)])
def test_node_feature_terms_travel_with_x_bit_identically(flavour, N, L, monkeypatch):
    """Round 6: the kernel that produces a block's node features x (the mixer for the first block, the previous block's tail afterwards) also writes them as two
    fp16 terms, and node_frags reads those instead of splitting x again in each of its 24 (head, half) workgroups.  The split is the same function on the same
    values, so nothing may change by a bit: EpsilonNet with ABOPT_X_TERMS=0 (every node_frags splits for itself) against the default, 16-row and 32-row kernels,
    fused and two-launch tails, ragged lengths."""
    from ab_opt_amd import hip
    d = standalone_abdesign_dpm(100, 2).to(DEV) if flavour == 'abdesign' else build_model(100, 2, device=DEV).diffusion
    lens = [L - (5 * i) % max(1, L // 3) for i in range(N)]
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lens, 9100 + N, [(5, 14), (22, 30)])
    beta = d.trans_pos.var_sched.betas[37].expand([N]).contiguous()
    arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
    pbc = hip.pair_bias_cache(arr, 6, pf)
    terms = hip.pair_terms(pf) if hip.pair_terms_used(N, L) else None
    outs = []
    for xt in ('0', '1'):
        monkeypatch.setenv('ABOPT_X_TERMS', xt)
        for fuse in ('0', '1'):
            monkeypatch.setenv('ABOPT_FUSE_TAIL', fuse)
            o = hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc, pair_terms=terms)
            outs.append({k: a.clone() for k, a in o.items() if a is not None})
    for o in outs[1:]:
        for k in outs[0]:
            assert torch.isfinite(o[k]).all() and torch.equal(o[k], outs[0][k]), k


def _pair_terms_statement(z, L):
    """torch statement of abopt_pair_terms (include/abopt.h): per (row, channel) power-of-two scale, two fp16 terms, K-packed layout."""
    N = z.shape[0]
    nch = (L + 15) // 16
    zp = torch.zeros(N, L, nch * 16, 64, device=z.device)
    zp[:, :, :L] = z
    am = zp.abs().amax(dim=2)                                             # [N, L, 64]
    ex = ((am.view(torch.int32) >> 23) & 0xff).clamp(64, 190)
    S = ((267 - ex) << 23).view(torch.float32)
    invS = ((ex - 27) << 23).view(torch.float32)                          # 2^-14 / S: the consumer's probabilities carry 2^14
    zs = zp * S[:, :, None, :]
    h = zs.half()
    l = (zs - h.float()).half()
    # [n, i, ch, kq, e, fm, mt] -> [ch, n, i, mt, kq, fm, term, e]: chunk-major over the whole batch (every workgroup of a launch reads the same chunk of its rows at the same time)
    t = torch.stack([h.view(N, L, nch, 4, 4, 16, 4), l.view(N, L, nch, 4, 4, 16, 4)], dim=-1).permute(2, 0, 1, 6, 3, 5, 7, 4).contiguous()
    return t, invS, S


def test_pair_terms_layout_and_scales():
    """abopt_pair_terms against its torch statement, bit for bit: scales (powers of two, max |z| S in [2^13, 2^14) per (row, channel)), the two fp16 terms,
    the K-packed operand order, zeros for keys past L, columns of zeros, magnitudes from 1e-6 to 1e4 across channels and rows."""
    from ab_opt_amd import hip
    for N, L in ((2, 50), (1, 16), (3, 129)):
        z = dev(synth.hash_tensor((N, L, L, 64), 9000 + L, scale=2.0))
        z = z * (10.0 ** torch.linspace(-6, 4, 64, device=DEV)) * (10.0 ** torch.linspace(-2, 2, L, device=DEV))[None, :, None, None]
        z[0, 3, :, 5] = 0.0
        z[0, 4] = 0.0
        blob = hip.pair_terms(z)
        nch = (L + 15) // 16
        nt = N * L * nch * 1024
        assert blob.numel() == nt + N * L * 64 == hip.pair_terms_bytes(N, L) // 4
        t, invS, S = _pair_terms_statement(z, L)
        assert torch.equal(blob[nt:].view(N, L, 64), invS)
        assert torch.equal(blob[:nt].view(torch.float16).view(nch, N, L, 4, 4, 16, 2, 4), t)
        am = z.abs().amax(dim=2)
        nz = am > 0
        assert ((am * S)[nz] >= 2.0 ** 13).all() and ((am * S)[nz] < 2.0 ** 14).all() and torch.equal((S * invS), torch.full_like(S, 2.0 ** -14))
        # the terms carry 22 bits of every value within 2^-17 of its column's largest
        rec = (t[..., 0, :].float() + t[..., 1, :].float())                # [ch, n, i, mt, kq, fm, e]
        zr = rec.permute(1, 2, 0, 4, 6, 5, 3).reshape(N, L, nch * 16, 64)[:, :, :L] * (invS * 2.0 ** 14)[:, :, None, :]
        big = z.abs() >= am[:, :, None, :] * 2.0 ** -17
        assert ((zr - z).abs()[big] <= z.abs()[big] * 2.0 ** -21).all()
        assert ((zr - z).abs() <= am[:, :, None, :] * 2.0 ** -37).logical_or(big).all()


@pytest.mark.parametrize('N,L,lengths', [(3, 100, [100, 77, 1]), (9, 130, None), (32, 256, None)])
def test_pair_aggregation_on_fp16_terms_vs_fp64(N, L, lengths, monkeypatch):
    """Round 6: the 32-row block kernels aggregate sum_j alpha[i,j,h] z[i,j,c] (ga.py:114-118) on the fp16 matrix instructions when handed abopt_pair_terms
    (z and the probabilities as two fp16 terms each, packed along the key index; per-(row, channel) power-of-two scales).  Stated accuracy: that of the fp32
    statement.  Checked on the pair features of a block (feat[:, :, :768]) against an fp64 contraction of the kernel's own alpha with z, with z scaled over
    SIX orders of magnitude across channels and over four across query rows: per channel, the error of the term path is at most 3x the error of the fp32
    path (+ 2e-7 of the channel's range); the other feature groups and the block output stay within 3x the fp32 path's error against the oracle in fp64."""
    from ab_opt_amd import hip
    monkeypatch.setenv('ABOPT_CORE32', '1')
    monkeypatch.setenv('ABOPT_CORE_NO_SPLIT', '1')
    blk = _block_on_device(seed=23)
    lens = [L] * N if lengths is None else lengths
    R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(N, L, lens, salt=8100 + N)]
    cs = 10.0 ** torch.linspace(-3, 3, 64, device=DEV)[torch.randperm(64, generator=torch.Generator().manual_seed(5)).to(DEV)]
    z = z * cs * (10.0 ** torch.linspace(-2, 2, L, device=DEV))[None, :, None, None]
    with torch.no_grad():
        blk.proj_pair_bias.weight.div_(cs * 30.0)                             # keep the pair logits O(1): the softmax must not collapse onto one key
    tns, st = blk.packed()
    arr = (hip.GaWeights * 1)(st)
    pbc = hip.pair_bias_cache(arr, 1, z)
    terms = hip.pair_terms(z)
    out32, feat32 = hip.ga_block_forward_cached(st, R, t, x, z, mask, pbc, None, want_feat=True)
    out16, feat16 = hip.ga_block_forward_cached(st, R, t, x, z, mask, pbc, terms, want_feat=True)
    fused = hip.ga_block_forward_cached(st, R, t, x, z, mask, pbc, terms)                     # core + tail in one launch: the same bits
    assert torch.isfinite(out16).all() and torch.equal(fused, out16)
    # the whole block in fp64 (the oracle's statements on double tensors)
    from oracle import ipa as oipa
    sd64 = {k: a.detach().cpu().double() for k, a in blk.state_dict().items()}
    parts = {}
    out64 = oipa.ga_block(sd64, '', R.cpu().double(), t.cpu().double(), x.cpu().double(), z.cpu().double(), mask.cpu(), mode='mm', parts=parts)
    ref64 = parts['feat'].to(DEV)
    valid = mask[:, :, None]
    e16 = ((feat16.double() - ref64).abs() * valid)[..., :768].reshape(-1, 12, 64).amax(dim=(0, 1))
    e32 = ((feat32.double() - ref64).abs() * valid)[..., :768].reshape(-1, 12, 64).amax(dim=(0, 1))
    rng = (ref64.abs() * valid)[..., :768].reshape(-1, 12, 64).amax(dim=(0, 1))
    assert (e16 <= 3.0 * e32 + 2e-7 * rng).all(), (e16 / (e32 + 1e-30)).max().item()
    # per element, relative to what the row's column of z can produce at all
    zmax = z.abs().amax(dim=2)[:, :, None, :].expand(N, L, 12, 64).reshape(N, L, 768).double()
    assert (((feat16[..., :768].double() - ref64[..., :768]).abs() * valid) <= 2e-6 * zmax + 1e-30).all()
    # node features, aggregated points, distances, directions: the logits' q . k part runs on fp16 terms too (22 bits against fp32's 24)
    for lo, hi in ((768, 1152), (1152, 1440), (1440, 1536), (1536, 1824)):
        f16, f32, f64 = [(a.double() * valid)[..., lo:hi] for a in (feat16, feat32, ref64)]
        assert (f16 - f64).abs().max() <= 3.0 * (f32 - f64).abs().max() + 1e-6 * max(1.0, f64.abs().max().item()), (lo, hi)
    assert max_abs(out16.cpu().double(), out64) <= 3.0 * max_abs(out32.cpu().double(), out64) + 2e-6


@pytest.mark.parametrize('xscale', [1.0, 0.05, 6.0])
def test_attention_logits_on_fp16_terms_vs_fp64(xscale, monkeypatch):
    """Round 6: handed pair terms, the 32-row block kernels also multiply the 32 channels of q / sqrt(D) and k as two fp16 terms each (three 16-cycle products
    per (head, row tile, chunk) instead of eight fp32 steps; the point part of the logit stays on fp32).  The logits themselves never leave the chip, so
    the check is on what they produce: every group of IPA features (pair | node | points | distances | directions) against the oracle's statements in
    fp64, with node features x scaled so that |q . k| is tiny, ordinary and large (sharply peaked softmax): the term path's error is at most 3x the fp32
    path's (+ 1e-6 of the group's range)."""
    from ab_opt_amd import hip
    from oracle import ipa as oipa
    monkeypatch.setenv('ABOPT_CORE32', '1')
    monkeypatch.setenv('ABOPT_CORE_NO_SPLIT', '1')
    N, L = 4, 112
    blk = _block_on_device(seed=29)
    R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(N, L, [112, 97, 50, 16], salt=8300)]
    x = x * xscale
    _, st = blk.packed()
    pbc = hip.pair_bias_cache((hip.GaWeights * 1)(st), 1, z)
    terms = hip.pair_terms(z)
    out32, feat32 = hip.ga_block_forward_cached(st, R, t, x, z, mask, pbc, None, want_feat=True)
    out16, feat16 = hip.ga_block_forward_cached(st, R, t, x, z, mask, pbc, terms, want_feat=True)
    assert torch.equal(hip.ga_block_forward_cached(st, R, t, x, z, mask, pbc, terms), out16) and not torch.equal(feat16, feat32)
    sd64 = {k: a.detach().cpu().double() for k, a in blk.state_dict().items()}
    parts = {}
    out64 = oipa.ga_block(sd64, '', R.cpu().double(), t.cpu().double(), x.cpu().double(), z.cpu().double(), mask.cpu(), mode='mm', parts=parts)
    ref64 = parts['feat'].to(DEV)
    valid = mask[:, :, None]
    for lo, hi in ((0, 768), (768, 1152), (1152, 1440), (1440, 1536), (1536, 1824)):
        f16, f32, f64 = [(a.double() * valid)[..., lo:hi] for a in (feat16, feat32, ref64)]
        assert (f16 - f64).abs().max() <= 3.0 * (f32 - f64).abs().max() + 1e-6 * max(1.0, f64.abs().max().item()), (xscale, lo, hi)
    assert max_abs(out16.cpu().double(), out64) <= 3.0 * max_abs(out32.cpu().double(), out64) + 2e-6


@pytest.mark.parametrize('fscale', [1e-3, 1e-4, 1e-6])
def test_two_term_fp16_products_with_small_activations(fscale):
    """activations are split WITHOUT a scale, so the low fp16 term of a value below 2^-3 is a subnormal and every activation carries an
    ABSOLUTE rounding floor of 2^-25 (the rounding of an fp32 number of size 0.5) instead of fp32's relative 2^-24.  Here EVERY aggregated feature is
    small (1e-3 ... 1e-6, no large column to hide behind).  Stated and checked: (i) u = feat . W_out^T formed from the two terms is off the fp64 product by
    at most that floor times the row's sum of |W_out| -- relative to u itself the term path is WORSE than an fp32 GEMM on such inputs: the accuracy claim
    of DESIGN.md section 3.2 is for activations of O(0.1 ... 1e4); (ii) what the block returns -- LayerNorm(x + u) and the layers behind it with node
    features x of ordinary size -- stays within 3x the error of the fp32 statement."""
    from ab_opt_amd import hip
    import plain_statement
    N, L = 3, 70
    blk = _block_on_device(seed=19)
    _, _, x, _, mask = [dev(a) for a in cases.ipa_inputs(N, L, [70, 33, 1], salt=4300)]
    feat = dev(synth.hash_tensor((N, L, 1824), 4301, scale=1.0)) * fscale
    t_ = blk.packed()[0]
    out = hip.block_tail_forward(feat.reshape(-1, 1824), t_['w_out_frag'], t_['w_mlp_frag'], x.reshape(-1, 128), t_['b_out'], mask.reshape(-1), t_['ln1_gamma'],
                                 t_['ln1_beta'], t_['b_mlp0'], t_['b_mlp1'], t_['b_mlp2'], t_['ln2_gamma'], t_['ln2_beta']).reshape(N, L, 128)
    with torch.no_grad():
        ref32 = plain_statement.block_tail(blk, x, feat, mask)
        W = blk.out_transform.weight.detach()
        # (i) the product itself, from the terms the kernel forms: h = fp16(f), l = fp16(f - h) -- the same roundings, the sums in fp64
        h = feat.half().double()
        l = (feat.double() - h).float().half().double()
        u_terms = (h + l) @ W.double().t()
        u64 = feat.double() @ W.double().t()
        floor = 2.0 ** -25 * W.abs().sum(dim=1).double()                      # per output column
        assert ((u_terms - u64).abs() <= floor + 1e-300).all()
        blk.double()
        ref64 = plain_statement.block_tail(blk, x.double(), feat.double(), mask)
    e_hip, e_f32 = max_abs(out, ref64), max_abs(ref32, ref64)
    assert torch.isfinite(out).all() and e_hip <= 3.0 * e_f32 + 2e-7 * ref64.abs().max().item(), (fscale, e_hip, e_f32)


@pytest.mark.parametrize('flavour', ['abdesign', 'abdock'])
def test_fused_heads_match_gemm_path(flavour, monkeypatch):
    """heads.hip (the three denoiser heads as one kernel, time features as an affine term; the mixer as one kernel with the sequence
    embedding folded into a table) against the GEMM path through the C ABI:
    same R_next / eps_pos / c up to fp32 summation order; per-sample beta, row count not a multiple of the 32-row tile.
    The heads' geometric epilogue (dpm_full.py:95-107) runs as the tail of the heads kernel; ABOPT_FUSE_HEADS=0 launches it on its own
    (rows.hip: heads_epilogue_kernel, the same device function): bit-identical."""
    from ab_opt_amd import hip
    T, t, N, L = 100, 41, 3, 70
    d = standalone_abdesign_dpm(T, 2).to(DEV) if flavour == 'abdesign' else build_model(T, 2, device=DEV).diffusion      # (never .to() a cached model: it moves in place)
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, [70, 33, 1], 3100, [(5, 14), (22, 30)])
    beta = d.trans_pos.var_sched.betas[torch.tensor([t, 7, 93], device=DEV)].contiguous()
    ew = d.eps_net.packed()
    assert ew.w_heads_frag
    plain = hip.EpsWeights()
    for name, _ in hip.EpsWeights._fields_:
        setattr(plain, name, getattr(ew, name))
    plain.w_heads_frag = None
    plain.w_mix_frag = None
    a = hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False)
    monkeypatch.setenv('ABOPT_FUSE_HEADS', '0')
    a2 = hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False)
    monkeypatch.delenv('ABOPT_FUSE_HEADS')
    for k in ('R_next', 'v_next', 'eps_pos', 'c'):
        assert torch.isfinite(a[k]).all() and torch.equal(a[k], a2[k]), k
    b = hip.eps_net_forward(plain, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False)
    for k in ('R_next', 'eps_pos', 'c'):          # v_next = log(R_next) amplifies near theta = pi; R_next pins it
        assert max_abs(a[k], b[k]) < 2e-5, (k, max_abs(a[k], b[k]))
    assert max_abs(a['v_next'], b['v_next']) < 1e-3


def test_ga_block_and_cache_above_2048_residues():
    """L = 2085 (> 2048, not a multiple of the 16-key chunk; round 1 fell back to a superseded kernel there): the sampler's block (fused
    projections, core with and without the pair-bias cache, fused tail) against the plain torch statement of the block evaluated on the
    same GPU -- which the reference's recorded gradients pin (test_ipa_core_autograd_function_vs_torch_statement)."""
    from ab_opt_amd import hip
    import plain_statement
    N, L = 1, 2085
    blk = _block_on_device(seed=17)
    R, t, x, z, mask = [dev(a) for a in cases.ipa_inputs(N, L, [2003], salt=4100)]
    with torch.no_grad():
        ref = plain_statement.ga_block(blk, R, t, x, z, mask)
    _, s_full = blk.packed()
    out = hip.ga_block_forward(s_full, R, t, x, z, mask)
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    assert max_abs(out, ref) <= tol, max_abs(out, ref)
    # the cached variant at the same length: EpsilonNet with and without the per-call cache, bit for bit
    d = standalone_abdesign_dpm(100, 2).to(DEV)
    v, p, s_, rf, pf, gen, mres = _rand_eps_inputs(N, L, [2003], 4200, [(25, 33), (1500, 1530)])
    beta = d.trans_pos.var_sched.betas[50].expand([N]).contiguous()
    pbc = hip.pair_bias_cache(d.eps_net.encoder.packed_array(), 6, pf)
    a = hip.eps_net_forward(d.eps_net.packed(), v, p, s_, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc)
    b = hip.eps_net_forward(d.eps_net.packed(), v, p, s_, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False)
    for k in ('R_next', 'eps_pos', 'c'):
        assert torch.isfinite(a[k]).all() and torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('N,L,lengths', [(32, 256, None), (40, 100, None), (24, 250, 'ragged'), (96, 48, 'ragged'), (136, 32, None), (72, 64, 'ragged')])
def test_persistent_core_is_bit_identical(N, L, lengths, monkeypatch):
    """With the pair-bias cache and more query blocks than CUs the sampler's core runs as ONE persistent workgroup per CU that walks
    several query blocks with the roles' pipeline kept across block boundaries (csrc/ipa_core.hip: ipa_core_persist_kernel).  Same
    arithmetic in the same order: every sample must come out bit-identical to the same sample run in a small batch (one block per
    workgroup, the plain kernel) -- odd block counts per workgroup, chunk counts not divisible by 3 and ragged lengths included.
    L <= 64: blocks of 2..4 positions, where the per-block pieces (q swap, epilogues) dominate."""
    from ab_opt_amd import hip
    monkeypatch.setenv('ABOPT_CORE_NO_SPLIT', '1')             # the small batches below would otherwise take the key-split form (other summation order)
    monkeypatch.setenv('ABOPT_CORE32', '0')                    # ... and the full-round batches the 32-row kernel (test_core32_is_bit_identical)
    d = standalone_abdesign_dpm(100, 2).to(DEV)
    lens = [L] * N if lengths is None else [L - (7 * i) % min(60, L // 2) for i in range(N)]
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lens, 5200 + N, [(5, 14), (22, 30)])
    beta = d.trans_pos.var_sched.betas[37].expand([N]).contiguous()
    arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
    big = hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=hip.pair_bias_cache(arr, 6, pf))
    for lo in (0, N - 3):
        sl = slice(lo, lo + 3)
        c = lambda a: a[sl].contiguous()
        small = hip.eps_net_forward(ew, c(v), c(p), c(s), c(rf), c(pf), c(beta), c(gen), c(mres), d.abdock, d.num_bins, False,
                                    pair_bias_cache=hip.pair_bias_cache(arr, 6, c(pf)))
        for k in ('R_next', 'eps_pos', 'c'):
            assert torch.isfinite(big[k]).all() and torch.equal(big[k][sl], small[k]), (k, lo)


@pytest.mark.parametrize('N,L,lengths', [(32, 256, None), (3, 100, None), (5, 250, 'ragged'), (8, 48, 'ragged'), (4, 40, None), (2, 33, None), (16, 64, 'ragged'), (9, 130, None)])
def test_core32_is_bit_identical(N, L, lengths, monkeypatch):
    """The 32-row form of the cached core (csrc/ipa_core.hip: ipa_core32_kernel, 8 waves x 256 registers, q' in registers) does the
    per-row arithmetic of the 16-row kernels in the same order: the whole EpsilonNet output must be bit-identical with and without it --
    lengths that leave the second row tile of the last block empty or partial, chunk counts not divisible by 3, ragged masks."""
    from ab_opt_amd import hip
    monkeypatch.setenv('ABOPT_CORE_NO_SPLIT', '1')
    monkeypatch.setenv('ABOPT_FUSE_TAIL', '0')                  # the 32-row core on its own (its fused form: test_fused_block_is_bit_identical)
    d = standalone_abdesign_dpm(100, 2).to(DEV)
    lens = [L] * N if lengths is None else [L - (7 * i) % min(60, L // 2) for i in range(N)]
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lens, 6100 + N, [(5, 14), (22, 30)])
    beta = d.trans_pos.var_sched.betas[37].expand([N]).contiguous()
    arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
    pbc = hip.pair_bias_cache(arr, 6, pf)
    monkeypatch.setenv('ABOPT_CORE32', '0')
    ref = hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc)
    monkeypatch.setenv('ABOPT_CORE32', '1')
    got = hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc)
    for k in ('R_next', 'eps_pos', 'c'):
        assert torch.isfinite(got[k]).all() and torch.equal(got[k], ref[k]), k
    monkeypatch.delenv('ABOPT_CORE32')                          # the library's own choice (the 32-row kernel at the bench shape): same bits either way
    auto = hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc)
    assert all(torch.equal(auto[k], ref[k]) for k in ('R_next', 'eps_pos', 'c'))


@pytest.mark.gpu
@pytest.mark.parametrize('N,L,lengths', [(32, 256, None), (32, 256, 'ragged'), (3, 100, None), (5, 250, 'ragged'), (8, 48, 'ragged'), (4, 40, None), (2, 33, None), (9, 130, None),
                                         (24, 256, 'ragged')])
def test_fused_block_is_bit_identical(N, L, lengths, monkeypatch):
    """Round 4: core + tail of a GABlock as ONE kernel (csrc/ipa_core.hip: ipa_core32_kernel<true> -- out_transform, LayerNorm,
    mlp_transition, LayerNorm of ga.py:174-177 as the epilogue of the 32-row core, `feat` never written).  Its sums run in the order of
    the stand-alone tail kernel (tail_common.h), so the whole EpsilonNet output is bit-identical to the two-launch form and to the 16-row
    kernels: lengths that leave the last block's second row tile empty or partial, ragged masks (masked query rows: u = 0), the
    library's own choice at the bench shape."""
    from ab_opt_amd import hip
    monkeypatch.setenv('ABOPT_CORE_NO_SPLIT', '1')
    d = standalone_abdesign_dpm(100, 2).to(DEV)
    lens = [L] * N if lengths is None else [L - (7 * i) % min(60, L // 2) for i in range(N)]
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lens, 7100 + N, [(5, 14), (22, 30)])
    beta = d.trans_pos.var_sched.betas[37].expand([N]).contiguous()
    arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
    pbc = hip.pair_bias_cache(arr, 6, pf)
    run = lambda: hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc)
    monkeypatch.setenv('ABOPT_CORE32', '0')
    ref16 = run()
    monkeypatch.setenv('ABOPT_CORE32', '1')
    monkeypatch.setenv('ABOPT_FUSE_TAIL', '0')
    ref32 = run()
    monkeypatch.setenv('ABOPT_FUSE_TAIL', '1')
    got = run()
    for k in ('R_next', 'eps_pos', 'c'):
        assert torch.isfinite(got[k]).all() and torch.equal(got[k], ref32[k]) and torch.equal(got[k], ref16[k]), k
    monkeypatch.delenv('ABOPT_CORE32')
    monkeypatch.delenv('ABOPT_FUSE_TAIL')
    auto = run()
    assert all(torch.equal(auto[k], ref16[k]) for k in ('R_next', 'eps_pos', 'c'))


def test_residue_features_native_vs_torch_statement():
    """ResidueEmbedding on the training path: features from abopt_residue_features (+ bucketed row sums for the embedding tables) against
    the torch statement of residue.py:33-88 in the same module -- output and every parameter gradient, both flavours (hotspot table),
    ragged lengths, two chains, structure / sequence masks."""
    import plain_statement
    for flavour in ('abdock', 'abdesign'):
        m = build_model(10, 3, flavour=flavour, device=DEV).train()
        re_ = m.residue_embed
        b = {k: dev(v) for k, v in synth.make_batch(3, synth.LAYOUT_128, seed=31, lengths=[128, 77, 50]).items()}
        b['chain_nb'][:, 40:] = 1
        smask = b['mask'] & ~b['generate_flag']
        args = (b['aa'], b['res_nb'], b['chain_nb'], b['pos_heavyatom'], b['mask_heavyatom'], b['fragment_type'])
        kw = dict(structure_mask=smask, sequence_mask=smask)
        if re_.hotspot_embed is not None:
            kw['hotspot'] = (b['fragment_type'] == 3).long()
        w = dev(synth.hash_tensor((3, 128, 128), 9, scale=1.0))
        out = {}
        for native in (True, False):
            re_.zero_grad(set_to_none=True)
            with torch.enable_grad():
                y = re_(*args, **kw) if native else plain_statement.residue_embedding(re_, *args, **kw)
                (y * w).sum().backward()
            out[native] = (y.detach(), {n: p.grad.clone() for n, p in re_.named_parameters() if p.grad is not None})
        re_.zero_grad(set_to_none=True)
        (ya, ga), (yb, gb) = out[True], out[False]
        assert (ya - yb).abs().max().item() <= 2e-5 * max(1.0, yb.abs().max().item())
        assert set(ga) == set(gb)
        for n in ga:
            assert (ga[n] - gb[n]).abs().max().item() <= 2e-4 * max(1e-6, gb[n].abs().max().item()), (flavour, n)


def test_dpm_losses_autograd_function_vs_torch_statement():
    """training.DpmLosses (abopt_dpm_losses: rotation cosine, position MSE and sequence KL losses of dpm_full.py:199-231 with their
    gradients in one launch) against the torch statement of the same lines (cosine_embedding_loss, mse_loss, kl_div on the posteriors):
    sums and the three input gradients; sequence states outside 0..19 and masked rows included."""
    from ab_opt_amd import training, hip
    import plain_statement
    g = torch.Generator().manual_seed(11)
    N, L = 4, 100
    R0 = hip.so3_exp(dev(torch.randn(N, L, 3, generator=g)))
    Rp = (hip.so3_exp(dev(torch.randn(N, L, 3, generator=g))) + dev(torch.randn(N, L, 3, 3, generator=g) * 0.1)).requires_grad_()
    pp = dev(torch.randn(N, L, 3, generator=g)).requires_grad_()
    pt = dev(torch.randn(N, L, 3, generator=g))
    cd = torch.softmax(dev(torch.randn(N, L, 20, generator=g) * 2), -1).requires_grad_()
    st, s0 = dev(torch.randint(0, 22, (N, L), generator=g)), dev(torch.randint(0, 20, (N, L), generator=g))
    ab = dev(torch.tensor([0.999, 0.7, 0.2, 0.011]))
    gen = dev(torch.rand(N, L, generator=g) < 0.6)
    w = dev(torch.tensor([0.7, -1.3, 2.1]))
    sums = training.DpmLosses.apply(Rp, R0, pp, pt, cd, st, s0, ab, gen)
    (sums * w).sum().backward()
    got = (sums.detach(), Rp.grad.clone(), pp.grad.clone(), cd.grad.clone())
    Rp.grad = pp.grad = cd.grad = None
    genf = gen.float()
    cp, ct = Rp.transpose(-2, -1).reshape(-1, 3), R0.transpose(-2, -1).reshape(-1, 3)
    lr = F.cosine_embedding_loss(cp, ct, torch.ones(cp.shape[0], dtype=torch.long, device=DEV), reduction='none').reshape(N, L, 3).sum(-1)
    t_idx = torch.arange(N, device=DEV)
    post_true = plain_statement.posterior(ab, st, s0, t_idx)
    log_pred = torch.log(plain_statement.posterior(ab, st, cd, t_idx) + 1e-8)
    kl = F.kl_div(input=log_pred, target=post_true, reduction='none', log_target=False).sum(-1)
    ref = torch.stack([(lr * genf).sum(), (F.mse_loss(pp, pt, reduction='none').sum(-1) * genf).sum(), (kl * genf).sum()])
    (ref * w).sum().backward()
    for a, b, name in zip(got, (ref.detach(), Rp.grad, pp.grad, cd.grad), ('sums', 'd R_pred', 'd p_pred', 'd c_den')):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item()), (name, (a - b).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize('pred_x0', [True, False])
def test_abdock_losses_and_layer_norm_autograd_functions_vs_torch_statement(pred_x0):
    """training.AbdockLosses (abopt_abdock_losses: the prmsd cross entropy on the binned, detached RMSD and the dist loss of
    dpm_full.py:180-198,369-378, with their gradients in one launch) and training.NativeLayerNorm (the prmsd head's LayerNorm,
    layers.py:146-155) against the torch statement of the same lines: values and input gradients; samples whose first residue is / is not
    generated (the m0 mask), ragged residue masks, both objectives."""
    from ab_opt_amd import training
    g = torch.Generator().manual_seed(17)
    N, L, nb, scale = 5, 70, 40, 10.0
    logits = dev(torch.randn(N, nb, generator=g)).requires_grad_()
    p_pred = dev(torch.randn(N, L, 3, generator=g)).requires_grad_()
    p0n = dev(torch.randn(N, L, 3, generator=g))
    gen = dev(torch.rand(N, L, generator=g) < 0.3)
    gen[:, 0] = dev(torch.tensor([True, False, True, True, False]))
    mres = dev(synth.mask_from_lengths([70, 66, 70, 41, 70], L))
    gen &= mres
    ca, cb = dev(torch.rand(N, generator=g) + 1.0), dev(torch.rand(N, generator=g))
    off = dev(torch.linspace(0.5, 19.5, nb))
    w = dev(torch.tensor([0.7, -1.3]))
    prmsd, dist = training.AbdockLosses.apply(logits, p_pred, p0n, None if pred_x0 else ca, None if pred_x0 else cb, gen, mres, off, scale, pred_x0)
    (prmsd * w[0] + dist * w[1]).backward()
    got = (prmsd.detach(), dist.detach(), logits.grad.clone(), p_pred.grad.clone())
    logits.grad = p_pred.grad = None
    pred_p0 = p_pred if pred_x0 else torch.where(gen[..., None], ca.view(-1, 1, 1) * p0n - cb.view(-1, 1, 1) * p_pred, p0n)
    pa, pb = pred_p0 * scale * gen.unsqueeze(-1), p0n * scale * gen.unsqueeze(-1)
    rmsd = torch.sqrt(((pa - pb) ** 2).sum(-1).sum(-1) / gen.sum(-1)).detach()
    diff = torch.abs(rmsd.unsqueeze(-1) - off)
    onehot = torch.zeros_like(diff).scatter_(-1, torch.argmin(diff, -1, keepdim=True), 1.0)
    err = -(onehot * F.log_softmax(logits, dim=-1)).sum(-1)
    m0 = gen[:, 0]
    ref_prmsd = (err * m0).sum() / (m0.sum() + 1e-10)
    if pred_x0:
        dp, dt = torch.cdist(p_pred, p_pred, compute_mode='donot_use_mm_for_euclid_dist'), torch.cdist(p0n, p0n, compute_mode='donot_use_mm_for_euclid_dist')
        sel = gen[:, :, None].expand_as(dp) & (mres[:, :, None] & mres[:, None, :])
        ref_dist = F.smooth_l1_loss(torch.masked_select(dp, sel), torch.masked_select(dt, sel), reduction='none').mean()
    else:
        ref_dist = p_pred.sum() * 0.0
    (ref_prmsd * w[0] + ref_dist * w[1]).backward()
    for a, b, name in zip(got, (ref_prmsd.detach(), ref_dist.detach(), logits.grad, p_pred.grad), ('prmsd', 'dist', 'd logits', 'd p_pred')):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), (name, (a - b).abs().max().item())
    # LayerNorm of the prmsd head (131 columns)
    x = dev(torch.randn(3, 37, 131, generator=g) * 3).requires_grad_()
    gm, bt = dev(torch.randn(131, generator=g)).requires_grad_(), dev(torch.randn(131, generator=g)).requires_grad_()
    wy = dev(torch.randn(3, 37, 131, generator=g))
    y = training.NativeLayerNorm.apply(x, gm, bt, 1e-10)
    (y * wy).sum().backward()
    gotl = (y.detach(), x.grad.clone(), gm.grad.clone(), bt.grad.clone())
    x.grad = gm.grad = bt.grad = None
    yr = F.layer_norm(x, (131,), gm, bt, eps=1e-10)
    (yr * wy).sum().backward()
    for a, b, name in zip(gotl, (yr.detach(), x.grad, gm.grad, bt.grad), ('y', 'dx', 'd gamma', 'd beta')):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), (name, (a - b).abs().max().item())


def test_heads_epilogue_autograd_function_vs_torch_statement():
    """training.HeadsEpilogue (abopt_heads_epilogue_forward / _backward: eps_pos = gen ? R eps_crd : 0, R_next = R U(eps_rot),
    dpm_full.py:95-101 under autograd) against the torch statement of the same lines: values and both input gradients, small and
    large quaternion vectors, masked rows."""
    from ab_opt_amd import training, hip
    import plain_statement
    g = torch.Generator().manual_seed(3)
    N, L = 5, 77
    R = hip.so3_exp(dev(torch.randn(N, L, 3, generator=g) * 1.5))
    crd = dev(torch.randn(N, L, 3, generator=g)).requires_grad_()
    rot = dev(torch.randn(N, L, 3, generator=g) * torch.tensor([0.01, 1.0, 8.0])[torch.randint(0, 3, (N, L, 1), generator=g)]).requires_grad_()
    gen = dev(torch.rand(N, L, generator=g) < 0.7)
    wR, wp = dev(torch.randn(N, L, 3, 3, generator=g)), dev(torch.randn(N, L, 3, generator=g))
    Rn, ep = training.HeadsEpilogue.apply(R, crd, rot, gen)
    ((Rn * wR).sum() + (ep * wp).sum()).backward()
    got = (Rn.detach(), ep.detach(), crd.grad.clone(), rot.grad.clone())
    crd.grad = rot.grad = None
    gen3 = gen[:, :, None].expand(N, L, 3)
    ep2 = torch.where(gen3, torch.einsum('nlab,nlb->nla', R, crd), torch.zeros_like(crd))
    Rn2 = R @ plain_statement.quat1ijk_to_rot(rot)
    ((Rn2 * wR).sum() + (ep2 * wp).sum()).backward()
    for a, b, name in zip(got, (Rn2.detach(), ep2.detach(), crd.grad, rot.grad), ('R_next', 'eps_pos', 'd eps_crd', 'd eps_rot')):
        assert (a - b).abs().max().item() <= 3e-6 * max(1.0, b.abs().max().item()), name
    assert (got[2][~gen] == 0).all()


def test_pair_embed_backward_recomputes_T_bit_identically():
    """abopt_pair_embed_backward without the saved T = dg / d softplus(coef) (pair.py:62-73 under autograd): the kernel recomputes it from
    the atoms with the forward's arithmetic -- dys and dsoftplus must equal, bit for bit, the run that reads the forward's dump
    (ragged lengths, masked atoms, two chains, L not a multiple of the 64-pair strips)."""
    from ab_opt_amd import hip
    m = synth.fresh_model(10, 3, device=DEV)                                   # a private model: the cached one is shared by the whole session
    with torch.no_grad():
        m.pair_embed.aapair_to_distcoef.weight.copy_(dev(synth.hash_tensor(tuple(m.pair_embed.aapair_to_distcoef.weight.shape), 23, scale=2.0)))
    b = {k: dev(v) for k, v in synth.make_batch(3, synth.LAYOUT_128, seed=12, lengths=[75, 128, 40]).items()}
    b['chain_nb'][:, 30:] = 1
    inp, keep = hip.encode_inputs(b['aa'], b['res_nb'], b['chain_nb'], b['pos_heavyatom'], b['mask_heavyatom'], 15, structure_mask=b['mask'] & ~b['generate_flag'])
    w = m.pair_embed._hip_weights()
    out, acts, G, T = hip.pair_embed_forward(inp, w, save_activations=True, save_T=True)
    out2, acts2, G2, T2 = hip.pair_embed_forward(inp, w, save_activations=True)
    assert T2 is None and torch.equal(out, out2) and torch.equal(G, G2) and torch.equal(acts, acts2)
    dout = dev(synth.hash_tensor(tuple(out.shape), 5, scale=1.0))
    dys_a, ds_a = hip.pair_embed_backward(inp, w, dout, acts, T)
    dys_b, ds_b, db = hip.pair_embed_backward(inp, w, dout, acts, colsum=True)
    assert torch.isfinite(ds_b).all() and ds_a.abs().max().item() > 0
    assert torch.equal(dys_a, dys_b) and torch.equal(ds_a, ds_b)
    # the five bias gradients from the kernel's per-wave partial sums = column sums of dys (fp64 reference; padded rows of the strips excluded)
    ref = dys_a.double().reshape(-1, hip.PAIR_DY).sum(0)
    assert (db.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize('weight_decay,max_norm', [(0.0, None), (0.0, 0.5), (0.01, 100.0)])
def test_fused_adam_vs_torch_adam(weight_decay, max_norm):
    """training.FusedAdam (csrc/optim.hip: clip_grad_norm_ + Adam for the whole parameter list in a handful of launches) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.Adam, the pair the reference's training loop runs (A/train.py:116-117,
    A/diffab/utils/train.py:28-36): six steps on 40 tensors from 1 element to 300k (more tensors than one launch carries, sizes that are
    not multiples of the block), parameters / moments / returned norm equal to fp32 rounding."""
    from ab_opt_amd import training
    g = torch.Generator().manual_seed(31)
    shapes = [(1,), (3,), (128,), (127, 3), (4096,), (4097,), (300, 1000), (64, 64), (2016, 128)] + [(17 + i, 5) for i in range(31)]
    ref = [torch.nn.Parameter(torch.randn(sh, generator=g).to(DEV)) for sh in shapes]
    got = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    a = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    b = training.FusedAdam(got, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    for it in range(6):
        grads = [torch.randn(sh, generator=g).to(DEV) * (0.1 + it) for sh in shapes]
        for p, q, gr in zip(ref, got, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm) if max_norm is not None else None
        a.step()
        n_got = b.step(max_grad_norm=max_norm)
        if max_norm is not None:
            assert abs(n_got.item() - n_ref.item()) <= 2e-6 * n_ref.item()
            assert torch.equal(got[0].grad, grads[0])          # gradients are read, not rescaled in place
        else:
            assert n_got is None
        for p, q in zip(ref, got):
            assert (p - q).abs().max().item() <= 2e-6 * max(1.0, p.abs().max().item()), it
    for p, q in zip(ref, got):
        sa, sb = a.state[p], b.state[q]
        if 'exp_avg' in sb:
            assert (sa['exp_avg'] - sb['exp_avg']).abs().max().item() <= 1e-6 * max(1.0, sa['exp_avg'].abs().max().item())
            assert (sa['exp_avg_sq'] - sb['exp_avg_sq']).abs().max().item() <= 1e-6 * max(1.0, sa['exp_avg_sq'].abs().max().item())
    assert int(b.state[got[0]]['step'].item()) == 6
    import copy
    sd = copy.deepcopy(b.state_dict())                          # torch's layout (as if it had been to disk: load_state_dict shares tensors otherwise)
    cp = [torch.nn.Parameter(p.detach().clone()) for p in got]
    c = training.FusedAdam(cp, lr=3e-3, weight_decay=weight_decay)
    c.load_state_dict(sd)
    grads = [torch.randn(sh, generator=g).to(DEV) for sh in shapes]
    for p, q, gr in zip(got, cp, grads):
        p.grad, q.grad = gr.clone(), gr.clone()
    b.step(max_grad_norm=max_norm); c.step(max_grad_norm=max_norm)                 # the reloaded optimizer continues at step 7, bit for bit
    assert all(torch.equal(p, q) for p, q in zip(got, cp)) and int(c.state[cp[0]]['step'].item()) == 7


@pytest.mark.gpu
def test_fused_adam_hyper_parameters_follow_the_scheduler_under_graph_replay():
    """A captured step must not freeze lr (round-3 advisor finding): FusedAdam's kernels read {lr, betas, eps, weight_decay, clip norm}
    from a device buffer that is refreshed from param_groups outside the capture.  Three replays with MultiStepLR-style changes of lr
    against the same steps taken eagerly with torch.optim.Adam; and the parameters' _version moves (packed-weight caches key on it)."""
    from ab_opt_amd import training
    g = torch.Generator().manual_seed(3)
    shapes = [(257,), (64, 33), (5,)]
    ref = [torch.nn.Parameter(torch.randn(sh, generator=g).to(DEV)) for sh in shapes]
    got = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    grads = [torch.randn(sh, generator=g).to(DEV) for sh in shapes]
    for p, q, gr in zip(ref, got, grads):
        p.grad, q.grad = gr.clone(), gr.clone()
    a = torch.optim.Adam(ref, lr=1e-2)
    b = training.FusedAdam(got, lr=1e-2)
    b.step(max_grad_norm=1.0)                                                   # eager warm-up step (allocates state)
    torch.nn.utils.clip_grad_norm_(ref, 1.0); a.step()
    for p, gr in zip(ref, grads):
        p.grad = gr.clone()
    v0 = got[0]._version
    graph = torch.cuda.CUDAGraph()
    snap = [q.detach().clone() for q in got]
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        b.step(max_grad_norm=1.0)
    for q, s0 in zip(got, snap):                                                # a capture records, it does not execute
        assert torch.equal(q, s0)
    for lr, clip in ((1e-2, 1.0), (1e-3, 1.0), (5e-4, 0.25)):
        for grp in b.param_groups:
            grp['lr'] = lr
        for grp in a.param_groups:
            grp['lr'] = lr
        b.refresh_hyper(clip)
        graph.replay()
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, clip); a.step()
        for p, q in zip(ref, got):
            assert (p - q).abs().max().item() <= 2e-6 * max(1.0, p.abs().max().item()), (lr, clip)
    assert int(b.state[got[0]]['step'].item()) == 4
    b.step(max_grad_norm=1.0)
    assert got[0]._version > v0


@pytest.mark.gpu
def test_loop_graph_cache_is_bounded_and_skips_unchanged_pair_feat():
    """graph_mode='auto' keeps at most `max_graphs` captured loops (round-3 advisor finding: one pinned pool per shape seen twice, never
    evicted) and a replay with the very same, unmodified pair_feat tensor skips the copy into the graph's static buffer -- but never
    serves stale contents: an in-place change (new _version) or another tensor object is copied."""
    m = build_model(10, 3, device=DEV)
    dpm = m.diffusion
    dpm.max_graphs = 2
    outs = {}
    for L in (32, 48, 64, 32):
        args = synth.eps_inputs(2, L, [L, L - 5], [(4, 12)], num_steps=10, t=7)
        v, p, s, rf, pf, beta, gen, mres = [dev(a) for a in args]
        for rep in range(3):
            tv = dpm.sample(v, p, s, rf, pf, gen, mres, seed=5, graph='auto')
        outs[L] = tv[0][1].clone()
        assert len(dpm._graphs) <= 2
    assert len(dpm._graphs) == 2
    g = next(reversed(dpm._graphs.values()))
    assert g._pf_src is not None and g._pf_src[0]() is pf
    ref = dpm.sample(v, p, s, rf, pf, gen, mres, seed=5, graph=False)[0][1]
    assert torch.equal(ref, outs[32])
    pf.mul_(0.5)                                                                # same object, new version: must be copied again
    a = dpm.sample(v, p, s, rf, pf, gen, mres, seed=5, graph='auto')[0][1]
    b = dpm.sample(v, p, s, rf, pf, gen, mres, seed=5, graph=False)[0][1]
    assert torch.equal(a, b) and not torch.equal(a, ref)
    pf2 = pf * 2.0                                                              # another object
    a = dpm.sample(v, p, s, rf, pf2, gen, mres, seed=5, graph='auto')[0][1]
    assert torch.equal(a, ref)
    dpm.clear_graphs()
    assert len(dpm._graphs) == 0


def test_abopt_gemm_views_splitk_bias_relu_vs_fp64():
    """abopt_gemm (the training path's strided-batched fp32 product C = alpha a b^T): plain, transposed VIEWS read in place, batch
    broadcasting, split-K (few output tiles, long K), and the bias / ReLU epilogue, against fp64 matmul; ragged sizes; deterministic."""
    from ab_opt_amd import hip
    g = torch.Generator().manual_seed(5)
    rnd = lambda *sh: dev(torch.randn(*sh, generator=g))
    ok = lambda got, ref: (got.double() - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
    a, b = rnd(70, 33), rnd(45, 33)
    assert ok(hip.gemm(a, b)[0], a.double() @ b.double().t())
    at, bt = rnd(33, 70), rnd(33, 45)                                            # both operands as transposed views
    assert ok(hip.gemm(at.t(), bt.t(), alpha=0.5)[0], 0.5 * (at.double().t() @ bt.double()))
    A3, B2 = rnd(6, 40, 129), rnd(17, 129)                                      # batch broadcasting of a 2-D operand
    assert ok(hip.gemm(A3, B2), A3.double() @ B2.double().t())
    x, dy = rnd(5000, 96), rnd(5000, 64)                                        # weight gradient: 2 output tiles over K = 5000 rows -> split-K
    dw = hip.gemm(dy.t(), x.t())[0]
    assert ok(dw, dy.double().t() @ x.double()) and torch.equal(dw, hip.gemm(dy.t(), x.t())[0])
    w, bias = rnd(200, 96), rnd(200)                                            # y = relu(x W^T + b) in the epilogue
    assert ok(hip.gemm(x, w, bias=bias)[0], x.double() @ w.double().t() + bias.double())
    assert ok(hip.gemm(x, w, bias=bias, relu=True)[0], (x.double() @ w.double().t() + bias.double()).clamp_min(0))
    assert ok(hip.gemm(x, w, relu=True)[0], (x.double() @ w.double().t()).clamp_min(0))
    col = rnd(300, 320)[:, 64:128]                                              # a column slice (row stride 320) as operand
    assert ok(hip.gemm(col, w[:10, :64])[0], col.double() @ w[:10, :64].double().t())
    # C as a column slice of a wider matrix (ldc > N) with a split-K shape: the columns outside the slice, and the memory past the last
    # row, stay untouched (round-3 advisor finding: the slab sum used to write all M * ldc elements)
    wide = torch.full((64 + 1, 200), 7.0, device=DEV)
    hip.gemm(dy.t(), x.t(), out=wide[:64, 50:146])
    assert ok(wide[:64, 50:146], dy.double().t() @ x.double())
    assert (wide[:64, :50] == 7).all() and (wide[:64, 146:] == 7).all() and (wide[64] == 7).all()
    cb = torch.full((3, 20, 40), 7.0, device=DEV)                                # batched, gaps between the batches of C
    a3, b3 = rnd(3, 20, 2048), rnd(3, 24, 2048)
    hip.gemm(a3, b3, out=cb[:, :, 8:32])
    assert ok(cb[:, :, 8:32], a3.double() @ b3.double().transpose(1, 2)) and (cb[:, :, :8] == 7).all() and (cb[:, :, 32:] == 7).all()


def test_bucket_colsum_vs_index_add():
    """abopt_bucket_colsum (the relative-position table's gradient on the training path, D/modules/encoders/pair.py:46-53 under autograd):
    rows of a strided column slice summed by bucket, rows with a negative index skipped, against torch.index_add_ in fp64; deterministic."""
    from ab_opt_amd import hip
    g = torch.Generator().manual_seed(77)
    for rows, cols, nb, ld in ((70000, 64, 65, 320), (1000, 128, 96, 128), (5, 64, 3, 64), (4097, 20, 7, 33)):
        y = torch.randn(rows, ld, generator=g).to(DEV)
        x = y[:, ld - cols:]                                     # a column slice, read in place
        idx = torch.randint(-1, nb, (rows,), generator=g).to(torch.int32).to(DEV)
        out = hip.bucket_colsum(x, idx, nb)
        keep = idx >= 0
        ref = torch.zeros(nb, cols, dtype=torch.float64, device=DEV).index_add_(0, idx[keep].long(), x[keep].double())
        assert (out.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), (rows, cols, nb)
        assert torch.equal(out, hip.bucket_colsum(x, idx, nb))
    with pytest.raises(RuntimeError):
        hip.bucket_colsum(torch.zeros(4, 64, device=DEV), torch.zeros(4, dtype=torch.int32, device=DEV), 97)
    # the segmented form + the plain one = the sum by PAIR of residue types (a of i, b of j) of the amino-acid-pair tables' gradients
    # (pair.py:46-53,66 under autograd): against the one-hot contraction autograd would run, in fp64; strided operand read in place
    for N, L, cols, ld in ((3, 37, 64, 320), (2, 64, 80, 80), (1, 5, 240, 240), (35000, 2, 64, 64)):       # (the last: 70 000 segments, more than one grid holds -- ADVICE r05)
        nt = 22
        y = torch.randn(N * L * L, ld, generator=g).to(DEV)
        x = y[:, ld - cols:]
        aa = torch.randint(0, nt, (N, L), generator=g).to(DEV)
        aa32 = aa.to(torch.int32)
        s1 = hip.segment_bucket_colsum(x, N * L, aa32, L, nt)
        assert s1.shape == (N * L, nt, cols)
        got = hip.bucket_colsum(s1.view(N * L, -1), aa32.view(-1), nt).view(nt, nt, cols)
        oh = torch.nn.functional.one_hot(aa, nt).double()
        ref = torch.einsum('nia,njb,nijc->abc', oh, oh, x.double().view(N, L, L, cols))
        assert (got.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), (N, L, cols)
        assert torch.equal(s1, hip.segment_bucket_colsum(x, N * L, aa32, L, nt))


# ------------------------------------------------------------------------------------------ round 3: graph replay, bench N>1 path, edge cases
def test_graph_replay_is_bit_identical_to_eager_launches():
    """FullDPM._run(graph=True): the captured hipGraph of the sampling loop must reproduce the eager loop bit for bit, for the seed
    it was captured with AND for later seeds (the Philox position is read from device memory at replay), for new inputs copied into
    its static buffers, and through model.sample()'s automatic mode (eager first call, captured from the second)."""
    m = build_model(10, 3, device=DEV)
    dpm = m.diffusion
    dpm._graphs.clear(); dpm._graph_seen.clear()
    batch = {k: dev(v) for k, v in synth.make_batch(4, synth.LAYOUT_128, seed=9, lengths=[128, 117, 128, 90]).items()}
    res_feat, pair_feat, R0, p0 = m.encode(dict(batch), True, True)
    from ab_opt_amd import hip
    v0 = hip.so3_log(R0)
    gen, mres = batch['generate_flag'], batch['mask']

    def run(seed, graph, pf=pair_feat, off=0):
        st = hip.sample_init(v0, p0, batch['aa'], gen, None, seed, off, 10.0, [0.0, 0.0, 0.0], True, True)
        out = dpm._run(st, 10, res_feat, pf, gen, mres, True, True, True, None, seed, off, False, graph=graph)
        return [a.clone() for a in out[:3]]
    for seed, off in ((5, 0), (6, 0), (5, 4096)):
        ref = run(seed, False, off=off)
        got = run(seed, True, off=off)
        assert all(torch.equal(a, b) for a, b in zip(ref, got)), (seed, off)
    assert len(dpm._graphs) == 1 and dpm.last_run_info['graph']
    pf2 = pair_feat.flip(0).contiguous()                                  # other inputs through the same graph
    assert all(torch.equal(a, b) for a, b in zip(run(7, False, pf2), run(7, True, pf2)))
    assert len(dpm._graphs) == 1
    # automatic mode through the model boundary
    dpm._graphs.clear(); dpm._graph_seen.clear()
    opt = {'sample_structure': True, 'sample_sequence': True, 'contig': '', 'seed': 31}
    eager = m.sample(dict(batch), dict(opt, graph=False))
    first = m.sample(dict(batch), dict(opt))              # auto: this call is still eager
    assert len(dpm._graphs) == 0
    second = m.sample(dict(batch), dict(opt))             # captured + replayed
    assert len(dpm._graphs) == 1
    third = m.sample(dict(batch), dict(opt, seed=32))
    for t in eager:
        for a, b, c in zip(eager[t], first[t], second[t]):
            assert torch.equal(a.cpu(), b.cpu()) and torch.equal(a.cpu(), c.cpu()), t
    assert not torch.equal(third[0][1], second[0][1]) and torch.equal(second[0][1], eager[0][1])     # slot 0 owns its storage
    dpm._graphs.clear(); dpm._graph_seen.clear()


def test_bench_two_ranks_on_one_gpu():
    """The N>1 branch of bench.py (torch.distributed launch, weak scaling, candidate all_gather inside the timed region, max over
    ranks) executed with two ranks sharing cuda:0: the line must carry n_gpus=2 and a whole-job rate of the order of the 1-GPU rate."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    common = ['--steps', '6', '--warmup', '2', '--repeats', '3', '--batch', '8', '--no-cpu-baseline', '--no-secondary']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + common, capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + common,
                         capture_output=True, text=True, env=env, timeout=1200)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, two.stdout                                    # rank 0 prints ONE line
    l2 = json.loads(lines[0])
    assert l2['n_gpus'] == 2 and l2['steps'] == 6 and l2['repeats'] == 3 and l2['scaling'] == 'weak'
    assert l2['config']['backend'] in ('gloo', 'nccl') and l2['config']['ranks_per_device'] == 2
    rs = l2['config']['ranks']                                          # every rank reports its device and its own clock through the process group
    assert [r['rank'] for r in rs] == [0, 1] and all(r['device'] == 0 and 0 < r['ms_per_step'] <= l2['ms_per_step_max'] * 1.001 for r in rs)
    assert l2['roofline']['launches'] > 0 and l2['roofline']['frac'] > 0
    # two ranks time-share one GPU: the whole-job rate stays within a factor of the single-rank rate (never 2x, never collapsed)
    assert 0.3 * l1['value'] < l2['value'] < 1.6 * l1['value'], (l1['value'], l2['value'])


@pytest.mark.gpu
def test_bench_plain_command_self_launches_two_ranks():
    """The driver's N = 1 command shape with N = 2 -- `python bench.py --gpus 2 ...`, no launcher, no WORLD_SIZE -- must start its own two
    ranks (bench.self_launch: torch.distributed.run on 127.0.0.1) and print ONE line that carries both of them."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--repeats', '2', '--batch', '8',
                        '--no-cpu-baseline', '--no-secondary'], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 4 and line['value'] > 0
    assert [x['rank'] for x in line['config']['ranks']] == [0, 1]


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two devices: one rank per GPU over RCCL')
def test_two_devices_rccl_bench_sampling_and_graphed_ddp(tmp_path):
    """The first multi-GPU run must be boring.  With >= 2 devices: (i) `bench.py --gpus 2` as the driver launches it --
    backend nccl, one rank per device, both ranks in the line; (ii) sharded sampling and the by-complex test-set driver over RCCL equal
    the single-process results bit for bit; (iii) DDP(nccl) + FusedAdam eager and captured (GraphedTrainStep) keeps the replicas identical."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import mp_workers
    from conftest import ROOT
    from ab_opt_amd import sampler
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--repeats', '3'],
                         capture_output=True, text=True, env=env, timeout=1800)
    assert two.returncode == 0, two.stderr[-3000:]
    line = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['backend'] == 'nccl' and line['config']['ranks_per_device'] == 1
    assert sorted(r['device'] for r in line['config']['ranks']) == [0, 1] and line['roofline']['frac'] > 0.2
    m = build_model(10, 3, device=DEV)
    b = {k: dev(v) for k, v in synth.make_batch(5, synth.LAYOUT_128, seed=11, replicate=True).items()}
    traj, _, top, cand = sampler.sample_sharded(m, b, dict(sample_structure=True, sample_sequence=True, contig=''), k=2, seed=42)
    cx = [{k: dev(v) for k, v in synth.make_batch(1, synth.LAYOUT_128, seed=100 + c).items()} for c in range(3)]
    ref = sampler.design_testset_sharded(m, cx, 4, k=2, seed=7, complexes_per_launch=1)
    _spawn2(mp_workers.two_device_worker, tmp_path)
    got = [torch.load(tmp_path / f'twodev_{r}.pt', weights_only=False) for r in range(2)]
    assert [g_['device'] for g_ in got] == [0, 1] and all(g_['backend'] == 'nccl' for g_ in got)
    for g_ in got:
        assert torch.equal(g_['cand'], cand.cpu()) and torch.equal(g_['top'], top.cpu())
        for x, r_ in zip(g_['testset'], ref):
            assert torch.equal(x['ca'], r_['ca']) and torch.equal(x['top'], r_['top'])
        assert g_['graph_error'] is None, g_['graph_error']
        assert len(g_['losses']) == 4 and all(math.isfinite(x) for x in g_['losses'])
    assert torch.equal(torch.cat([got[0]['p0'], got[1]['p0']]), traj[0][1].cpu())
    for n, p0 in got[0]['params'].items():
        assert torch.equal(p0, got[1]['params'][n]), n                     # replicas in lock step after 2 eager + 2 replayed steps


@pytest.mark.gpu
def test_eps_net_shape_fuzz_vs_oracle():
    """Twelve pseudo-random shapes through the sampler's launch path -- batch size, length (also not a multiple of 16 or 32), ragged masks, flavour,
    pair features per sample / one complex / groups of four, with and without the pair-bias cache -- against the oracle on the first and the last
    sample: whichever form of the core the library picks for a shape (one-block, key-split, persistent, 32-row, fused with the tail, grouped)
    must stay within the per-step tolerance.  (tools/r05/fuzz_eps.py is the long form: 96 cases in round 5, worst R_next 3.6e-6, profiles/r05_fuzz_eps_seed7.txt.)"""
    import random
    from ab_opt_amd import hip
    from oracle import dpm
    rnd = random.Random(20)
    models = {'abdesign': (standalone_abdesign_dpm(100, 2), standalone_abdesign_dpm(100, 2).to(DEV)),
              'abdock': (build_model(100, 2).diffusion, build_model(100, 2, device=DEV).diffusion)}
    seen = set()
    for case in range(12):
        flavour = rnd.choice(['abdesign', 'abdock'])
        L = rnd.choice([rnd.randint(1, 40), rnd.randint(41, 130), rnd.randint(131, 280), 256, 48, 200])
        group = rnd.choice([0, 0, 1, 4])
        N = rnd.randint(1, max(1, min(66, 30000 // (L * L // 64 + 1))))
        if group > 1:
            N = max(group, N // group * group)
        lengths = [max(1, L - rnd.randint(0, L // 3)) if rnd.random() < 0.5 else L for _ in range(N)]
        d_cpu, d = models[flavour]
        a, b = sorted(rnd.sample(range(L + 1), 2))
        v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lengths, 6000 + case, [(a, b)])
        Nc = N if group == 0 else (1 if group == 1 else N // group)
        if group > 1:
            mres = mres[::group].repeat_interleave(group, 0).contiguous()
            gen = gen & mres
        pfc = pf[:Nc].contiguous()
        t = rnd.choice([100, 63, 21, 2])
        beta = d.trans_pos.var_sched.betas[t].expand([N]).contiguous()
        use_cache = rnd.random() < 0.8 or group != 0
        pbc = hip.pair_bias_cache(d.eps_net.encoder.packed_array(), 6, pfc) if use_cache else None
        net = hip.eps_net_forward(d.eps_net.packed(), v, p, s, rf, pfc, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc, pair_feat_shared=group)
        sd = {k: x.cpu() for k, x in d_cpu.state_dict().items()}
        inv = d_cpu.trans_rot.angular_distrib_inv
        den = dpm.Denoiser(sd, num_steps=100, variant=flavour, obj='pred_x0', mode='mm', pre='', tables=(None, dict(stddevs=inv.stddevs, approx_flag=inv.approx_flag, X=inv.X, Y=None)))
        for n in sorted({0, N - 1}):
            c = lambda x_: x_[n:n + 1].cpu()
            cpf = pfc[(n if group == 0 else (0 if group == 1 else n // group))][None].cpu()
            ref = den._eps(c(v), c(p), c(s), c(rf), cpf, c(beta), c(gen), c(mres), False)
            what = (case, flavour, N, L, group, use_cache, n)
            assert max_abs(c(net['R_next']), ref[1]) < 3e-5 and max_abs(c(net['eps_pos']), ref[2]) < 3e-5 and max_abs(c(net['c']), ref[3]) < 1e-5, what
            if d.abdock:
                assert max_abs(c(net['prmsd_logits']), ref[4]) < 3e-5, what
        assert all(torch.isfinite(x).all() for x in net.values() if x is not None)
        seen.add((flavour, group > 0, L >= 192))
    assert len(seen) >= 5                                        # the draw covered both flavours, shared and distinct pair features, short and long crops


@pytest.mark.gpu
def test_launch_spans_time_the_dominant_kernel_inside_a_graph_replay():
    """abopt_prof_spans: the 32-row launches of a captured loop carry span slots; after a reset, one replay yields exactly steps x layers
    launches whose in-kernel wall-clock spans sum to a plausible duration (within a factor of the HIP-event timing of the same steps launched
    eagerly), and a second reset + replay gives the same count again."""
    from ab_opt_amd import hip
    import bench
    dpm, state, rf, pf, gen, mres = bench.build_workload(DEV, 32, 256, 100, seed=3)
    run = lambda n, graph: dpm._run(state, 100, rf, pf, gen, mres, True, True, True, None, 7, 0, False, stop_after=n, graph=graph)
    run(2, False)
    hip.GRAPH_CAPTURE_SPANS = True
    try:
        run(3, True)
    finally:
        hip.GRAPH_CAPTURE_SPANS = False
    for _ in range(2):
        hip.prof_spans_reset()
        run(3, True)
        n, ms = hip.prof_spans()
        assert n == 3 * 6, n
    hip.prof_enable(True)
    run(3, False)
    torch.cuda.synchronize()
    n_e, ms_e = hip.prof_collect()
    hip.prof_enable(False)
    assert n_e == 18 and 0.7 * ms_e < ms < 1.4 * ms_e, (ms, ms_e)
    assert 0.05 < ms / n < 0.6                                # 50 .. 600 us per launch at this shape
    dpm.clear_graphs()


@pytest.mark.gpu
def test_bench_nccl_branch_single_rank():
    """bench.py's RCCL branch (init_process_group('nccl', device_id=...), candidate all_gather on device buffers, MAX all-reduce of the
    step time, barriers) executed with ONE rank on the one GPU of the box (ABOPT_BENCH_FORCE_DIST=1): no 8-GPU node exists for the builder,
    but the code the driver's N > 1 runs take is no longer unexecuted."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', ABOPT_BENCH_FORCE_DIST='1', ABOPT_BENCH_BACKEND='nccl', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '1', '--repeats', '2', '--batch', '8',
                        '--no-cpu-baseline', '--no-secondary'], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1])
    assert line['config']['backend'] == 'nccl' and line['n_gpus'] == 1 and line['value'] > 0


@pytest.mark.gpu
def test_two_rank_ddp_with_fused_adam_and_graph_guard(tmp_path):
    """Data-parallel training with the native optimizer: two DDP ranks on cuda:0 (gloo) take three FusedAdam steps on different samples and
    end with bit-identical parameters (the all-reduced gradients are the same on both, and so is the clipping norm); GraphedTrainStep
    refuses the DDP model there (a gloo all-reduce inside backward cannot be captured; with find_unused_parameters=True neither could an
    RCCL one) instead of capturing a step that would silently skip the gradient exchange."""
    import mp_workers
    _spawn2(mp_workers.ddp_fused_adam_worker, tmp_path)
    a, b = [torch.load(tmp_path / f'ddpadam_{r}.pt') for r in range(2)]
    assert a['refused'] and 'nccl' in a['refused'] and b['refused']
    assert all(math.isfinite(x) for x in a['losses'] + b['losses'])
    ref = build_model(10, 3).state_dict()
    moved = 0
    for n in a['params']:
        assert torch.equal(a['params'][n], b['params'][n]), n
        moved += int(not torch.equal(a['params'][n], ref[n]))
    assert moved > 150


def test_add_noise_probs_without_sequence_noise_and_dockq_empty_selection():
    """abopt_add_noise(c_noisy) with noise_sequence=0 returns c_0 = onehot(s_0) (it used to leave the buffer unwritten);
    abopt_dockq_lite marks a candidate without common CA atoms in a chain with -1 and the binding raises like calc_DockQ's asserts."""
    from ab_opt_amd import hip, sampler
    m = build_model(10, 3, device=DEV)
    N, L = 2, 40
    s0 = dev((synth.hash_tensor((N, L), 3) + 0.5).mul(22).long().clamp(0, 21))
    gen = dev(cases.gen_from_ranges(N, L, [(4, 20)]))
    t = torch.full([N], 5, dtype=torch.long, device=DEV)
    tr = m.diffusion
    out = hip.add_noise(t, tr.trans_pos.var_sched.alpha_bars, tr.trans_rot.angular_distrib_fwd, None, 3, 0,
                        dev(synth.hash_tensor((N, L, 3), 1)), dev(synth.hash_tensor((N, L, 3), 2, scale=20.0)), s0, gen, 10.0, [0.0, 0.0, 0.0],
                        noise_structure=True, noise_sequence=False, want_probs=True)
    probs = out[-1]
    want = torch.nn.functional.one_hot(s0.clamp(0, 20), 21)[..., :20].float() * (s0 < 20)[..., None]
    assert torch.equal(probs, want) and torch.equal(out[2], s0)
    pos, mask, group, models = [dev(a) for a in cases.dockq_case()]
    mm = mask[None].expand(models.shape[0], -1, -1).clone()
    mm[2, group == 2, 1] = False                      # candidate 2: no CA atom of the antigen chain present
    raw = hip.dockq_lite(models, mm, pos, mask, group, check=False)
    assert raw[2, 2] == -1 and raw[2, 3] == -1 and raw[2, 1] >= 0 and (raw[[0, 1, 3, 4, 5], 1:] >= 0).all()     # Lrms has no ligand atoms
    with pytest.raises(ValueError):
        sampler.dockq_scores(models, mm, pos, mask, group=group)


def test_dockq_superposition_corner_cases():
    """abopt_dockq_lite (Horn quaternion + Jacobi eigen-solver in fp64) on the corner-case fixture dockq_edge, whose expected values
    two independent CPU algorithms agree on: mirror-image candidates (proper rotations only), 3..4-residue interfaces, collinear CA
    atoms (iRMS / Fnat only: the roll about the line is arbitrary, so LRMS is not defined there)."""
    from ab_opt_amd import sampler
    g = load_golden('dockq_edge')
    for c in cases.dockq_edge_cases():
        out = sampler.dockq_scores(dev(c['models']), dev(c['mask']), dev(c['pos']), dev(c['mask']), group=dev(c['group']))
        want = g[c['name']]
        assert max_abs(out['fnat'].cpu(), want[:, 0]) < 1e-6, c['name']
        assert max_abs(out['irms'].cpu(), want[:, 1]) < 1e-4, (c['name'], out['irms'].cpu(), want[:, 1])
        if c['lrms_defined']:
            assert max_abs(out['Lrms'].cpu(), want[:, 2]) < 1e-4, (c['name'], out['Lrms'].cpu(), want[:, 2])
            assert max_abs(out['DockQ'].cpu(), want[:, 3]) < 1e-5, c['name']
        else:
            assert torch.isfinite(out['Lrms']).all()


def test_eps_net_reference_headline_shape_vs_oracle(monkeypatch):
    """The reference's own headline invocation (AbDock/README.md:61: dock_pdb.py -n 1000 -b 1000; configs/train/dock_single.yml:12-14
    crops the CDR plus 20 antigen residues): N = 1000 poses of ONE complex, L = 48, AbDock flavour, the context passed once
    (pair_feat (1,L,L,C), shared pair-bias cache).  Since round 5 this shape takes the fused 32-row core + tail kernel (2000 workgroups, the second
    block of every pose half empty); with ABOPT_CORE32=0 the persistent 16-row core with 3000 three-position query blocks on 256 workgroups + the
    stand-alone tail.  Per-pose masks are ragged (the kernels take one per pose).  A subset of the poses against the oracle; then the whole batch
    must be bit-identical to the 16-row form and to the same poses evaluated in small batches through the plain kernel."""
    from ab_opt_amd import hip
    from oracle import dpm
    N, L, T, t = 1000, 48, 100, 63
    d_cpu = build_model(T, 2).diffusion
    d = build_model(T, 2, device=DEV).diffusion
    sd = {k: v.cpu() for k, v in d_cpu.state_dict().items()}
    lens = [L - (5 * i) % 17 for i in range(N)]
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lens, 7100, [(8, 26)])
    rf1, pf1 = rf[:1].contiguous(), pf[:1].contiguous()                       # one complex
    beta = d.trans_pos.var_sched.betas[t].expand([N]).contiguous()
    arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
    pbc = hip.pair_bias_cache(arr, 6, pf1)
    rfN = rf1.expand(N, -1, -1).contiguous()
    net = hip.eps_net_forward(ew, v, p, s, rfN, pf1, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc, pair_feat_shared=True)
    net = {k: (a.clone() if a is not None else None) for k, a in net.items()}
    monkeypatch.setenv('ABOPT_CORE32', '0')
    net16 = hip.eps_net_forward(ew, v, p, s, rfN, pf1, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc, pair_feat_shared=True)
    monkeypatch.delenv('ABOPT_CORE32')
    for k in ('R_next', 'eps_pos', 'c', 'prmsd_logits'):
        assert torch.equal(net[k], net16[k]), k
    ids = [0, 1, 499, 998, 999]
    ix = torch.tensor(ids, device=DEV)
    c = lambda a: a[ix].cpu()
    inv = d_cpu.trans_rot.angular_distrib_inv
    den = dpm.Denoiser(sd, num_steps=T, variant='abdock', obj='pred_x0', mode='mm', pre='',
                       tables=(None, dict(stddevs=inv.stddevs, approx_flag=inv.approx_flag, X=inv.X, Y=None)))
    k5 = len(ids)
    ref = den._eps(c(v), c(p), c(s), rf1.cpu().expand(k5, -1, -1), pf1.cpu().expand(k5, -1, -1, -1), c(beta), c(gen), c(mres), False)
    assert max_abs(c(net['R_next']), ref[1]) < 3e-5
    assert max_abs(c(net['eps_pos']), ref[2]) < 3e-5
    assert max_abs(c(net['c']), ref[3]) < 1e-5
    assert max_abs(c(net['prmsd_logits']), ref[4]) < 3e-5
    for lo in (0, 497, N - 3):                                                # 3 poses: 9 query blocks, the plain one-block kernel
        sl = slice(lo, lo + 3)
        cc = lambda a: a[sl].contiguous()
        small = hip.eps_net_forward(ew, cc(v), cc(p), cc(s), cc(rfN), pf1, cc(beta), cc(gen), cc(mres), d.abdock, d.num_bins, False,
                                    pair_bias_cache=pbc, pair_feat_shared=True)
        for k in ('R_next', 'eps_pos', 'c'):
            assert torch.equal(net[k][sl], small[k]), (k, lo)


@pytest.mark.gpu
@pytest.mark.parametrize('N,L,shared', [(64, 256, 1), (64, 256, 0), (128, 256, 16), (40, 200, 0)])
def test_fused_block_repeats_bit_for_bit_over_several_rounds(N, L, shared):
    """Race detector for the fused core + tail kernel: its epilogue re-uses the LDS of the S/P tile, the key mask region and the aggregated
    points for staging buffers, u, the LayerNorm rows and the activation planes, separated only by workgroup barriers.  Shapes with
    several rounds of workgroups per CU (N = 64: two, N = 128 grouped: four) and a partial last block (L = 200), twelve runs each: every
    output equal to the first run's, bit for bit, and to the two-launch form's."""
    import os
    from ab_opt_amd import hip
    d = standalone_abdesign_dpm(100, 2).to(DEV)
    lens = [L - (3 * i) % 30 for i in range(N)]
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lens, 9100 + N, [(5, 14), (22, 30)])
    Nc = N if not shared else (1 if shared == 1 else N // shared)
    pfc = pf[:Nc].contiguous()
    beta = d.trans_pos.var_sched.betas[37].expand([N]).contiguous()
    arr, ew = d.eps_net.encoder.packed_array(), d.eps_net.packed()
    pbc = hip.pair_bias_cache(arr, 6, pfc)
    run = lambda: {k: (a.clone() if a is not None else None) for k, a in
                   hip.eps_net_forward(ew, v, p, s, rf, pfc, beta, gen, mres, d.abdock, d.num_bins, False, pair_bias_cache=pbc, pair_feat_shared=shared).items()}
    first = run()
    for rep in range(11):
        again = run()
        for k in ('R_next', 'eps_pos', 'c'):
            assert torch.equal(again[k], first[k]), (rep, k)
    os.environ['ABOPT_FUSE_TAIL'] = '0'
    try:
        two = run()
    finally:
        del os.environ['ABOPT_FUSE_TAIL']
    for k in ('R_next', 'eps_pos', 'c'):
        assert torch.isfinite(first[k]).all() and torch.equal(two[k], first[k]), k


def test_training_step_config5_vs_oracle_at_full_size():
    """BASELINE config 5 at its own size (AbDesign flavour FullDPM.forward, N = 16, L = 256): losses and the gradient of EVERY parameter
    of the denoiser, res_feat and pair_feat from the native path (HIP noising, IPA core forward / backward, block tail, abopt_gemm
    everywhere, shared d pair_feat buffer) against the ORACLE under CPU autograd with the same step indices and injected noise -- not
    against another evaluation by this package.  The oracle runs in float64 (the yardstick) and in float32 (what the reference's own
    arithmetic leaves of it): losses within 2e-5, every gradient within 3e-4 of its maximum of the float64 value, block 0 and the mixer
    included -- or no further from it than twice the float32 oracle is (single layers whose transition-MLP ReLUs sit on a kink)."""
    from oracle import dpm as odpm
    N, L = 16, 256
    d = standalone_abdesign_dpm(100, 2).to(DEV).train()
    d.zero_grad(set_to_none=True)
    lens = [L - (5 * i) % 40 for i in range(N)]
    v, p, s, res_feat, pair_feat, _, gen, mres = synth.eps_inputs(N, L, lens, [(25, 33), (51, 57), (94, 106), (133, 144), (159, 166), (198, 207)], salt=900)
    s = s.clamp(max=19)
    g = torch.Generator().manual_seed(77)
    t = torch.randint(1, 100, (N,), generator=g)
    noise = dict(axis=torch.randn(N, L, 3, generator=g), bin=torch.randint(0, 8191, (N, L), generator=g), ubin=torch.rand(N, L, generator=g),
                 gauss=torch.randn(N, L, generator=g), pos=torch.randn(N, L, 3, generator=g), s_noisy=torch.randint(0, 20, (N, L), generator=g))
    rf = dev(res_feat).clone().requires_grad_(True)
    pf = dev(pair_feat).clone().requires_grad_(True)
    loss = d(dev(v), dev(p) * 10, dev(s), rf, pf, dev(gen), dev(mres), True, True, t=dev(t), noise={k: dev(a) for k, a in noise.items()})
    sum(loss.values()).backward()
    got = {n: q.grad.detach().cpu() for n, q in d.named_parameters() if q.grad is not None}
    got['res_feat'], got['pair_feat'] = rf.grad.cpu(), pf.grad.cpu()
    got_loss = {k: a.item() for k, a in loss.items()}
    d.zero_grad(set_to_none=True)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    f = d.trans_rot.angular_distrib_fwd
    ref = {}
    for dt in (torch.float64, torch.float32):
        cv = lambda a: (a.detach().cpu().to(dt) if a.is_floating_point() else a.detach().cpu())
        sd = {k: cv(a).requires_grad_(a.is_floating_point() and 'eps_net' in k) for k, a in d.state_dict().items()}
        den = odpm.Denoiser(sd, num_steps=100, variant='abdesign', pre='', tables=(None, None), mode='mm')
        den.sch = {k: cv(a) for k, a in den.sch.items()}
        den.tab_fwd = dict(stddevs=cv(f.stddevs), approx_flag=f.approx_flag.cpu(), X=cv(f.X), Y=cv(f.Y))
        r_, p_ = cv(res_feat).requires_grad_(True), cv(pair_feat).requires_grad_(True)
        nz = dict(rot=dict(axis=cv(noise['axis']), bin=noise['bin'], ubin=cv(noise['ubin']), gauss=cv(noise['gauss'])), pos=cv(noise['pos']), s_noisy=noise['s_noisy'])
        with torch.enable_grad():
            lo = den.loss(cv(v), cv(p) * 10, s, r_, p_, gen, mres, t, nz)
            sum(lo.values()).backward()
        gr = {k: a.grad for k, a in sd.items() if a.grad is not None}
        gr['res_feat'], gr['pair_feat'] = r_.grad, p_.grad
        ref[dt] = ({k: a.item() for k, a in lo.items()}, gr)
    (l64, g64), (l32, g32) = ref[torch.float64], ref[torch.float32]
    assert set(got_loss) == {'rot', 'pos', 'seq'}
    for k in got_loss:
        assert abs(got_loss[k] - l64[k]) <= 2e-5 * max(1.0, abs(l64[k])), (k, got_loss[k], l64[k])
    assert set(g64) == set(got) and len(got) > 140
    # Error of a tensor = max over its rows (output units of a weight / residues of an activation gradient) of |got - fp64| / max |fp64|.
    # ONE ReLU of the 3 x 6 transition / head layers sitting within an ulp of its kink for ONE of the 4096 residues flips between two fp32
    # evaluations and moves exactly one row (measured: eps_seq_net.0 unit 29 at 2.3e-3, every other unit of that layer <= 5e-6; the oracle's
    # own fp32 evaluation happens to land on the fp64 side): the two worst rows of a tensor are held to 5e-3, all the others to 3e-4.
    rows, exempt = [], []
    for n in sorted(got):
        mx = g64[n].abs().max().item() + 1e-30
        e = ((got[n].double() - g64[n]).abs() / mx)
        e = e.reshape(-1, e.shape[-1]).max(1).values if e.dim() > 1 else e
        srt = torch.sort(e.flatten(), descending=True).values
        e_all, e_rest = srt[0].item(), (srt[2].item() if srt.numel() > 2 else 0.0)
        e_o32 = (g32[n].double() - g64[n]).abs().max().item() / mx
        rows.append((e_rest, e_all, e_o32, n))
        if e_all > 3e-4:
            exempt.append((n, e_all, e_rest))
    print('config 5 at full size vs the float64 oracle -- tensors with a row above 3e-4 (name, worst row, third-worst row):', exempt)
    print('worst tensors (all rows but the two worst | worst row | oracle fp32 | name):', ' ; '.join('%.1e | %.1e | %.1e | %s' % w for w in sorted(rows)[-4:]))
    # ... and only where a ReLU follows the layer (the flipped unit is a row of THAT layer's weight / bias gradient): every other tensor holds 3e-4 on all of its rows
    # (the exemption was a blanket over all 145 tensors; measured in round 6: no tensor needs it, worst row of any tensor 4.5e-5)
    relu_fed = lambda n: any(k in n for k in ('mlp_transition.0.', 'mlp_transition.2.', 'res_feat_mixer.0.', '_net.0.', '_net.2.'))
    assert all(a <= 3e-4 and b <= (5e-3 if relu_fed(n) else 3e-4) for a, b, c, n in rows), [w for w in rows if w[0] > 3e-4 or w[1] > (5e-3 if relu_fed(w[3]) else 3e-4)][:6]
    assert len(exempt) <= 2, exempt
    block0 = [w for w in rows if 'blocks.0.' in w[3] or 'mixer' in w[3]]
    assert len(block0) > 20 and max(w[1] for w in block0) <= 3e-4, sorted(block0)[-3:]


@pytest.mark.parametrize('N,L,lengths', [(8, 256, 'ragged'), (3, 256, None), (2, 100, 'ragged'), (5, 128, None)])
def test_key_split_core_vs_unsplit_and_oracle(N, L, lengths, monkeypatch):
    """Small batches (fewer query blocks than half the CUs) run the cached core with the keys of every query block split over 2 or 4
    workgroups and a softmax merge (csrc/ipa_core.hip: ipa_core_kernel<.., SPLIT>, ipa_split_merge_kernel).  Against the unsplit kernel
    (ABOPT_CORE_NO_SPLIT=1) EpsilonNet agrees to fp32 rounding (another summation order, six layers deep), against the oracle to the
    usual 3e-5 -- ragged lengths included."""
    from ab_opt_amd import hip
    from oracle import dpm
    T, t = 100, 37
    d_cpu = standalone_abdesign_dpm(T, 2)
    d = standalone_abdesign_dpm(T, 2).to(DEV)
    sd = {k: v.cpu() for k, v in d_cpu.state_dict().items()}
    lens = [L] * N if lengths is None else [L - (37 * i) % (L // 2) for i in range(N)]
    v, p, s, rf, pf, gen, mres = _rand_eps_inputs(N, L, lens, 8200 + N, [(5, 14), (22, 30)])
    beta = d.trans_pos.var_sched.betas[t].expand([N]).contiguous()
    ew = d.eps_net.packed()
    pbc = hip.pair_bias_cache(d.eps_net.encoder.packed_array(), 6, pf)
    run = lambda: {k: a.clone() for k, a in hip.eps_net_forward(ew, v, p, s, rf, pf, beta, gen, mres, False, 0, False, pair_bias_cache=pbc).items() if a is not None}
    split = run()
    monkeypatch.setenv('ABOPT_CORE_NO_SPLIT', '1')
    plain = run()
    differs = False
    for k in ('R_next', 'eps_pos', 'c'):
        assert torch.isfinite(split[k]).all()
        assert max_abs(split[k], plain[k]) < 1e-5, (k, max_abs(split[k], plain[k]))
        differs |= not torch.equal(split[k], plain[k])
    assert differs                                              # otherwise the split form did not run and this test tests nothing
    ids = list(range(min(N, 3)))
    c = lambda a: a[ids].cpu()
    inv = d_cpu.trans_rot.angular_distrib_inv
    den = dpm.Denoiser(sd, num_steps=T, variant='abdesign', obj='pred_noise', mode='mm', pre='',
                       tables=(None, dict(stddevs=inv.stddevs, approx_flag=inv.approx_flag, X=inv.X, Y=None)))
    ref = den._eps(c(v), c(p), c(s), c(rf), c(pf), c(beta), c(gen), c(mres), False)
    assert max_abs(c(split['R_next']), ref[1]) < 3e-5
    assert max_abs(c(split['eps_pos']), ref[2]) < 3e-5
    assert max_abs(c(split['c']), ref[3]) < 1e-5


# ------------------------------------------------------------------------------------------ sequence-design mode (seq_design.yml / fixbb.yml)
def _tables_inv(d_cpu):
    inv = d_cpu.trans_rot.angular_distrib_inv
    return (None, dict(stddevs=inv.stddevs, approx_flag=inv.approx_flag, X=inv.X, Y=None))


def test_sequence_design_steps_teacher_forced_vs_reference():
    """AbDock/configs/test/seq_design.yml:4-6 (sample_structure=False, sample_sequence=True, contig; per pose from optimize_ab.py:14-36):
    every one of the 10 recorded reference steps.  The structure must be handed on untouched (dpm_full.py:294-295), the posterior is
    the reference's at 2e-6, the sequence follows the injected draws; encode() of this mode and the init state as well; then the
    whole model.sample() on the device RNG path."""
    from ab_opt_amd import hip
    from ab_opt_amd.model import generate_mask_from_str
    g = load_golden('trajectory_abdock_T10_seqdesign')
    _, m, batch = _traj_setup()
    d = m.diffusion
    b = {k: dev(v) for k, v in batch.items()}
    gen = torch.logical_and(b['generate_flag'], generate_mask_from_str('31-36', b['generate_flag']))
    assert torch.equal(gen.cpu(), g['gen'])
    b['generate_flag'] = gen
    with torch.no_grad():
        rf, pf, R0, p0 = m.encode(dict(b), False, True)                      # remove_structure=False (diffab.py:132-136)
    assert max_abs(rf.cpu(), g['res_feat']) < 2e-4 * g['res_feat'].abs().max().item()
    assert max_abs(pf.cpu()[:, ::7, ::5], g['pair_feat_sub']) < 2e-4 * g['pair_feat_sub'].abs().max().item()
    assert max_abs(R0.cpu(), g['R0']) < 1e-5 and max_abs(p0.cpu(), g['p0']) == 0
    rf, mres = dev(g['res_feat']), b['mask']
    v0 = hip.so3_log(dev(g['R0']), False)
    v_i, p_i, s_i = hip.sample_init(v0, dev(g['p0']), b['aa'], gen, dict(s=dev(g['init_s'])), 0, 0, 10.0, [0.0, 0.0, 0.0], False, True)
    assert torch.equal(v_i, v0) and max_abs(p_i.cpu(), g['traj10_p']) < 1e-5 and torch.equal(s_i.cpu(), g['traj10_s'])
    for t in range(10, 0, -1):
        state = (dev(g[f'traj{t}_v']), dev(g[f'traj{t}_p']), dev(g[f'traj{t}_s']))
        noise_t = {k: dev(g[f't{t}_{k}']) for k in ('axis', 'bin', 'ubin', 'gauss', 'z', 's_next')}
        tv, tp, ts, tpr, tpp = d._run(tuple(a.clone() for a in state), t, rf, pf, gen, mres, False, True, True, {t: noise_t}, 0, 0, False, stop_after=1)
        assert torch.equal(tv[t - 1], state[0]) and torch.equal(tv[t - 1].cpu(), g[f'traj{t - 1}_v']), t     # bit for bit
        assert max_abs(tp[t - 1].cpu(), g[f'traj{t - 1}_p']) < 2e-6 and max_abs(tp[t - 1], state[1]) < 2e-6, t
        assert torch.equal(ts[t - 1].cpu(), g[f'traj{t - 1}_s']), t
        assert max_abs(tpr[t - 1].cpu(), g[f'traj{t - 1}_prmsd']) < 1e-4 and max_abs(tpp[t - 1].cpu(), g[f'traj{t - 1}_ppl']) < 1e-5, t
        post, out = _device_step_with_posterior(d, t, state, rf, pf, gen, mres, noise_t, sample_structure=False)
        assert max_abs(post.cpu() + 1e-8, g[f't{t}_probs']) < 2.5e-5, t             # (max over the steps 2.2e-6 / 7.5e-6 / 1.25e-5 in three builds of round 6 that differ only in fp32 roundings -- packed or plain VALU instructions, two forms of the dihedral: one probability of the softmax of a head whose inputs carry 4e-7)
        assert torch.equal(out['v'], state[0])
    # the whole call on the device's own RNG: structure untouched from t = 10 to 0, context sequence untouched, designed residues valid
    bb = {k: dev(v) for k, v in batch.items()}
    traj = m.sample(bb, sample_opt=dict(sample_structure=False, sample_sequence=True, contig='31-36'))
    assert torch.equal(bb['generate_flag'], gen)
    genc = gen.cpu()
    ctx = ~genc & batch['mask']        # padded residues (aa = 21) are RE-DRAWN by the reference itself every step: clampped_one_hot(21) = 0, multinomial(0 + 1e-8), transition.py:240-245
    v0_dev = hip.so3_log(R0, False)        # of the device's own encode(): equal to the fixture's frames to 1e-5 (above), handed on bit for bit
    for t in (10, 5, 0):
        assert torch.equal(traj[t][0].cpu(), v0_dev.cpu()), t            # (entries t > 0 live on the host, t = 0 on the device)
        assert max_abs(traj[t][1].cpu(), g['p0']) < 1e-4, t
        assert torch.equal(traj[t][2].cpu()[ctx], batch['aa'][ctx]), t
        sg = traj[t][2].cpu()[genc]
        assert bool(((sg >= 0) & (sg < 20)).all())


def test_fixbb_abdesign_steps_teacher_forced_vs_reference():
    """AbDesign/configs/test/fixbb.yml:7 mode at FullDPM level (A/.../dpm_full.py:193-254 with sample_structure=False): 10 recorded steps."""
    g = load_golden('trajectory_abdesign_T10_fixbb')
    d = standalone_abdesign_dpm(10, 4).to(DEV)
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)], num_steps=10, t=3)
    rf, pf, gen, mres = dev(res_feat), dev(pair_feat), dev(gen), dev(mres)
    for t in range(10, 0, -1):
        state = (dev(g[f'traj{t}_v']), dev(g[f'traj{t}_p']), dev(g[f'traj{t}_s']))
        noise_t = {k: dev(g[f't{t}_{k}']) for k in ('axis', 'bin', 'ubin', 'gauss', 'z', 's_next')}
        tv, tp, ts, _, _ = d._run(tuple(a.clone() for a in state), t, rf, pf, gen, mres, False, True, True, {t: noise_t}, 0, 0, False, stop_after=1)
        assert torch.equal(tv[t - 1].cpu(), g[f'traj{t - 1}_v']), t
        assert max_abs(tp[t - 1].cpu(), g[f'traj{t - 1}_p']) < 2e-6, t
        assert torch.equal(ts[t - 1].cpu(), g[f'traj{t - 1}_s']), t
        post, _ = _device_step_with_posterior(d, t, state, rf, pf, gen, mres, noise_t, sample_structure=False)
        assert max_abs(post.cpu() + 1e-8, g[f't{t}_probs']) < 5e-6, t
    traj = d.sample(dev(v), dev(p) * 10, dev(s), rf, pf, gen, mres, sample_structure=False, sample_sequence=True, seed=3)
    assert torch.equal(traj[0][0].cpu(), v) and max_abs(traj[0][1].cpu(), p * 10) < 1e-4
    ctx = ~gen.cpu() & mres.cpu()      # (padded residues, s = 21, are re-drawn by the reference too: transition.py:240-245)
    assert torch.equal(traj[0][2].cpu()[ctx], s[ctx])
    assert torch.equal(g['traj0_s'][ctx], s[ctx]) and not torch.equal(g['traj0_s'][~mres.cpu()], s[~mres.cpu()])     # ... as the fixture shows


@pytest.mark.parametrize('flavour', ['abdock', 'abdesign'])
def test_training_sequence_only_vs_reference(flavour):
    """train_structure=False, train_sequence=True (AbDock/configs/train/seq_design.yml:11-12; dpm_full.py:163-178): the structure enters
    un-noised, eps_p = 0, the one draw of the mode (s_noisy) injected; losses (2e-5 rel) and gradients (3e-4 of max) against the
    reference's recorded values, both trees; add_noise's categorical against the recorded multinomial input."""
    from ab_opt_amd import hip
    from test_oracle_golden import _sub
    g = load_golden(f'training_seqonly_{flavour}')
    d = (build_model(100, 2, device=DEV).diffusion if flavour == 'abdock' else standalone_abdesign_dpm(100, 2).to(DEV)).train()
    d.zero_grad()
    N, L = 2, 48
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
    s = s.clamp(max=19)
    res_feat = dev(res_feat).clone().requires_grad_(True)
    pair_feat = dev(pair_feat).clone().requires_grad_(True)
    tt = torch.tensor([37, 80], device=DEV)
    vs, h = d.trans_pos.var_sched, d._sched_host()
    v_n, p_n, s_n, eps, probs = hip.add_noise(tt, vs.alpha_bars, d.trans_rot.angular_distrib_fwd, dict(s_noisy=dev(g['s_noisy'])), 0, 0, dev(v), dev(p) * 10,
                                              dev(s), dev(gen), h['scale'], h['mean'], noise_structure=False, noise_sequence=True, grad_mode=True,
                                              want_eps=True, want_probs=True)
    assert torch.equal(v_n.cpu(), v) and max_abs(p_n.cpu(), p * 10) < 2e-6 and not eps.any() and torch.equal(s_n.cpu(), g['s_noisy'])
    assert max_abs(probs.cpu() + 1e-8, g['addnoise_probs']) < 1e-7
    loss = d(dev(v), dev(p) * 10, dev(s), res_feat, pair_feat, dev(gen), dev(mres), False, True, t=tt, noise=dict(s_noisy=dev(g['s_noisy'])))
    assert set(loss) == ({'prmsd', 'dist', 'rot', 'pos', 'seq'} if flavour == 'abdock' else {'rot', 'pos', 'seq'})
    for k, val in loss.items():
        ref = g['loss_' + k].item()
        assert abs(val.item() - ref) <= 2e-5 * max(1.0, abs(ref)), (k, val.item(), ref)
    sum(loss.values()).backward()
    params = dict(d.named_parameters())
    n = 0
    for k in g:
        if k.startswith('grad_eps_net'):
            got = _sub(params[k[len('grad_'):]].grad.cpu(), g[k])
            assert max_abs(got, g[k]) <= 3e-4 * g[k].abs().max().item() + 1e-7, k
            n += 1
    assert n == 11
    assert max_abs(res_feat.grad.cpu(), g['grad_res_feat']) <= 3e-4 * g['grad_res_feat'].abs().max().item()
    assert max_abs(pair_feat.grad.cpu()[:, ::5, ::3], g['grad_pair_feat_sub']) <= 3e-4 * g['grad_pair_feat_sub'].abs().max().item()
    d.zero_grad()


# ------------------------------------------------------------------------------------------ the headline schedule, T = 100
@pytest.mark.parametrize('flavour', ['abdock', 'abdesign'])
def test_single_steps_T100_vs_reference(flavour):
    """Recorded reference steps of the T = 100 sampler at t = 100, 64, 22 (last histogram row of the inverse IGSO(3) distribution), 21, 10, 2
    (its Gaussian branch |2 sigma + sigma randn| mod pi, so3.py:127-135: t = 2..21, 20 of the 100 steps) and 1 (no noise), both trees,
    draws injected into abopt_denoise_step: positions 1e-4 A, orientations under rot_close, posterior 2e-6."""
    from oracle import dpm, geometry as G
    from test_oracle_golden import STEP_TS, step_noise
    g = load_golden(f'steps_T100_{flavour}')
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)], num_steps=100, t=3)
    if flavour == 'abdock':
        full = build_model(100, 2)
        d_cpu, d = full.diffusion, build_model(100, 2, device=DEV).diffusion
        den = dpm.Denoiser(full.state_dict(), num_steps=100, variant='abdock', obj='pred_x0', mode='mm', tables=_tables_inv(d_cpu))
        a, b = 'traj{}', 'traj{}'
    else:
        d_cpu, d = standalone_abdesign_dpm(100, 2), standalone_abdesign_dpm(100, 2).to(DEV)
        den = dpm.Denoiser(d_cpu.state_dict(), num_steps=100, variant='abdesign', pre='', mode='mm', tables=_tables_inv(d_cpu))
        a, b = 'in{}', 'out{}'
    rf, pf, gend, mresd = dev(res_feat), dev(pair_feat), dev(gen), dev(mres)
    worst_p = 0.0
    for t in STEP_TS:
        i, o = a.format(t), b.format(t - 1 if flavour == 'abdock' else t)
        nz = step_noise(g, t)
        state = (dev(g[i + '_v']), dev(g[i + '_p']), dev(g[i + '_s']))
        noise_t = {k: dev(x) for k, x in nz.items()}
        tv, tp, ts, tpr, tpp = d._run(tuple(x.clone() for x in state), t, rf, pf, gend, mresd, True, True, True, {t: noise_t}, 0, 0, False, stop_after=1)
        e = dpm.so3_noise(den.tab_inv, torch.full((2, 40), t), nz)
        if t <= 1:
            e = torch.zeros_like(e)
        ref = den._eps(g[i + '_v'], den.norm(g[i + '_p']), g[i + '_s'], res_feat, pair_feat, den.sch['betas'][t].expand([2]), gen, mres, False)
        n, worst = rot_close(tv[t - 1].cpu(), g[o + '_v'], G.so3_exp(e) @ G.so3_exp(ref[0]), R_upstream=ref[1])
        assert n >= 70 and worst < 1.0, (t, n, worst)
        dp = max_abs(tp[t - 1].cpu(), g[o + '_p'])
        worst_p = max(worst_p, dp)
        assert dp < 1e-4, (t, dp)
        assert torch.equal(ts[t - 1].cpu(), g[o + '_s']), t
        if flavour == 'abdock':
            assert max_abs(tpr[t - 1].cpu(), g[o + '_prmsd']) < 1e-4 and max_abs(tpp[t - 1].cpu(), g[o + '_ppl']) < 1e-5, t
        post, _ = _device_step_with_posterior(d, t, state, rf, pf, gend, mresd, noise_t)
        assert max_abs(post.cpu() + 1e-8, g[f't{t}_probs']) < 5e-6, t
    print('worst T=100 step position error (Angstrom):', worst_p)


@pytest.mark.parametrize('t', [10, 21, 2])
def test_inverse_igso3_gaussian_branch_on_device_rng(t):
    """The device's OWN draws where the inverse IGSO(3) distribution takes its Gaussian branch (sigma_t <= 0.1: t = 2..21 at T = 100):
    theta = |2 sigma + sigma N(0,1)| mod pi (so3.py:127-135), chi-square against the folded normal; no histogram bin is involved."""
    from ab_opt_amd import hip
    d = build_model(100, 2, device=DEV).diffusion
    N, L = 64, 256
    sp = d._step_params(t, True, True, True)
    inv = d.trans_rot.angular_distrib_inv
    sig = float(inv.stddevs[t])
    assert sp.igso3_gaussian == 1 and 0 < sig <= 0.1
    z3 = torch.zeros(N, L, 3, device=DEV)
    c_net = torch.full((N, L, 20), 0.05, device=DEV)
    s_t = torch.zeros(N, L, dtype=torch.int64, device=DEV)
    gen = torch.ones(N, L, dtype=torch.bool, device=DEV)
    out = dict(v=torch.empty(N, L, 3, device=DEV), p=torch.empty(N, L, 3, device=DEV), s=torch.empty(N, L, dtype=torch.int64, device=DEV),
               prmsd=torch.empty(N, device=DEV), ppl=torch.empty(N, device=DEV))
    # a CDF row of NaNs: the branch must not read it
    hip.denoise_step(sp, None, 4321, 0, z3, z3, s_t, z3, z3, c_net, torch.zeros(N, 40, device=DEV), gen, inv.X[t], torch.full((8191,), float('nan'), device=DEV), 40, out)
    theta = out['v'].norm(dim=-1).flatten().cpu().double()              # v_next = log(exp(e) exp(0)) = e
    assert bool(torch.isfinite(theta).all())
    Phi = lambda x: 0.5 * (1 + torch.erf(x / math.sqrt(2)))
    edges = torch.linspace(0, 6 * sig, 33, dtype=torch.float64)
    lo, hi = edges[:-1], edges[1:]
    probs = Phi((hi - 2 * sig) / sig) - Phi((lo - 2 * sig) / sig) + Phi((-lo - 2 * sig) / sig) - Phi((-hi - 2 * sig) / sig)
    hist = torch.histc(theta, bins=32, min=0.0, max=6 * sig)
    expected = probs * theta.numel()
    keep = expected > 20
    chi2 = (((hist - expected) ** 2 / expected)[keep]).sum().item()
    assert chi2 < 3 * int(keep.sum()), (t, chi2)
    assert abs(theta.mean().item() - 2 * sig) < 0.02 * sig + 0.05 * sig            # folded tail moves the mean by 0.0085 sigma
    axis = (out['v'] / out['v'].norm(dim=-1, keepdim=True)).reshape(-1, 3).cpu()
    assert axis.mean(0).abs().max() < 0.02


# ------------------------------------------------------------------------------------------ grouped weight-gradient products
@pytest.mark.gpu
def test_gemm_tn_grouped_vs_fp64():
    """abopt_gemm_tn_grouped (include/abopt.h): C_p = A_p^T B_p for a group of tall operand pairs in one product launch + one slab-sum
    launch -- odd sizes (M, N not multiples of 64 or 4, K not a multiple of 32), column slices of wider matrices read in place, K too short
    to split, more problems than one launch carries (24), a single fat problem that is not split at all -- against float64, and bit for bit
    against itself on a second call and against the same products issued one per call."""
    from ab_opt_amd import hip
    g = torch.Generator().manual_seed(12)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(DEV)
    wide_a, wide_b = rnd(4096, 300), rnd(4096, 2100)
    pairs = [(rnd(4096, 128), rnd(4096, 128)), (wide_a[:, 4:135], wide_b[:, 64:64 + 1824]), (rnd(4099, 26), rnd(4099, 64)),
             (rnd(200, 64), rnd(200, 240)), (wide_a[:, 200:264], wide_b[:, 0:2016]), (rnd(777, 5), rnd(777, 3))]
    pairs += [(rnd(1024 + 32 * i, 128), rnd(1024 + 32 * i, 131)) for i in range(22)]          # 28 problems: two launches
    got = hip.gemm_tn_grouped(pairs)
    again = hip.gemm_tn_grouped(pairs)
    for (a, b), c, c2 in zip(pairs, got, again):
        ref = a.double().t() @ b.double()
        assert c.shape == ref.shape and torch.equal(c, c2)
        assert (c.double() - ref).abs().max().item() <= 2e-6 * (a.double().abs().t() @ b.double().abs()).max().item(), (a.shape, b.shape)
    one = hip.gemm_tn_grouped([(wide_b, wide_b[:, :512])])[0]                                   # 33 x 8 tiles: no K split, no slab sum
    assert (one.double() - wide_b.double().t() @ wide_b[:, :512].double()).abs().max().item() <= 2e-6 * 4096 * 16
    outs = [torch.full((p[0].shape[1], p[1].shape[1]), float('nan'), device=DEV) for p in pairs[:3]]
    res = hip.gemm_tn_grouped(pairs[:3], outs=outs)
    assert all(r is o and torch.isfinite(o).all() for r, o in zip(res, outs))
    with pytest.raises(TypeError):
        hip.gemm_tn_grouped([(rnd(64, 8).t(), rnd(8, 8))])


@pytest.mark.gpu
def test_wgrad_group_check_detects_a_second_use_of_a_parameter():
    """WgradGroup hands autograd an UNFILLED tensor as a leaf's gradient and fills it at the flush -- correct only if nothing else contributes to
    that leaf in the same pass.  ABOPT_WGRAD_CHECK (WgradGroup.check) verifies it after every pass: with the package's own loss every queued leaf ends the pass
    holding the very tensor the group computed; a regulariser written with plain torch operations on a queued weight makes the engine add to / clone the
    unfinished tensor, and the check raises instead of returning a wrong gradient.  Non-contiguous or hooked parameters are never queued (they run at once)."""
    from ab_opt_amd import training
    N, L = 2, 64
    d = standalone_abdesign_dpm(100, 2).to(DEV).train()
    v, p, s, res_feat, pair_feat, _, gen, mres = synth.eps_inputs(N, L, [64, 50], [(25, 33), (51, 57)], salt=951)
    s = s.clamp(max=19)
    t = dev(torch.tensor([3, 77]))
    w = d.eps_net.encoder.blocks[0].out_transform.weight

    def run(extra):
        d.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        loss = sum(d(dev(v), dev(p) * 10, dev(s), dev(res_feat), dev(pair_feat), dev(gen), dev(mres), True, True, t=t).values())
        if extra:
            loss = loss + 1e-3 * (w * w).sum()                              # a second, plain-torch use of a weight whose gradient is queued
        loss.backward()
        torch.cuda.synchronize()
        return w.grad.detach().clone()
    was_on, was_check = training.WgradGroup.enabled, training.WgradGroup.check
    try:
        training.WgradGroup.enabled, training.WgradGroup.check = False, False
        ref, ref_extra = run(False), run(True)
        training.WgradGroup.enabled, training.WgradGroup.check = True, True
        got = run(False)                                                    # the package's own pass: the check is silent
        assert (got - ref).abs().max().item() <= 3e-4 * ref.abs().max().item()
        with pytest.raises(RuntimeError, match='WgradGroup'):
            run(True)
        training.WgradGroup.sync()
        # a hook on the parameter: its products run at once, the second use is then summed by the engine as usual
        h = w.register_hook(lambda g: g)
        try:
            got_extra = run(True)
        finally:
            h.remove()
        assert (got_extra - ref_extra).abs().max().item() <= 3e-4 * ref_extra.abs().max().item()
    finally:
        training.WgradGroup.sync()
        training.WgradGroup.enabled, training.WgradGroup.check = was_on, was_check
        d.zero_grad(set_to_none=True)


def test_wgrad_group_matches_ungrouped_backward():
    """training.WgradGroup: the weight-gradient products of a backward pass queue and run as grouped launches (about one per GABlock + one
    when the engine finishes the pass).  Against the one-product-per-launch form: every gradient within fp32 summation-order distance
    (another K split), the gradients that do not go through the queue (res_feat, pair_feat) bit for bit; the grouped form
    repeats bit for bit, pass after pass (an unflushed queue or an operand freed early would show up here); a second backward without
    zero_grad accumulates like the ungrouped form (those products run at once); a step captured by GraphedTrainStep replays
    deterministically."""
    from ab_opt_amd import training
    N, L = 4, 128
    d = standalone_abdesign_dpm(100, 2).to(DEV).train()
    v, p, s, res_feat, pair_feat, _, gen, mres = synth.eps_inputs(N, L, [128, 120, 97, 128], [(25, 33), (51, 57), (94, 106)], salt=901)
    s = s.clamp(max=19)
    t = dev(torch.tensor([3, 40, 77, 99]))

    def grads(accumulate=False):
        d.zero_grad(set_to_none=True)
        out = None
        for rep in range(2 if accumulate else 1):
            rf, pf = dev(res_feat).clone().requires_grad_(True), dev(pair_feat).clone().requires_grad_(True)
            torch.manual_seed(123 + rep)                                    # the noising kernel's Philox seed comes from torch's host generator
            loss = d(dev(v), dev(p) * 10, dev(s), rf, pf, dev(gen), dev(mres), True, True, t=t)
            sum(loss.values()).backward()
            out = {n: q.grad for n, q in d.named_parameters() if q.grad is not None}
            out['res_feat'], out['pair_feat'] = rf.grad, pf.grad
        torch.cuda.synchronize()
        return {k: a.detach().clone() for k, a in out.items()}

    was = training.WgradGroup.enabled
    try:
        training.WgradGroup.enabled = False
        ref, ref_acc = grads(), grads(True)
        training.WgradGroup.enabled = True
        assert training.WgradGroup.active()
        first = grads()
        assert training.WgradGroup._state is None                          # closed by the engine's final callback
        assert set(first) == set(ref) and len(first) > 140
        moved = 0
        for k in ref:
            if k in ('res_feat', 'pair_feat'):
                assert torch.equal(first[k], ref[k]), k                      # nothing on their way goes through the queue
            else:                                                           # weights; biases, LayerNorm, pair-bias and spatial-coefficient gradients (column sums = products with a column of ones)
                assert (first[k] - ref[k]).abs().max().item() <= 2e-5 * ref[k].abs().max().item() + 1e-12, k
                moved += int(not torch.equal(first[k], ref[k]))
        assert moved > 20                                                   # otherwise the queue did not run and this test tests nothing
        for rep in range(6):
            got = grads()
            for k in first:
                assert torch.equal(got[k], first[k]), (rep, k)
        acc = grads(True)
        for k in ref_acc:
            assert (acc[k] - ref_acc[k]).abs().max().item() <= 2e-5 * ref_acc[k].abs().max().item() + 1e-12, k
        # a step captured by GraphedTrainStep (the flushes are ordinary launches of the capture): two identical builds replay to identical parameters
        from ab_opt_amd.utils import synth as sy
        finals = []
        for build in range(2):
            m = sy.fresh_model(10, 3, device=DEV).train()
            opt = training.FusedAdam(m.parameters(), lr=1e-3)
            batch = {k: dev(a) for k, a in sy.make_batch(2, sy.LAYOUT_128, seed=5, lengths=[64, 57]).items()}
            torch.manual_seed(5); torch.cuda.manual_seed(5)                 # step indices (device generator) and Philox seeds (host generator)
            step = training.GraphedTrainStep(m, opt, batch, max_grad_norm=100.0, warmup=1)
            for _ in range(3):
                losses = step(batch)
            torch.cuda.synchronize()
            assert all(torch.isfinite(a) for a in losses.values())
            finals.append({n: q.detach().clone() for n, q in m.named_parameters()})
        for n in finals[0]:
            assert torch.equal(finals[0][n], finals[1][n]), n
    finally:
        training.WgradGroup.enabled = was
        d.zero_grad(set_to_none=True)

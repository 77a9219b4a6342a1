#!/usr/bin/env python
"""Denoising steps/sec of the HIP sampler on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 100 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the sampling loop (EpsilonNet forward + rotation/position/sequence transitions,
device RNG included) over one batch of synthetic complexes.  Workload at every N: BASELINE.json configs[1]
(AbDesign codesign_single model block, L=256, 6 CDR segments, batch 32 per GPU, T=100); the K timed steps are
steps T..T-K+1 of that sampler.  Inputs (res_feat, pair_feat, initial state) are resident in HBM before the timed
region; encode() and PDB I/O are outside the metric.  value = samples*K / max-over-ranks time.
Samples are independent, so ranks share nothing during the loop (weak scaling, 32 samples per GPU); the only
exchange is the candidate all_gather of the batched-sampling reduction after the last step, which is inside the
timed region at N>1.

Roofline accounting: the dominant kernel is the IPA core -- since round 4 with the block's
tail (out_transform + LayerNorm + MLP + LayerNorm) fused into it; its ALGORITHMIC bytes per launch are N*(256*L^2 + 1076*L) (pair
features once + node features in/out + frames/mask: the survey's figure for the whole block), its duration is measured live with HIP
events recorded on the launch stream around every launch (DESIGN.md section 5 says where they sit when the loop is replayed from a
hipGraph).  `roofline.two_launch_form` times the core alone (ABOPT_FUSE_TAIL=0) in one extra pass; `box` carries the device-to-device
copy rate and the clock the dominant kernel sustained, so a reader can tell a slow box from a slow kernel.  The timed region is run
`--repeats` times and the median is reported (min / max alongside).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy ceiling)
FP32_PEAK_TFLOPS = 157.3   # fp32 vector == fp32 MFMA peak (same guide)
NUM_LAYERS = 6
# HBM bytes per ipa_core launch from the PMC counters of the committed profile (separate rocprofv3 --pmc passes on this
# command; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md section HBM).  A constant from that profile,
# not something this run measured: `traffic_source` names the file.
MEASURED_TRAFFIC = {}
TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'ipa_core_traffic.json')
if os.path.exists(TRAFFIC_JSON):
    with open(TRAFFIC_JSON) as fh:
        for rec in json.load(fh):
            MEASURED_TRAFFIC[(rec['N'], rec['L'])] = (rec['bytes_per_launch'], rec['source'])


def ipa_algorithmic_bytes(N, L):
    """per sample-layer 4*C*L^2 (z once) + 2*4*F*L (x in/out) + 52*L (R, t, mask) = 256 L^2 + 1076 L."""
    return N * (256 * L * L + 1076 * L)


def ipa_kernel_io_bytes(N, L, C=64):
    """What the IPA-core KERNEL itself must move (its own operands, not the survey's per-layer figure): z once + node
    projections in (2016 floats/residue) + features out (1824) + frames/mask (52 B).  Reported next to the survey figure."""
    return N * (4 * C * L * L + (2016 * 4 + 1824 * 4 + 52) * L)


def step_algorithmic_bytes(N, L):
    """6*N*(256 L^2 + 1076 L) + 13.0e6 (weights once per step)."""
    return NUM_LAYERS * ipa_algorithmic_bytes(N, L) + 13.0e6


def step_flops(N, L):
    """per sample-layer node GEMMs (2*L*128*(2016+128*3) + 2*L*1824*128) + pairwise terms
    2*L^2*12*(64 + 32 + 24 + 64 + 32 + 24) + softmax ~ 4 flops per logit; + mixer/heads ~0.09 GF per sample."""
    node = 2 * L * 128 * (2016 + 3 * 128) + 2 * L * 1824 * 128
    pair = 2 * L * L * 12 * (64 + 32 + 24 + 64 + 32 + 24) + 4 * L * L * 12
    return N * (NUM_LAYERS * (node + pair) + 0.09e9 * L / 256)


def build_workload(dev, N, L, T, seed, abdesign=True, shared_context=False, cdrs=None):
    from ab_opt_amd.dpm import FullDPM
    from ab_opt_amd.utils import synth
    kw = dict(_abdesign=True) if abdesign else dict(obj='pred_x0', num_bins=40, dist_min=0.5, dist_max=19.5)
    dpm = FullDPM(128, 64, num_steps=T, eps_net_opt=dict(num_layers=NUM_LAYERS), **kw).eval()
    synth.fill_module_(dpm, seed=2)
    dpm = dpm.to(dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    if cdrs is None:
        cdrs = (synth.LAYOUT_256 if L == 256 else synth.LAYOUT_128)['cdrs']
    gen = torch.zeros(N, L, dtype=torch.bool, device=dev)
    for a, b in cdrs:
        gen[:, a:b] = True
    mres = torch.ones(N, L, dtype=torch.bool, device=dev)
    Nc = 1 if shared_context else N                                        # shared_context: ONE complex, N poses of it (the reference's runners)
    res_feat = torch.randn(Nc, L, 128, device=dev, generator=g)
    pair_feat = torch.randn(Nc, L, L, 64, device=dev, generator=g)         # distinct per sample unless shared_context: no replication shortcut
    v = torch.randn(N, L, 3, device=dev, generator=g)
    p = torch.randn(N, L, 3, device=dev, generator=g) * 10
    s = torch.randint(0, 20, (N, L), device=dev, generator=g)
    return dpm, (v, p, s), res_feat, pair_feat, gen, mres


def cpu_baseline(L, T, budget_s=45.0, check=None):
    """The oracle ('port' of the reference, test infrastructure) on the host cores: a bounded sample of the same
    workload.  Threads are capped at 32: the op mix is elementwise/bandwidth bound and slows down badly beyond that on
    many-core hosts; `cores` reports the threads actually used.

    `check` (optional): (dpm, t, state, res_feat, pair_feat, gen, mres, gpu_out, sample_ids) -- the oracle also recomputes
    EpsilonNet for a few samples of the bench batch at the bench's first step and the caller's GPU outputs are compared with
    it (the bench launch geometry, N % 8 == 0, is the one under test)."""
    from oracle import dpm as odpm
    from ab_opt_amd.dpm import FullDPM
    from ab_opt_amd.utils import synth
    host_cores = os.cpu_count() or 1
    threads = min(host_cores, 32)
    torch.set_num_threads(threads)
    out = {}
    if check is not None:
        dpm_gpu, t, state, res_feat, pair_feat, gen, mres, gpu_out, ids = check
        sd = {k: v.detach().cpu() for k, v in dpm_gpu.state_dict().items()}
        idx = torch.tensor(ids)
        v, p, s = [a[idx.to(a.device)].cpu() for a in state]
        beta = dpm_gpu.trans_pos.var_sched.betas[t].cpu().expand([len(ids)])
        ref = odpm.eps_net(sd, 'eps_net.', v, (p - 0.0) / 10.0, s, res_feat[idx.to(res_feat.device)].cpu(), pair_feat[idx.to(pair_feat.device)].cpu(), beta,
                           gen[idx.to(gen.device)].cpu(), mres[idx.to(mres.device)].cpu(), num_layers=NUM_LAYERS, prmsd_head=False, mode='mm')
        errs = {}
        for name, k in (('R_next', 1), ('eps_pos', 2), ('c', 3)):
            errs[name] = float((gpu_out[name][idx.to(gpu_out[name].device)].cpu() - ref[k]).abs().max())
        out['parity_first_step'] = dict(samples=list(ids), max_abs_err=errs, tol=5e-5, ok=all(e < 5e-5 for e in errs.values()))
        assert out['parity_first_step']['ok'], out['parity_first_step']
    m = FullDPM(128, 64, num_steps=10, eps_net_opt=dict(num_layers=NUM_LAYERS), _abdesign=True).eval()
    synth.fill_module_(m, seed=2)
    sch = odpm.variance_schedule(10)
    den = odpm.Denoiser(m.state_dict(), num_steps=10, variant='abdesign', pre='', mode='ref',
                        tables=(None, odpm.igso3_tables(sch['sigmas'].tolist())))
    best, detail, t_all = 0.0, {}, time.perf_counter()
    for mode, N, reps in (('ref', 1, 2), ('mm', 1, 4), ('mm', 4, 2)):
        if time.perf_counter() - t_all > budget_s:
            detail[f'{mode}_N{N}'] = 'skipped (time budget)'
            continue
        den.mode = mode
        v, p, s, res_feat, pair_feat, _, gen, mres = synth.eps_inputs(N, L, [L] * N, [(25, 33), (51, 57), (94, 106)], num_steps=10, t=7)
        nz = dict(axis=torch.randn(N, L, 3), bin=torch.randint(0, 8191, (N, L)), ubin=torch.rand(N, L), gauss=torch.randn(N, L),
                  z=torch.randn(N, L, 3), s_next=torch.randint(0, 20, (N, L)))
        den.step(7, v, p, s, res_feat, pair_feat, gen, mres, nz)             # warm
        t0 = time.perf_counter()
        done = 0
        for _ in range(reps):
            den.step(7, v, p, s, res_feat, pair_feat, gen, mres, nz)
            done += 1
            if time.perf_counter() - t_all > budget_s:
                break
        rate = N * done / (time.perf_counter() - t0)
        detail[f'{mode}_N{N}'] = round(rate, 3)
        best = max(best, rate)
    ref_rate = detail.get('ref_N1') if isinstance(detail.get('ref_N1'), float) else None
    # `value` is the reference's OWN op order (what the reference's CPU path executes; oracle mode 'ref' is bit-equal to it,
    # tests/test_oracle_golden.py); the matmul-contracted restatement of the same arithmetic is the fastest CPU port we have.
    out.update(value=ref_rate if ref_rate is not None else round(best, 3), unit='sample-steps/s', cores=threads, threads=threads,
               host_cores=host_cores, kind='port', ref_order_value=ref_rate, matmul_contracted_value=round(best, 3),
               sample=f'oracle Denoiser.step (EpsilonNet + transitions) at L={L} on {threads} threads of a {host_cores}-core host '
                      f'(the op mix is bandwidth bound and slows down beyond 32 threads): N=1 in the reference op order (= value), N=1 and N=4 with '
                      f'matmul contractions (a few steps each, {budget_s:.0f}s budget); per-variant: {detail}')
    return out


def secondary_measurements(dev, L):
    """The other BASELINE.json configs that fit one GPU, each as a single number so the driver's record carries them
    (they are not the headline metric):
      train_step_ms               config 5: AbDesign flavour model(batch) -> losses -> backward -> Adam, N=16, L=256
      sample_e2e_ms               config 2 end to end: model.sample(batch) incl. encode(), the pair-bias cache and the trajectory hand-over, N=32
      config3_sample_steps_per_s  config 3: AbDock pose sampling (prmsd head, pred_x0, sample_sequence=False), N=64, the sampling loop alone"""
    from ab_opt_amd.utils.synth import build_model, make_batch, LAYOUT_256, LAYOUT_128
    layout = LAYOUT_256 if L == 256 else LAYOUT_128
    res = {}
    # ---- config 2 end to end
    model = build_model(100, 7, flavour='abdesign', device=dev).eval()
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(32, layout).items()}
    opt = {'sample_structure': True, 'sample_sequence': True, 'contig': ''}
    e2e = {}
    for mode, g in (('eager', False), ('graph', True)):
        model.sample(dict(batch), dict(opt, graph=g))                       # (graph: capture + first replay)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.sample(dict(batch), dict(opt, graph=g))
        torch.cuda.synchronize()
        e2e[mode] = round((time.perf_counter() - t0) * 1e3, 2)
    res['sample_e2e_ms'] = min(e2e.values())
    res['sample_e2e_eager_vs_graph_ms'] = e2e
    res['sample_e2e_config'] = (f'model.sample, AbDesign flavour, N=32, L={L}, T=100: encode + pair-bias cache + 100 steps + D2H of the trajectory '
                                '(graph: the loop replayed from a hipGraph captured by an earlier call, inputs copied into its static buffers)')
    model.diffusion.clear_graphs()
    torch.cuda.empty_cache()
    # ---- config 5 training step: eager launches and the whole step replayed from one hipGraph (training.GraphedTrainStep)
    from ab_opt_amd import training
    model.train()
    NT = 16
    tb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(NT, layout).items()}
    adam = training.FusedAdam(model.parameters(), lr=1e-4)       # clip_grad_norm_ + torch.optim.Adam of A/train.py:116-117 as one native call

    def step():
        adam.zero_grad(set_to_none=True)
        loss = sum(model(dict(tb)).values())
        loss.backward()
        adam.step(max_grad_norm=100.0)                              # codesign_single.yml:21
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    iters = 5
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / iters * 1e3
    graph_ms = None
    try:
        gstep = training.GraphedTrainStep(model, adam, tb, max_grad_norm=100.0)
        for _ in range(2):
            gstep(tb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters):
            losses = gstep(tb)
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / iters * 1e3
        assert all(torch.isfinite(v) for v in losses.values())
        gstep.close()
        del gstep
    except Exception as e:                   # the eager number stands on its own
        res['train_graph_error'] = repr(e)
    best = min(eager_ms, graph_ms) if graph_ms is not None else eager_ms
    res['train_step_ms'] = round(best, 2)
    res['train_step_eager_ms'], res['train_step_graph_ms'] = round(eager_ms, 2), (round(graph_ms, 2) if graph_ms is not None else None)
    res['train_step_config'] = f'AbDesign flavour model(batch) fwd + bwd + gradient clipping + Adam (training.FusedAdam), N={NT}, L={L} (encode() inside, as train.py runs it); best of eager / hipGraph replay'
    # SURVEY 8(d), "training (config 5) extra": per sample 285 MB (forward 102 MB + backward 6 z reads + 6 dz writes + 5 dz reads of 16.8 MB) and
    # 3 x (4.15 GFLOP denoiser + 5.3 GFLOP encode) (backward = 2 x forward)
    tb_bytes, tb_flops = NT * 285e6 * (L / 256) ** 2, NT * 3 * (4.15e9 + 5.3e9) * (L / 256) ** 2
    res['train_roofline'] = {'algorithmic_bytes': tb_bytes, 'flops': tb_flops, 'ms': round(best, 2),
                             'hbm_frac': round(tb_bytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             'tflops': round(tb_flops / (best * 1e-3) / 1e12, 2), 'flop_frac': round(tb_flops / (best * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                             'bound': ('fp32 matrix / vector pipes' if tb_flops / (FP32_PEAK_TFLOPS * 1e12) > tb_bytes / (HBM_PEAK_GBS * 1e9) else 'hbm') +
                                      ' (%.2f ms of HBM time against %.2f ms of fp32 FLOPs at peak)' % (tb_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, tb_flops / (FP32_PEAK_TFLOPS * 1e12) * 1e3),
                             'kernel_share': 'profiles/r06_e_train_steady_step_kernel_stats.txt: steady-state share of GPU time in abopt:: kernels (95 %)'}
    model.zero_grad(set_to_none=True)
    model.eval()
    del adam, tb
    torch.cuda.empty_cache()
    # ---- config 3: AbDock poses
    N3, T, K3 = 64, 100, 20
    dpm, state, res_feat, pair_feat, gen, mres = build_workload(dev, N3, L, T, seed=77, abdesign=False)
    run = lambda n: dpm._run(state, T, res_feat, pair_feat, gen, mres, True, False, True, None, 99, 0, False, stop_after=n)
    run(3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(K3)
    torch.cuda.synchronize()
    res['config3_sample_steps_per_s'] = round(N3 * K3 / (time.perf_counter() - t0), 1)
    res['config3_config'] = f'AbDock dock_single model block (prmsd head, pred_x0), structure-only sampling, N={N3} poses, L={L}, {K3} timed steps'
    del dpm, state, res_feat, pair_feat
    torch.cuda.empty_cache()
    # ---- config 4, ONE rank's leg at its own size: 8 complexes x 16 samples, L = 256, T = 100, through the test-set driver (encode of
    # the 8 complexes, ONE grouped launch per step for the 128 samples, per-complex commonness ranking and DockQ on the device)
    try:
        from ab_opt_amd import sampler
        m4 = build_model(100, 7, flavour='abdesign', device=dev).eval()
        cx = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(1, layout, seed=500 + c).items()} for c in range(8)]
        leg = {}
        for name, per in (('grouped', 8), ('per_complex', 1)):
            sampler.design_testset_sharded(m4, cx, 16, k=3, seed=3, native=True, complexes_per_launch=per)        # warm (and, for the sampler's graph mode, the eager first call)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out4 = sampler.design_testset_sharded(m4, cx, 16, k=3, seed=3, native=True, complexes_per_launch=per)
            torch.cuda.synchronize()
            leg[name] = time.perf_counter() - t0
            assert len(out4) == 8 and all(torch.isfinite(r['ca']).all() for r in out4)
        res['config4_rank_leg_sample_steps_per_s'] = round(8 * 16 * 100 / leg['grouped'], 1)
        res['config4_rank_leg_s'] = {k: round(v, 4) for k, v in leg.items()}
        res['config4_config'] = (f'sampler.design_testset_sharded on one rank: 8 complexes x 16 samples, L={L}, T=100, AbDesign flavour; wall time of the WHOLE leg '
                                 '(encode, 100 steps, trajectory hand-over, per-complex commonness ranking, backbone reconstruction + DockQ of all 128 designs); '
                                 'grouped = one launch of 128 samples sharing pair features per complex (pair_feat_shared=16), per_complex = eight launches of 16 '
                                 '(the round-3 path); value = 8*16*100 / grouped wall time')
        m4.diffusion.clear_graphs()
        del m4, cx, out4
        torch.cuda.empty_cache()
    except Exception as e:
        res['config4_error'] = repr(e)
    # ---- the reference's own headline invocation (AbDock/README.md:61: dock_pdb.py -n 1000 -b 1000 with configs/test/dock_cdr.yml): 1000 poses of
    # ONE complex cropped to the CDR-H3 + 20 antigen residues (dock_single.yml:12-14 antigen_size 20, initial_patch_size 0) -> L ~ 30..48
    res.update(poses_measurement(dev, 1000, 48, 'poses1000'))
    # ---- config 3 as BASELINE.json words it ("64 poses/complex"): 64 poses of ONE 256-residue complex, i.e. shared pair features (the number above
    # gives every pose its own pair features, the conservative reading); with shared features z is cache-resident and the fp16 term forms of round 6 run
    res.update(poses_measurement(dev, 64, L, 'config3_one_complex'))
    # ---- small batches at L=256 (latency-bound regime)
    res.update(poses_measurement(dev, 8, 256, 'n8_L256', abdesign=True))
    return res


def poses_measurement(dev, N, L, key, abdesign=False, K=20):
    """N poses of one complex of L residues (shared context: pair_feat (1,L,L,C)), AbDock dock flavour, structure-only sampling."""
    T = 100
    if abdesign:
        dpm, state, res_feat, pair_feat, gen, mres = build_workload(dev, N, L, T, seed=79, abdesign=True)
        flags = (True, True)
    else:
        dpm, state, res_feat, pair_feat, gen, mres = build_workload(dev, N, L, T, seed=78, abdesign=False, shared_context=True, cdrs=[(8, min(26, L))])
        flags = (True, False)
    run = lambda n, graph: dpm._run(state, T, res_feat, pair_feat, gen, mres, flags[0], flags[1], True, None, 99, 0, False, stop_after=n, graph=graph)
    out = {}
    run(3, False)
    for mode, graph in (('eager', False), ('graph', True)):
        run(K, graph)                                   # (graph: capture + first replay)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(K, graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[mode] = (round(N * K / dt, 1), round(dt / K * 1e3, 4))
    best = max(out.values())
    Nz = pair_feat.shape[0]
    alg = NUM_LAYERS * (Nz * 256 * L * L + N * 1076 * L) + 13.0e6          # SURVEY 8(d) with z counted once per DISTINCT complex
    return {f'{key}_sample_steps_per_s': best[0], f'{key}_ms_per_step': best[1], f'{key}_eager_vs_graph': out,
            f'{key}_step_hbm_frac': (round(alg / (best[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if Nz == N else None),   # (one shared complex: z lives in L2 / MALL, HBM is not the bound)
            f'{key}_step_tflops': round(step_flops(N, L) / (best[1] * 1e-3) / 1e12, 2),
            f'{key}_config': (f'{"AbDesign codesign" if abdesign else "AbDock dock_single (prmsd head, pred_x0), structure-only"} sampling loop, '
                              f'N={N}, L={L}, {"one complex shared by all poses" if Nz == 1 else "distinct complexes"}, {K} timed steps; '
                              'step_hbm_frac = SURVEY 8(d) bytes (z once per distinct complex) / time / 8 TB/s; step_tflops = 8(d) FLOPs / time')}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command under torch.distributed.run with N ranks on 127.0.0.1 (a free port),
    stdout / stderr inherited.  Returns the launcher's exit code."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] --gpus %d without WORLD_SIZE: launching %s' % (n, ' '.join(cmd[1:8])), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def log(*a):
    print('[bench %.1fs]' % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=7, help='the K-step timed region is run this many times; the median is reported')
    ap.add_argument('--batch', type=int, default=32, help='samples per GPU')
    ap.add_argument('--length', type=int, default=256)
    ap.add_argument('--graph', choices=('on', 'off'), default='on', help='replay the K-step loop from a hipGraph captured before the timed region')
    ap.add_argument('--graph-events', action='store_true', help='(experiment) record the per-launch HIP events inside the captured graph')
    ap.add_argument('--no-prof', action='store_true', help='no HIP events around the IPA-core launches (A/B of their cost)')
    ap.add_argument('--replay-only', action='store_true', help='skip the eager_events and two_launch_form passes: under rocprofv3 every dominant-kernel launch is then a replayed one (profiles/*_kernel_stats_replay_only.txt)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip train_step_ms / sample_e2e_ms / config3 / poses1000 (N=1 only)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # the plain command shape `python bench.py --gpus N ...`: launch the N ranks ourselves (one process per GPU over RCCL, the command of
        # the module docstring) and hand their output through -- rank 0 alone prints the JSON line
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; start it as `python bench.py --gpus {args.gpus} ...` '
                 f'(self-launching) or under torch.distributed.run with --nproc-per-node {args.gpus}')
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    ndev = torch.cuda.device_count()
    local_dev = local % ndev                   # more ranks than devices (the 2-ranks-on-one-GPU test of this path): ranks share a device
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    dist, backend = None, None
    # ABOPT_BENCH_FORCE_DIST=1: initialise the process group (and run the gather / max-over-ranks code) even with ONE rank, so the RCCL
    # branch executes on a 1-GPU box (tests/test_hip_parity.py::test_bench_nccl_branch_single_rank)
    if world > 1 or os.environ.get('ABOPT_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        # RCCL needs one device per rank; with fewer devices than ranks the (tiny) exchange goes through gloo and host memory
        backend = os.environ.get('ABOPT_BENCH_BACKEND', 'nccl' if ndev >= world else 'gloo')
        if world == 1 and 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'), RANK='0', WORLD_SIZE='1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from ab_opt_amd import hip
    from ab_opt_amd.sampler import all_gather_candidates
    hip.lib()
    N, L, T, K, W, R = args.batch, args.length, 100, args.steps, args.warmup, max(1, args.repeats)
    assert 1 <= K <= T and W <= T
    use_graph = args.graph == 'on'
    log('building workload', dict(N=N, L=L, T=T, K=K, W=W, repeats=R, graph=use_graph), hip.device_info())
    dpm, state, res_feat, pair_feat, gen, mres = build_workload(dev, N, L, T, seed=2022 + rank)
    log('workload ready')
    run = lambda n, graph=False: dpm._run(state, T, res_feat, pair_feat, gen, mres, True, True, True, None, 1234 + rank, rank * N * L, False,
                                          stop_after=n, graph=graph)

    # the network outputs of the first step (same launch geometry as the timed steps), kept for the oracle check below
    first = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        beta = dpm.trans_pos.var_sched.betas[T].expand([N]).contiguous()
        first = hip.eps_net_forward(dpm.eps_net.packed(), state[0], state[1] / 10.0, state[2], res_feat, pair_feat, beta, gen, mres, False, 0, False,
                                    pair_bias_cache=hip.pair_bias_cache(dpm.eps_net.encoder.packed_array(), NUM_LAYERS, pair_feat),
                                    pair_terms=hip.pair_terms(pair_feat) if dpm._pair_terms_wanted(N, L, N, dev) else None)     # (what the timed loop runs)
        first = {k: (v.clone() if v is not None else None) for k, v in first.items()}
    if W > 0:
        run(W)
    torch.cuda.synchronize()
    graph_events = False
    if use_graph:
        # capture (outside the timed region, like the warmup): K steps of the loop, the one-off pair-bias cache build included
        t_cap = time.perf_counter()
        hip.GRAPH_CAPTURE_EVENTS = bool(args.graph_events and not args.no_prof)
        hip.GRAPH_CAPTURE_SPANS = not args.no_prof            # in-kernel launch spans: the dominant kernel timed inside the replayed graph
        run(K, graph=True)
        hip.GRAPH_CAPTURE_SPANS = False
        torch.cuda.synchronize()
        graph_events = hip.GRAPH_CAPTURE_EVENTS
        log('graph captured + first replay: %.3f s' % (time.perf_counter() - t_cap))
    log('warmup done')

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_pass(graph, events):
        """One K-step timed region: barrier + synchronize on both sides, max over ranks."""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        if not graph:
            hip.prof_enable(events)
        t0 = time.perf_counter()
        tv, tp, ts, _, _ = run(K, graph=graph)
        if dist is not None:
            # batched-sampling reduction: gather generated-residue CA candidates of every rank (design_for_pdb.py:326-336)
            cand = tp[T - K][gen].reshape(N, -1, 3).contiguous()
            all_gather_candidates(cand)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(tp[T - K]).all()
        local_times.append(dt)                 # this rank's own clock (the reported time is the MAX over ranks)
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # ---- the timed region, R times.  Eager mode: HIP events around every IPA-core launch inside it.  Graph mode: host-recorded
    # event pairs cannot live in a replayed graph, so the kernel is timed in ONE extra eager pass of the same K steps right after
    # the repeats (same process, same box, same clocks), unless --graph-events put the records into the graph itself.
    times, launches, ipa_ms, local_times = [], 0, 0.0, []
    for r in range(R):
        times.append(timed_pass(use_graph, events=not args.no_prof))
        if not use_graph and not args.no_prof:
            n_, ms_ = hip.prof_collect()
            launches, ipa_ms = launches + n_, ipa_ms + ms_
        elif graph_events:
            n_, ms_ = hip.prof_collect(keep=True)
            launches, ipa_ms = launches + n_, ipa_ms + ms_
    hip.prof_enable(False)
    log('timed region x%d: %s ms per step' % (R, ', '.join('%.4f' % (t / K * 1e3) for t in times)))
    # who took part: every rank's device and its OWN median step time, gathered through the process group the timing used -- a SCALE line
    # can be checked at a glance (all ranks present, one device each, nobody far off the others)
    ranks_seen = None
    if dist is not None:
        prop = torch.cuda.get_device_properties(dev)
        mine = dict(rank=rank, device=local_dev, name=prop.name, pci_bus_id=getattr(prop, 'pci_bus_id', None), uuid=str(getattr(prop, 'uuid', '')) or None,
                    ms_per_step=round(sorted(local_times[:R])[R // 2] / K * 1e3, 4))
        ranks_seen = [None] * dist.get_world_size()
        dist.all_gather_object(ranks_seen, mine)
    instrumented, eager_events = None, None
    if use_graph and not graph_events and not args.no_prof:
        # (i) one more replay of the SAME graph, its dominant-kernel launches timed by their in-kernel wall-clock spans (first workgroup in,
        #     last workgroup out: what rocprofv3 reports) -- host-recorded events cannot sit in a replayed graph, and an eager pass is not the
        #     same thing: its launch gaps let the chip clock up (kernels 5-10 % faster than in the back-to-back replay on a power-limited box)
        hip.prof_spans_reset()
        dt_i = timed_pass(True, events=False)
        launches, ipa_ms = hip.prof_spans()
        instrumented = round(dt_i / K * 1e3, 4)
        log('instrumented graph replay: %.4f ms per step, %d launches, %.1f us per launch' % (instrumented, launches, ipa_ms / max(launches, 1) * 1e3))
        # (ii) the eager pass with HIP events around every launch (round 3's method), kept next to it
        n_e, ms_e = 0, 0.0
        if not args.replay_only or launches == 0:
            dt_e = timed_pass(False, events=True)
            n_e, ms_e = hip.prof_collect()
            hip.prof_enable(False)
            eager_events = dict(ms_per_step=round(dt_e / K * 1e3, 4), avg_launch_ms=round(ms_e / max(n_e, 1), 4), launches=n_e)
        if launches == 0:                                   # (a shape that does not take the 32-row kernels: events are all there is)
            launches, ipa_ms, instrumented = n_e, ms_e, eager_events['ms_per_step']
    clock = hip.prof_clock()                                    # the clock wave 0 / workgroup 0 of the last dominant-kernel launch ran at
    # the WHOLE loop the metric names (all T steps from noise to t = 1, one graph replay each), next to the K-step headline: the headline's K
    # steps are t = T .. T-K+1 and carry the one-off pair-bias cache build, so a reader can tell how the 20-step line relates to a full call
    loop_full = None
    if rank == 0 and world == 1 and use_graph and K < T and not args.no_prof:
        run(T, graph=True)                                      # capture + first replay
        torch.cuda.synchronize()
        lt = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(T, graph=True)
            torch.cuda.synchronize()
            lt.append((time.perf_counter() - t0) / T * 1e3)
        loop_full = dict(loop100_ms_per_step=round(sorted(lt)[1], 4), loop100_sample_steps_per_s=round(N / (sorted(lt)[1] * 1e-3), 1), repeats_ms=[round(x, 4) for x in lt],
                         note=f'all T={T} steps of the sampler (t = {T} .. 1) replayed from one hipGraph, pair-bias cache build included once; median of 3')
    # the two-launch form of a block (32-row core, then the tail kernel) in one more eager pass: the IPA core ALONE, for continuity with
    # the rounds before the tail was fused into it (bit-identical results; not part of the timed region)
    two_launch = None
    if rank == 0 and world == 1 and not args.no_prof and not args.replay_only and os.environ.get('ABOPT_FUSE_TAIL') is None:
        os.environ['ABOPT_FUSE_TAIL'] = '0'
        try:
            run(2)
            dt_u = timed_pass(False, events=True)
            n_u, ms_u = hip.prof_collect()
            hip.prof_enable(False)
            two_launch = dict(ms_per_step_eager=round(dt_u / K * 1e3, 4), ipa_core_avg_launch_ms=round(ms_u / max(n_u, 1), 4), launches=n_u)
        finally:
            del os.environ['ABOPT_FUSE_TAIL']
    # what this box gives: device-to-device copy rate (read + write bytes) and the one-off cost of a sample() call that the K steps share
    box = None
    if rank == 0:
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev)           # 1 GiB
        b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            b.copy_(a)
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * a.numel() * 4 / (time.perf_counter() - t0) / 1e9
        del a, b
        arr = dpm.eps_net.encoder.packed_array()
        hip.pair_bias_cache(arr, NUM_LAYERS, pair_feat); torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.pair_bias_cache(arr, NUM_LAYERS, pair_feat); torch.cuda.synchronize()
        cache_ms = (time.perf_counter() - t0) * 1e3
        box = dict(copy_GBs=round(copy_gbs, 1), copy_note='1 GiB device-to-device copy, read + write bytes / time',
                   dominant_kernel_clock_GHz=round(clock[2], 3) if clock else None,
                   dominant_kernel_wave_us=round(clock[1] * 1e6, 1) if clock else None,
                   clock_note='shader cycles / 100 MHz wall ticks of wave 0, workgroup 0 of the last dominant-kernel launch (abopt_prof_clock)',
                   one_off_ms={'pair_bias_cache': round(cache_ms, 3),
                               'note': 'built once per sample() call (pair_feat and the weights are constant over the T steps); it IS inside every '
                                       'timed K-step region here (conservative: a 100-step call amortises it over 100 steps, this run over K)'})

    if rank == 0:
        ts_sorted = sorted(times)
        dt = ts_sorted[len(ts_sorted) // 2] if R % 2 else 0.5 * (ts_sorted[R // 2 - 1] + ts_sorted[R // 2])
        per_launch_ms = ipa_ms / max(launches, 1)
        alg = ipa_algorithmic_bytes(N, L)
        ach = alg / (per_launch_ms * 1e-3) / 1e9 if launches else 0.0
        step_s = dt / K
        traffic, traffic_src = MEASURED_TRAFFIC.get((N, L), (None, None))
        timing = ('HIP events on the launch stream around every ipa_core launch, all %d repeats of the timed region' % R) if not use_graph else (
            'HIP event records captured into the replayed graph around every ipa_core launch' if graph_events else
            'in-kernel launch spans (abopt_prof_spans: 100 MHz wall clock of the first workgroup in / last workgroup out of every launch) of ONE more '
            'replay of the timed graph, run right after the repeats (it took instrumented_ms_per_step); eager_events = the same K steps launched '
            'eagerly with HIP events around every launch, for comparison')
        # in-kernel spans exist for the 32-row launches only, and the sampler's 32-row launch is the fused core + tail kernel unless ABOPT_FUSE_TAIL=0
        fused = use_graph and instrumented is not None and os.environ.get('ABOPT_FUSE_TAIL') != '0' and hip.lib().abopt_pair_terms_used(N, L, 0) == 1
        kernel_name = ('ipa_core32_kernel<true, *>: IPA core + block tail (out_transform, LayerNorm, MLP, LayerNorm) as ONE kernel -- the survey\'s per-layer bytes are '
                       'those of the whole block, so they apply unchanged' if fused else 'ipa_core')
        line = {
            'metric': 'denoising steps/sec (256-res complex, 100-step sampler)', 'value': round(world * N * K / dt, 2),
            'unit': 'sample-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(step_s * 1e3, 4),
            'repeats': R, 'ms_per_step_min': round(ts_sorted[0] / K * 1e3, 4), 'ms_per_step_max': round(ts_sorted[-1] / K * 1e3, 4),
            'value_min': round(world * N * K / ts_sorted[-1], 2), 'value_max': round(world * N * K / ts_sorted[0], 2),
            'aggregate': 'median of the repeats; each repeat times exactly K steps (barrier + synchronize on both sides, max over ranks)',
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'AbDesign codesign_single model block (F=128, C=64, 6 IPA layers, T=100), L={L}, 6 CDR segments, '
                                   f'batch {N} per GPU, distinct pair features per sample, device Philox RNG',
                       'samples_per_gpu': N, 'residues': L, 'sampler_steps': T, 'parallelism': f'independent samples x{world}',
                       'launch': ('hipGraph replay of the K-step loop (captured once, before the timed region; Philox position from device memory)'
                                  if use_graph else 'eager launches'),
                       'arithmetic': 'fp32 storage and accumulation everywhere.  fp32 x fp32 products as two fp16 terms per operand (three products, 22 significant bits, '
                                     'power-of-two scales) on the fp16 matrix instructions in the dense layers (node projections, out_transform + MLP, heads, mixer: '
                                     'tests/test_hip_parity.py::test_two_term_fp16_products_are_fp32_accurate) and, since round 6 (pair_terms: ' + str(bool(dpm.last_run_info.get('pair_terms'))) + '), in the pair '
                                     'aggregation sum_j alpha z (z pre-split once per call) and the q . k channel part of the attention logits '
                                     '(test_pair_aggregation_on_fp16_terms_vs_fp64, test_attention_logits_on_fp16_terms_vs_fp64); point distances, softmax, value / point aggregation on the fp32 matrix instructions',
                       'backend': backend, 'ranks_per_device': (world + ndev - 1) // ndev if world > 1 else 1,
                       'ranks': ranks_seen, 'ranks_note': None if ranks_seen is None else 'gathered through the process group: device and OWN median ms_per_step of every rank (the line\'s ms_per_step is the max over ranks per repeat)'},
            'roofline': {'bound': 'hbm', 'kernel': kernel_name, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src, 'launches': launches,
                         'avg_launch_ms': round(per_launch_ms, 4), 'rocprofv3_note': 'a rocprofv3 --kernel-trace --stats summary of this command averages the replayed launches (what avg_launch_ms measures) TOGETHER with the launches of the eager_events pass (eager_events.avg_launch_ms each; eager launches run slower than replayed ones: L2 is written back and invalidated around every eager kernel, so the fragments node_frags has just produced come from HBM) and a few warm-up launches', 'timing': timing, 'instrumented_ms_per_step': instrumented, 'eager_events': eager_events,
                         'algorithmic_bytes_per_launch': alg,
                         'algorithmic_bytes_formula': 'N*(256*L^2 + 1076*L)  [SURVEY 8(d)]',
                         'kernel_io_bytes_per_launch': ipa_kernel_io_bytes(N, L),
                         'kernel_io_frac': round(ipa_kernel_io_bytes(N, L) / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if launches else 0.0,
                         'step': {'algorithmic_bytes': step_algorithmic_bytes(N, L), 'hbm_frac': round(step_algorithmic_bytes(N, L) / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                  'flops': step_flops(N, L), 'tflops': round(step_flops(N, L) / step_s / 1e12, 2),
                                  'fp32_peak_tflops': FP32_PEAK_TFLOPS, 'flop_frac': round(step_flops(N, L) / step_s / 1e12 / FP32_PEAK_TFLOPS, 4),
                                  'dominant_kernel_share_of_step': (round(per_launch_ms * NUM_LAYERS / instrumented, 4) if (launches and instrumented) else
                                                                    (round(per_launch_ms * NUM_LAYERS / (step_s * 1e3), 4) if launches else None)),
                                  'share_note': 'kernel time and step time of the SAME (instrumented, eager) pass'}},
            'box': box,
        }
        if loop_full is not None:
            line['loop100_ms_per_step'] = loop_full['loop100_ms_per_step']
            line['loop100'] = loop_full
        elif K == T:
            line['loop100_ms_per_step'] = round(step_s * 1e3, 4)
        if two_launch:
            u = two_launch['ipa_core_avg_launch_ms']
            two_launch.update(ipa_core_frac=round(alg / (u * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              note='ABOPT_FUSE_TAIL=0: the 32-row IPA core as its own kernel (HIP events, one eager pass of the same K steps); '
                                   'same arithmetic, bit-identical results -- the timed region above runs the fused form')
            line['roofline']['two_launch_form'] = two_launch
        if world == 1 and not args.no_cpu_baseline:
            log('cpu baseline on', os.cpu_count(), 'cores ...')
            check = (dpm, T, state, res_feat, pair_feat, gen, mres, first, [0, N // 2 + 1] if N > 2 else [0]) if first is not None else None
            line['cpu_baseline'] = cpu_baseline(L, T, check=check)
        if world == 1 and not args.no_secondary:
            log('secondary configs ...')
            dpm.clear_graphs()
            torch.cuda.empty_cache()
            try:
                line['secondary'] = secondary_measurements(dev, L)
            except Exception as e:           # never lose the headline line to a secondary measurement
                line['secondary'] = {'error': repr(e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

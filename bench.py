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
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy ceiling)
# HBM bytes per ipa_core launch from the PMC counters (separate rocprofv3 --pmc passes on this command, FETCH_SIZE doubled per
# the gfx950 correction): profiles/r01_g_pmc_ipa_core.txt.  745 MB read + 60 MB written; 142 MB above the algorithmic
# bytes = the per-call pair-bias cache stream that replaces the in-kernel pair-bias contraction.
MEASURED_TRAFFIC = {(32, 256): 363911.5 * 1024 * 2 + 58368.1 * 1024}


def ipa_core_bytes(N, L, C=64):
    """Algorithmic HBM bytes of ONE launch of the IPA-core kernel (DESIGN.md 'Roofline accounting'):
    z once + node projections in (2016 floats/residue) + features out (1824) + frames/mask (52 B)."""
    return N * (4 * C * L * L + (2016 * 4 + 1824 * 4 + 52) * L)


def build_workload(dev, N, L, T, seed):
    from conftest import AttrDict
    import cases
    from ab_opt_amd.dpm import FullDPM
    from ab_opt_amd.utils import synth
    dpm = FullDPM(128, 64, num_steps=T, eps_net_opt=dict(num_layers=6), _abdesign=True).eval()
    synth.fill_module_(dpm, seed=2)
    dpm = dpm.to(dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    layout = synth.LAYOUT_256 if L == 256 else synth.LAYOUT_128
    gen = torch.zeros(N, L, dtype=torch.bool, device=dev)
    for a, b in layout['cdrs']:
        gen[:, a:b] = True
    mres = torch.ones(N, L, dtype=torch.bool, device=dev)
    res_feat = torch.randn(N, L, 128, device=dev, generator=g)
    pair_feat = torch.randn(N, L, L, 64, device=dev, generator=g)          # distinct per sample: no replication shortcut
    v = torch.randn(N, L, 3, device=dev, generator=g)
    p = torch.randn(N, L, 3, device=dev, generator=g) * 10
    s = torch.randint(0, 20, (N, L), device=dev, generator=g)
    return dpm, (v, p, s), res_feat, pair_feat, gen, mres


def cpu_baseline(L, T, budget_s=45.0):
    """The oracle ('port' of the reference, test infrastructure) on the host cores: a bounded sample of the same
    workload.  Threads are capped at 32: the op mix is elementwise/bandwidth bound and slows down badly beyond that on
    many-core hosts; `cores` reports the threads actually used."""
    from oracle import dpm as odpm
    from ab_opt_amd.dpm import FullDPM
    from ab_opt_amd.utils import synth
    import cases
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    m = FullDPM(128, 64, num_steps=10, eps_net_opt=dict(num_layers=6), _abdesign=True).eval()
    synth.fill_module_(m, seed=2)
    sch = odpm.variance_schedule(10)
    den = odpm.Denoiser(m.state_dict(), num_steps=10, variant='abdesign', pre='', mode='ref',
                        tables=(None, odpm.igso3_tables(sch['sigmas'].tolist())))
    best, detail, t_all = 0.0, {}, time.perf_counter()
    for mode, N, reps in (('mm', 1, 4), ('ref', 1, 2), ('mm', 4, 2)):
        if time.perf_counter() - t_all > budget_s:
            detail[f'{mode}_N{N}'] = 'skipped (time budget)'
            continue
        den.mode = mode
        v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [L] * N, [(25, 33), (51, 57), (94, 106)], num_steps=10, t=7)
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
    return dict(value=round(best, 3), unit='sample-steps/s', cores=threads, kind='port',
                sample=f'oracle Denoiser.step (EpsilonNet + transitions) at L={L}, {threads} threads: N=1 with matmul contractions, '
                       f'N=1 in the reference op order, N=4 with matmul contractions (a few steps each, {budget_s:.0f}s budget); '
                       f'best rate reported; per-variant: {detail}')


def log(*a):
    print('[bench %.1fs]' % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='samples per GPU')
    ap.add_argument('--length', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)

    from ab_opt_amd import hip
    hip.lib()
    N, L, T, K, W = args.batch, args.length, 100, args.steps, args.warmup
    assert 1 <= K <= T and W <= T
    log('building workload', dict(N=N, L=L, T=T, K=K, W=W), hip.device_info())
    dpm, state, res_feat, pair_feat, gen, mres = build_workload(dev, N, L, T, seed=2022 + rank)
    log('workload ready')
    run = lambda n: dpm._run(state, T, res_feat, pair_feat, gen, mres, True, True, True, None, 1234 + rank, rank * N * L, False, stop_after=n)

    if W > 0:
        run(W)
    torch.cuda.synchronize()
    log('warmup done')
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    hip.prof_enable(True)
    t0 = time.perf_counter()
    tv, tp, ts, _, _ = run(K)
    if dist is not None:
        # batched-sampling reduction: gather generated-residue CA candidates of every rank (design_for_pdb.py:326-336)
        cand = tp[T - K][gen].reshape(N, -1, 3).contiguous()
        allc = [torch.empty_like(cand) for _ in range(world)]
        dist.all_gather(allc, cand)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    log('timed region done: %.3f s' % dt)
    launches, ipa_ms = hip.prof_collect()
    hip.prof_enable(False)
    assert torch.isfinite(tp[T - K]).all()

    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        per_launch_ms = ipa_ms / max(launches, 1)
        ach = ipa_core_bytes(N, L) / (per_launch_ms * 1e-3) / 1e9 if launches else 0.0
        line = {
            'metric': 'denoising steps/sec (256-res complex, 100-step sampler)', 'value': round(world * N * K / dt, 2),
            'unit': 'sample-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(dt / K * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'AbDesign codesign_single model block (F=128, C=64, 6 IPA layers, T=100), L={L}, 6 CDR segments, '
                                   f'batch {N} per GPU, distinct pair features per sample, device Philox RNG',
                       'samples_per_gpu': N, 'residues': L, 'sampler_steps': T, 'parallelism': f'independent samples x{world}'},
            'roofline': {'bound': 'hbm', 'kernel': 'ipa_core', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': MEASURED_TRAFFIC.get((N, L)), 'launches': launches,
                         'avg_launch_ms': round(per_launch_ms, 4), 'algorithmic_bytes_per_launch': ipa_core_bytes(N, L)},
        }
        if world == 1 and not args.no_cpu_baseline:
            log('cpu baseline on', os.cpu_count(), 'cores ...')
            line['cpu_baseline'] = cpu_baseline(L, T)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

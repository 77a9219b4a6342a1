#!/usr/bin/env python
"""Generate the golden fixtures by running the REFERENCE (pengzhangzhi/ab_opt) in this container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz / *.json

Runs only where /root/reference exists (the build container).  Nothing here is imported by the
tests; the tests read the .npz files.  Weights are the integer-hash fill of
ab_opt_amd.utils.synth.fill_module_, inputs come from tests/cases.py, so only outputs are stored.
Random draws made by the reference are captured by wrapping the torch RNG entry points and are
stored with the outputs, so the build can replay ("teacher-force") the same noise.
"""
import os, sys, json, types, importlib.machinery, contextlib
from unittest import mock
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'

from ab_opt_amd.utils import synth      # noqa: E402
import cases                             # noqa: E402


class AttrDict(dict):
    """Minimal EasyDict stand-in (easydict is not installed here)."""
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, len(out), 'arrays')


# ------------------------------------------------------------------ RNG capture
class Tape:
    NAMES = ['randn', 'randn_like', 'rand_like', 'multinomial', 'randint_like', 'randint']

    def __init__(self):
        self.log = []
        self.cat_probs = []        # inputs of every 20-class torch.multinomial call (the categorical the reference samples from)

    @contextlib.contextmanager
    def recording(self):
        orig = {n: getattr(torch, n) for n in self.NAMES}

        def wrap(n):
            def f(*a, **k):
                out = orig[n](*a, **k)
                self.log.append((n, out.clone()))
                if n == 'multinomial' and a[0].shape[-1] == 20:
                    self.cat_probs.append(a[0].detach().clone())
                return out
            return f
        with contextlib.ExitStack() as st:
            for n in self.NAMES:
                st.enter_context(mock.patch.object(torch, n, wrap(n)))
            yield self

    def pop(self, kind):
        n, v = self.log.pop(0)
        assert n == kind, (n, kind)
        return v


def abdock_model(num_steps, seed):
    sys.path.insert(0, os.path.join(REF, 'AbDock'))
    from src.models import get_model
    torch.manual_seed(0)
    m = get_model(AttrDict(cases.cfg_abdock(num_steps))).eval()
    synth.fill_module_(m, seed=seed)
    return m


def abdesign_fulldpm(num_steps, seed):
    """AbDesign's FullDPM needs a few absent third-party modules stubbed at import time only."""
    import importlib.abc

    class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        ROOTS = ('easydict', 'torch_scatter', 'lmdb', 'abnumber', 'wandb', 'Bio', 'joblib')

        def find_spec(self, name, path, target=None):
            if name.split('.')[0] in self.ROOTS:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            mm = mock.MagicMock(name=spec.name)
            mm.__path__, mm.__spec__, mm.__name__ = [], spec, spec.name
            return mm

        def exec_module(self, module):
            pass
    if not any(type(f).__name__ == '_StubFinder' for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())
    sys.path.insert(0, os.path.join(REF, 'AbDesign'))
    from diffab.modules.diffusion.dpm_full import FullDPM
    # AbDesign's local_to_global lost its `.view(N, L, -1, 3)` (AbDesign/diffab/modules/common/geometry.py:86-88)
    # and raises on the 5-D value points of ga.py:131, so the AbDesign IPA path cannot execute as shipped.
    # Run it with the sibling tree's intact function (AbDock/src/modules/common/geometry.py:72-91), which is
    # the only difference between the two geometry.py files.  Documented in DESIGN.md ("AbDesign parity").
    import diffab.modules.encoders.ga as _ga_mod
    from src.modules.common.geometry import local_to_global as _l2g
    import diffab.modules.common.geometry as _geo_mod
    _ga_mod.local_to_global = _l2g
    _geo_mod.local_to_global = _l2g       # apply_rotation_to_vector (dpm_full.py:89) resolves it at call time
    torch.manual_seed(0)
    m = FullDPM(128, 64, num_steps=num_steps, eps_net_opt=dict(num_layers=6)).eval()
    synth.fill_module_(m, seed=seed)
    return m


# ------------------------------------------------------------------ cases
def case_so3():
    from src.modules.common import so3, geometry
    w = cases.SO3_EDGE_W
    R = so3.so3vec_to_rotation(w)
    with torch.no_grad():
        lg_nograd = so3.rotation_to_so3vec(R)
    with torch.enable_grad():
        lg_grad = so3.rotation_to_so3vec(R.clone().requires_grad_(True)).detach()
    q = synth.hash_tensor((16, 4), 7, scale=2.0)
    e = synth.hash_tensor((16, 3), 8, scale=3.0)
    save('so3', exp=R, log_nograd=lg_nograd, log_grad=lg_grad,
         quat=geometry.quaternion_to_rotation_matrix(q), quat1ijk=geometry.quaternion_1ijk_to_rotation_matrix(e))


def case_ga_block():
    from src.modules.encoders.ga import GABlock
    blk = synth.fill_module_(GABlock(128, 64), seed=1).eval()
    R, t, x, z, mask = cases.ipa_inputs(2, 24, [24, 19])
    with torch.no_grad():
        ln, lp, ls = blk._node_logits(x), blk._pair_logits(z), blk._spatial_logits(R, t, x)
        from src.modules.encoders.ga import _alpha_from_logits
        alpha = _alpha_from_logits((ln + lp + ls) * np.sqrt(1 / 3), mask)
        feat = torch.cat([blk._pair_aggregation(alpha, z), blk._node_aggregation(alpha, x),
                          blk._spatial_aggregation(alpha, R, t, x)], dim=-1)
        out = blk(R, t, x, z, mask)
    save('ga_block', l_node=ln, l_pair=lp, l_spat=ls, alpha=alpha, feat=feat, out=out)
    # a bigger, output-only case with a fully-masked sample tail
    R, t, x, z, mask = cases.ipa_inputs(2, 128, [128, 101], salt=150)
    with torch.no_grad():
        save('ga_block_L128', out=blk(R, t, x, z, mask))


def case_eps_net():
    m = abdock_model(100, seed=2)
    net = m.diffusion.eps_net
    for tag, (N, L, lens, gr) in dict(small=(2, 40, [40, 33], [(5, 14), (22, 30)]),
                                      L128=(1, 128, [128], [(30, 42)])).items():
        args = cases.eps_inputs(N, L, lens, gr)
        with torch.no_grad():
            v_next, R_next, eps_pos, c, pl = net(*args)
        save(f'eps_net_abdock_{tag}', v_next=v_next, R_next=R_next, eps_pos=eps_pos, c=c, prmsd_logits=pl)
    d = abdesign_fulldpm(100, seed=2)
    args = cases.eps_inputs(2, 40, [40, 33], [(5, 14), (22, 30)])
    with torch.no_grad():
        v_next, R_next, eps_pos, c = d.eps_net(*args)
    save('eps_net_abdesign_small', v_next=v_next, R_next=R_next, eps_pos=eps_pos, c=c)
    keys = {k: list(v.shape) for k, v in d.state_dict().items()}
    json.dump(keys, open(os.path.join(HERE, 'state_dict_abdesign_fulldpm.json'), 'w'), indent=0)


def case_schedule_tables():
    m = abdock_model(100, seed=2)
    sd = m.state_dict()
    vs = {k.split('.')[-1]: v for k, v in sd.items() if k.startswith('diffusion.trans_pos.var_sched.')}
    out = dict(vs)
    for d in ['fwd', 'inv']:
        pre = f'diffusion.trans_rot.angular_distrib_{d}.'
        Y = sd[pre + 'Y']
        out[f'{d}_stddevs'] = sd[pre + 'stddevs']
        out[f'{d}_approx_flag'] = sd[pre + 'approx_flag']
        out[f'{d}_X0'] = sd[pre + 'X'][0]
        out[f'{d}_Ysum'] = Y.double().sum(1)
        out[f'{d}_Ysub'] = Y[:, ::64]
        out[f'{d}_Yrow50'] = Y[50]
    save('schedule_T100', **out)
    keys = {k: list(v.shape) for k, v in sd.items()}
    json.dump(keys, open(os.path.join(HERE, 'state_dict_abdock.json'), 'w'), indent=0)
    m10 = abdock_model(10, seed=2)
    sd = m10.state_dict()
    out = {k.split('.')[-1]: v for k, v in sd.items() if k.startswith('diffusion.trans_pos.var_sched.')}
    for d in ['fwd', 'inv']:
        pre = f'diffusion.trans_rot.angular_distrib_{d}.'
        out[f'{d}_stddevs'] = sd[pre + 'stddevs']
        out[f'{d}_Ysub'] = sd[pre + 'Y'][:, ::16]
        out[f'{d}_Ysum'] = sd[pre + 'Y'].double().sum(1)
    save('schedule_T10', **out)


def _noise_from_tape(tape, N, L, T, with_init=True):
    noise = {}
    if with_init:
        noise['init_q4'] = tape.pop('randn')
        noise['init_p'] = tape.pop('randn_like')
        noise['init_s'] = tape.pop('randint_like')
    for t in range(T, 0, -1):
        noise[f't{t}_axis'] = tape.pop('randn')
        noise[f't{t}_bin'] = tape.pop('multinomial').reshape(N, L)
        noise[f't{t}_ubin'] = tape.pop('rand_like').reshape(N, L)
        noise[f't{t}_gauss'] = tape.pop('randn_like').reshape(N, L)
        noise[f't{t}_z'] = tape.pop('randn_like')
        noise[f't{t}_s_next'] = tape.pop('multinomial').reshape(N, L)
    assert not tape.log, len(tape.log)
    return noise


def case_trajectory():
    """BASELINE config 1 shape: N=2, L=128 (64 Ab + 64 Ag), one CDR, 10 steps, full model.sample()."""
    T = 10
    m = abdock_model(T, seed=3)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=2022, lengths=[128, 117])
    torch.manual_seed(2022)
    tape = Tape()
    with tape.recording():
        traj = m.sample({k: v.clone() for k, v in batch.items()},
                        sample_opt=dict(sample_structure=True, sample_sequence=True, contig=''))
    out = _noise_from_tape(tape, 2, 128, T)
    for t in range(T, -1, -1):
        e = traj[t]
        out[f'traj{t}_v'], out[f'traj{t}_p'], out[f'traj{t}_s'] = e[0], e[1], e[2]
        if t < T:
            out[f'traj{t}_prmsd'], out[f'traj{t}_ppl'] = e[3], e[4]
    with torch.no_grad():
        rf, pf, R0, p0 = m.encode({k: v.clone() for k, v in batch.items()}, True, True)
    out.update(res_feat=rf, pair_feat_sub=pf[:, ::7, ::5], pair_feat_sum=pf.double().sum((1, 2)), R0=R0, p0=p0)
    save('trajectory_abdock_T10', **out)

    # optimize(): noise to step 4 then denoise
    torch.manual_seed(11)
    tape = Tape()
    with tape.recording():
        traj = m.optimize({k: v.clone() for k, v in batch.items()}, 4,
                          optimize_opt=dict(sample_structure=True, sample_sequence=True))
    o = {}
    o['rot_axis'] = tape.pop('randn'); o['rot_bin'] = tape.pop('multinomial').reshape(2, 128)
    o['rot_ubin'] = tape.pop('rand_like').reshape(2, 128); o['rot_gauss'] = tape.pop('randn_like').reshape(2, 128)
    o['pos'] = tape.pop('randn_like'); o['s_noisy'] = tape.pop('multinomial').reshape(2, 128)
    o.update(_noise_from_tape(tape, 2, 128, 4, with_init=False))
    for t in range(4, -1, -1):
        e = traj[t]
        o[f'traj{t}_v'], o[f'traj{t}_p'], o[f'traj{t}_s'] = e[0], e[1], e[2]
    save('optimize_abdock_T10_k4', **o)


def case_structonly():
    """BASELINE config 3 mode: AbDock pose diffusion, sample_sequence=False (structure-only), every step recorded."""
    T = 10
    m = abdock_model(T, seed=3)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=2022, lengths=[128, 117])
    torch.manual_seed(7)
    tape = Tape()
    with tape.recording():
        traj = m.sample({k: v.clone() for k, v in batch.items()},
                        sample_opt=dict(sample_structure=True, sample_sequence=False, contig=''))
    n2 = {}
    n2['init_q4'] = tape.pop('randn'); n2['init_p'] = tape.pop('randn_like')
    for t in range(T, 0, -1):
        n2[f't{t}_axis'] = tape.pop('randn')
        n2[f't{t}_bin'] = tape.pop('multinomial').reshape(2, 128)
        n2[f't{t}_ubin'] = tape.pop('rand_like').reshape(2, 128)
        n2[f't{t}_gauss'] = tape.pop('randn_like').reshape(2, 128)
        n2[f't{t}_z'] = tape.pop('randn_like')
        n2[f't{t}_s_next'] = tape.pop('multinomial').reshape(2, 128)
    for t in range(T, -1, -1):
        e = traj[t]
        n2[f'traj{t}_v'], n2[f'traj{t}_p'], n2[f'traj{t}_s'] = e[0], e[1], e[2]
        if t < T:
            n2[f'traj{t}_prmsd'], n2[f'traj{t}_ppl'] = e[3], e[4]
    with torch.no_grad():      # sample_sequence=False keeps the sequence in encode() (diffab.py:128-131)
        rf, pf, _, _ = m.encode({k: v.clone() for k, v in batch.items()}, True, False)
    n2['res_feat'] = rf
    save('trajectory_abdock_T10_structonly', **n2)


def case_posterior():
    """The categorical distributions the reference samples from (AminoacidCategoricalTransition._sample input = post + 1e-8,
    transition.py:171-181,202-245): the 10 denoising posteriors of the config-1 trajectory and, for optimize(), the forward
    categorical c_t of add_noise plus its 4 posteriors.  Same models / seeds as case_trajectory, so the runs are replayed and
    asserted equal to the committed trajectories before anything is written."""
    T = 10
    m = abdock_model(T, seed=3)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=2022, lengths=[128, 117])
    old = np.load(os.path.join(HERE, 'trajectory_abdock_T10.npz'))
    torch.manual_seed(2022)
    tape = Tape()
    with tape.recording():
        traj = m.sample({k: v.clone() for k, v in batch.items()}, sample_opt=dict(sample_structure=True, sample_sequence=True, contig=''))
    assert np.array_equal(traj[0][1].numpy(), old['traj0_p']) and np.array_equal(traj[3][2].numpy(), old['traj3_s'])
    assert len(tape.cat_probs) == T
    out = {f't{t}_probs': tape.cat_probs[T - t].reshape(2, 128, 20) for t in range(T, 0, -1)}
    old = np.load(os.path.join(HERE, 'optimize_abdock_T10_k4.npz'))
    torch.manual_seed(11)
    tape = Tape()
    with tape.recording():
        traj = m.optimize({k: v.clone() for k, v in batch.items()}, 4, optimize_opt=dict(sample_structure=True, sample_sequence=True))
    assert np.array_equal(traj[0][1].numpy(), old['traj0_p'])
    assert len(tape.cat_probs) == 5
    out['opt_addnoise_probs'] = tape.cat_probs[0].reshape(2, 128, 20)
    for i, t in enumerate(range(4, 0, -1)):
        out[f'opt_t{t}_probs'] = tape.cat_probs[1 + i].reshape(2, 128, 20)
    save('posterior_abdock_T10', **out)


def case_abdesign_sample():
    """AbDesign FullDPM.sample (pred_noise, no prmsd) at FullDPM level, T=10."""
    T = 10
    d = abdesign_fulldpm(T, seed=4)
    N, L = 2, 40
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [40, 33], [(5, 14), (22, 30)], num_steps=T, t=3)
    torch.manual_seed(5)
    tape = Tape()
    with tape.recording():
        traj = d.sample(v, p * 10, s, res_feat, pair_feat, gen, mres)
    out = _noise_from_tape(tape, N, L, T)
    for t in range(T, -1, -1):
        out[f'traj{t}_v'], out[f'traj{t}_p'], out[f'traj{t}_s'] = traj[t]
    save('trajectory_abdesign_T10', **out)


def case_training():
    T = 100
    m = abdock_model(T, seed=2).train()
    dpm = m.diffusion
    N, L = 2, 48
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
    s = s.clamp(max=19)
    t = torch.tensor([37, 80])
    res_feat = res_feat.clone().requires_grad_(True)
    pair_feat = pair_feat.clone().requires_grad_(True)
    torch.manual_seed(13)
    tape = Tape()
    with tape.recording():
        loss = dpm(v, p * 10, s, res_feat, pair_feat, gen, mres, denoise_structure=True, denoise_sequence=True, t=t)
    o = {}
    o['rot_axis'] = tape.pop('randn'); o['rot_bin'] = tape.pop('multinomial').reshape(N, L)
    o['rot_ubin'] = tape.pop('rand_like').reshape(N, L); o['rot_gauss'] = tape.pop('randn_like').reshape(N, L)
    o['pos'] = tape.pop('randn_like'); o['s_noisy'] = tape.pop('multinomial').reshape(N, L)
    assert not tape.log
    total = sum(loss.values())
    total.backward()
    for k, val in loss.items():
        o['loss_' + k] = val
    names = ['eps_net.encoder.blocks.0.proj_pair_bias.weight', 'eps_net.encoder.blocks.0.spatial_coef',
             'eps_net.encoder.blocks.5.out_transform.weight', 'eps_net.encoder.blocks.5.proj_query_point.weight',
             'eps_net.res_feat_mixer.0.weight', 'eps_net.eps_rot_net.4.weight']
    params = dict(dpm.named_parameters())
    for n_ in names:
        o['grad_' + n_] = params[n_].grad
    o['grad_res_feat'] = res_feat.grad
    o['grad_pair_feat_sub'] = pair_feat.grad[:, ::5, ::3]
    o['grad_pair_feat_sum'] = pair_feat.grad.double().sum((1, 2))
    save('training_abdock', **o)


def case_training_abdesign():
    """AbDesign FullDPM.forward (BASELINE config 5 flavour: rot, pos(eps), seq losses; A/modules/diffusion/dpm_full.py:138-191) with
    fixed t and recorded noise; losses + a few parameter gradients + input gradients."""
    T = 100
    dpm = abdesign_fulldpm(T, seed=2).train()
    N, L = 2, 48
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
    s = s.clamp(max=19)
    t = torch.tensor([37, 80])
    res_feat = res_feat.clone().requires_grad_(True)
    pair_feat = pair_feat.clone().requires_grad_(True)
    torch.manual_seed(17)
    tape = Tape()
    with tape.recording():
        loss = dpm(v, p * 10, s, res_feat, pair_feat, gen, mres, denoise_structure=True, denoise_sequence=True, t=t)
    o = {}
    o['rot_axis'] = tape.pop('randn'); o['rot_bin'] = tape.pop('multinomial').reshape(N, L)
    o['rot_ubin'] = tape.pop('rand_like').reshape(N, L); o['rot_gauss'] = tape.pop('randn_like').reshape(N, L)
    o['pos'] = tape.pop('randn_like'); o['s_noisy'] = tape.pop('multinomial').reshape(N, L)
    assert not tape.log
    assert set(loss) == {'rot', 'pos', 'seq'}
    sum(loss.values()).backward()
    for k, val in loss.items():
        o['loss_' + k] = val
    names = ['eps_net.encoder.blocks.0.proj_pair_bias.weight', 'eps_net.encoder.blocks.0.spatial_coef',
             'eps_net.encoder.blocks.5.out_transform.weight', 'eps_net.encoder.blocks.3.proj_key_point.weight',
             'eps_net.encoder.blocks.2.mlp_transition.2.weight', 'eps_net.encoder.blocks.4.layer_norm_1.gamma',
             'eps_net.res_feat_mixer.0.weight', 'eps_net.eps_crd_net.4.weight', 'eps_net.eps_seq_net.0.weight', 'eps_net.current_sequence_embedding.weight']
    params = dict(dpm.named_parameters())
    for n_ in names:
        g_ = params[n_].grad
        o['grad_' + n_] = g_[::3, ::5] if g_.numel() > 50000 else g_            # big matrices: strided sample (tests use the same stride)
    o['grad_res_feat'] = res_feat.grad
    o['grad_pair_feat_sub'] = pair_feat.grad[:, ::5, ::3]
    save('training_abdesign', **o)


def _step_draws(tape, N, L, t, out, pre=''):
    out[f'{pre}t{t}_axis'] = tape.pop('randn')
    out[f'{pre}t{t}_bin'] = tape.pop('multinomial').reshape(N, L)
    out[f'{pre}t{t}_ubin'] = tape.pop('rand_like').reshape(N, L)
    out[f'{pre}t{t}_gauss'] = tape.pop('randn_like').reshape(N, L)
    out[f'{pre}t{t}_z'] = tape.pop('randn_like')
    out[f'{pre}t{t}_s_next'] = tape.pop('multinomial').reshape(N, L)


def case_seqdesign():
    """The sequence-design mode the reference ships (AbDock/configs/test/seq_design.yml:4-6, driven per pose by optimize_ab.py:14-36):
    model.sample with sample_structure=False, sample_sequence=True and a contig.  dpm_full.py:255-267 (only the sequence is
    initialised), :290-297 (every transition still draws; the structure keeps v_t, p_t).  Every step, every draw and the
    categorical the reference sampled from are stored; the encode() of this mode (remove_structure=False) too."""
    T = 10
    m = abdock_model(T, seed=3)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=2022, lengths=[128, 117])
    b = {k: v.clone() for k, v in batch.items()}
    torch.manual_seed(23)
    tape = Tape()
    with tape.recording():
        traj = m.sample(b, sample_opt=dict(sample_structure=False, sample_sequence=True, contig='31-36'))
    out = dict(gen=b['generate_flag'])                      # the contig-restricted flag model.sample left in the batch (diffab.py:125-129)
    assert int(out['gen'].sum()) == 12 and bool(out['gen'][:, 30:36].all())
    out['init_s'] = tape.pop('randint_like')
    for t in range(T, 0, -1):
        _step_draws(tape, 2, 128, t, out)
    assert not tape.log and len(tape.cat_probs) == T
    for t in range(T, -1, -1):
        e = traj[t]
        out[f'traj{t}_v'], out[f'traj{t}_p'], out[f'traj{t}_s'] = e[0], e[1], e[2]
        if t < T:
            out[f'traj{t}_prmsd'], out[f'traj{t}_ppl'] = e[3], e[4]
            out[f't{t + 1}_probs'] = tape.cat_probs[T - t - 1].reshape(2, 128, 20)
    with torch.no_grad():
        rf, pf, R0, p0 = m.encode({k: v.clone() for k, v in b.items()}, False, True)
    out.update(res_feat=rf, pair_feat_sub=pf[:, ::7, ::5], pair_feat_sum=pf.double().sum((1, 2)), R0=R0, p0=p0)
    save('trajectory_abdock_T10_seqdesign', **out)

    # AbDesign fixbb (AbDesign/configs/test/fixbb.yml:7) at FullDPM level
    d = abdesign_fulldpm(T, seed=4)
    N, L = 2, 40
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [40, 33], [(5, 14), (22, 30)], num_steps=T, t=3)
    torch.manual_seed(29)
    tape = Tape()
    with tape.recording():
        traj = d.sample(v, p * 10, s, res_feat, pair_feat, gen, mres, sample_structure=False, sample_sequence=True)
    out = dict(init_s=tape.pop('randint_like'))
    for t in range(T, 0, -1):
        _step_draws(tape, N, L, t, out)
    assert not tape.log and len(tape.cat_probs) == T
    for t in range(T, -1, -1):
        out[f'traj{t}_v'], out[f'traj{t}_p'], out[f'traj{t}_s'] = traj[t]
        if t < T:
            out[f't{t + 1}_probs'] = tape.cat_probs[T - t - 1].reshape(N, L, 20)
    save('trajectory_abdesign_T10_fixbb', **out)


def case_training_seqonly():
    """train_structure=False, train_sequence=True (AbDock/configs/train/seq_design.yml:11-12; dpm_full.py:163-178): the structure
    enters un-noised, eps_p = 0, only the sequence is noised.  FullDPM.forward of both trees, fixed t, recorded draw, losses +
    parameter / input gradients."""
    T = 100
    N, L = 2, 48
    names = ['eps_net.encoder.blocks.0.proj_pair_bias.weight', 'eps_net.encoder.blocks.0.spatial_coef',
             'eps_net.encoder.blocks.5.out_transform.weight', 'eps_net.encoder.blocks.3.proj_key_point.weight',
             'eps_net.encoder.blocks.2.mlp_transition.2.weight', 'eps_net.encoder.blocks.4.layer_norm_1.gamma',
             'eps_net.res_feat_mixer.0.weight', 'eps_net.eps_crd_net.4.weight', 'eps_net.eps_seq_net.0.weight',
             'eps_net.eps_rot_net.4.weight', 'eps_net.current_sequence_embedding.weight']
    for tag, dpm, seed in (('abdock', abdock_model(T, seed=2).train().diffusion, 31), ('abdesign', abdesign_fulldpm(T, seed=2).train(), 37)):
        v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [48, 41], [(6, 17), (30, 37)], salt=300)
        s = s.clamp(max=19)
        t = torch.tensor([37, 80])
        res_feat = res_feat.clone().requires_grad_(True)
        pair_feat = pair_feat.clone().requires_grad_(True)
        dpm.zero_grad()
        torch.manual_seed(seed)
        tape = Tape()
        with tape.recording():
            loss = dpm(v, p * 10, s, res_feat, pair_feat, gen, mres, denoise_structure=False, denoise_sequence=True, t=t)
        o = dict(s_noisy=tape.pop('multinomial').reshape(N, L))
        assert not tape.log                                    # the structure draws nothing in this mode
        o['addnoise_probs'] = tape.cat_probs[0].reshape(N, L, 20)
        sum(loss.values()).backward()
        for k, val in loss.items():
            o['loss_' + k] = val
        params = dict(dpm.named_parameters())
        for n_ in names:
            g_ = params[n_].grad
            o['grad_' + n_] = g_[::3, ::5] if g_.numel() > 50000 else g_
        o['grad_res_feat'] = res_feat.grad
        o['grad_pair_feat_sub'] = pair_feat.grad[:, ::5, ::3]
        save(f'training_seqonly_{tag}', **o)


STEP_TS = (100, 64, 22, 21, 10, 2, 1)


def case_steps_T100():
    """T = 100 (the headline sampler's schedule), N = 2, L = 40, single denoising steps of the reference's own loops at t in STEP_TS:
    state at t, all six draws of step t, the categorical sampled from, state at t - 1 (+ prmsd / perplexity).  t = 22 is the last
    histogram row of the inverse IGSO(3) distribution, t = 2..21 take the Gaussian branch |2 sigma + sigma randn| mod pi
    (so3.py:127-135), t = 1 adds no noise (transition.py:95,152).
      AbDock:   FullDPM.sample run free from t = 100 to 0 (dpm_full.py:236-302); the steps of that one run are stored.
      AbDesign: with hash-filled weights a free run's generated positions leave the fp32 range a 1e-4 A check means anything in
                (3900 A at t = 64), so each step is the FIRST denoising step of FullDPM.optimize(opt_step = t)
                (A/.../dpm_full.py:256-319: the same loop body as sample, entered from add_noise of a sane structure)."""
    T = 100
    N, L = 2, 40
    v, p, s, res_feat, pair_feat, _, gen, mres = cases.eps_inputs(N, L, [40, 33], [(5, 14), (22, 30)], num_steps=T, t=3)
    s = torch.where(mres, s.clamp(max=19), s)

    d = abdock_model(T, seed=2).diffusion
    inv = d.trans_rot.angular_distrib_inv
    flags = inv.approx_flag.tolist()
    assert flags[22] is False and all(flags[2:22]) and abs(float(inv.stddevs[21]) - 0.1) > 1e-4
    torch.manual_seed(41)
    tape = Tape()
    with tape.recording(), torch.no_grad():
        traj = d.sample(v, p * 10, s, res_feat, pair_feat, gen, mres)
    tape.pop('randn'); tape.pop('randn_like'); tape.pop('randint_like')
    out = {}
    for t in range(T, 0, -1):
        tmp = {}
        _step_draws(tape, N, L, t, tmp)
        if t in STEP_TS:
            out.update(tmp)
            out[f't{t}_probs'] = tape.cat_probs[T - t].reshape(N, L, 20)
            for tt in (t, t - 1):
                e = traj[tt]
                out[f'traj{tt}_v'], out[f'traj{tt}_p'], out[f'traj{tt}_s'] = e[0], e[1], e[2]
                if tt < T:
                    out[f'traj{tt}_prmsd'], out[f'traj{tt}_ppl'] = e[3], e[4]
    assert not tape.log
    for k, a in out.items():
        assert torch.isfinite(a.float()).all() and (a.float().abs().max() < 100 or k.endswith('_bin')), k
    save('steps_T100_abdock', **out)

    d = abdesign_fulldpm(T, seed=2)
    out = {}
    for i, t in enumerate(STEP_TS):
        torch.manual_seed(43 + i)
        tape = Tape()
        with tape.recording(), torch.no_grad():
            traj = d.optimize(v, p * 10, s, t, res_feat, pair_feat, gen, mres)
        for _ in range(6):
            tape.log.pop(0)                                    # add_noise draws (rot: 4, pos: 1, seq: 1)
        _step_draws(tape, N, L, t, out)
        out[f't{t}_probs'] = tape.cat_probs[1].reshape(N, L, 20)
        out[f'in{t}_v'], out[f'in{t}_p'], out[f'in{t}_s'] = traj[t]
        out[f'out{t}_v'], out[f'out{t}_p'], out[f'out{t}_s'] = traj[t - 1]
    for k, a in out.items():
        assert torch.isfinite(a.float()).all() and (a.float().abs().max() < 100 or k.endswith('_bin')), k
    save('steps_T100_abdesign', **out)


def case_encode():
    m = abdock_model(10, seed=3)
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=99, lengths=[24, 19])
    batch['generate_flag'][:, 8:13] = True
    batch['fragment_type'][:, :12] = 1
    batch['fragment_type'][:, 12:] = 3
    batch['fragment_type'] = batch['fragment_type'] * batch['mask']
    batch['chain_nb'][:, 12:] = 1
    with torch.no_grad():
        rf, pf, R0, p0 = m.encode({k: v.clone() for k, v in batch.items()}, True, True)
        rf2, pf2, _, _ = m.encode({k: v.clone() for k, v in batch.items()}, True, False)
    # gradients of encode(): d/dparams of <res_feat, w1> + <pair_feat, w2> with hash weights w (training path of the reference)
    m.zero_grad()
    rfg, pfg, _, _ = m.encode({k: v.clone() for k, v in batch.items()}, True, True)
    w1, w2 = synth.hash_tensor(tuple(rfg.shape), 71, scale=1.0), synth.hash_tensor(tuple(pfg.shape), 72, scale=1.0)
    ((rfg * w1).sum() + (pfg * w2).sum()).backward()
    P = dict(m.named_parameters())
    grads = {'grad_' + k.replace('.', '__'): P[k].grad for k in (
        'pair_embed.aa_pair_embed.weight', 'pair_embed.relpos_embed.weight', 'pair_embed.aapair_to_distcoef.weight',
        'pair_embed.distance_embed.0.weight', 'pair_embed.distance_embed.2.bias', 'pair_embed.out_mlp.0.weight', 'pair_embed.out_mlp.4.weight',
        'residue_embed.aatype_embed.weight', 'residue_embed.mlp.6.weight')}
    grads['grad_residue_embed__mlp__0__weight_sub'] = P['residue_embed.mlp.0.weight'].grad[::4, ::7]
    save('encode_small', res_feat=rf, pair_feat=pf, R0=R0, p0=p0, res_feat_seqkept=rf2, pair_feat_seqkept_sum=pf2.double().sum((1, 2)), **grads)


def case_encode_full():
    """encode() at config-5 sample size (L = 256): the reference's forward values (sub-sampled) and the gradients of EVERY embedding
    parameter of d/dparams [<res_feat, w1> + <pair_feat, w2>] -- the training path of D/models/diffab.py:39-83 at the size the bench runs it
    (round 4 pinned encode()'s backward only on the 24 / 19-residue fixture)."""
    m = abdock_model(10, seed=3)
    with torch.no_grad():
        m.pair_embed.aapair_to_distcoef.weight.copy_(cases.encode_full_distcoef(m.pair_embed.aapair_to_distcoef.weight.shape))
    batch = cases.encode_full_batch()
    m.zero_grad()
    rf, pf, R0, p0 = m.encode({k: v.clone() for k, v in batch.items()}, True, True)
    w1, w2 = synth.hash_tensor(tuple(rf.shape), 71, scale=1.0), synth.hash_tensor(tuple(pf.shape), 72, scale=1.0)
    ((rf * w1).sum() + (pf * w2).sum()).backward()
    P = dict(m.named_parameters())
    grads = {'grad_' + k.replace('.', '__'): P[k].grad for k in cases.ENCODE_FULL_PARAMS}
    grads['grad_residue_embed__mlp__0__weight_sub'] = P['residue_embed.mlp.0.weight'].grad[::4, ::7]
    save('encode_L256', res_feat=rf, pair_feat_sub=pf[:, ::9, ::7], pair_feat_sum=pf.double().sum((1, 2)), R0=R0, **grads)


def case_reconstruct():
    """reconstruct_backbone_partially (geometry.py:404-480) on a ragged two-chain batch; also dumps the ideal backbone
    tables (constants.py:310-320: data, 21 residue types) the function reads."""
    from src.modules.common import geometry as G
    from src.utils.protein import constants as K
    batch = synth.make_batch(2, synth.LAYOUT_128, seed=41, lengths=[40, 33])
    batch['chain_nb'][:, 22:] = 1
    batch['res_nb'][:, 22:] = batch['res_nb'][:, 22:] - 21
    batch['generate_flag'][:] = False
    batch['generate_flag'][:, 5:12] = True
    batch['generate_flag'][:, 20:25] = True                         # spans the chain break
    batch['generate_flag'] &= batch['mask']
    N, L = batch['aa'].shape
    v_new = synth.hash_tensor((N, L, 3), 61, scale=2.0)
    t_new = batch['pos_heavyatom'][:, :, 1] + synth.hash_tensor((N, L, 3), 62, scale=1.5)
    aa_new = (synth.hash_tensor((N, L), 63, scale=10.0).abs() * 2).long().clamp(max=19)
    from src.modules.common.so3 import so3vec_to_rotation
    R_new = so3vec_to_rotation(v_new)
    pos_new, mask_new = G.reconstruct_backbone_partially(pos_ctx=batch['pos_heavyatom'], R_new=R_new, t_new=t_new, aa=aa_new,
                                                         chain_nb=batch['chain_nb'], res_nb=batch['res_nb'],
                                                         mask_atoms=batch['mask_heavyatom'], mask_recons=batch['generate_flag'])
    save('reconstruct_small', R_new=R_new, t_new=t_new, aa_new=aa_new, pos_new=pos_new, mask_new=mask_new,
         bb_table=K.backbone_atom_coordinates_tensor, o_table=K.bb_oxygen_coordinate_tensor)


def case_rank():
    sys.path.insert(0, os.path.join(REF, 'AbDock'))
    # design_for_testset imports heavy deps at module import; restate-free: load the three pure functions by exec of
    # nothing -- instead call through torch with the documented formula is NOT a golden.  Import guarded:
    src = open(os.path.join(REF, 'AbDock/src/tools/runner/design_for_testset.py')).read()
    start = src.index('def calc_per_rmsd')
    end = src.index('def ', src.index('def rank_commoness') + 10) if 'def ' in src[src.index('def rank_commoness') + 10:] else len(src)
    ns = {'torch': torch}
    exec(compile(src[start:end], 'design_for_testset_excerpt', 'exec'), ns)   # executed here only; never stored
    structs = synth.hash_tensor((16, 36, 3), 55, scale=8.0)
    structs[3] = structs[5] + 0.01
    rank = ns['rank_commoness'](structs, 5)
    save('rank_commoness', rank=rank, avg_rmsd=ns['calc_avg_rmsd'](structs))


def case_dockq():
    """DockQ scoring of docked candidates (design_for_pdb.py:316-321).  Fnat / contact counts / interface residues come from the
    REFERENCE's own scorer: its `fnat` C program built from /root/reference/AbDock/DockQ/src (oracle/Makefile -> oracle/_ref/fnat) and
    run on PDB files written from the case tensors.  iRMS / LRMS follow DockQ.py:296-366; Biopython is absent here, so they are
    computed with the oracle's SVD fit and cross-checked against scipy.spatial.transform.Rotation.align_vectors."""
    import subprocess, tempfile
    from scipy.spatial.transform import Rotation
    from oracle import dockq as DQ
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True)
    fnat_bin = os.path.join(ROOT, 'oracle', '_ref', 'fnat')
    pos, mask, group, models = cases.dockq_case()
    pos, mask, group, models = pos.numpy().astype(np.float64), mask.numpy(), group.numpy(), models.numpy().astype(np.float64)
    out = dict(fnat=[], nat_correct=[], nat_total=[], irms=[], Lrms=[], DockQ=[])
    with tempfile.TemporaryDirectory() as d:
        DQ.write_pdb(d + '/native.pdb', pos, mask, group)
        for k in range(models.shape[0]):
            DQ.write_pdb(d + '/model.pdb', models[k], mask, group)
            run = lambda cut: DQ.parse_reference_fnat(subprocess.run([fnat_bin, d + '/model.pdb', d + '/native.pdb', cut, '-all'],
                                                                     capture_output=True, text=True, check=True).stdout)
            r5, r10 = run('5'), run('10')
            o = DQ.dockq(models[k], mask, pos, mask, group)
            inter = set((i + 1, 'AB'[group[i] - 1]) for i in np.nonzero(o['interface'])[0])
            assert inter == r10['inter'] and (o['nat_correct'], o['nat_total']) == (r5['nat_correct'], r5['nat_total'])
            # independent superposition check (scipy): interface fit and receptor fit
            both = mask[:, 1] & (group > 0)
            sel = o['interface'] & both
            x, y = pos[:, 1], models[k][:, 1]
            cx, cy = x[sel].mean(0), y[sel].mean(0)
            _, rssd = Rotation.align_vectors(x[sel] - cx, y[sel] - cy)
            assert abs(rssd / np.sqrt(sel.sum()) - o['irms']) < 1e-6, (rssd / np.sqrt(sel.sum()), o['irms'])
            n1, n2 = (both & (group == 1)).sum(), (both & (group == 2)).sum()
            rec, lig = (1, 2) if n1 > n2 else (2, 1)
            rs, ls = both & (group == rec), both & (group == lig)
            cx, cy = x[rs].mean(0), y[rs].mean(0)
            Rm, _ = Rotation.align_vectors(x[rs] - cx, y[rs] - cy)
            lr = np.sqrt((((Rm.apply(y[ls] - cy) + cx) - x[ls]) ** 2).sum(-1).mean())
            assert abs(lr - o['Lrms']) < 1e-6, (lr, o['Lrms'])
            out['fnat'].append(r5['fnat']); out['nat_correct'].append(r5['nat_correct']); out['nat_total'].append(r5['nat_total'])
            out['irms'].append(o['irms']); out['Lrms'].append(o['Lrms'])
            out['DockQ'].append((r5['nat_correct'] / r5['nat_total'] + 1 / (1 + (o['irms'] / 1.5) ** 2) + 1 / (1 + (o['Lrms'] / 8.5) ** 2)) / 3)
        out['interface'] = o['interface']
    save('dockq_small', **{k: np.asarray(v) for k, v in out.items()})


def case_dockq_edge():
    """Superposition corner cases (cases.dockq_edge_cases): expected Fnat / iRMS / LRMS / DockQ from the oracle, accepted only where its
    two independent superposition algorithms (SVD-Kabsch and QCP) and scipy agree; Fnat also from the reference's `fnat` binary."""
    import subprocess, tempfile
    from scipy.spatial.transform import Rotation
    from oracle import dockq as DQ
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True)
    fnat_bin = os.path.join(ROOT, 'oracle', '_ref', 'fnat')
    out = {}
    for c in cases.dockq_edge_cases():
        pos, mask, group, models = c['pos'].numpy().astype(np.float64), c['mask'].numpy(), c['group'].numpy(), c['models'].numpy().astype(np.float64)
        rows = []
        with tempfile.TemporaryDirectory() as d:
            DQ.write_pdb(d + '/native.pdb', pos, mask, group)
            for k in range(models.shape[0]):
                o = DQ.dockq(models[k], mask, pos, mask, group)
                DQ.write_pdb(d + '/model.pdb', models[k], mask, group)
                r5 = DQ.parse_reference_fnat(subprocess.run([fnat_bin, d + '/model.pdb', d + '/native.pdb', '5', '-all'], capture_output=True, text=True, check=True).stdout)
                assert (o['nat_correct'], o['nat_total']) == (r5['nat_correct'], r5['nat_total']), c['name']
                both = mask[:, 1] & (group > 0)
                sel = o['interface'] & both
                x, y = pos[:, 1], models[k][:, 1]
                q_rmsd, _, _ = DQ.qcp(x[sel], y[sel])
                assert abs(q_rmsd - o['irms']) < 2e-6, (c['name'], q_rmsd, o['irms'])
                _, rssd = Rotation.align_vectors(x[sel] - x[sel].mean(0), y[sel] - y[sel].mean(0))
                assert abs(rssd / np.sqrt(sel.sum()) - o['irms']) < 2e-6
                if c['lrms_defined']:
                    n1, n2 = (both & (group == 1)).sum(), (both & (group == 2)).sum()
                    rec, lig = (1, 2) if n1 > n2 else (2, 1)
                    rs, ls = both & (group == rec), both & (group == lig)
                    _, Rq, tq = DQ.qcp(x[rs], y[rs])
                    lr = np.sqrt((((y[ls] @ Rq.T + tq) - x[ls]) ** 2).sum(-1).mean())
                    assert abs(lr - o['Lrms']) < 1e-6, (c['name'], lr, o['Lrms'])
                rows.append([o['fnat'], o['irms'], o['Lrms'] if c['lrms_defined'] else -1.0, o['DockQ'] if c['lrms_defined'] else -1.0, o['n_interface']])
        out[c['name']] = np.asarray(rows)
        print(c['name'], out[c['name']])
    save('dockq_edge', **out)


if __name__ == '__main__':
    assert os.path.isdir(REF), 'reference mount not present: goldens can only be generated in the build container'
    sys.path.insert(0, os.path.join(REF, 'AbDock'))
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['so3', 'ga_block', 'eps_net', 'schedule_tables', 'trajectory', 'structonly', 'abdesign_sample',
                             'training', 'training_abdesign', 'seqdesign', 'training_seqonly', 'steps_T100', 'encode', 'encode_full', 'rank', 'reconstruct', 'posterior', 'dockq', 'dockq_edge']
    for w in which:
        print('==', w)
        globals()['case_' + w]()

"""Deterministic synthetic inputs and weights (no datasets or checkpoints exist offline).

* `hash_uniform` / `fill_module_` -- integer-hash weight fill: the value
  of element `idx` of the tensor with ordinal `salt` is a pure function of (salt, idx), computed in
  uint64 with numpy, so the reference model (in the build container), the oracle and the HIP
  model get bit-identical parameters without shipping a checkpoint.
* `make_complex` / `make_batch` -- synthetic antibody-antigen complexes with the batch schema the
  model boundary consumes (reference: AbDock/src/utils/data.py:60-76,
  AbDock/src/utils/transforms/merge.py:62-87; shapes in SURVEY.md section 8b/8d).
"""
import math
import numpy as np
import torch

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def hash_uniform(n, salt):
    """n floats in [-0.5, 0.5), exactly representable in fp32 (24-bit), from a splitmix64-style mix."""
    with np.errstate(over='ignore'):
        x = np.arange(n, dtype=np.uint64) + np.uint64(salt + 1) * _GOLD
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5).astype(np.float32)


def _param_scale(name, shape):
    leaf = name.split('.')[-1]
    if leaf == 'gamma':
        return 'affine', 1.0, 0.2          # LayerNorm scale: 1 + 0.2u
    if leaf == 'beta':
        return 'affine', 0.0, 0.2
    if leaf == 'spatial_coef':
        return 'affine', math.log(math.e - 1), 0.5
    if leaf == 'bias':
        return 'affine', 0.0, 0.2
    if len(shape) == 2 and ('embed' in name or 'aapair_to_distcoef' in name):
        return 'affine', 0.0, 2.0          # embedding tables: U(-1, 1)
    fan_in = shape[-1] if len(shape) >= 2 else 1
    return 'affine', 0.0, 2.0 / math.sqrt(fan_in)


@torch.no_grad()
def fill_module_(module, seed=0):
    """Fill every parameter of `module`, in named_parameters() order, with the hash stream."""
    for k, (name, p) in enumerate(module.named_parameters()):
        _, off, scale = _param_scale(name, tuple(p.shape))
        u = hash_uniform(p.numel(), seed * 1000003 + k)
        p.copy_(torch.from_numpy(off + scale * u).reshape(p.shape))
    return module


def hash_tensor(shape, salt, scale=1.0, offset=0.0):
    n = int(np.prod(shape))
    return torch.from_numpy((offset + scale * hash_uniform(n, salt)).astype(np.float32)).reshape(shape)


# ---------------------------------------------------------------------------- complexes
# Ideal backbone in the residue frame (CA origin, C on +x, N in the xy plane), Angstrom.
_LOCAL = np.array([[-0.525, 1.363, 0.0],     # N
                   [0.0, 0.0, 0.0],          # CA
                   [1.526, 0.0, 0.0],        # C
                   [2.153, -1.062, 0.0],     # O
                   [-0.529, -0.774, -1.205]], dtype=np.float64)  # CB
MAX_ATOMS = 15
PAD_AA = 21

LAYOUT_256 = dict(L=256, chains=[(0, 110, 1), (110, 216, 2), (216, 256, 3)],
                  cdrs=[(25, 33), (51, 57), (94, 106), (133, 144), (159, 166), (198, 207)])
LAYOUT_128 = dict(L=128, chains=[(0, 64, 1), (64, 128, 3)], cdrs=[(30, 42)])


def _rand_rot(rs):
    q = rs.normal(size=4)
    q /= np.linalg.norm(q)
    r, i, j, k = q
    return np.array([[1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r)],
                     [2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r)],
                     [2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)]])


def _ca_walk(rs, n, step=3.8, min_sep=4.0):
    pts = np.zeros((n, 3))
    for i in range(1, n):
        for _ in range(200):
            d = rs.normal(size=3)
            d /= np.linalg.norm(d)
            cand = pts[i - 1] + step * d
            if i < 2 or np.min(np.linalg.norm(pts[:i - 1] - cand, axis=1)) >= min_sep:
                break
        pts[i] = cand
    return pts


def make_complex(layout=LAYOUT_256, seed=2022, length=None):
    """One synthetic complex as a dict of unbatched tensors (pre-collate schema)."""
    rs = np.random.RandomState(seed)
    L = layout['L'] if length is None else length
    ca = _ca_walk(rs, L)
    ca -= ca.mean(axis=0, keepdims=True)
    pos = np.zeros((L, MAX_ATOMS, 3), dtype=np.float32)
    for i in range(L):
        pos[i, :5] = (ca[i][None, :] + _LOCAL @ _rand_rot(rs).T).astype(np.float32)
    matom = np.zeros((L, MAX_ATOMS), dtype=bool)
    matom[:, :5] = True
    aa = rs.randint(0, 20, size=L).astype(np.int64)
    res_nb = np.zeros(L, dtype=np.int64)
    chain_nb = np.zeros(L, dtype=np.int64)
    frag = np.zeros(L, dtype=np.int64)
    for c, (a, b, f) in enumerate(layout['chains']):
        a, b = min(a, L), min(b, L)
        res_nb[a:b] = np.arange(1, b - a + 1)
        chain_nb[a:b] = c
        frag[a:b] = f
    gen = np.zeros(L, dtype=bool)
    for a, b in layout['cdrs']:
        gen[min(a, L):min(b, L)] = True
    t = torch.from_numpy
    return dict(aa=t(aa), res_nb=t(res_nb), chain_nb=t(chain_nb), pos_heavyatom=t(pos),
                mask_heavyatom=t(matom), fragment_type=t(frag), generate_flag=t(gen))


def collate(items, pad_to=None):
    """Pad to the longest item and stack; adds 'mask' (reference PaddingCollate semantics:
    aa padded with 21, everything else with 0)."""
    Lm = max(d['aa'].shape[0] for d in items) if pad_to is None else pad_to
    out = {}
    for k in items[0]:
        vs = []
        for d in items:
            v = d[k]
            n = Lm - v.shape[0]
            if n > 0:
                padv = PAD_AA if k == 'aa' else 0
                v = torch.cat([v, torch.full((n,) + tuple(v.shape[1:]), padv, dtype=v.dtype)], 0)
            vs.append(v)
        out[k] = torch.stack(vs, 0)
    out['mask'] = torch.stack([torch.arange(Lm) < d['aa'].shape[0] for d in items], 0)
    return out


def make_batch(n, layout=LAYOUT_256, seed=2022, lengths=None, replicate=False):
    """Batch of n complexes.  replicate=True mimics the reference runner (one crop repeated n times,
    AbDock/src/tools/runner/design_for_pdb.py:141); otherwise each sample is a different complex."""
    if replicate:
        one = make_complex(layout, seed)
        return collate([one] * n)
    lengths = lengths or [None] * n
    return collate([make_complex(layout, seed + 7919 * i, lengths[i]) for i in range(n)])


# ---------------------------------------------------------------------------- model + denoiser inputs (bench.py, smoke(), tests)
MODEL_CFG_ABDOCK = dict(
    type='diffab', res_feat_dim=128, pair_feat_dim=64,
    diffusion=dict(num_steps=100, eps_net_opt=dict(num_layers=6), obj='pred_x0'),
    train_structure=True, train_sequence=False, num_bins=40, dist_min=0.5, dist_max=19.5,
)   # AbDock/configs/train/dock_single.yml:2-17


def cfg_abdock(num_steps=100, **over):
    c = {k: (dict(v) if isinstance(v, dict) else v) for k, v in MODEL_CFG_ABDOCK.items()}
    c['diffusion'] = dict(c['diffusion'], num_steps=num_steps, eps_net_opt=dict(num_layers=6))
    c.update(over)
    return c


class AttrDict(dict):
    """dict with attribute access, nested: stands in for the reference's EasyDict configs."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v
    __setattr__ = dict.__setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None         # (pickle / copy probe for dunder attributes this way)


_MODELS = {}
_MODEL_FP = {}            # fingerprint of every cached model as built (tests/conftest.py checks after each test that nobody changed or moved it)


def model_fingerprint(m, device):
    import torch
    with torch.no_grad():
        ts = list(m.parameters()) + list(m.buffers())
        assert all(t.device.type == torch.device(device).type for t in ts), 'a cached model was moved to another device (module.to() works in place)'
        return float(sum(t.double().abs().sum() for t in ts if t.is_floating_point()).item())


def fresh_model(num_steps, seed, flavour='abdock', device='cpu'):
    """A NEW get_model(cfg) of the dock_single / codesign_single model block with hash-filled weights (the same fill the reference got when
    the golden vectors were made): for callers that change weights or move the module."""
    from .. import get_model
    cfg = cfg_abdock(num_steps)
    if flavour == 'abdesign':
        for k in ('num_bins', 'dist_min', 'dist_max'):
            cfg.pop(k)
        cfg['diffusion'].pop('obj')
    m = get_model(AttrDict(cfg)).eval()
    fill_module_(m, seed=seed)
    return m.to(device)


def build_model(num_steps, seed, flavour='abdock', device='cpu'):
    """fresh_model, cached per (num_steps, seed, flavour, device): every caller of a session shares ONE module -- do not write to its
    weights and do not call .to() on it (tests/conftest.py checks the fingerprint after every test)."""
    key = (num_steps, seed, flavour, str(device))
    if key not in _MODELS:
        _MODELS[key] = fresh_model(num_steps, seed, flavour, device)
        _MODEL_FP[key] = model_fingerprint(_MODELS[key], device)
    return _MODELS[key]


def mask_from_lengths(lengths, L):
    return torch.stack([torch.arange(L) < n for n in lengths], 0)


def gen_from_ranges(N, L, ranges):
    g = torch.zeros(N, L, dtype=torch.bool)
    for a, b in ranges:
        g[:, a:b] = True
    return g


def eps_inputs(N, L, lengths, gen_ranges, salt=200, F=128, C=64, t=37, num_steps=100):
    """Hash-filled inputs of one EpsilonNet call: v, p, s, res_feat, pair_feat, beta, mask_generate, mask_res (CPU tensors)."""
    from ..modules import VarianceSchedule
    v = hash_tensor((N, L, 3), salt + 0, scale=4.0)
    p = hash_tensor((N, L, 3), salt + 1, scale=3.0)
    s = (hash_tensor((N, L), salt + 2) + 0.5).mul(21).long().clamp(0, 20)
    mres = mask_from_lengths(lengths, L)
    s = torch.where(mres, s, torch.full_like(s, 21))
    res_feat = hash_tensor((N, L, F), salt + 3, scale=2.0)
    pair_feat = hash_tensor((N, L, L, C), salt + 4, scale=2.0)
    beta = VarianceSchedule(num_steps).betas[t].expand([N]).clone()
    gen = gen_from_ranges(N, L, gen_ranges) & mres
    return v, p, s, res_feat, pair_feat, beta, gen, mres

"""Model boundary: `get_model(cfg)` -> nn.Module with forward / sample / optimize.

Drop-in for AbDock/src/models/{_base.py:1-13, diffab.py:19-205} and AbDesign/diffab/models/{_base.py,
diffab.py:19-142}: same registry, constructor cfg keys, method signatures, batch-dict schema, trajectory
layout and state_dict keys.  Everything under `self.diffusion` executes in libabopt_hip.so.
"""
import torch
import torch.nn as nn

from . import hip
from .dpm import FullDPM
from .embed import ResidueEmbedding, PairEmbedding, ATOM_CA

_MODEL_DICT = {}
max_num_heavyatoms = 15
resolution_to_num_atoms = {'backbone+CB': 5, 'full': max_num_heavyatoms}


def register_model(name):
    def decorator(cls):
        _MODEL_DICT[name] = cls
        return cls
    return decorator


def _cfg_get(cfg, key, default=None):
    if hasattr(cfg, 'get'):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def get_model(cfg):
    """_base.py:12-13.  cfg.type 'diffab' resolves to the AbDock flavour when the config carries the prmsd
    keys (`num_bins`), else to the AbDesign flavour; 'diffab_abdock' / 'diffab_abdesign' force one."""
    name = _cfg_get(cfg, 'type')
    if name == 'diffab':
        name = 'diffab_abdock' if _cfg_get(cfg, 'num_bins') is not None else 'diffab_abdesign'
    return _MODEL_DICT[name](cfg)


def generate_mask_from_str(str_input, tensor):
    """'start-end' (1-based, inclusive) -> bool mask like `tensor` (diffab.py:184-205)."""
    start, end = str_input.split('-')
    mask = torch.zeros_like(tensor, dtype=torch.bool)
    mask[..., int(start) - 1:int(end)] = True
    return mask


def generate_random_mask_from(tensor, mask_ratio_min, mask_ratio_max):
    """diffab.py:166-180."""
    ratio = float(torch.empty(1).uniform_(mask_ratio_min, mask_ratio_max))
    return torch.bernoulli(torch.zeros_like(tensor.float()).fill_(ratio)).bool()


class _DiffabBase(nn.Module):
    ABDOCK = True

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        g = lambda k, d=None: _cfg_get(cfg, k, d)
        num_atoms = resolution_to_num_atoms[g('resolution', 'full')]
        F, Cp = g('res_feat_dim'), g('pair_feat_dim')
        self.residue_embed = ResidueEmbedding(F, num_atoms, hotspot=not self.ABDOCK)
        self.pair_embed = PairEmbedding(Cp, num_atoms)
        dcfg = dict(g('diffusion'))
        if self.ABDOCK:
            self.diffusion = FullDPM(F, Cp, **dcfg, num_bins=g('num_bins'), dist_min=g('dist_min'), dist_max=g('dist_max'))
        else:
            self.diffusion = FullDPM(F, Cp, **dcfg, _abdesign=True)

    def encode(self, batch, remove_structure, remove_sequence):
        """diffab.py:39-83 -> res_feat (N,L,F), pair_feat (N,L,L,C), R (N,L,3,3), p (N,L,3)."""
        ctx = torch.logical_and(batch['mask_heavyatom'][:, :, ATOM_CA], ~batch['generate_flag'])
        sm = ctx if remove_structure else None
        qm = ctx if remove_sequence else None
        extra = {} if self.ABDOCK else dict(hotspot=batch.get('hotspot'))
        if not torch.is_grad_enabled():          # sample / optimize / validation: the inference kernels (csrc/embed.hip), nothing saved for a backward
            inp, keep = hip.encode_inputs(batch['aa'], batch['res_nb'], batch['chain_nb'], batch['pos_heavyatom'], batch['mask_heavyatom'],
                                          self.residue_embed.max_num_atoms, fragment_type=batch['fragment_type'], hotspot=extra.get('hotspot'),
                                          structure_mask=sm, sequence_mask=qm)
            res_feat, R, p = self.residue_embed.forward_hip(inp)
            return res_feat, self.pair_embed.forward_hip(inp), R, p
        # training: the same kernels under custom autograd functions (embed.py); the frames come out of the residue-feature kernel
        res_feat, R, p = self.residue_embed.forward_with_frames(aa=batch['aa'], res_nb=batch['res_nb'], chain_nb=batch['chain_nb'],
                                                                pos_atoms=batch['pos_heavyatom'], mask_atoms=batch['mask_heavyatom'],
                                                                fragment_type=batch['fragment_type'], structure_mask=sm, sequence_mask=qm, **extra)
        pair_feat = self.pair_embed(aa=batch['aa'], res_nb=batch['res_nb'], chain_nb=batch['chain_nb'],
                                    pos_atoms=batch['pos_heavyatom'], mask_atoms=batch['mask_heavyatom'],
                                    structure_mask=sm, sequence_mask=qm)
        return res_feat, pair_feat, R, p

    def forward(self, batch):
        """diffab.py:85-112 -> loss dict (AbDock: prmsd, dist, rot, pos, seq; AbDesign: rot, pos, seq)."""
        g = lambda k, d=None: _cfg_get(self.cfg, k, d)
        mask_generate = batch['generate_flag']
        if self.ABDOCK and g('mask_ratio_min', False):
            mask_generate = torch.logical_and(mask_generate, generate_random_mask_from(mask_generate, g('mask_ratio_min'), g('mask_ratio_max')))
            batch['generate_flag'] = mask_generate
        mask_res = batch['mask']
        res_feat, pair_feat, R_0, p_0 = self.encode(batch, remove_structure=g('train_structure', True), remove_sequence=g('train_sequence', True))
        v_0 = hip.so3_log(R_0.detach(), grad_mode=torch.is_grad_enabled())     # frames come from input coordinates: no gradient flows here;
        # the clamp follows autograd state like the reference's log_rotation (so3.py:12-16): validation runs under no_grad
        return self.diffusion(v_0, p_0, batch['aa'], res_feat, pair_feat, mask_generate, mask_res,
                              denoise_structure=g('train_structure', True), denoise_sequence=g('train_sequence', True))

    @torch.no_grad()
    def sample(self, batch, sample_opt={'sample_structure': True, 'sample_sequence': True, 'contig': ''}):
        """diffab.py:114-140."""
        mask_generate = batch['generate_flag']
        if self.ABDOCK and sample_opt.get('sample_sequence', False) and sample_opt['contig'] != '':
            mask_generate = torch.logical_and(mask_generate, generate_mask_from_str(sample_opt['contig'], mask_generate))
            batch['generate_flag'] = mask_generate
        mask_res = batch['mask']
        res_feat, pair_feat, R_0, p_0 = self.encode(batch, remove_structure=sample_opt.get('sample_structure', True),
                                                    remove_sequence=sample_opt.get('sample_sequence', True))
        v_0 = hip.so3_log(R_0, grad_mode=False)
        opt = {k: v for k, v in sample_opt.items() if k != 'contig'}
        return self.diffusion.sample(v_0, p_0, batch['aa'], res_feat, pair_feat, mask_generate, mask_res, **opt)

    @torch.no_grad()
    def optimize(self, batch, opt_step, optimize_opt={'sample_structure': True, 'sample_sequence': True}):
        """diffab.py:142-163."""
        mask_generate, mask_res = batch['generate_flag'], batch['mask']
        res_feat, pair_feat, R_0, p_0 = self.encode(batch, remove_structure=optimize_opt.get('sample_structure', True),
                                                    remove_sequence=optimize_opt.get('sample_sequence', True))
        v_0 = hip.so3_log(R_0, grad_mode=False)
        return self.diffusion.optimize(v_0, p_0, batch['aa'], opt_step, res_feat, pair_feat, mask_generate, mask_res, **optimize_opt)


@register_model('diffab_abdock')
class DiffusionAntibodyDesign(_DiffabBase):
    """AbDock flavour: prmsd head, obj in {pred_x0, pred_noise}, contig masks (AbDock/src/models/diffab.py)."""
    ABDOCK = True


@register_model('diffab_abdesign')
class DiffusionAntibodyDesignAbDesign(_DiffabBase):
    """AbDesign flavour (AbDesign/diffab/models/diffab.py): no prmsd head, noise-prediction objective."""
    ABDOCK = False

"""Frame / backbone utilities the reference's runners call around the sampler, on the HIP device.

`reconstruct_backbone_partially` mirrors AbDock/src/modules/common/geometry.py:453-480 (same argument names, same returns);
`so3vec_to_rotation` / `rotation_to_so3vec` mirror AbDock/src/modules/common/so3.py:54-63.  All three execute in
libabopt_hip.so (no CPU path)."""
import torch

from . import hip


def so3vec_to_rotation(so3vec):
    return hip.so3_exp(so3vec)


def rotation_to_so3vec(R):
    return hip.so3_log(R, grad_mode=torch.is_grad_enabled())


def reconstruct_backbone_partially(pos_ctx, R_new, t_new, aa, chain_nb, res_nb, mask_atoms, mask_recons):
    """-> (pos_new (N,L,A,3), mask_new (N,L,A)): backbone N, CA, C, O of the residues in `mask_recons` rebuilt from the frames
    (R_new, t_new) and residue types `aa`; every other atom copied from the context."""
    return hip.reconstruct_backbone_partially(pos_ctx, R_new, t_new, aa, chain_nb, res_nb, mask_atoms, mask_recons)

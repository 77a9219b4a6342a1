"""ab_opt_amd: MI355X-native denoising hot path of pengzhangzhi/ab_opt (AbDock / AbDesign).

`get_model(cfg)` returns a module with the reference's forward / sample / optimize contract whose
diffusion path runs as hand-written HIP kernels (libabopt_hip.so, C ABI in include/abopt.h).
"""
__version__ = "0.1.0"

from .model import get_model, register_model, DiffusionAntibodyDesign, DiffusionAntibodyDesignAbDesign  # noqa: F401

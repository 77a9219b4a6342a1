"""ctypes binding of libabopt_hip.so (the C ABI in include/abopt.h).

PyTorch is used only as the owner of device memory and streams: every call passes
`tensor.data_ptr()` and `torch.cuda.current_stream().cuda_stream`.  There is no fallback:
if the library is missing or a tensor is not on a HIP device, these functions raise.
"""
import ctypes as C
import os
import threading
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ABOPT_LIB_PATH: developer override to load a variant build of the same ABI (csrc/Makefile VARIANT=...: timing / ablation / A-B builds)
LIB_PATH = os.environ.get('ABOPT_LIB_PATH') or os.path.join(_HERE, 'libabopt_hip.so')
ABI_VERSION = 41

c_f = C.c_void_p        # device float*
c_i64 = C.c_void_p      # device int64*
c_u8 = C.c_void_p       # device uint8*


class GemmTnProblem(C.Structure):
    """abopt_gemm_tn_problem (include/abopt.h)"""
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p), ('c', C.c_void_p), ('lda', C.c_int), ('ldb', C.c_int), ('m', C.c_int), ('n', C.c_int), ('k', C.c_int)]


class GaWeights(C.Structure):
    _fields_ = [(n, c_f) for n in (
        'w_node', 'w_pair_bias', 'spatial_coef', 'w_out', 'b_out', 'ln1_gamma', 'ln1_beta',
        'w_mlp0', 'b_mlp0', 'w_mlp1', 'b_mlp1', 'w_mlp2', 'b_mlp2', 'ln2_gamma', 'ln2_beta', 'w_node_frag', 'w_out_frag', 'w_mlp_frag', 'w_out_terms')]


class GaDebug(C.Structure):
    _fields_ = [('logits', c_f), ('alpha', c_f), ('feat', c_f)]


class EpsWeights(C.Structure):
    _fields_ = ([(n, c_f) for n in ('seq_embed', 'w_mix0', 'b_mix0', 'w_mix1', 'b_mix1')] +
                [('blocks', C.POINTER(GaWeights)), ('num_layers', C.c_int)] +
                [(n, c_f) for n in ('w_head1', 'b_head1', 'w_crd2', 'b_crd2', 'w_crd3', 'b_crd3',
                                    'w_rot2', 'b_rot2', 'w_rot3', 'b_rot3', 'w_seq2', 'b_seq2', 'w_seq3', 'b_seq3',
                                    'prmsd_ln_gamma', 'prmsd_ln_beta', 'w_prmsd1', 'b_prmsd1', 'w_prmsd2', 'b_prmsd2',
                                    'w_prmsd3', 'b_prmsd3')] +
                [('num_bins', C.c_int), ('w_heads_frag', c_f), ('w_mix_frag', c_f), ('mix_table', c_f)])


class EncodeInputs(C.Structure):
    _fields_ = ([('N', C.c_int), ('L', C.c_int), ('atoms_in', C.c_int), ('atoms', C.c_int)] +
                [(n, C.c_void_p) for n in ('aa', 'res_nb', 'chain_nb', 'fragment_type', 'hotspot', 'pos_atoms', 'mask_atoms',
                                           'structure_mask', 'sequence_mask')])


class ResidueEmbedWeights(C.Structure):
    _fields_ = [(n, c_f) for n in ('aatype_embed', 'type_embed', 'hotspot_embed', 'freq_bands', 'w0', 'b0', 'w1', 'b1', 'w2', 'b2', 'w3', 'b3')]


class PairEmbedWeights(C.Structure):
    _fields_ = [(n, c_f) for n in ('aa_pair_embed', 'relpos_embed', 'aapair_to_distcoef', 'freq_bands', 'wd0', 'bd0', 'wd1', 'bd1',
                                   'wo0', 'bo0', 'wo1', 'bo1', 'wo2', 'bo2')]


class StepParams(C.Structure):
    _fields_ = [('t', C.c_int), ('alpha_clamped', C.c_float), ('alpha_bar', C.c_float), ('sigma', C.c_float),
                ('sqrt_recip_abar', C.c_float), ('sqrt_recipm1_abar', C.c_float), ('igso3_std', C.c_float),
                ('igso3_gaussian', C.c_int), ('position_scale', C.c_float), ('position_mean', C.c_float * 3),
                ('pred_x0', C.c_int), ('sample_structure', C.c_int), ('sample_sequence', C.c_int),
                ('dist_min', C.c_float), ('dist_max', C.c_float), ('ppl_masked', C.c_int)]


class AddNoiseNoise(C.Structure):
    _fields_ = [('axis', c_f), ('bin', c_i64), ('ubin', c_f), ('gauss', c_f), ('pos', c_f), ('s_noisy', c_i64)]


class StepNoise(C.Structure):
    _fields_ = [('axis', c_f), ('bin', c_i64), ('ubin', c_f), ('gauss', c_f), ('z', c_f), ('s_next', c_i64)]


EXPORTS = ['abopt_abi_version', 'abopt_last_error', 'abopt_device_info', 'abopt_so3_exp', 'abopt_so3_log',
           'abopt_ga_workspace_bytes', 'abopt_ga_block_forward', 'abopt_ga_block_forward_cached', 'abopt_ga_encoder_forward',
           'abopt_eps_workspace_bytes', 'abopt_eps_net_forward', 'abopt_pair_bias_cache_bytes', 'abopt_pair_bias_cache', 'abopt_pair_terms_bytes', 'abopt_pair_terms', 'abopt_pair_terms_used', 'abopt_nonfinite_flag', 'abopt_denoise_step', 'abopt_sample_init',
           'abopt_add_noise', 'abopt_gemm', 'abopt_gemm_tn_grouped', 'abopt_colsum', 'abopt_adam_step', 'abopt_adam_ws_floats', 'abopt_bucket_colsum', 'abopt_segment_bucket_colsum', 'abopt_heads_epilogue_forward', 'abopt_heads_epilogue_backward', 'abopt_dpm_losses', 'abopt_abdock_losses', 'abopt_layer_norm_forward', 'abopt_layer_norm_backward', 'abopt_residue_features', 'abopt_residue_features_workspace_bytes', 'abopt_commonness_score', 'abopt_prof_enable', 'abopt_prof_collect', 'abopt_prof_peek', 'abopt_prof_clock', 'abopt_prof_spans_reset', 'abopt_prof_spans',
           'abopt_reconstruct_backbone_partially', 'abopt_ipa_train_workspace_bytes', 'abopt_ipa_core_train_forward', 'abopt_ipa_points_backward', 'abopt_ipa_backward_operands', 'abopt_ipa_backward_assemble', 'abopt_ipa_pair_backward', 'abopt_ipa_dz_assemble',
           'abopt_residue_embed_workspace_bytes', 'abopt_residue_embed_forward', 'abopt_pair_embed_workspace_bytes', 'abopt_pair_embed_forward',
           'abopt_pair_embed_backward_workspace_bytes', 'abopt_pair_embed_backward', 'abopt_dockq_workspace_bytes', 'abopt_dockq_lite', 'abopt_node_frag_source_row', 'abopt_node_frag_floats',
           'abopt_out_frag_floats', 'abopt_out_terms_floats', 'abopt_out_frag_terms', 'abopt_heads_frag_floats', 'abopt_mixer_frag_floats', 'abopt_mlp_frag_floats', 'abopt_pack_tail_weights', 'abopt_block_tail_forward', 'abopt_block_tail_backward']

_lib = None
_lock = threading.Lock()


def lib():
    """Load the shared library once; raise (never fall back) if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                               '(or `make -C ab_opt_amd/csrc`).  ab_opt_amd has no non-HIP execution path.')
        L = C.CDLL(LIB_PATH)
        L.abopt_abi_version.restype = C.c_int
        L.abopt_last_error.restype = C.c_char_p
        L.abopt_ga_workspace_bytes.restype = C.c_size_t
        L.abopt_ga_workspace_bytes.argtypes = [C.c_int] * 4
        L.abopt_eps_workspace_bytes.restype = C.c_size_t
        L.abopt_eps_workspace_bytes.argtypes = [C.c_int] * 4
        L.abopt_device_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
        L.abopt_so3_exp.argtypes = [c_f, c_f, C.c_int64, C.c_void_p]
        L.abopt_so3_log.argtypes = [c_f, c_f, C.c_int64, C.c_int, C.c_void_p]
        L.abopt_ga_block_forward.argtypes = [C.POINTER(GaWeights), c_f, c_f, c_f, c_f, c_u8, c_f, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(GaDebug), C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_ga_encoder_forward.argtypes = [C.POINTER(GaWeights), C.c_int, c_f, c_f, c_f, c_f, c_u8, c_f, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_eps_net_forward.argtypes = [C.POINTER(EpsWeights), c_f, c_f, c_i64, c_f, c_f, c_f, c_u8, c_u8,
                                            c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f, C.c_int, c_f,
                                            C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_ga_block_forward_cached.argtypes = [C.POINTER(GaWeights), c_f, c_f, c_f, c_f, c_u8, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f, C.c_int, c_f,
                                                    C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_pair_terms_bytes.restype = C.c_size_t
        L.abopt_pair_terms_bytes.argtypes = [C.c_int] * 2
        L.abopt_pair_terms.argtypes = [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.abopt_pair_terms_used.argtypes = [C.c_int] * 3
        L.abopt_nonfinite_flag.argtypes = [C.c_int, C.c_void_p]
        L.abopt_pair_bias_cache_bytes.restype = C.c_size_t
        L.abopt_pair_bias_cache_bytes.argtypes = [C.c_int] * 3
        L.abopt_pair_bias_cache.argtypes = [C.POINTER(GaWeights), C.c_int, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.abopt_denoise_step.argtypes = [C.POINTER(StepParams), C.POINTER(StepNoise), C.c_uint64, C.c_uint64,
                                         c_f, c_f, c_i64, c_f, c_f, c_f, c_f, c_u8, c_f, c_f, C.c_int, C.c_int,
                                         c_f, c_f, c_i64, c_f, c_f, c_f, c_f, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.abopt_sample_init.argtypes = [c_f, c_f, c_i64, c_u8, c_f, c_f, c_i64, C.c_uint64, C.c_uint64,
                                        C.c_float, C.POINTER(C.c_float), C.c_int, C.c_int, c_f, c_f, c_i64, C.c_int, C.c_int, C.c_void_p]
        L.abopt_commonness_score.argtypes = [c_f, c_f, C.c_int, C.c_int, C.c_void_p]
        L.abopt_add_noise.argtypes = [c_i64, c_f, c_f, c_u8, c_f, c_f, C.c_int, C.c_int, C.POINTER(AddNoiseNoise), C.c_uint64, C.c_uint64,
                                      c_f, c_f, c_i64, c_u8, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                                      c_f, c_f, c_i64, c_f, c_f, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.abopt_gemm.argtypes = [c_f, C.c_int, C.c_int64, C.c_int, c_f, C.c_int, C.c_int64, C.c_int, c_f, C.c_int, C.c_int64,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_colsum.argtypes = [c_f, C.c_int, C.c_int64, C.c_int, c_f, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_gemm_tn_grouped.argtypes = [C.POINTER(GemmTnProblem), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_prof_enable.argtypes = [C.c_int]
        L.abopt_prof_collect.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.abopt_prof_peek.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.abopt_reconstruct_backbone_partially.argtypes = [c_f, c_f, c_f, c_i64, c_i64, c_i64, c_u8, c_u8, c_f, c_f, c_f, c_u8] + [C.c_int] * 3 + [C.c_void_p]
        L.abopt_ipa_train_workspace_bytes.restype = C.c_size_t
        L.abopt_ipa_train_workspace_bytes.argtypes = [C.c_int] * 2
        L.abopt_ipa_core_train_forward.argtypes = [c_f, c_f, c_f, c_f, c_u8, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_ipa_points_backward.argtypes = [c_f, C.c_int, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_void_p]
        L.abopt_ipa_backward_operands.argtypes = [c_f] * 6 + [C.c_int, C.c_int, C.c_void_p]
        L.abopt_ipa_backward_assemble.argtypes = [c_f] * 9 + [C.c_int, C.c_int, C.c_void_p]
        L.abopt_ipa_pair_backward.argtypes = [c_f, c_f, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.abopt_residue_embed_workspace_bytes.restype = C.c_size_t
        L.abopt_residue_features_workspace_bytes.restype = C.c_size_t
        L.abopt_residue_features_workspace_bytes.argtypes = [C.c_int] * 2
        L.abopt_residue_features.argtypes = [C.POINTER(EncodeInputs), C.POINTER(ResidueEmbedWeights), c_f, c_f, c_f, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_residue_embed_workspace_bytes.argtypes = [C.c_int] * 4
        L.abopt_pair_embed_workspace_bytes.restype = C.c_size_t
        L.abopt_pair_embed_workspace_bytes.argtypes = [C.c_int] * 3
        L.abopt_residue_embed_forward.argtypes = [C.POINTER(EncodeInputs), C.POINTER(ResidueEmbedWeights), c_f, c_f, c_f, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_pair_embed_forward.argtypes = [C.POINTER(EncodeInputs), C.POINTER(PairEmbedWeights), c_f, c_f, c_f, c_f, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_pair_embed_backward_workspace_bytes.restype = C.c_size_t
        L.abopt_pair_embed_backward_workspace_bytes.argtypes = [C.c_int] * 3
        L.abopt_pair_embed_backward.argtypes = [C.POINTER(EncodeInputs), C.POINTER(PairEmbedWeights), c_f, c_f, c_f, c_f, c_f, c_f, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_node_frag_source_row.argtypes = [C.c_int] * 3
        L.abopt_node_frag_floats.restype = C.c_size_t
        L.abopt_abdock_losses.argtypes = [c_f, c_f, c_f, c_f, c_f, c_u8, c_u8, c_f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, c_f, c_f, c_f, C.c_void_p]
        L.abopt_layer_norm_forward.argtypes = [c_f, c_f, c_f, C.c_int, C.c_float, C.c_int64, c_f, c_f, c_f, C.c_void_p]
        L.abopt_layer_norm_backward.argtypes = [c_f, c_f, c_f, c_f, C.c_int, C.c_int64, c_f, c_f, C.c_void_p]
        L.abopt_adam_ws_floats.restype = C.c_size_t
        L.abopt_adam_ws_floats.argtypes = [C.c_int, C.c_void_p]
        L.abopt_adam_step.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_double] * 6 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.abopt_dockq_workspace_bytes.restype = C.c_size_t
        L.abopt_dockq_workspace_bytes.argtypes = [C.c_int]
        L.abopt_dockq_lite.argtypes = [c_f, c_u8, C.c_int, c_f, c_u8, C.c_void_p, C.c_int, C.c_int, C.c_int, c_f, C.c_void_p, C.c_size_t, C.c_void_p]
        L.abopt_out_frag_floats.restype = C.c_size_t
        L.abopt_out_terms_floats.restype = C.c_size_t
        L.abopt_out_frag_terms.argtypes = [c_f, c_f, C.c_void_p]
        L.abopt_heads_frag_floats.restype = C.c_size_t
        L.abopt_mixer_frag_floats.restype = C.c_size_t
        L.abopt_mlp_frag_floats.restype = C.c_size_t
        L.abopt_pack_tail_weights.argtypes = [c_f] * 7 + [C.c_void_p]
        L.abopt_block_tail_forward.argtypes = [c_f] * 5 + [c_u8] + [c_f] * 9 + [C.c_int64, C.c_void_p]
        L.abopt_block_tail_backward.argtypes = [c_f] * 3 + [c_u8] + [c_f] * 6 + [C.c_int64, C.c_void_p]
        for name in EXPORTS:
            getattr(L, name)          # AttributeError here = a symbol of include/abopt.h is missing
        if L.abopt_abi_version() != ABI_VERSION:
            raise RuntimeError(f'libabopt_hip.so ABI {L.abopt_abi_version()} != expected {ABI_VERSION}; rebuild it')
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f'abopt error {rc}: {lib().abopt_last_error().decode()}')


_DT = {torch.float32: 'f32', torch.int64: 'i64', torch.bool: 'u8', torch.uint8: 'u8'}


def ptr(t, dtype=None, optional=False, strided=False):
    """Device pointer of a contiguous HIP tensor (None -> NULL when optional).  strided=True: the caller passes the strides itself."""
    if t is None:
        if optional:
            return None
        raise ValueError('required tensor is None')
    if not t.is_cuda:
        raise RuntimeError('ab_opt_amd kernels run on a HIP device only; got a CPU tensor (there is no CPU path)')
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f'tensor lives on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: kernels launch on '
                           'the current device/stream -- call torch.cuda.set_device() (one process per GPU) or wrap the call in '
                           '`with torch.cuda.device(tensor.device):`')
    if not strided and not t.is_contiguous():
        raise ValueError('tensor must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'expected {dtype}, got {t.dtype}')
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _contig(*ts):
    """Contiguous versions of the arguments, returned as objects the CALLER binds to locals: a temporary
    `x.contiguous()` inside an argument list is freed as soon as its address is taken, and the caching allocator hands
    the same block to the next temporary -- two kernel arguments would then alias."""
    return [t if (t is None or t.is_contiguous()) else t.contiguous() for t in ts]


class Workspace:
    """Grow-only scratch buffer per (device, stream)."""
    _bufs = {}

    @classmethod
    def get(cls, nbytes, device):
        key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


def device_info():
    cu, lds = C.c_int(), C.c_int()
    arch = C.create_string_buffer(64)
    _check(lib().abopt_device_info(C.byref(cu), C.byref(lds), arch, 64))
    return dict(cu_count=cu.value, lds_bytes_per_cu=lds.value, arch=arch.value.decode())


# ------------------------------------------------------------------------------- thin wrappers
def so3_exp(w):
    w = w.contiguous()
    R = torch.empty(w.shape[:-1] + (3, 3), dtype=torch.float32, device=w.device)
    _check(lib().abopt_so3_exp(ptr(w, torch.float32), ptr(R), w.numel() // 3, stream()))
    return R


def so3_log(R, grad_mode=False):
    R = R.contiguous()
    w = torch.empty(R.shape[:-2] + (3,), dtype=torch.float32, device=R.device)
    _check(lib().abopt_so3_log(ptr(R, torch.float32), ptr(w), R.numel() // 9, int(grad_mode), stream()))
    return w


def ga_weights_struct(t):
    """t: dict name -> contiguous device tensor with the field names of GaWeights (w_node_frag optional)."""
    s = GaWeights()
    for name, _ in GaWeights._fields_:
        setattr(s, name, ptr(t.get(name), torch.float32, optional=name in ('w_node_frag', 'w_out_frag', 'w_mlp_frag', 'w_out_terms')))
    return s


def pack_mfma_operand(w):
    """w [32 B, K] (K a multiple of 16) -> [B, K/16, 2, 64, 4] flattened + {S, 1 / S, 0, 0}: 32x32x16 MFMA operand order, every weight as the two
    fp16 terms of S w (fp32 container of 8 fp16 per lane), S = tail_weight_scale(w):
    [cb][step][term][lane = 32 khalf + c][i] = term(S w[32 cb + c][16 step + 8 khalf + i])."""
    R, K = w.shape
    S = tail_weight_scale(w)
    g = w.float() * S
    h16 = g.half()
    terms = torch.stack([h16, (g - h16.float()).half()], 0)   # [term, R, K] fp16
    g = terms.reshape(2, R // 32, 32, K // 16, 2, 8)          # [term, cb, col, step, k half, i]
    out = g.permute(1, 3, 0, 4, 2, 5).contiguous()            # [cb, step, term, k half, col, i]
    return torch.cat([out.view(-1).view(torch.float32), torch.tensor([S, 1.0 / S, 0.0, 0.0], dtype=torch.float32, device=w.device)])


def pack_heads_weights(w_head1, w_crd2, w_rot2, w_seq2, w_crd3, w_rot3, w_seq3):
    """-> w_heads_frag [27, 8, 2, 64, 4] + {S, 1 / S, 0, 0} (include/abopt.h: abopt_eps_weights.w_heads_frag)."""
    pad32 = lambda w: torch.cat([w, torch.zeros(32 - w.shape[0], w.shape[1], dtype=w.dtype, device=w.device)], 0)
    rows = torch.cat([w_head1[:, :128], w_crd2, w_rot2, w_seq2, pad32(w_crd3), pad32(w_rot3), pad32(w_seq3)], 0)      # [27 * 32, 128]
    return pack_mfma_operand(rows.contiguous())


def out_feat_order():
    """The order in which the 1824 feature columns enter the tail's K index (csrc/tail_common.h: ot_feat_col): the 768 pair-feature columns
    (head h, channel ch) sit at K index 192 * ((ch % 16) // 4) + 16 h + 4 (ch // 16) + ch % 4, the other columns keep their place."""
    k = torch.arange(1824)
    c, j = k // 192, k % 192
    perm = (j >> 4) * 64 + ((j >> 2) & 3) * 16 + 4 * c + (j & 3)
    return torch.where(k < 768, perm, k)


def tail_weight_scale(w):
    """Power of two S with max |w| S in [2^14, 2^15) (csrc/mlp.hip: tail_weight_absmax_kernel): the weights of the forward tail are split into two
    fp16 terms of S w, so that the low terms stay normal; 1 for an all-zero matrix."""
    a = w.detach().float().abs()
    m = float(a[torch.isfinite(a)].max()) if bool(torch.isfinite(a).any()) else 0.0
    if m <= 0.0:
        return 1.0
    import math
    e = 15 - math.frexp(m)[1]
    return math.ldexp(1.0, max(-100, min(100, e)))


def _fp16_terms(v):
    """[..., 8] fp32 -> (h, l) as [..., 4] fp32 words holding 8 fp16 each: h = fp16(v), l = fp16(v - h), round to nearest (csrc/ipa_common.h: split_pair2)."""
    h = v.half()
    lo = (v - h.float()).half()
    return h.contiguous().view(torch.float32), lo.contiguous().view(torch.float32)


def pack_out_weights(w_out):
    """w_out [128, 1824] -> w_out_frag [4 cb, 114 k-steps, 2 terms, 64 lanes, 8 fp16] (as 128 * 1824 fp32 words) in 32x32x16 operand order
    (include/abopt.h: abopt_ga_weights.w_out_frag): lane 32 kh + c holds the two fp16 terms of S w_out[32 cb + c][col(16 s + 8 kh + i)]."""
    w = w_out.float()[:, out_feat_order().to(w_out.device)] * tail_weight_scale(w_out)
    v = w.reshape(4, 32, 114, 2, 8).permute(0, 2, 3, 1, 4).reshape(4, 114, 64, 8)           # [cb, s, lane = 32 kh + c, i]
    h, lo = _fp16_terms(v)
    return torch.stack([h, lo], dim=2).contiguous()                                              # [cb, s, term, lane, 4 words]


def pack_mlp_weights(w0, w1, w2):
    """three [128, 128] layers -> w_mlp_frag (include/abopt.h: abopt_ga_weights.w_mlp_frag): [layer][ct][s][term][lane = 16 kq + m] x 8 fp16 =
    the two fp16 terms of S_layer w[16 ct + m][32 s + 8 kq + i], then {S_out = 0 here, S_0, S_1, S_2, 1 / S ...} -- the caller of the device packer
    gets S_out filled in; this host statement takes it from `pack_mlp_weights.s_out` when set -- zero-padded to abopt_mlp_frag_floats() floats."""
    out, scales = [], []
    for w in (w0, w1, w2):
        S = tail_weight_scale(w)
        scales.append(S)
        v = (w.float() * S).reshape(8, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).reshape(8, 4, 64, 8)   # [ct, m, s, kq, i] -> [ct, s, lane = 16 kq + m, i]
        h, lo = _fp16_terms(v)
        out.append(torch.stack([h, lo], dim=2).reshape(-1))
    flat = torch.cat(out)
    return flat, scales


def pack_tail_weights_host(w_out, w0, w1, w2):
    """Host statement of abopt_pack_tail_weights' forward buffers: (w_out_frag, w_mlp_frag), bit for bit what the device packer writes."""
    wof = pack_out_weights(w_out).flatten()
    flat, scales = pack_mlp_weights(w0, w1, w2)
    S = [tail_weight_scale(w_out)] + scales
    sc = torch.tensor(S + [1.0 / v for v in S], dtype=torch.float32, device=flat.device)
    pad = torch.zeros(lib().abopt_mlp_frag_floats() - flat.numel() - 8, dtype=torch.float32, device=flat.device)
    return wof, torch.cat([flat, sc, pad])


_NODE_FRAG_INDEX = None

def split_bf16x3(w):
    """fp32 tensor -> (h, m, l) int16 bf16 bit patterns with h + m + l == w exactly (csrc/node_frags.hip: split3): each term is the
    round-to-nearest bf16 of what the previous terms left."""
    def rne(v):
        return v.to(torch.bfloat16).to(torch.float32)
    h = rne(w)
    r1 = w - h
    m = rne(r1)
    l = rne(r1 - m)
    return [(v.contiguous().view(torch.int32) >> 16).to(torch.int16) for v in (h, m, l)]


def pack_node_weights(w_node):
    """w_node [2016, 128] -> w_node_frag [12 heads, 12 tiles, 4 k-steps, 2 fp16 terms, 64 lanes, 4] (fp32 container of 8 fp16 per lane) +
    {S, 1 / S, 0, 0} (include/abopt.h: abopt_node_frag_source_row): the per-head, tile-ordered, operand-order copy the fused projection kernel keeps
    in LDS, every weight as h = fp16(S w), l = fp16(S w - h) with S = tail_weight_scale(w_node).  Index tables come from the library so the layout
    has one owner."""
    global _NODE_FRAG_INDEX
    if _NODE_FRAG_INDEX is None:
        L_ = lib()
        rows = torch.tensor([[[L_.abopt_node_frag_source_row(h, T, m) for m in range(16)] for T in range(12)] for h in range(12)], dtype=torch.long)
        _NODE_FRAG_INDEX = rows
    rows = _NODE_FRAG_INDEX.to(w_node.device)
    S = tail_weight_scale(w_node)
    wz = torch.cat([w_node, torch.zeros(1, w_node.shape[1], dtype=w_node.dtype, device=w_node.device)], 0)
    g = wz[torch.where(rows >= 0, rows, torch.full_like(rows, w_node.shape[0]))].float() * S   # [12, 12, 16 (m), 128]
    h16 = g.half()
    terms = torch.stack([h16, (g - h16.float()).half()], 0)                                # [2, h, T, m, 128] fp16
    terms = terms.reshape(2, 12, 12, 16, 4, 4, 8)                                          # [term, h, T, m, s, kq, i]: k = 32 s + 8 kq + i
    out = terms.permute(1, 2, 4, 0, 5, 3, 6).contiguous()                                  # [h, T, s, term, kq, m, i] == [h][T][s][term][lane = 16 kq + m][i]
    return torch.cat([out.view(-1).view(torch.float32), torch.tensor([S, 1.0 / S, 0.0, 0.0], dtype=torch.float32, device=w_node.device)])


def ga_block_forward(ws, R, t, x, z, mask, debug=False):
    N, L, F = x.shape
    Cd = z.shape[-1]
    out = torch.empty_like(x)
    dbg, extras = None, {}
    if debug:
        extras = dict(logits=torch.empty(N, L, L, 12, device=x.device), alpha=torch.empty(N, L, L, 12, device=x.device),
                      feat=torch.empty(N, L, 1824, device=x.device))
        dbg = GaDebug(ptr(extras['logits']), ptr(extras['alpha']), ptr(extras['feat']))
    nb = lib().abopt_ga_workspace_bytes(N, L, F, Cd)
    buf = Workspace.get(nb, x.device)
    R, t, x, z, mask = _contig(R, t, x, z, mask)      # copies (if any) stay alive until the launch is enqueued
    _check(lib().abopt_ga_block_forward(C.byref(ws), ptr(R, torch.float32), ptr(t, torch.float32), ptr(x, torch.float32), ptr(z, torch.float32),
                                        ptr(mask, torch.bool), ptr(out), N, L, F, Cd,
                                        C.byref(dbg) if dbg is not None else None, ptr(buf), buf.numel(), stream()))
    return (out, extras) if debug else out


def ga_block_forward_cached(ws, R, t, x, z, mask, pair_bias_cache, pair_terms=None, pair_feat_shared=0, want_feat=False):
    """One block with this block's slice of the bias cache and (optionally) the pair terms (include/abopt.h: abopt_ga_block_forward_cached)."""
    N, L, F = x.shape
    Cd = z.shape[-1]
    out = torch.empty_like(x)
    feat = torch.empty(N, L, 1824, device=x.device) if want_feat else None
    nb = lib().abopt_ga_workspace_bytes(N, L, F, Cd)
    buf = Workspace.get(nb, x.device)
    R, t, x, z, mask = _contig(R, t, x, z, mask)
    _check(lib().abopt_ga_block_forward_cached(C.byref(ws), ptr(R, torch.float32), ptr(t, torch.float32), ptr(x, torch.float32), ptr(z, torch.float32),
                                               ptr(mask, torch.bool), ptr(out), N, L, F, Cd, ptr(pair_bias_cache, torch.float32),
                                               ptr(pair_terms, torch.float32, optional=True), int(pair_feat_shared), ptr(feat, optional=True),
                                               ptr(buf), buf.numel(), stream()))
    return (out, feat) if want_feat else out


def ga_encoder_forward(ws_array, num_layers, R, t, x, z, mask):
    N, L, F = x.shape
    Cd = z.shape[-1]
    out = torch.empty_like(x)
    nb = lib().abopt_ga_workspace_bytes(N, L, F, Cd)
    buf = Workspace.get(nb, x.device)
    R, t, x, z, mask = _contig(R, t, x, z, mask)
    _check(lib().abopt_ga_encoder_forward(ws_array, num_layers, ptr(R, torch.float32), ptr(t, torch.float32), ptr(x, torch.float32),
                                          ptr(z, torch.float32), ptr(mask, torch.bool), ptr(out), N, L, F, Cd, ptr(buf), buf.numel(), stream()))
    return out


def eps_net_forward(ew, v_t, p_t, s_t, res_feat, pair_feat, beta, mask_generate, mask_res, has_prmsd, num_bins, grad_mode=False, out=None,
                    pair_bias_cache=None, pair_feat_shared=False, pair_terms=None):
    N, L = mask_res.shape
    F, Cd = res_feat.shape[-1], pair_feat.shape[-1]
    dev = res_feat.device
    if out is None:
        out = dict(v_next=torch.empty(N, L, 3, device=dev), R_next=torch.empty(N, L, 3, 3, device=dev),
                   eps_pos=torch.empty(N, L, 3, device=dev), c=torch.empty(N, L, 20, device=dev),
                   prmsd_logits=torch.empty(N, num_bins, device=dev) if has_prmsd else None)
    nb = lib().abopt_eps_workspace_bytes(N, L, F, Cd)
    buf = Workspace.get(nb, dev)
    v_t, p_t, s_t, res_feat, pair_feat, beta, mask_generate, mask_res = _contig(v_t, p_t, s_t, res_feat, pair_feat, beta, mask_generate, mask_res)
    _check(lib().abopt_eps_net_forward(C.byref(ew), ptr(v_t, torch.float32), ptr(p_t, torch.float32), ptr(s_t, torch.int64), ptr(res_feat, torch.float32),
                                       ptr(pair_feat, torch.float32), ptr(beta, torch.float32), ptr(mask_generate, torch.bool), ptr(mask_res, torch.bool),
                                       ptr(out['v_next']), ptr(out['R_next']), ptr(out['eps_pos']), ptr(out['c']),
                                       ptr(out['prmsd_logits'], optional=True), N, L, F, Cd, int(grad_mode),
                                       ptr(pair_bias_cache, torch.float32, optional=True), int(pair_feat_shared),
                                       ptr(pair_terms, torch.float32, optional=True), ptr(buf), buf.numel(), stream()))
    return out


def pair_bias_cache_bytes(N, L, num_layers):
    return lib().abopt_pair_bias_cache_bytes(N, L, num_layers)


def nonfinite_flag(reset=True):
    """Synchronises the current stream and returns whether any eps_net_forward since the last reset produced a non-finite head output
    (include/abopt.h: abopt_nonfinite_flag -- the range guard of the two-term fp16 layers)."""
    r = lib().abopt_nonfinite_flag(int(reset), stream())
    if r < 0:
        raise RuntimeError('abopt_nonfinite_flag: ' + last_error())
    return bool(r)


def pair_terms_bytes(N, L):
    return lib().abopt_pair_terms_bytes(N, L)


def pair_terms_used(N, L, pair_feat_shared=0):
    """Whether eps_net_forward(N, L) with a bias cache takes the kernels that read pair terms on this device (include/abopt.h: abopt_pair_terms_used)."""
    return bool(lib().abopt_pair_terms_used(N, L, int(pair_feat_shared)))


def pair_terms(pair_feat):
    """pair_feat as K-packed two-term fp16 operands of the pair aggregation + one power-of-two scale per query row, once per sampling call
    (include/abopt.h: abopt_pair_terms).  Returns the blob eps_net_forward(pair_terms=...) takes."""
    N, L = pair_feat.shape[:2]
    blob = torch.empty(lib().abopt_pair_terms_bytes(N, L) // 4, dtype=torch.float32, device=pair_feat.device)
    pair_feat, = _contig(pair_feat)
    _check(lib().abopt_pair_terms(ptr(pair_feat, torch.float32), ptr(blob), N, L, pair_feat.shape[-1], stream()))
    return blob


def pair_bias_cache(blocks_array, num_layers, pair_feat):
    """proj_pair_bias(pair_feat) of every block, once per sampling call (include/abopt.h: abopt_pair_bias_cache)."""
    N, L = pair_feat.shape[:2]
    nb = lib().abopt_pair_bias_cache_bytes(N, L, num_layers)
    cache = torch.empty(nb // 4, dtype=torch.float32, device=pair_feat.device)
    pair_feat, = _contig(pair_feat)
    _check(lib().abopt_pair_bias_cache(blocks_array, num_layers, ptr(pair_feat, torch.float32), ptr(cache), N, L,
                                       pair_feat.shape[-1], stream()))
    return cache


def denoise_step(sp, noise, seed, offset, v_t, p_t, s_t, v_net, p_net, c_net, prmsd_logits, mask_generate,
                 ig_X_row, ig_cdf_row, num_bins, out, want_post=False, seed_dev=None):
    """seed_dev (optional): int64 device tensor {seed, offset} read by the kernel in place of the two host values (graph replays)."""
    N, L = mask_generate.shape
    nz = None
    if noise is not None:
        nz = StepNoise(ptr(noise['axis'], torch.float32), ptr(noise['bin'], torch.int64), ptr(noise['ubin'], torch.float32),
                       ptr(noise['gauss'], torch.float32), ptr(noise['z'], torch.float32), ptr(noise['s_next'], torch.int64))
    post = torch.empty(N, L, 20, device=v_t.device) if want_post else None
    _check(lib().abopt_denoise_step(C.byref(sp), C.byref(nz) if nz is not None else None, seed, offset,
                                    ptr(v_t, torch.float32), ptr(p_t, torch.float32), ptr(s_t, torch.int64),
                                    ptr(v_net, torch.float32), ptr(p_net, torch.float32), ptr(c_net, torch.float32),
                                    ptr(prmsd_logits, optional=True), ptr(mask_generate, torch.bool),
                                    ptr(ig_X_row, torch.float32), ptr(ig_cdf_row, optional=True), ig_X_row.numel(), num_bins,
                                    ptr(out['v']), ptr(out['p']), ptr(out['s']), ptr(out.get('prmsd'), optional=True),
                                    ptr(out.get('ppl'), optional=True), ptr(post, optional=True), ptr(out.get('p_norm'), optional=True),
                                    ptr(seed_dev, torch.int64, optional=True), N, L, stream()))
    return post


def sample_init(v, p, s, mask_generate, init_noise, seed, offset, scale, mean, sample_structure, sample_sequence):
    N, L = mask_generate.shape
    v_i, p_i, s_i = torch.empty_like(v), torch.empty_like(p), torch.empty_like(s)
    q4 = pn = sr = None
    if init_noise is not None:
        q4, pn, sr = init_noise.get('q4'), init_noise.get('p'), init_noise.get('s')
    mean_arr = (C.c_float * 3)(*[float(m) for m in mean])
    v, p, s, mask_generate = _contig(v, p, s, mask_generate)
    _check(lib().abopt_sample_init(ptr(v, torch.float32), ptr(p, torch.float32), ptr(s, torch.int64),
                                   ptr(mask_generate, torch.bool), ptr(q4, optional=True), ptr(pn, optional=True),
                                   ptr(sr, optional=True), seed, offset, float(scale), mean_arr, int(sample_structure), int(sample_sequence),
                                   ptr(v_i), ptr(p_i), ptr(s_i), N, L, stream()))
    return v_i, p_i, s_i


def add_noise(t, alpha_bars, fwd, noise, seed, offset, v_0, p_0, s_0, mask_generate, scale, mean,
              noise_structure=True, noise_sequence=True, grad_mode=False, want_eps=False, want_probs=False, seed_dev=None):
    """fwd: ApproxAngularDistribution of the forward process (buffers stddevs, approx_flag, X + cdf())."""
    N, L = mask_generate.shape
    v_n, p_n, s_n = torch.empty_like(v_0), torch.empty_like(p_0), torch.empty_like(s_0)
    eps = torch.empty_like(p_0) if want_eps else None
    probs = torch.empty(N, L, 20, dtype=torch.float32, device=p_0.device) if want_probs else None
    nz = None
    if noise is not None:
        o = not noise_structure                  # sequence-only noising draws s_noisy alone (dpm_full.py:163-178)
        nz = AddNoiseNoise(ptr(noise.get('axis'), torch.float32, optional=o), ptr(noise.get('bin'), torch.int64, optional=o),
                           ptr(noise.get('ubin'), torch.float32, optional=o), ptr(noise.get('gauss'), torch.float32, optional=o),
                           ptr(noise.get('pos'), torch.float32, optional=o), ptr(noise.get('s_noisy'), torch.int64, optional=True))
    mean_arr = (C.c_float * 3)(*[float(m) for m in mean])
    cdf = fwd.cdf() if noise is None else None
    t, v_0, p_0, s_0, mask_generate = _contig(t, v_0, p_0, s_0, mask_generate)
    _check(lib().abopt_add_noise(ptr(t, torch.int64), ptr(alpha_bars, torch.float32), ptr(fwd.stddevs, torch.float32),
                                 ptr(fwd.approx_flag, torch.bool), ptr(fwd.X, torch.float32), ptr(cdf, optional=True), fwd.X.shape[1], fwd.X.shape[0],
                                 C.byref(nz) if nz is not None else None, seed, offset,
                                 ptr(v_0, torch.float32), ptr(p_0, torch.float32), ptr(s_0, torch.int64),
                                 ptr(mask_generate, torch.bool), float(scale), mean_arr, int(noise_structure), int(noise_sequence),
                                 int(grad_mode), ptr(v_n), ptr(p_n), ptr(s_n), ptr(eps, optional=True), ptr(probs, optional=True),
                                 ptr(seed_dev, torch.int64, optional=True), N, L, stream()))
    out = (v_n, p_n, s_n) + ((eps,) if want_eps else ()) + ((probs,) if want_probs else ())
    return out


def commonness_score(structs):
    structs = structs.contiguous().float()
    B, n, _ = structs.shape
    score = torch.empty(B, device=structs.device)
    _check(lib().abopt_commonness_score(ptr(structs), ptr(score), B, n, stream()))
    return score


def pair_bias_cache_layers(w_pair_bias_list, pair_feat):
    """The cache from bare proj_pair_bias weights (training: built once per step for all blocks) -> list of per-layer views."""
    n = len(w_pair_bias_list)
    ws = _contig(*[w.detach().float() for w in w_pair_bias_list])
    arr = (GaWeights * n)()
    for i, w in enumerate(ws):
        arr[i].w_pair_bias = ptr(w, torch.float32)
    cache = pair_bias_cache(arr, n, pair_feat)
    return list(cache.view(n, -1).unbind(0))


def ipa_core_train_forward(proj_local, R, t, z, mask, w_pair_bias, spatial_coef, pbc=None):
    """Training-mode IPA core: -> feat (N,L,1824), alpha head-major (N,12,L,L)  (include/abopt.h: abopt_ipa_core_train_forward)."""
    N, L = mask.shape
    dev = z.device
    feat = torch.empty(N, L, 1824, device=dev)
    alpha = torch.empty(N, 12, L, L, device=dev)
    nb = lib().abopt_ipa_train_workspace_bytes(N, L)
    buf = Workspace.get(nb, dev)
    proj_local, R, t, z, mask, w_pair_bias, spatial_coef = _contig(proj_local, R, t, z, mask, w_pair_bias, spatial_coef)
    _check(lib().abopt_ipa_core_train_forward(ptr(proj_local, torch.float32), ptr(R, torch.float32), ptr(t, torch.float32),
                                              ptr(z, torch.float32), ptr(mask, torch.bool),
                                              ptr(w_pair_bias, torch.float32), ptr(spatial_coef, torch.float32), ptr(pbc, torch.float32, optional=True),
                                              ptr(feat), ptr(alpha), N, L, z.shape[-1], ptr(buf), buf.numel(), stream()))
    return feat, alpha


def ipa_points_backward(dfeat, feat, R, t):
    """-> dout_cat (N,12,L,56) = [d feat_node | d aggregated points] head-major, delta (N,L,12)  (abopt_ipa_points_backward)."""
    N, L = feat.shape[:2]
    dout_cat = torch.empty(N, 12, L, 56, device=feat.device)
    delta = torch.empty(N, L, 12, device=feat.device)
    dfeat, feat, R, t = _contig(dfeat, feat, R, t)
    _check(lib().abopt_ipa_points_backward(ptr(dfeat, torch.float32), dfeat.shape[-1], ptr(feat, torch.float32),
                                           ptr(R, torch.float32), ptr(t, torch.float32), ptr(dout_cat), ptr(delta), N, L, stream()))
    return dout_cat, delta


def ipa_backward_operands(proj_local, R, t):
    """-> Aq, Ak (N,12,L,57), Av (N,12,L,56): head-major GEMM operands of the IPA backward (abopt_ipa_backward_operands)."""
    N, L = proj_local.shape[:2]
    dev = proj_local.device
    Aq, Ak, Av = torch.empty(N, 12, L, 57, device=dev), torch.empty(N, 12, L, 57, device=dev), torch.empty(N, 12, L, 56, device=dev)
    proj_local, R, t = _contig(proj_local, R, t)
    _check(lib().abopt_ipa_backward_operands(ptr(proj_local, torch.float32), ptr(R, torch.float32), ptr(t, torch.float32),
                                             ptr(Aq), ptr(Ak), ptr(Av), N, L, stream()))
    return Aq, Ak, Av


def ipa_backward_assemble(P1, P2, P3, Aq, Ak, R, spatial_coef):
    """-> d proj_local (N,L,2016), e (N,L,12) with e.sum((0, 1)) = d loss / d spatial_coef  (abopt_ipa_backward_assemble)."""
    N, _, L, _ = P1.shape
    dproj = torch.empty(N, L, 2016, device=P1.device)
    e = torch.empty(N, L, 12, device=P1.device)
    P1, P2, P3, R, spatial_coef = _contig(P1, P2, P3, R, spatial_coef)
    _check(lib().abopt_ipa_backward_assemble(ptr(P1, torch.float32), ptr(P2, torch.float32), ptr(P3, torch.float32),
                                             ptr(Aq, torch.float32), ptr(Ak, torch.float32), ptr(R, torch.float32),
                                             ptr(spatial_coef, torch.float32), ptr(dproj), ptr(e), N, L, stream()))
    return dproj, e


def ipa_pair_backward(z, alpha, dalpha_node, delta, dfeat, w_pair_bias, dz_into=None, want_dz=True, reduce=None):
    """alpha, dalpha_node head-major (N,12,L,L) -> g (N,12,L,L), dz (N,L,L,C), dWb (12,C)  (include/abopt.h: abopt_ipa_pair_backward).
    dz_into: an existing d pair_feat buffer this block's gradient is ADDED to (returned as dz).  want_dz=False: dz is None -- the caller
    assembles d pair_feat of all blocks at once (ipa_dz_assemble).  reduce: the column-sum function for the per-row partials of dWb (default
    hip.colsum; training passes WgradGroup's, which rides in the grouped weight-gradient launch)."""
    N, L = z.shape[:2]
    g = torch.empty_like(alpha)
    dz = None if not want_dz else (torch.empty_like(z) if dz_into is None else dz_into)
    dwb_rows = torch.empty(N * L, 12 * z.shape[-1], device=z.device)
    z, alpha, dalpha_node, delta, dfeat, w_pair_bias = _contig(z, alpha, dalpha_node, delta, dfeat, w_pair_bias)
    _check(lib().abopt_ipa_pair_backward(ptr(z, torch.float32), ptr(alpha, torch.float32), ptr(dalpha_node, torch.float32),
                                         ptr(delta, torch.float32), ptr(dfeat, torch.float32), dfeat.shape[-1],
                                         ptr(w_pair_bias, torch.float32), ptr(g), ptr(dz, torch.float32, optional=True), ptr(dwb_rows), int(dz_into is not None), N, L, z.shape[-1], stream()))
    return g, dz, (reduce or colsum)(dwb_rows).view(12, z.shape[-1])


def ipa_dz_assemble(alphas, gs, dfeats, wbs, like):
    """d pair_feat (shape of `like`) of all blocks of an encoder from their (alpha, g, d feat, proj_pair_bias.weight): abopt_ipa_dz_assemble."""
    nl = len(alphas)
    N, L, _, Cd = like.shape
    keep = [_contig(*ts) for ts in (alphas, gs, dfeats, [w.detach().float() for w in wbs])]
    ld = keep[2][0].shape[-1]
    assert all(t.shape[-1] == ld for t in keep[2])
    arr = lambda ts: (C.c_void_p * nl)(*[ptr(t, torch.float32).value for t in ts])
    dz = torch.empty_like(like)
    _check(lib().abopt_ipa_dz_assemble(nl, arr(keep[0]), arr(keep[1]), arr(keep[2]), ld, arr(keep[3]), ptr(dz), N, L, Cd, stream()))
    return dz


def encode_inputs(aa, res_nb, chain_nb, pos_atoms, mask_atoms, atoms, fragment_type=None, hotspot=None, structure_mask=None, sequence_mask=None):
    """-> (EncodeInputs, keepalive list).  Tensors are made contiguous; the struct only borrows their pointers."""
    N, L = aa.shape
    keep = [aa.contiguous(), res_nb.contiguous(), chain_nb.contiguous(), pos_atoms.contiguous(), mask_atoms.contiguous()]
    opt = [None if t is None else t.contiguous() for t in (fragment_type, hotspot, structure_mask, sequence_mask)]
    s = EncodeInputs(N, L, pos_atoms.shape[2], atoms, ptr(keep[0], torch.int64), ptr(keep[1], torch.int64), ptr(keep[2], torch.int64),
                     ptr(opt[0], torch.int64, optional=True), ptr(opt[1], torch.int64, optional=True), ptr(keep[3], torch.float32),
                     ptr(keep[4], torch.bool), ptr(opt[2], torch.bool, optional=True), ptr(opt[3], torch.bool, optional=True))
    return s, keep + opt


def residue_embed_forward(inp, weights, has_hotspot):
    N, L = inp.N, inp.L
    dev = torch.device('cuda', torch.cuda.current_device())
    res_feat = torch.empty(N, L, 128, device=dev)
    R = torch.empty(N, L, 3, 3, device=dev)
    p = torch.empty(N, L, 3, device=dev)
    nb = lib().abopt_residue_embed_workspace_bytes(N, L, inp.atoms, int(has_hotspot))
    buf = Workspace.get(nb, dev)
    _check(lib().abopt_residue_embed_forward(C.byref(inp), C.byref(weights), ptr(res_feat), ptr(R), ptr(p), ptr(buf), buf.numel(), stream()))
    return res_feat, R, p


PAIR_ACT = 288


def residue_features(inp, weights, in_dim):
    """-> features (N*L, in_dim) (a view of a buffer whose rows are padded to a multiple of 4 floats), R (N,L,3,3), p (N,L,3):
    abopt_residue_features, the inputs of ResidueEmbedding's MLP (residue.py:33-88)."""
    N, L = inp.N, inp.L
    dev = torch.device('cuda', torch.cuda.current_device())
    ld = (in_dim + 3) & ~3
    feat = torch.empty(N * L, ld, device=dev)
    R, p = torch.empty(N, L, 3, 3, device=dev), torch.empty(N, L, 3, device=dev)
    nb = lib().abopt_residue_features_workspace_bytes(N, L)
    buf = Workspace.get(nb, dev)
    _check(lib().abopt_residue_features(C.byref(inp), C.byref(weights), ptr(feat), ptr(R), ptr(p), ptr(buf), buf.numel(), stream()))
    return feat[:, :in_dim], R, p


def pair_embed_forward(inp, weights, save_activations=False, save_T=False):
    """-> pair_feat (N,L,L,64) [, activations (N,L,L,288), G, T (N,L,L,atoms*16) for the training backward; T is None unless save_T:
    the backward recomputes it from the atoms]."""
    N, L = inp.N, inp.L
    dev = torch.device('cuda', torch.cuda.current_device())
    pair_feat = torch.empty(N, L, L, 64, device=dev)
    acts = G = T = None
    if save_activations:
        acts = torch.empty(N, L, L, PAIR_ACT, device=dev)
        G = torch.empty(N, L, L, inp.atoms * 16, device=dev)
        T = torch.empty(N, L, L, inp.atoms * 16, device=dev) if save_T else None
    nb = lib().abopt_pair_embed_workspace_bytes(N, L, inp.atoms)
    buf = Workspace.get(nb, dev)
    _check(lib().abopt_pair_embed_forward(C.byref(inp), C.byref(weights), ptr(pair_feat), ptr(acts, optional=True), ptr(G, optional=True),
                                          ptr(T, optional=True), ptr(buf), buf.numel(), stream()))
    return (pair_feat, acts, G, T) if save_activations else pair_feat


PAIR_DY = 320


def pair_embed_backward(inp, weights, dpair_feat, acts, T=None, colsum=False):
    """-> dys (N,L,L,320), ds (N,L,L,atoms*16) [, column sums of dys (320) when colsum]  (include/abopt.h: abopt_pair_embed_backward).
    T None: recomputed in the kernel."""
    N, L = inp.N, inp.L
    dev = acts.device
    dys = torch.empty(N, L, L, PAIR_DY, device=dev)
    ds = torch.empty(N, L, L, inp.atoms * 16, device=dev)
    db = torch.empty(PAIR_DY, device=dev) if colsum else None
    nb = lib().abopt_pair_embed_backward_workspace_bytes(N, L, inp.atoms)
    buf = Workspace.get(nb, dev)
    dpair_feat, = _contig(dpair_feat)
    _check(lib().abopt_pair_embed_backward(C.byref(inp), C.byref(weights), ptr(dpair_feat, torch.float32), ptr(acts, torch.float32),
                                           ptr(T, torch.float32, optional=True), ptr(dys), ptr(ds), ptr(db, optional=True), ptr(buf), buf.numel(), stream()))
    return (dys, ds, db) if colsum else (dys, ds)


_BB_TABLES = {}


def reconstruct_backbone_partially(pos_ctx, R_new, t_new, aa, chain_nb, res_nb, mask_atoms, mask_recons):
    """geometry.py:404-480 on the device -> (pos_new (N,L,A,3), mask_new (N,L,A) bool)."""
    dev = pos_ctx.device
    if dev not in _BB_TABLES:
        import numpy as np
        d = np.load(os.path.join(_HERE, 'data', 'backbone_ideal.npz'))
        _BB_TABLES[dev] = (torch.from_numpy(d['bb_table']).to(dev).contiguous(), torch.from_numpy(d['o_table']).to(dev).contiguous())
    bb, ot = _BB_TABLES[dev]
    N, L, A = mask_atoms.shape
    pos_new = torch.empty(N, L, A, 3, device=dev)
    mask_new = torch.empty(N, L, A, dtype=torch.bool, device=dev)
    pos_ctx, R_new, t_new, aa, chain_nb, res_nb, mask_atoms, mask_recons = _contig(pos_ctx, R_new, t_new, aa, chain_nb, res_nb, mask_atoms, mask_recons)
    _check(lib().abopt_reconstruct_backbone_partially(ptr(pos_ctx, torch.float32), ptr(R_new, torch.float32),
                                                      ptr(t_new, torch.float32), ptr(aa, torch.int64),
                                                      ptr(chain_nb, torch.int64), ptr(res_nb, torch.int64),
                                                      ptr(mask_atoms, torch.bool), ptr(mask_recons, torch.bool),
                                                      ptr(bb), ptr(ot), ptr(pos_new), ptr(mask_new), N, L, A, stream()))
    return pos_new, mask_new


def dockq_lite(model_pos, model_mask, native_pos, native_mask, group, check=True):
    """DockQ of S candidates against one native (include/abopt.h: abopt_dockq_lite) -> (S, 4) = fnat, irms, Lrms, DockQ.
    model_pos (S,L,A,3); model_mask (S,L,A) or (L,A) shared; native_pos (L,A,3); native_mask (L,A); group (L,) int {0,1,2}.
    check=True (one device read-back) raises, like the reference's asserts (DockQ.py:150-188), when a candidate has no CA atom
    common to model and native in its interface, receptor or ligand; check=False leaves the kernel's -1 markers in place."""
    S, L, A, _ = model_pos.shape
    shared = model_mask.dim() == 2
    model_pos, model_mask, native_pos, native_mask = _contig(model_pos.float(), model_mask, native_pos.float(), native_mask)
    group = group.to(torch.int32).contiguous()
    out = torch.empty(S, 4, dtype=torch.float32, device=model_pos.device)
    nb = lib().abopt_dockq_workspace_bytes(L)
    buf = Workspace.get(nb, model_pos.device)
    _check(lib().abopt_dockq_lite(ptr(model_pos, torch.float32), ptr(model_mask, torch.bool), int(shared), ptr(native_pos, torch.float32),
                                  ptr(native_mask, torch.bool), ptr(group, torch.int32), S, L, A, ptr(out), ptr(buf), buf.numel(), stream()))
    if check and S > 0 and bool((out[:, 1:3] < 0).any()):
        bad = (out[:, 1:3] < 0).any(1).nonzero().flatten().tolist()
        raise ValueError(f'dockq_lite: candidates {bad} have an empty interface / receptor / ligand CA selection (no atoms in both model and native)')
    return out


def pack_tail_weights(w_out, w0, w1, w2, transposed=False):
    """(w_out_frag, w_mlp_frag[, w_mlpT_frag]) packed on the device in one launch (include/abopt.h: abopt_pack_tail_weights)."""
    dev = w_out.device
    w_out, w0, w1, w2 = _contig(w_out.detach().float(), w0.detach().float(), w1.detach().float(), w2.detach().float())
    wof = torch.empty(lib().abopt_out_frag_floats(), dtype=torch.float32, device=dev)
    wmf = torch.empty(lib().abopt_mlp_frag_floats(), dtype=torch.float32, device=dev)
    wmt = torch.empty(lib().abopt_mlp_frag_floats(), dtype=torch.float32, device=dev) if transposed else None
    _check(lib().abopt_pack_tail_weights(ptr(w_out, torch.float32), ptr(w0, torch.float32), ptr(w1, torch.float32), ptr(w2, torch.float32),
                                         ptr(wof), ptr(wmf), ptr(wmt, torch.float32, optional=True), stream()))
    return (wof, wmf, wmt) if transposed else (wof, wmf)


def out_frag_terms(wof):
    """w_out_frag -> w_out_terms, what the fused core + tail kernel streams (abopt_out_frag_terms; since ABI 39 the two layouts are the same and
    this is a copy)."""
    wot = torch.empty(lib().abopt_out_terms_floats(), dtype=torch.float32, device=wof.device)
    _check(lib().abopt_out_frag_terms(ptr(wof, torch.float32), ptr(wot), stream()))
    return wot


def block_tail_forward(feat, wof, wmf, x, b_out, mask, g1, be1, b0, b1, b2, g2, be2, save=False):
    """out = LN2(y + MLP(y)), y = LN1(x + mask * (feat W_out^T + b_out)) for [rows, .] inputs (abopt_block_tail_forward).
    save=True also returns the [5, rows, 128] activation dump the backward consumes."""
    rows = x.numel() // 128
    feat, x, mask, b_out, g1, be1, b0, b1, b2, g2, be2 = _contig(feat, x, mask, b_out, g1, be1, b0, b1, b2, g2, be2)
    out = torch.empty_like(x)
    saved = torch.empty(5, rows, 128, dtype=torch.float32, device=x.device) if save else None
    f = lambda t: ptr(t, torch.float32)
    _check(lib().abopt_block_tail_forward(f(feat), f(wof), f(wmf), f(x), f(b_out), ptr(mask, torch.bool), f(g1), f(be1), f(b0), f(b1), f(b2), f(g2), f(be2),
                                          ptr(out), ptr(saved, torch.float32, optional=True), rows, stream()))
    return (out, saved) if save else out


def block_tail_backward(dout, saved, wmt, mask, g1, g2, reduce=None):
    """Row-local backward of the tail (abopt_block_tail_backward) -> dpre [3, rows, 128], da1, du [rows, 128], colsum [8, 128].
    reduce: the column-sum function for the per-workgroup partials (default hip.colsum; see ipa_pair_backward)."""
    rows = saved.shape[1]
    dout, mask, g1, g2 = _contig(dout, mask, g1, g2)
    dev = dout.device
    dpre = torch.empty(3, rows, 128, dtype=torch.float32, device=dev)
    da1 = torch.empty(rows, 128, dtype=torch.float32, device=dev)
    du = torch.empty(rows, 128, dtype=torch.float32, device=dev)
    colpart = torch.empty((rows + 31) // 32, 8, 128, dtype=torch.float32, device=dev)
    f = lambda t: ptr(t, torch.float32)
    _check(lib().abopt_block_tail_backward(f(dout), f(saved), f(wmt), ptr(mask, torch.bool), f(g1), f(g2), ptr(dpre), ptr(da1), ptr(du), ptr(colpart),
                                           rows, stream()))
    return dpre, da1, du, (reduce or colsum)(colpart.view(colpart.shape[0], -1)).view(8, 128)


def _operand(t):
    """(tensor, ld, batch stride, transposed) of a 2-D / 3-D fp32 operand whose matrices are row- or column-major; anything else is copied."""
    if t.dim() == 2:
        t = t.unsqueeze(0)
    b, r, c = t.shape
    sb, sr, sc_ = t.stride()
    if b == 1:
        sb = 0
    if sc_ == 1 and sr >= max(c, 1):
        return t, sr, sb, 0
    if sr == 1 and sc_ >= max(r, 1):
        return t, sc_, sb, 1
    t = t.contiguous()
    return t, t.stride(1), (t.stride(0) if b > 1 else 0), 0


def gemm(a, b, alpha=1.0, out=None, bias=None, relu=False):
    """C = alpha * a @ b^T on libabopt_hip.so (include/abopt.h: abopt_gemm).  a (M,K) or (B,M,K); b (N,K) or (B,N,K); either may be
    a transposed VIEW (x.t(), x.transpose(1, 2)): the kernel reads k-strided operands in place.  Batch broadcasting: a 2-D operand
    serves every batch.  bias (N,) and relu: y = relu(a b^T + bias) in the product's epilogue.  out (optional): (M,N) / (B,M,N) fp32 with
    unit column stride; it may be a column slice of a wider matrix (ldc = its row stride > N) -- the other columns are left untouched."""
    nb = max(a.shape[0] if a.dim() == 3 else 1, b.shape[0] if b.dim() == 3 else 1)
    a, lda, sa, at = _operand(a.float())
    b, ldb, sb, bt = _operand(b.float())
    M, K, N = a.shape[1], a.shape[2], b.shape[1]
    assert b.shape[2] == K, (a.shape, b.shape)
    c = torch.empty(nb, M, N, dtype=torch.float32, device=a.device) if out is None else out
    ldc, sc = N, M * N
    if out is not None:
        o3 = out if out.dim() == 3 else out.unsqueeze(0)
        if o3.shape != (nb, M, N) or o3.stride(2) != 1 or out.dtype != torch.float32 or not out.is_cuda:
            raise TypeError('gemm: out must be fp32 (B,M,N) on the device with unit column stride')
        ldc, sc = o3.stride(1), (o3.stride(0) if nb > 1 else 0)
    tiles = ((M + 63) // 64) * ((N + 63) // 64) * nb
    ws = None
    # split-K slabs: the C side (gemm.hip: launch_gemm_batched) splits below 256 output tiles, for a densely packed C only, into at most
    # min(1024 / tiles, K / 256) slabs -- size the workspace for exactly that (an env-tuned ABOPT_GEMM_TMAX / _KDIV build clamps to what it gets)
    dense_c = ldc == N and (nb == 1 or sc == M * N)
    if tiles < 256 and K >= 1024 and bias is None and not relu and dense_c:
        ws = Workspace.get(max(1, min(1024 // tiles, K // 256)) * nb * M * N * 4, a.device)
    if bias is not None:
        bias = bias.detach().float().contiguous()
        assert bias.numel() == N
    _check(lib().abopt_gemm(ptr(a, torch.float32, strided=True), lda, sa, at, ptr(b, torch.float32, strided=True), ldb, sb, bt, ptr(c, torch.float32, strided=True), ldc, sc, M, N, K, nb, float(alpha),
                            ptr(bias, torch.float32, optional=True), int(bool(relu)), ptr(ws, optional=True), ws.numel() if ws is not None else 0, stream()))
    return c


GEMM_TN_GROUP_MAX = 24


def gemm_tn_grouped(pairs, outs=None):
    """[a_p^T @ b_p for (a_p, b_p) in pairs] -- tall 2-D fp32 operands (K_p, M_p), (K_p, N_p) with unit column stride, column slices of wider
    matrices read in place -- as ONE product launch + ONE slab-sum launch per 24 pairs (include/abopt.h: abopt_gemm_tn_grouped): the
    weight-gradient products of a backward pass.  Returns the list of (M_p, N_p) results (`outs`: contiguous fp32 tensors to write, else fresh ones)."""
    res = []
    for i0 in range(0, len(pairs), GEMM_TN_GROUP_MAX):
        chunk = pairs[i0:i0 + GEMM_TN_GROUP_MAX]
        arr = (GemmTnProblem * len(chunk))()
        keep, tiles = [], 0
        for j, (q, (a, b)) in enumerate(zip(arr, chunk)):
            if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0] or a.stride(1) != 1 or b.stride(1) != 1 or a.shape[0] == 0:
                raise TypeError(f'gemm_tn_grouped: operands must be 2-D (K, M) / (K, N) with unit column stride and K > 0, got {tuple(a.shape)} {tuple(a.stride())} / {tuple(b.shape)} {tuple(b.stride())}')
            c = torch.empty(a.shape[1], b.shape[1], dtype=torch.float32, device=a.device) if outs is None else outs[i0 + j]
            if tuple(c.shape) != (a.shape[1], b.shape[1]):
                raise TypeError(f'gemm_tn_grouped: out {tuple(c.shape)} for operands {tuple(a.shape)} / {tuple(b.shape)}')
            q.a, q.b, q.c = ptr(a, torch.float32, strided=True), ptr(b, torch.float32, strided=True), ptr(c)
            q.lda, q.ldb, q.m, q.n, q.k = a.stride(0), b.stride(0), a.shape[1], b.shape[1], a.shape[0]
            tiles += ((q.m + 63) // 64) * ((q.n + 63) // 64)
            keep.append(c)
        want = max(1024 // tiles, 1) if tiles < 256 else 1
        need = sum(min(want, q.k // 256) * q.m * q.n for q in arr if min(want, q.k // 256) > 1)
        ws = Workspace.get(need * 4, chunk[0][0].device) if need else None
        _check(lib().abopt_gemm_tn_grouped(arr, len(chunk), ptr(ws, optional=True), ws.numel() if ws is not None else 0, stream()))
        res += keep
    return res


def colsum(x):
    """x.sum(0) for a 2-D fp32 tensor with unit column stride (a row-sliced / column-sliced view is read in place): abopt_colsum."""
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32 or not x.is_cuda or x.shape[0] == 0:
        return x.sum(0)
    rows, cols = x.shape
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    ws = Workspace.get(1024 * max(cols, 128) * 4, x.device)
    _check(lib().abopt_colsum(ptr(x, torch.float32, strided=True), x.stride(0), rows, cols, ptr(out), ptr(ws), ws.numel(), stream()))
    return out


def dpm_losses(R_pred, R_0, p_pred, p_target, c_den, s_t, s_0, abar_t, mask_generate):
    """-> sums (3,) of (rot, pos, seq) over the generated residues, and d(sum)/d(R_pred, p_pred, c_den) (abopt_dpm_losses)."""
    R_pred, R_0, p_pred, p_target, c_den, s_t, s_0, abar_t, mask_generate = _contig(R_pred.float(), R_0.float(), p_pred.float(), p_target.float(), c_den.float(),
                                                                                    s_t, s_0, abar_t.float(), mask_generate)
    N, L = mask_generate.shape
    nblk = (N * L + 255) // 256
    part = torch.empty(nblk, 3, dtype=torch.float32, device=R_pred.device)
    gR, gp, gc = torch.empty_like(R_pred), torch.empty_like(p_pred), torch.empty_like(c_den)
    _check(lib().abopt_dpm_losses(ptr(R_pred, torch.float32), ptr(R_0, torch.float32), ptr(p_pred, torch.float32), ptr(p_target, torch.float32), ptr(c_den, torch.float32),
                                  ptr(s_t, torch.int64), ptr(s_0, torch.int64), ptr(abar_t, torch.float32), ptr(mask_generate, torch.bool), N, L, ptr(part), ptr(gR), ptr(gp),
                                  ptr(gc), stream()))
    return colsum(part), gR, gp, gc


def abdock_losses(prmsd_logits, p_pred, p0n, coef_a, coef_b, mask_generate, mask_res, offsets, scale, pred_x0):
    """-> parts (N,4) = {CE_n, m0_n, smooth-l1 sum_n, count_n}, d prmsd_logits (N,nb) = softmax - onehot, d p_pred (N,L,3) of the smooth-l1 sum
    (abopt_abdock_losses: the prmsd and dist losses of the AbDock flavour, dpm_full.py:180-198,369-378)."""
    N, L = mask_generate.shape
    nb = prmsd_logits.shape[-1]
    prmsd_logits, p_pred, p0n, mask_generate, mask_res, offsets = _contig(prmsd_logits.float(), p_pred.float(), p0n.float(), mask_generate, mask_res, offsets.float())
    ca = cb = None
    if not pred_x0:
        ca, cb = _contig(coef_a.float(), coef_b.float())
    part = torch.empty(N, 4, dtype=torch.float32, device=p_pred.device)
    gl, gp = torch.empty_like(prmsd_logits), torch.empty_like(p_pred)
    _check(lib().abopt_abdock_losses(ptr(prmsd_logits, torch.float32), ptr(p_pred, torch.float32), ptr(p0n, torch.float32), ptr(ca, torch.float32, optional=True),
                                     ptr(cb, torch.float32, optional=True), ptr(mask_generate, torch.bool), ptr(mask_res, torch.bool), ptr(offsets, torch.float32), nb, N, L,
                                     C.c_float(float(scale)), int(bool(pred_x0)), ptr(part), ptr(gl), ptr(gp), stream()))
    return part, gl, gp


def layer_norm_forward(x, gamma, beta, eps, save=True):
    """Rows of x (.., cols <= 256) through the reference's LayerNorm (layers.py:146-155) -> y [, xhat, rstd for the backward] (abopt_layer_norm_forward)."""
    cols = x.shape[-1]
    x2, gamma, beta = _contig(x.float().reshape(-1, cols), gamma.detach().float(), beta.detach().float())
    rows = x2.shape[0]
    y = torch.empty_like(x2)
    xhat = torch.empty_like(x2) if save else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x2.device) if save else None
    _check(lib().abopt_layer_norm_forward(ptr(x2, torch.float32), ptr(gamma, torch.float32), ptr(beta, torch.float32), cols, C.c_float(float(eps)), C.c_int64(rows), ptr(y),
                                          ptr(xhat, torch.float32, optional=True), ptr(rstd, torch.float32, optional=True), stream()))
    return y.view(x.shape), xhat, rstd


def layer_norm_backward(dy, xhat, rstd, gamma):
    """-> dx (shape of dy), d gamma, d beta (abopt_layer_norm_backward + two column sums)."""
    cols = dy.shape[-1]
    dy2, gamma = _contig(dy.float().reshape(-1, cols), gamma.detach().float())
    rows = dy2.shape[0]
    dx, dyx = torch.empty_like(dy2), torch.empty_like(dy2)
    _check(lib().abopt_layer_norm_backward(ptr(dy2, torch.float32), ptr(xhat, torch.float32), ptr(rstd, torch.float32), ptr(gamma, torch.float32), cols, C.c_int64(rows),
                                           ptr(dx), ptr(dyx), stream()))
    return dx.view(dy.shape), colsum(dyx), colsum(dy2)


def heads_epilogue_forward(R, eps_crd, eps_rot, mask_generate):
    """-> R_next (.., 3, 3), eps_pos (.., 3): dpm_full.py:95-101 without the log map (abopt_heads_epilogue_forward)."""
    R, eps_crd, eps_rot, mask_generate = _contig(R.float(), eps_crd.float(), eps_rot.float(), mask_generate)
    rows = mask_generate.numel()
    R_next, eps_pos = torch.empty_like(R), torch.empty_like(eps_crd)
    _check(lib().abopt_heads_epilogue_forward(ptr(R, torch.float32), None, ptr(eps_crd, torch.float32), ptr(eps_rot, torch.float32), ptr(mask_generate, torch.bool),
                                              None, ptr(R_next), ptr(eps_pos), C.c_int64(rows), 0, stream()))
    return R_next, eps_pos


def heads_epilogue_backward(R, eps_rot, mask_generate, dR_next, deps_pos):
    """-> d eps_crd, d eps_rot (.., 3) (abopt_heads_epilogue_backward); dR_next / deps_pos may be None (= zero)."""
    R, eps_rot, mask_generate = _contig(R.float(), eps_rot.float(), mask_generate)
    dR_next = None if dR_next is None else dR_next.float().contiguous()
    deps_pos = None if deps_pos is None else deps_pos.float().contiguous()
    rows = mask_generate.numel()
    d_crd, d_rot = torch.empty_like(eps_rot), torch.empty_like(eps_rot)
    _check(lib().abopt_heads_epilogue_backward(ptr(R, torch.float32), ptr(eps_rot, torch.float32), ptr(mask_generate, torch.bool), ptr(dR_next, torch.float32, optional=True),
                                               ptr(deps_pos, torch.float32, optional=True), ptr(d_crd), ptr(d_rot), C.c_int64(rows), stream()))
    return d_crd, d_rot


def bucket_colsum(x, idx, buckets):
    """out[b] = sum of the rows of x (2-D fp32, unit column stride, read in place) whose idx is b; rows with idx < 0 are skipped."""
    rows, cols = x.shape
    if x.stride(1) != 1 or x.dtype != torch.float32 or not x.is_cuda or idx.dtype != torch.int32 or idx.numel() != rows:
        raise TypeError('bucket_colsum: fp32 [rows, cols] with unit column stride and an int32 index per row')
    out = torch.empty(buckets, cols, dtype=torch.float32, device=x.device)
    ws = Workspace.get(max(1, min(rows // 64, 2048 // ((cols + 63) // 64))) * buckets * cols * 4, x.device)      # the slices launch_bucket_colsum takes (gemm.hip)
    _check(lib().abopt_bucket_colsum(ptr(x, torch.float32, strided=True), x.stride(0), rows, cols, ptr(idx.contiguous(), torch.int32), buckets, ptr(out),
                                     ptr(ws), ws.numel(), stream()))
    return out


def segment_bucket_colsum(x, segments, idx, idx_div, buckets):
    """x (2-D fp32 [segments * rows_per_segment, cols], unit column stride, read in place) -> [segments, buckets, cols]: per segment, the rows
    summed by bucket; the bucket of row j of segment s is idx.flatten()[(s // idx_div) * rows_per_segment + j] (abopt_segment_bucket_colsum)."""
    rows, cols = x.shape
    rps = rows // segments
    if x.stride(1) != 1 or x.dtype != torch.float32 or not x.is_cuda or idx.dtype != torch.int32 or rps * segments != rows or idx.numel() * idx_div != rows:
        raise TypeError('segment_bucket_colsum: fp32 [segments * rows_per_segment, cols] with unit column stride and an int32 index row per idx_div segments')
    out = torch.empty(segments, buckets, cols, dtype=torch.float32, device=x.device)
    _check(lib().abopt_segment_bucket_colsum(ptr(x, torch.float32, strided=True), x.stride(0), segments, rps, cols, ptr(idx.contiguous(), torch.int32), idx_div, buckets,
                                             ptr(out), stream()))
    return out


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay=0.0, max_grad_norm=None, grad_norm_out=None, ws=None,
              hyper_dev=None):
    """clip_grad_norm_ + torch.optim.Adam.step for a list of fp32 tensors in a handful of launches (abopt_adam_step; A/train.py:116-117).
    step: int64 device tensor of one element, incremented by the call.  grad_norm_out (1 float on the device) receives the unclipped norm
    when max_grad_norm is given.  The gradient tensors are read, not rewritten.  ws: the caller's scratch (adam_ws_bytes), else the shared one.
    hyper_dev: 6 float64 on the device {lr, beta1, beta2, eps, weight_decay, max_grad_norm}, read by the kernels instead of the scalars
    (graph replays follow a scheduler)."""
    n = len(params)
    if n == 0:
        return
    for group in (params, grads, exp_avg, exp_avg_sq):
        for t in group:
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise TypeError('adam_step: contiguous fp32 HIP tensors only')
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    numel = (C.c_int64 * n)(*[t.numel() for t in params])
    need = lib().abopt_adam_ws_floats(n, numel)
    if ws is None or ws.numel() * ws.element_size() < need * 4:
        ws = Workspace.get(need * 4, params[0].device)
    _check(lib().abopt_adam_step(n, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), numel, lr, beta1, beta2, eps, weight_decay,
                                 float(max_grad_norm) if max_grad_norm is not None else 0.0, ptr(step, torch.int64), ptr(ws), need,
                                 ptr(grad_norm_out, torch.float32, optional=True), ptr(hyper_dev, torch.float64, optional=True), stream()))


def adam_ws_bytes(params):
    n = len(params)
    return 4 * lib().abopt_adam_ws_floats(n, (C.c_int64 * n)(*[t.numel() for t in params])) if n else 0


def prof_enable(on=True):
    _check(lib().abopt_prof_enable(int(on)))


GRAPH_CAPTURE_EVENTS = False      # bench.py experiment: keep the IPA-core event records inside a captured loop graph
GRAPH_CAPTURE_SPANS = False       # bench.py: give the dominant kernel's launches span slots while a loop graph is captured (abopt_prof_spans)


def prof_spans_reset():
    _check(lib().abopt_prof_spans_reset(stream()))


def prof_spans():
    """(launches, total_ms) of the 32-row IPA launches that ran since prof_spans_reset(), from their in-kernel wall-clock spans."""
    n, ms = C.c_int(), C.c_double()
    _check(lib().abopt_prof_spans(C.byref(n), C.byref(ms)))
    return n.value, ms.value


def prof_clock():
    """(cycles, seconds, GHz) of wave 0 / workgroup 0 of the most recent 32-row IPA launch (abopt_prof_clock); None before any."""
    c, w = C.c_longlong(), C.c_longlong()
    _check(lib().abopt_prof_clock(C.byref(c), C.byref(w)))
    if w.value <= 0:
        return None
    return c.value, w.value * 1e-8, c.value / (w.value * 10.0)


def prof_collect(keep=False):
    """(launches, total_ms) of the IPA-core kernel since prof_enable(True).  keep=True: do not forget the event pairs (they were
    captured into a hipGraph and every replay records them again)."""
    n, ms = C.c_int(), C.c_double()
    _check((lib().abopt_prof_peek if keep else lib().abopt_prof_collect)(C.byref(n), C.byref(ms)))
    return n.value, ms.value

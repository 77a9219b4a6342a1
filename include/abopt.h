/* abopt.h -- C ABI of libabopt_hip.so: the MI355X (gfx950) denoising hot path of
 * pengzhangzhi/ab_opt (AbDock / AbDesign).
 *
 * The reference is pure Python/PyTorch: it has no FFI of its own, so this is the boundary a
 * maintainer binds with ctypes under the reference's nn.Module methods (INTEGRATION.md shows the
 * stub).  Every entry point names the reference function(s) it replaces
 * (D/ = AbDock/src/, A/ = AbDesign/diffab/ in the reference tree).
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch types; all floating point is fp32, row-major,
 *     contiguous; masks are 1 byte per element (torch.bool storage); s_t / aa are int64.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); buffers are owned by
 *     the caller; `ws` is caller-owned scratch of at least the matching *_workspace_bytes().
 *   - no global mutable state: re-entrant across streams (one workspace per stream).
 *   - return 0 on success, else an ABOPT_E* code; abopt_last_error() gives the thread-local text.
 */
#ifndef ABOPT_H
#define ABOPT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABOPT_ABI_VERSION 41

enum { ABOPT_OK = 0, ABOPT_EINVAL = 1, ABOPT_EHIP = 2, ABOPT_EUNSUPPORTED = 3, ABOPT_EWORKSPACE = 4 };

/* Fixed architecture of the IPA block (D/modules/encoders/ga.py:42-43 defaults, used by every shipped config). */
enum { ABOPT_HEADS = 12, ABOPT_QK_DIM = 32, ABOPT_POINTS = 8, ABOPT_AA = 20,
       ABOPT_NODE_PROJ = 2016,   /* 3*H*D + 3*H*P*3: query|key|value|query_pt|key_pt|value_pt */
       ABOPT_IPA_FEAT = 1824 };  /* H*C + H*D + H*P*(3+1+3) at C = 64 */

typedef void* abopt_stream;

int abopt_abi_version(void);
const char* abopt_last_error(void);
/* name/CU count/LDS bytes of the current HIP device; arch receives e.g. "gfx950". */
int abopt_device_info(int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len);

/* ---- SO(3) maps: D/modules/common/so3.py:33-57 (so3vec_to_rotation), :10-30,60-63 (rotation_to_so3vec).
 * w (n,3) <-> R (n,3,3).  grad_mode!=0 selects the reference's autograd-on cosine clamp (-0.999). */
int abopt_so3_exp(const float* w, float* R, int64_t n, abopt_stream stream);
int abopt_so3_log(const float* R, float* w, int64_t n, int grad_mode, abopt_stream stream);

/* ---- One GABlock: D/modules/encoders/ga.py:40-178.  Weight layouts are torch's ([out,in]). */
typedef struct {
    const float* w_node;        /* [2016, F]  rows: proj_query|proj_key|proj_value|proj_query_point|proj_key_point|proj_value_point (ga.py:54-66) */
    const float* w_pair_bias;   /* [12, C]    proj_pair_bias.weight (ga.py:59) */
    const float* spatial_coef;  /* [12]       raw parameter; softplus applied on device (ga.py:62-63,108) */
    const float* w_out;         /* [F, 1824]  out_transform.weight (ga.py:69-73) */
    const float* b_out;         /* [F] */
    const float* ln1_gamma; const float* ln1_beta;   /* layer_norm_1 (D/modules/common/layers.py:109-155) */
    const float* w_mlp0; const float* b_mlp0;        /* mlp_transition.0/.2/.4, each [F,F] + [F] (ga.py:76-78) */
    const float* w_mlp1; const float* b_mlp1;
    const float* w_mlp2; const float* b_mlp2;
    const float* ln2_gamma; const float* ln2_beta;
    const float* w_node_frag;   /* optional, abopt_node_frag_floats() floats = [12, 12, 4, 2, 64, 4] + {S, 1 / S, 0, 0}: w_node re-laid out per head in MFMA
                                   operand order, every weight as the two fp16 terms of S w (layout below); when given, the fused projection kernel
                                   replaces the GEMM + fragment pass (same results up to fp32 summation order) */
    const float* w_out_frag;    /* optional, 128 * 1824 4-byte words = [4 cb][114 s][2 terms][64 lanes] x 8 fp16: w_out in MFMA operand order as the two
                                   fp16 terms h = fp16(S w), l = fp16(S w - h) of [cb][s][lane = 32 kh + c][i] = w_out[32 cb + c][col(16 s + 8 kh + i)],
                                   S = the power of two with max |w_out| S in [2^14, 2^15) (stored in w_mlp_frag, below; the kernels multiply the
                                   sums by 1 / S; activations are split the same way in the kernels, three products per k-step: csrc/ipa_common.h,
                                   split_pair2).  col(k) = k for k >= 768; the 768 pair-feature columns
                                   (head h, channel ch) enter the K index at k = 192 ((ch % 16) / 4) + 16 h + 4 (ch / 16) + ch % 4 -- the order in
                                   which the fused core + tail kernel's lanes hold them (csrc/tail_common.h: ot_feat_col; abopt_pack_tail_weights
                                   produces this layout);
                                   when given together with w_mlp_frag, out_transform runs inside the LayerNorm/MLP kernel (no split-K partial slabs) */
    const float* w_mlp_frag;    /* optional, abopt_mlp_frag_floats() floats: w_mlp0, w_mlp1, w_mlp2 in 16x16x32 MFMA operand order as two fp16 terms of
                                   S_layer w: [layer][ct][s][term][lane = 16 kq + m] x 8 fp16, entry i = term(S w[16 ct + m][32 s + 8 kq + i])
                                   (3 * 128 * 128 words), then 8 floats {S_out, S_0, S_1, S_2, 1 / S_out, 1 / S_0, 1 / S_1, 1 / S_2}, then 256 floats of
                                   packer scratch; the rest of the buffer is zero */
    const float* w_out_terms;   /* optional, abopt_out_terms_floats() floats: a copy of w_out_frag (abopt_out_frag_terms; until ABI 38 the fused kernel
                                   streamed a different layout).  When given (with w_mlp_frag and a pair-bias cache), the IPA core and the tail of the
                                   block run as ONE kernel wherever the 32-row core applies: feat never leaves the chip (bit-identical results) */
} abopt_ga_weights;

/* Host-side description of the w_node_frag layout (used by the binding to pack weights once): for head h, tile T (0,1 q | 2,3 k |
 * 4,5 q_pts | 6,7 k_pts | 8,9 v | 10,11 v_pts), tile row m (0..15) -> source row of w_node, or -1 for a zero row.  Point tiles hold 4
 * points as (x, y, z, pad): m = 4 p + c.  A weight w is stored as two fp16 numbers h = fp16(S w), l = fp16(S w - h) (round to nearest), S the power
 * of two with max |w_node| S in [2^14, 2^15).  Element [h][T][s][term][lane = 16 kq + m] is a 16-byte vector of 8 fp16:
 * entry i = term(S w_node[row][32 s + 8 kq + i]), term 0 = h, 1 = l; the four floats behind the last element are {S, 1 / S, 0, 0}.
 * abopt_node_frag_floats() = size in 4-byte units. */
int abopt_node_frag_source_row(int h, int T, int m);
size_t abopt_node_frag_floats(void);

/* The tail of a GABlock on its own -- out_transform, mask, residual, LayerNorm, mlp_transition, LayerNorm
 * (AbDock/src/modules/encoders/ga.py:174-177; LayerNorm: AbDock/src/modules/common/layers.py:146-155) -- for hosts that drive the
 * block piecewise (the training path: forward with an activation dump, then the row-local part of its backward).
 *   abopt_pack_tail_weights  : w_out [128,1824], w_mlp0..2 [128,128] -> w_out_frag (abopt_out_frag_floats() floats), w_mlp_frag and
 *                              (optional) w_mlpT_frag = the three transposed layers (abopt_mlp_frag_floats() floats each), on the device.
 *   abopt_block_tail_forward : out [rows,128] = LN2(y + MLP(y)), y = LN1(x + mask * (feat W_out^T + b_out)).  saved (optional) receives
 *                              five [rows,128] slabs: x + mask*u | y | h0 = relu(W0 y + b0) | h1 = relu(W1 h0 + b1) | y + W2 h1 + b2.
 *   abopt_block_tail_backward: from d out and `saved`: dpre [3,rows,128] = d loss / d (pre-activation) of the three MLP layers (so
 *                              d W_l = dpre[l]^T . input_l), da1 [rows,128] = gradient reaching x through the residual,
 *                              du = mask * da1 (so d feat = du . W_out, d W_out = du^T . feat), and colpart [ceil(rows/32), 8, 128] =
 *                              per-tile column sums: 0 d ln2.beta | 1 d ln2.gamma | 2 d b_mlp2 | 3 d b_mlp1 | 4 d b_mlp0 | 5 d ln1.beta |
 *                              6 d ln1.gamma | 7 d b_out; the caller sums over tiles (deterministic). */
size_t abopt_out_frag_floats(void);
size_t abopt_heads_frag_floats(void);
size_t abopt_mixer_frag_floats(void);
size_t abopt_mlp_frag_floats(void);
size_t abopt_out_terms_floats(void);
/* w_out_frag (abopt_pack_tail_weights) -> w_out_terms: a copy since ABI 39 (both kernels stream the same pre-split layout). */
int abopt_out_frag_terms(const float* w_out_frag, float* w_out_terms, abopt_stream stream);
int abopt_pack_tail_weights(const float* w_out, const float* w_mlp0, const float* w_mlp1, const float* w_mlp2, float* w_out_frag,
                            float* w_mlp_frag, float* w_mlpT_frag, abopt_stream stream);
int abopt_block_tail_forward(const float* feat, const float* w_out_frag, const float* w_mlp_frag, const float* x, const float* b_out,
                             const uint8_t* mask, const float* ln1_gamma, const float* ln1_beta, const float* b_mlp0, const float* b_mlp1,
                             const float* b_mlp2, const float* ln2_gamma, const float* ln2_beta, float* out, float* saved, int64_t rows,
                             abopt_stream stream);
int abopt_block_tail_backward(const float* dout, const float* saved, const float* w_mlpT_frag, const uint8_t* mask, const float* ln1_gamma,
                              const float* ln2_gamma, float* dpre, float* da1, float* du, float* colpart, int64_t rows, abopt_stream stream);

/* Optional intermediates of one block for parity tests (any pointer may be NULL):
 * logits = (node+pair+spatial)*sqrt(1/3) before masking, alpha after softmax/masking: [N,L,L,12]; feat [N,L,1824]. */
typedef struct { float* logits; float* alpha; float* feat; } abopt_ga_debug;

size_t abopt_ga_workspace_bytes(int N, int L, int F, int C);

/* GABlock.forward (ga.py:149-178): R [N,L,3,3], t [N,L,3] (normalised coords), x [N,L,F], z [N,L,L,C],
 * mask [N,L] -> x_out [N,L,F].  F must be 128 and C 64 in this build (the only shipped shapes). */
int abopt_ga_block_forward(const abopt_ga_weights* w, const float* R, const float* t, const float* x,
                           const float* z, const uint8_t* mask, float* x_out, int N, int L, int F, int C,
                           const abopt_ga_debug* dbg, void* ws, size_t ws_bytes, abopt_stream stream);

/* The same block with what a sampling loop carries from call to call (ABI 41): this block's slice of abopt_pair_bias_cache and, optionally, the
 * abopt_pair_terms of the same z (pair_feat_shared as in abopt_eps_net_forward).  feat_out (optional, [N,L,1824]): the IPA features feeding
 * out_transform (ga.py:141-145) -- asking for them keeps the core and the tail as two launches (same results bit for bit). */
int abopt_ga_block_forward_cached(const abopt_ga_weights* w, const float* R, const float* t, const float* x,
                                  const float* z, const uint8_t* mask, float* x_out, int N, int L, int F, int C,
                                  const float* pair_bias_cache, const float* pair_terms, int pair_feat_shared, float* feat_out,
                                  void* ws, size_t ws_bytes, abopt_stream stream);

/* GAEncoder.forward (ga.py:181-193): num_layers blocks over the same R, t, z. */
int abopt_ga_encoder_forward(const abopt_ga_weights* blocks, int num_layers, const float* R, const float* t,
                             const float* x, const float* z, const uint8_t* mask, float* x_out,
                             int N, int L, int F, int C, void* ws, size_t ws_bytes, abopt_stream stream);

/* ---- EpsilonNet: D/modules/diffusion/dpm_full.py:35-112 (A/...:33-102 without the prmsd head).
 * Head weights are passed packed (see ab_opt_amd/hip.py: pack_eps_weights):
 *   w_head1 [384,132]: rows eps_crd_net.0 | eps_rot_net.0 | eps_seq_net.0, K padded 131->132 with a zero column
 *   w_head3_* keep torch shapes.  prmsd_* may be NULL (AbDesign). */
typedef struct {
    const float* seq_embed;                 /* [25, F]  current_sequence_embedding.weight */
    const float* w_mix0; const float* b_mix0;   /* [F, 2F], [F]   res_feat_mixer.0 */
    const float* w_mix1; const float* b_mix1;   /* [F, F],  [F]   res_feat_mixer.2 */
    const abopt_ga_weights* blocks; int num_layers;
    const float* w_head1; const float* b_head1;     /* [3F, F+4], [3F] */
    const float* w_crd2; const float* b_crd2; const float* w_crd3; const float* b_crd3;   /* [F,F],[F],[3,F],[3] */
    const float* w_rot2; const float* b_rot2; const float* w_rot3; const float* b_rot3;   /* [F,F],[F],[3,F],[3] */
    const float* w_seq2; const float* b_seq2; const float* w_seq3; const float* b_seq3;   /* [F,F],[F],[20,F],[20] */
    const float* prmsd_ln_gamma; const float* prmsd_ln_beta;      /* [F+3] */
    const float* w_prmsd1; const float* b_prmsd1;                 /* [F, F+4] (K padded), [F] */
    const float* w_prmsd2; const float* b_prmsd2;                 /* [F, F], [F] */
    const float* w_prmsd3; const float* b_prmsd3; int num_bins;   /* [num_bins, F], [num_bins] */
    const float* w_heads_frag;  /* optional, abopt_heads_frag_floats() floats = [27, 8, 2, 64, 4] + {S, 1 / S, 0, 0}: the heads' weights as two fp16 terms of S w in
                                   MFMA operand order (block layout of w_out_frag with one column block per entry: [block][s][term][lane = 32 kh + c] -> 8 fp16,
                                   entry i = term(S W[32 blk + c][16 s + 8 kh + i]), S one power of two for the whole buffer): blocks 0..11 = w_head1[:, :F]
                                   (crd | rot | seq first layers), 12..15 w_crd2, 16..19 w_rot2, 20..23 w_seq2, 24 w_crd3, 25 w_rot3, 26 w_seq3 (rows zero-padded to 32).
                                   When given, the three heads run as one kernel (time features enter as an affine term from w_head1[:, F:F+3]). */
    const float* w_mix_frag;    /* optional, abopt_mixer_frag_floats() floats = [8, 8, 2, 64, 4] + {S, 1 / S, 0, 0}: blocks 0..3 = w_mix0[:, :F], 4..7 = w_mix1, same
                                   layout; with mix_table the mixer is one kernel */
    const float* mix_table;     /* optional [25, F]: row s = w_mix0[:, F:] . seq_embed[s] + b_mix0 (the embedding half of the mixer's first layer) */
} abopt_eps_weights;

size_t abopt_eps_workspace_bytes(int N, int L, int F, int C);

/* Per-call pair-bias cache: proj_pair_bias(z) of EVERY block (ga.py:88-90) in one pass over pair_feat.  pair_feat and
 * the weights are constant over the steps of FullDPM.sample / optimize (dpm_full.py:274-283), so the sampler builds the
 * cache once per call and passes it to every abopt_eps_net_forward; the per-step kernel then skips that contraction
 * (bit-identical results).  cache: abopt_pair_bias_cache_bytes(N, L, num_layers) bytes, caller-owned.  The cache is an opaque operand of this library for exactly the
 * (N, L) batch it was built from (per block: chunk of 16 keys outermost, then the rows of the whole batch): it cannot be sliced per sample. */
size_t abopt_pair_bias_cache_bytes(int N, int L, int num_layers);
int abopt_pair_bias_cache(const abopt_ga_weights* blocks, int num_layers, const float* pair_feat, float* cache,
                          int N, int L, int C, abopt_stream stream);

/* Range guard (ABI 41).  The dense layers of abopt_eps_net_forward (node projections, out_transform + MLP, heads, mixer) multiply fp32 operands as two fp16
 * terms when the packed weights (w_node_frag, w_out_frag / w_out_terms, w_mlp_frag, w_heads_frag, w_mix_frag) are given: an activation beyond 65504 -- which
 * the fp32 reference handles -- becomes inf there and reaches the outputs as inf / NaN.  Every abopt_eps_net_forward raises a device flag when a head output
 * of any row is not finite; abopt_nonfinite_flag synchronises `stream`, returns the flag (1 / 0; -1 on error) and clears it if `reset`.  A caller that
 * sees 1 repeats the work with the packed-weight pointers set to NULL: the same layers then run as fp32 GEMMs with fp32's range (ab_opt_amd/dpm.py does,
 * once per sample() / optimize() call; tests/test_hip_parity.py::test_fp16_range_guard_falls_back_to_fp32_layers). */
int abopt_nonfinite_flag(int reset, abopt_stream stream);

/* Per-call pair terms (ABI 41): pair_feat re-laid as the fp16 operands of the pair aggregation sum_j alpha[i,j,h] z[i,j,:] (ga.py:114-118), built once per
 * FullDPM.sample / optimize call next to the bias cache (same constancy argument).  Every value becomes two fp16 terms h = fp16(S_i z), l = fp16(S_i z - h)
 * with one power of two S_ic per (query row, channel) (max_j |z[n,i,j,c]| S_ic in [2^13, 2^14)): 22 significant bits for every value within 2^-17 of the
 * largest of its (row, channel) column over the keys, the same 4 bytes per value; terms are packed along the contraction (key) index in the operand order of
 * v_mfma_f32_16x16x32_f16, the factors 2^-14 / S_ic follow them (the consuming kernel multiplies its probabilities by 2^14 before their split).
 * With the terms the 32-row block kernels run that aggregation on the fp16 matrix instructions (products exact, fp32 accumulation; error against the fp32
 * statement: that of fp32 accumulation itself, tests/test_hip_parity.py::test_pair_aggregation_on_fp16_terms_vs_fp64); without them (NULL) on the fp32 ones.
 * terms: abopt_pair_terms_bytes(N, L) bytes (< 4 GB), caller-owned; N = number of DISTINCT pair_feat entries (as for the bias cache); L <= 2048.  Opaque like the cache and
 * laid out over the whole batch ([chunk of 16 keys][row of the batch][4 KB]): not sliceable per sample. */
size_t abopt_pair_terms_bytes(int N, int L);
int abopt_pair_terms(const float* pair_feat, float* terms, int N, int L, int C, abopt_stream stream);
/* 1 if abopt_eps_net_forward(N samples, L residues, a bias cache, pair_feat_shared) would launch the kernels that read the terms on the current device
 * (the 32-row block kernels: shapes that fill the chip in whole rounds), 0 otherwise -- a caller need not build terms nobody reads. */
int abopt_pair_terms_used(int N, int L, int pair_feat_shared);

/* EpsilonNet.forward (dpm_full.py:70-112).  beta [N].  Outputs: v_next [N,L,3], R_next [N,L,3,3],
 * eps_pos [N,L,3], c_denoised [N,L,20], prmsd_logits [N,num_bins] (NULL when the head is absent). */
int abopt_eps_net_forward(const abopt_eps_weights* w, const float* v_t, const float* p_t, const int64_t* s_t,
                          const float* res_feat, const float* pair_feat, const float* beta,
                          const uint8_t* mask_generate, const uint8_t* mask_res,
                          float* v_next, float* R_next, float* eps_pos, float* c_denoised, float* prmsd_logits,
                          int N, int L, int F, int C, int grad_mode,
                          const float* pair_bias_cache /* NULL: compute the pair bias inside the step */,
                          int pair_feat_shared /* 0: pair_feat is [N,L,L,C].  1: pair_feat is [1,L,L,C] (and the cache was built with N = 1) and is
                                                  shared by all N samples -- the replicated-complex batches of the reference's runners,
                                                  D/tools/runner/design_for_pdb.py:141-147.  g > 1 (g divides N): pair_feat is [N/g,L,L,C] and
                                                  samples g c .. g c + g - 1 share entry c -- a test set of complexes x g samples in ONE launch
                                                  (D/tools/runner/design_for_testset.py:556-589; BASELINE config 4) */,
                          const float* pair_terms /* optional (needs pair_bias_cache): abopt_pair_terms of the same pair_feat */,
                          void* ws, size_t ws_bytes, abopt_stream stream);

/* ---- Per-step transitions: D/modules/diffusion/transition.py:42-50,80-101 (position), :146-160
 * (rotation), :202-245 (amino acid), D/modules/common/so3.py:111-146 (IGSO(3) draw),
 * dpm_full.py:284-300 (loop body after eps_net), :380-399 (perplexity), prmsd.py:31-47.
 * Schedule scalars of step t (host reads them from the var_sched buffers): */
typedef struct {
    int   t;                 /* current step, T..1 */
    float alpha_clamped;     /* max(alphas[t], alphas[T-1])          transition.py:84-86 */
    float alpha_bar;         /* alpha_bars[t] */
    float sigma;             /* sigmas[t] */
    float sqrt_recip_abar;   /* sqrt_recip_alphas_cumprod[t]          transition.py:45 */
    float sqrt_recipm1_abar; /* sqrt_recipm1_alphas_cumprod[t]        transition.py:46 */
    float igso3_std;         /* angular_distrib_inv.stddevs[t] */
    int   igso3_gaussian;    /* angular_distrib_inv.approx_flag[t] */
    float position_scale;    /* FullDPM.position_scale (10.0) */
    float position_mean[3];
    int   pred_x0;           /* 1: eps_net output is x0 (AbDock obj=pred_x0), 0: it is the noise */
    int   sample_structure;  /* dpm_full.py:294-295 */
    int   sample_sequence;   /* dpm_full.py:296-297 */
    float dist_min, dist_max;/* prmsd bounds (prmsd.py:41) */
    int   ppl_masked;        /* 1: perplexity averaged over generated residues (sample, dpm_full.py:293); 0: over all L (optimize, :358) */
} abopt_step_params;

/* Explicit draws for teacher-forced replay, reference draw order; all NULL => the
 * device Philox stream (seed, offset) is used instead. */
typedef struct {
    const float*   axis;    /* [N,L,3]  randn  -> rotation axis before normalisation  (so3.py:143) */
    const int64_t* bin;     /* [N,L]    multinomial bin of the IGSO(3) histogram      (so3.py:122) */
    const float*   ubin;    /* [N,L]    rand   in-bin position                        (so3.py:125) */
    const float*   gauss;   /* [N,L]    randn  Gaussian branch                        (so3.py:130) */
    const float*   z;       /* [N,L,3]  randn  position noise                         (transition.py:94-98) */
    const int64_t* s_next;  /* [N,L]    the multinomial sample itself                 (transition.py:176-177) */
} abopt_step_noise;

/* One loop iteration after eps_net: state (v_t, p_t in Angstrom, s_t) -> (v_next, p_next in Angstrom, s_next),
 * plus per-sample prmsd [N] and perplexity [N] (either may be NULL).
 * igso3_X / igso3_cdf: row t of the inverse-process histogram, [bins] bin starts and [bins-1] normalised CDF
 * (cdf only used without injected noise). post_out [N,L,20] optional (posterior, for tests).  p_next_norm [N,L,3] optional:
 * (p_next - position_mean) / position_scale, the normalised positions the next step's network call takes (dpm_full.py:276). */
int abopt_denoise_step(const abopt_step_params* sp, const abopt_step_noise* noise,
                       uint64_t seed, uint64_t offset,
                       const float* v_t, const float* p_t, const int64_t* s_t,
                       const float* v_net, const float* p_net, const float* c_net, const float* prmsd_logits,
                       const uint8_t* mask_generate,
                       const float* igso3_X, const float* igso3_cdf, int igso3_bins, int num_bins,
                       float* v_next, float* p_next, int64_t* s_next, float* prmsd, float* perplexity,
                       float* post_out, float* p_next_norm,
                       const uint64_t* seed_offset_dev /* optional DEVICE pointer to {seed, offset}: read by the kernel instead of the
                                                          two by-value arguments, so a captured hipGraph of the loop can be replayed
                                                          with a fresh stream position (the values are read at execution time) */,
                       int N, int L, abopt_stream stream);

/* Initial state of FullDPM.sample (dpm_full.py:255-269): q4 [N,L,4], pn [N,L,3], sr [N,L] are the
 * reference's three draws (NULL => Philox).  p in/out in Angstrom.  position_mean is a HOST pointer to 3 floats. */
int abopt_sample_init(const float* v, const float* p, const int64_t* s, const uint8_t* mask_generate,
                      const float* q4, const float* pn, const int64_t* sr, uint64_t seed, uint64_t offset,
                      float position_scale, const float* position_mean, int sample_structure, int sample_sequence,
                      float* v_init, float* p_init, int64_t* s_init, int N, int L, abopt_stream stream);

/* ---- Forward noising: RotationTransition.add_noise (D/modules/diffusion/transition.py:120-144), PositionTransition.add_noise
 * (:62-78), AminoacidCategoricalTransition.add_noise (:179-200); used by FullDPM.optimize (dpm_full.py:320-339) and by the
 * training loss (dpm_full.py:162-178).  t [N] is per sample; alpha_bars/fwd_* are the schedule buffers ([T+1], [T+1,bins]);
 * fwd_cdf [T+1,bins-1] only for the device-RNG path.  p_0 / p_noisy in Angstrom.  noise: reference draw order
 * randn(N,L,3) axis, multinomial bin, rand ubin, randn gauss | randn(N,L,3) pos | multinomial s_noisy; all NULL => Philox.
 * With noise_structure = 0 (train_structure / sample_structure = False: dpm_full.py:169-173 draws nothing for the structure)
 * s_noisy alone may be injected.
 * c_noisy (optional, [N,L,20]): the categorical c_t the sequence sample is drawn from (transition.py:196-198), whether or not
 * the sample itself is injected; with noise_sequence = 0 it is onehot(s_0), the reference's c_0 (all-zero rows for s_0 outside 0..19). */
typedef struct {
    const float* axis; const int64_t* bin; const float* ubin; const float* gauss;   /* rotation (so3.py:141-146) */
    const float* pos;                                                               /* e_rand (transition.py:75) */
    const int64_t* s_noisy;                                                         /* the categorical sample itself */
} abopt_addnoise_noise;

int abopt_add_noise(const int64_t* t, const float* alpha_bars, const float* fwd_stddevs, const uint8_t* fwd_approx,
                    const float* fwd_X, const float* fwd_cdf, int bins, int num_sched,
                    const abopt_addnoise_noise* noise, uint64_t seed, uint64_t offset,
                    const float* v_0, const float* p_0, const int64_t* s_0, const uint8_t* mask_generate,
                    float position_scale, const float* position_mean, int noise_structure, int noise_sequence, int grad_mode,
                    float* v_noisy, float* p_noisy, int64_t* s_noisy, float* eps_p, float* c_noisy,
                    const uint64_t* seed_offset_dev /* optional device {seed, offset}, as in abopt_denoise_step */,
                    int N, int L, abopt_stream stream);

/* ---- DockQ scoring of docked candidates: D/tools/runner/design_for_pdb.py:316-321 calls calc_DockQ(model, native, use_CA_only=True)
 * (AbDock/DockQ/DockQ.py:98-385) per candidate, which runs the `fnat` program twice (DockQ/src/fnat.c:100-252: residue contacts over
 * all heavy atoms, 5 A for Fnat, 10 A for the interface) and superimposes CA atoms twice (interface -> iRMS; receptor -> LRMS).
 * Structures are tensors with the batch's residue indexing: pos [L,A,3] Angstrom, mask [L,A], group [L] (0 = not in the file,
 * 1 / 2 = the two chains); atom slot 1 = CA.  model_pos [S,L,A,3]; model_mask [S,L,A], or [L,A] with model_mask_shared = 1.
 * out [S,4] = (fnat, irms, Lrms, DockQ).  ws: abopt_dockq_workspace_bytes(L).
 * A candidate whose interface, receptor or ligand has NO CA atom present in both model and native has no superposition: its irms /
 * Lrms (and DockQ) come back as -1 (the reference asserts on such inputs); with 1 or 2 common atoms the fit is degenerate but defined. */
size_t abopt_dockq_workspace_bytes(int L);
int abopt_dockq_lite(const float* model_pos, const uint8_t* model_mask, int model_mask_shared, const float* native_pos,
                     const uint8_t* native_mask, const int32_t* group, int S, int L, int A, float* out,
                     void* ws, size_t ws_bytes, abopt_stream stream);

/* ---- Batched-sampling reduction: D/tools/runner/design_for_testset.py:556-589 (calc_per_rmsd +
 * rank_commoness score).  structs [B,n,3] -> score [B] = mean_{b'} RMSD(b,b') * B/(B-1). */
/* ---- Training side of the IPA core (FullDPM.forward, D/modules/diffusion/dpm_full.py:156-234; the autograd of
 * GABlock.forward, ga.py:81-147).  The host keeps the dense projections, LayerNorm/MLP and losses in its autograd graph
 * (library GEMMs) and calls these two for the part that streams pair_feat:
 *   forward : proj_local [N,L,2016] = x . [Wq|Wk|Wv|Wqp|Wkp|Wvp]^T (points still in the residue frames)
 *             -> feat [N,L,1824] (the input of out_transform, ga.py:174) and alpha (ga.py:166) HEAD-MAJOR [N,12,L,L]
 *   backward: g [N,12,L,L] = d loss / d logits (after the sqrt(1/3) scale) and dpair_feat [N,L,L,C] of this block, from
 *             alpha, dalpha_node[n,h,i,j] = <dfeat_node_ih, v_jh> + <dagg_pts_ih, v_pts_jh>, delta[n,i,h] = sum_j alpha dalpha,
 *             dfeat (its first 12*C columns are d feat_p2n, row stride ld_dfeat) and proj_pair_bias.weight. */
size_t abopt_ipa_train_workspace_bytes(int N, int L);
int abopt_ipa_core_train_forward(const float* proj_local, const float* R, const float* t, const float* pair_feat, const uint8_t* mask,
                                 const float* w_pair_bias, const float* spatial_coef,
                                 const float* pair_bias_cache /* optional: this block's slice of abopt_pair_bias_cache built from the same pair_feat and weights */,
                                 float* feat, float* alpha, int N, int L, int C, void* ws, size_t ws_bytes, abopt_stream stream);
/* prologue of the backward: the points epilogue (ga.py:133-139) differentiated, d feat_node re-laid out head-major next to
 * it (dout_cat [N,12,L,32+24]) and delta [N,L,12] = sum_j alpha dalpha (see above). */
int abopt_ipa_points_backward(const float* dfeat, int ld_dfeat, const float* feat, const float* R, const float* t,
                              float* dout_cat, float* delta, int N, int L, abopt_stream stream);
/* head-major operands of the backward's batched GEMMs: Aq/Ak [N,12,L,57] = [q|k (32) | points in the global frame (24) | 1],
 * Av [N,12,L,56] = [v | v points]; and the final assembly of d proj_local [N,L,2016] (+ e [N,L,12], whose sum over (N, L) is
 * d loss / d spatial_coef: the chain through -softplus(.) sqrt(2/(9P))/2 is applied in the kernel) from P1 = g Ak, P2 = g^T Aq, P3 = alpha^T dout_cat. */
int abopt_ipa_backward_operands(const float* proj_local, const float* R, const float* t, float* Aq, float* Ak, float* Av,
                                int N, int L, abopt_stream stream);
int abopt_ipa_backward_assemble(const float* P1, const float* P2, const float* P3, const float* Aq, const float* Ak, const float* R,
                                const float* spatial_coef, float* dproj, float* e, int N, int L, abopt_stream stream);
int abopt_ipa_pair_backward(const float* pair_feat, const float* alpha, const float* dalpha_node, const float* delta,
                            const float* dfeat, int ld_dfeat, const float* w_pair_bias, float* g, float* dpair_feat /* may be NULL: see abopt_ipa_dz_assemble */,
                            float* dw_pair_bias_rows /* [N*L, 12*C]: per-query-row partials of d proj_pair_bias.weight (sum over rows) */,
                            int dpair_feat_accumulate /* 1: dpair_feat += this block's gradient (the six blocks of the encoder share one
                                                         buffer instead of six 268 MB tensors that autograd adds up); 0: overwrite */,
                            int N, int L, int C, abopt_stream stream);

/* d pair_feat of ALL blocks of an encoder in one pass: dpair_feat[n,i,j,c] = sum over blocks l and heads h of
 * alpha_l[n,h,i,j] dfeat_l[n,i,h*C+c] + g_l[n,h,i,j] w_pair_bias_l[h,c] (the pair-aggregation and proj_pair_bias terms of ga.py:88-90,114-118
 * differentiated), from what abopt_ipa_pair_backward(..., dpair_feat = NULL, ...) of every block left behind.  No z read and ONE write of
 * d pair_feat instead of a read-modify-write per block.  alpha / g / dfeat / w_pair_bias: HOST arrays of num_blocks (<= 6) device pointers. */
int abopt_ipa_dz_assemble(int num_blocks, const float* const* alpha, const float* const* g, const float* const* dfeat, int ld_dfeat,
                          const float* const* w_pair_bias, float* dpair_feat, int N, int L, int C, abopt_stream stream);

/* ---- encode(): D/models/diffab.py:39-83.  ResidueEmbedding.forward (D/modules/encoders/residue.py:26-92; the AbDesign
 * variant adds hotspot_embed, A/modules/encoders/residue.py:19-21), PairEmbedding.forward (D/modules/encoders/pair.py:37-101),
 * construct_3d_basis (D/modules/common/geometry.py:47-69).  Feature widths are fixed: res_feat_dim 128, pair_feat_dim 64,
 * max_aa_types 22, max_relpos 32. */
typedef struct {
    int N, L;
    int atoms_in;                    /* atoms per residue in pos_atoms / mask_atoms (15) */
    int atoms;                       /* atoms the embeddings use: 15 ('full') or 5 ('backbone+CB'), D/models/diffab.py:13-16 */
    const int64_t* aa;               /* [N,L] */
    const int64_t* res_nb;           /* [N,L] */
    const int64_t* chain_nb;         /* [N,L] */
    const int64_t* fragment_type;    /* [N,L]  (residue embedding only) */
    const int64_t* hotspot;          /* [N,L] or NULL -> zeros (AbDesign residue embedding only) */
    const float*   pos_atoms;        /* [N,L,atoms_in,3] Angstrom */
    const uint8_t* mask_atoms;       /* [N,L,atoms_in] */
    const uint8_t* structure_mask;   /* [N,L] or NULL (diffab.py:47-55) */
    const uint8_t* sequence_mask;    /* [N,L] or NULL */
} abopt_encode_inputs;

typedef struct {
    const float* aatype_embed;       /* [22, 128] */
    const float* type_embed;         /* [10, 128] */
    const float* hotspot_embed;      /* [10, 128] or NULL (AbDock) */
    const float* freq_bands;         /* [6] dihed_embed.freq_bands */
    const float* w0; const float* b0;/* mlp.0 [256, in], in = 128 + 22*atoms*3 + 39 + 128 (+128 with hotspot) */
    const float* w1; const float* b1;/* mlp.2 [128, 256] */
    const float* w2; const float* b2;/* mlp.4 [128, 128] */
    const float* w3; const float* b3;/* mlp.6 [128, 128] */
} abopt_residue_embed_weights;

typedef struct {
    const float* aa_pair_embed;      /* [484, 64] */
    const float* relpos_embed;       /* [65, 64] */
    const float* aapair_to_distcoef; /* [484, atoms*atoms] */
    const float* freq_bands;         /* [6] dihedral_embed.freq_bands */
    const float* wd0; const float* bd0;   /* distance_embed.0 [64, atoms*atoms] */
    const float* wd1; const float* bd1;   /* distance_embed.2 [64, 64] */
    const float* wo0; const float* bo0;   /* out_mlp.0 [64, 218] */
    const float* wo1; const float* bo1;   /* out_mlp.2 [64, 64] */
    const float* wo2; const float* bo2;   /* out_mlp.4 [64, 64] */
} abopt_pair_embed_weights;

size_t abopt_residue_embed_workspace_bytes(int N, int L, int atoms, int hotspot);
/* -> res_feat [N,L,128]; R [N,L,3,3] = construct_3d_basis(CA, C, N); p [N,L,3] = CA (diffab.py:76-83) */
int abopt_residue_embed_forward(const abopt_encode_inputs* in, const abopt_residue_embed_weights* w, float* res_feat, float* R, float* p,
                                void* ws, size_t ws_bytes, abopt_stream stream);
/* Training path: the input features of the residue MLP on their own (residue.py:33-88: aa embedding | per-type local coordinates | dihedral
 * encoding | fragment-type embedding [| hotspot embedding]) -> features [N*L, ld], ld = in_dim rounded up to a multiple of 4
 * (in_dim = 128 + 22*atoms*3 + 39 + 128 [+ 128]); only the embedding tables and freq_bands of `w` are read.  R, p as in the fused forward. */
size_t abopt_residue_features_workspace_bytes(int N, int L);
int abopt_residue_features(const abopt_encode_inputs* in, const abopt_residue_embed_weights* w, float* features, float* R, float* p,
                           void* ws, size_t ws_bytes, abopt_stream stream);
size_t abopt_pair_embed_workspace_bytes(int N, int L, int atoms);
/* -> pair_feat [N,L,L,64].  activations: NULL for inference; for training a [N,L,L,ABOPT_PAIR_ACT] buffer that receives, per
 * pair, relu(distance_embed.0) (64) | f_dist = relu(distance_embed.2) x structure mask (64) | f_dih (26, padded to 32) |
 * relu(out_mlp.0) (64) | relu(out_mlp.2) (64): what the backward of the five linears needs (pair.py:74-99). */
enum { ABOPT_PAIR_ACT = 288 };
/* gauss / dgauss (training): [N,L,L,atoms,16] -- the Gaussian atom-pair features g (pair.py:62-73, atom b of residue j padded to 16)
 * and T = dg / d softplus(coef) = -d^2 g.  dgauss may be NULL with gauss given: the backward then recomputes T from the atoms (1 GB
 * less to write and to read back at N = 16, L = 256). */
int abopt_pair_embed_forward(const abopt_encode_inputs* in, const abopt_pair_embed_weights* w, float* pair_feat, float* activations,
                             float* gauss, float* dgauss, void* ws, size_t ws_bytes, abopt_stream stream);
/* Backward chain of the five linears for the training path: from dpair_feat [N,L,L,64] and the saved activations writes, per pair,
 * dys [N,L,L,ABOPT_PAIR_DY] = d loss / d pre-activation of out_mlp.4 | out_mlp.2 | out_mlp.0 | distance_embed.2 |
 * distance_embed.0 (64 each), and dsoftplus [N,L,L,atoms,16] = d loss / d softplus(aapair_to_distcoef) per atom pair.  The
 * weight gradients are tall GEMMs of dys against the activations (host side).  dgauss NULL: T is recomputed in the kernel with the
 * forward's arithmetic (same bits).  dys_colsum (optional, [ABOPT_PAIR_DY]): the column sums of dys = the five bias gradients, formed from
 * per-wave partial sums inside the kernel instead of a second pass over the 1.3 GB of dys. */
enum { ABOPT_PAIR_DY = 320 };
size_t abopt_pair_embed_backward_workspace_bytes(int N, int L, int atoms);
int abopt_pair_embed_backward(const abopt_encode_inputs* in, const abopt_pair_embed_weights* w, const float* dpair_feat,
                              const float* activations, const float* dgauss, float* dys, float* dsoftplus, float* dys_colsum,
                              void* ws, size_t ws_bytes, abopt_stream stream);

/* ---- Training path: the geometric epilogue of the heads on its own, D/modules/diffusion/dpm_full.py:95-101 (A: 86-92), and its backward.
 *   eps_pos = gen ? R eps_crd : 0;   R_next = R U(eps_rot) with U the rotation of the quaternion 1 + b i + c j + d k (geometry.py:215-233);
 *   v_next = gen ? log(R_next) : v_t  (optional: v_t / v_next may both be NULL -- the training losses do not use it).
 * eps_crd / eps_rot [rows,3] are the outputs of eps_crd_net / eps_rot_net; the backward takes d R_next [rows,3,3] and d eps_pos [rows,3]
 * (either may be NULL = zero) and writes d eps_crd, d eps_rot [rows,3].  In the reference these are ~45 elementwise ATen kernels forward
 * and ~90 backward per step. */
/* The three per-residue losses of FullDPM.forward and their gradients in one pass (D/modules/diffusion/dpm_full.py:199-231, A: 156-190):
 *   rot = sum_k 1 - cos(R_pred[:, k], R_0[:, k])  (rotation_matrix_cosine_loss, dpm_full.py:15-32);  pos = |p_pred - p_target|^2;
 *   seq = KL(posterior(s_t, s_0) || posterior(s_t, c_denoised))  (transition.py:217-229),  each summed over the generated residues.
 * block_sums [ceil(N L / 256)][3]: per-workgroup sums of (rot, pos, seq), to be added in order and divided by sum(mask_generate) + 1e-8;
 * dR_pred [N,L,3,3], dp_pred [N,L,3], dc_denoised [N,L,20]: d(sum)/d(input) per residue, zero outside mask_generate. */
int abopt_dpm_losses(const float* R_pred, const float* R_0, const float* p_pred, const float* p_target, const float* c_denoised, const int64_t* s_t,
                     const int64_t* s_0, const float* alpha_bar_t, const uint8_t* mask_generate, int N, int L, float* block_sums, float* dR_pred,
                     float* dp_pred, float* dc_denoised, abopt_stream stream);
/* The two losses only the AbDock flavour has, with their gradients, one workgroup per sample (D/modules/diffusion/dpm_full.py:180-198,
 * D/modules/common/prmsd.py:49-70 pRMSDCa, dpm_full.py:369-378 calc_dist_loss):
 *   prmsd = sum_n CE(prmsd_logits[n], bin of rmsd_n) m0_n / (sum_n m0_n + 1e-10), rmsd_n over the generated residues of position_scale (pred_p0 - p0),
 *           m0_n = mask_generate[n, 0]; pred_p0 = p_pred (pred_x0 = 1) or gen ? coef_a[n] p0 - coef_b[n] p_pred : p0 (pred_x0 = 0, transition.py:52-60);
 *   dist  = mean smooth_l1(cdist(p_pred) - cdist(p0)) over the pairs mask_generate_i & mask_res_i & mask_res_j (pred_x0 = 1 only).
 * sample_parts [N,4] = {CE_n, m0_n, sum of the sample's smooth-l1 terms, their count}; dprmsd_logits [N,num_bins] = softmax - onehot (scale by
 * m0_n / (sum m0 + 1e-10)); dp_pred [N,L,3] = d(sum of smooth-l1 terms)/d p_pred (scale by 1 / total count; zeros when pred_x0 = 0). */
int abopt_abdock_losses(const float* prmsd_logits, const float* p_pred, const float* p0_norm, const float* coef_a, const float* coef_b,
                        const uint8_t* mask_generate, const uint8_t* mask_res, const float* bin_offsets, int num_bins, int N, int L, float position_scale,
                        int pred_x0, float* sample_parts, float* dprmsd_logits, float* dp_pred, abopt_stream stream);
/* LayerNorm with the reference's definition (D/modules/common/layers.py:146-155: biased variance, sqrt(var + eps)) over rows of cols <= 256 values --
 * the prmsd head's layer_norm under autograd (D/modules/common/nn.py:179-188).  forward keeps xhat [rows,cols] and rstd [rows] (both or neither);
 * backward writes dx and dy_xhat = dy * xhat (d gamma = column sums of dy_xhat, d beta = column sums of dy: abopt_colsum). */
int abopt_layer_norm_forward(const float* x, const float* gamma, const float* beta, int cols, float eps, int64_t rows, float* y, float* xhat, float* rstd,
                             abopt_stream stream);
int abopt_layer_norm_backward(const float* dy, const float* xhat, const float* rstd, const float* gamma, int cols, int64_t rows, float* dx, float* dy_xhat,
                              abopt_stream stream);
int abopt_heads_epilogue_forward(const float* R, const float* v_t, const float* eps_crd, const float* eps_rot, const uint8_t* mask_generate,
                                 float* v_next, float* R_next, float* eps_pos, int64_t rows, int grad_mode, abopt_stream stream);
int abopt_heads_epilogue_backward(const float* R, const float* eps_rot, const uint8_t* mask_generate, const float* dR_next, const float* deps_pos,
                                  float* deps_crd, float* deps_rot, int64_t rows, abopt_stream stream);

/* ---- reconstruct_backbone_partially: D/modules/common/geometry.py:404-480 (called on every saved frame right after the
 * sampler, D/tools/runner/design_for_pdb.py:166-223).  pos_ctx/pos_new [N,L,A,3], mask_atoms/mask_new [N,L,A], R_new [N,L,3,3],
 * t_new [N,L,3] in Angstrom, mask_recons [N,L]; bb_table [21,3,3] / o_table [21,3] are the ideal local backbone coordinates
 * per residue type (D/utils/protein/constants.py:310-320; shipped as ab_opt_amd/data/backbone_ideal.npz). */
int abopt_reconstruct_backbone_partially(const float* pos_ctx, const float* R_new, const float* t_new, const int64_t* aa,
                                         const int64_t* chain_nb, const int64_t* res_nb, const uint8_t* mask_atoms,
                                         const uint8_t* mask_recons, const float* bb_table, const float* o_table,
                                         float* pos_new, uint8_t* mask_new, int N, int L, int A, abopt_stream stream);

/* ---- Training path: general strided-batched fp32 GEMM, C[b] = alpha * A[b] . B[b]^T (exact fp32 FMA chains on the matrix cores).
 *   A(m,k) = a_transposed ? A[k*lda + m] : A[m*lda + k]     B(n,k) = b_transposed ? B[k*ldb + n] : B[n*ldb + k]     C(m,n) = C[m*ldc + n]
 * It stands where the reference's autograd calls ATen GEMMs in the backward of the denoiser (D/modules/encoders/ga.py:54-66,81-147,
 * 174-177 and D/modules/diffusion/dpm_full.py:39-59 under torch.autograd).  ws (optional): scratch for split-K partial tiles, used when
 * the output has too few tiles to fill the chip and K >= 1024 (weight gradients); partials are summed in a fixed order.
 * bias (optional, [N]) is added to every row and relu != 0 clamps at zero in the product's epilogue: y = relu(x W^T + b), the nn.Linear +
 * nn.ReLU pairs of the heads / mixer / residue MLPs as one launch (no split-K then). */
int abopt_gemm(const float* A, int lda, int64_t stride_a, int a_transposed, const float* B, int ldb, int64_t stride_b, int b_transposed,
               float* C, int ldc, int64_t stride_c, int M, int N, int K, int batch, float alpha, const float* bias, int relu,
               void* ws, size_t ws_bytes, abopt_stream stream);

/* The weight-gradient products of one backward pass as a group: C_p = A_p^T . B_p for count <= 24 tall operand pairs, i.e.
 *   C_p[m*n_p + n] = sum over k < k_p of a_p[k*lda_p + m] * b_p[k*ldb_p + n]      (C_p dense [m_p, n_p]; a_p, b_p read in place)
 * -- d W = d y^T x of every nn.Linear the backward of the denoiser walks (D/modules/encoders/ga.py:54-79, D/modules/diffusion/dpm_full.py:39-65
 * under `loss.backward()`, D/train.py:101-114) -- in ONE launch plus one for the split-K slab sums of all of them, instead of two launches per
 * product.  Exact fp32 FMA chains on the matrix cores; the K split is chosen for the group's tile count; partial slabs are summed in a fixed order
 * (deterministic; the order differs from abopt_gemm's for the same product).  ws: scratch for the slabs, up to
 * sum_p min(1024 / total 64x64 tiles, k_p / 256) * m_p * n_p floats are used (less workspace: fewer slabs, never an error). */
typedef struct { const float* a; const float* b; float* c; int lda, ldb, m, n, k; } abopt_gemm_tn_problem;
int abopt_gemm_tn_grouped(const abopt_gemm_tn_problem* problems, int count, void* ws, size_t ws_bytes, abopt_stream stream);

/* out[c] = sum over rows of x[r*ld + c] (bias gradients and per-row partials of weight gradients on the training path); deterministic:
 * row slices are summed in a fixed order.  ws (optional): slices * cols floats of scratch. */
int abopt_colsum(const float* x, int ld, int64_t rows, int cols, float* out, void* ws, size_t ws_bytes, abopt_stream stream);

/* out[b*cols + c] = sum over the rows r with idx[r] == b of x[r*ld + c] (rows with idx outside [0, buckets) are skipped; buckets <= 96):
 * the gradient of an embedding table looked up once per row of a tall activation matrix -- the relative-position table of
 * D/modules/encoders/pair.py:46-53 over its N L^2 pair rows -- without the [rows, buckets] one-hot matrix autograd builds.  Deterministic. */
int abopt_bucket_colsum(const float* x, int ld, int64_t rows, int cols, const int32_t* idx, int buckets, float* out, void* ws, size_t ws_bytes,
                        abopt_stream stream);

/* The same sum per SEGMENT of rows_per_segment consecutive rows: out[(s*buckets + b)*cols + c] = sum over the rows j of segment s with bucket b of
 * x[(s*rows_per_segment + j)*ld + c], where the bucket of row j of segment s is idx[(s / idx_div)*rows_per_segment + j] (idx_div consecutive
 * segments share one index row).  With segment = (sample, query residue i), rows = key residues j and idx = the residue types of the sample this
 * is the first half of the gradient of the amino-acid-pair tables of D/modules/encoders/pair.py:46-53,66 (aa_pair_embed, aapair_to_distcoef),
 * summed over j by type(j); abopt_bucket_colsum over the (sample, i) rows by type(i) finishes it -- in place of the two one-hot batched products
 * (and a 268 MB contiguous copy of the strided operand) autograd would run.  Deterministic.  buckets <= 96, segments <= 65535. */
int abopt_segment_bucket_colsum(const float* x, int ld, int segments, int rows_per_segment, int cols, const int32_t* idx, int idx_div, int buckets,
                                float* out, abopt_stream stream);

/* ---- Training path: gradient clipping + Adam for a whole parameter list in a handful of launches.  Replaces, with the same arithmetic,
 *   orig_grad_norm = clip_grad_norm_(model.parameters(), config.train.max_grad_norm); optimizer.step()
 * of A/train.py:116-117 / D/train.py:112-113 with torch.optim.Adam(lr, betas, weight_decay) (A/diffab/utils/train.py:28-36; eps as given,
 * no amsgrad, no maximize), which torch runs as ~1000 small launches for the model's 207 tensors.
 *   params / grads / exp_avg / exp_avg_sq / numel : HOST arrays of `count` device pointers / element counts (fp32, contiguous)
 *   step  : device int64, the number of steps taken so far; incremented by the call (bias corrections use the incremented value)
 *   max_grad_norm > 0 : gradients are scaled by min(1, max_grad_norm / (||g||_2 + 1e-6)) inside the update (the gradient tensors themselves
 *                       are NOT rewritten) and the unclipped norm is stored in grad_norm_out[0] (device, may be NULL); <= 0: no clipping
 *   ws    : device scratch of abopt_adam_ws_floats(count, numel) floats.
 * Hyper-parameters are doubles, as Python holds them: 1 - beta and the bias corrections are formed in double and rounded once, as torch does.
 *   hyper_dev (optional, DEVICE, 6 doubles {lr, beta1, beta2, eps, weight_decay, max_grad_norm}): read by the kernels at execution time
 *   INSTEAD of the by-value arguments, so a captured hipGraph of the step follows a learning-rate scheduler (the reference's loops use
 *   ReduceLROnPlateau / MultiStepLR, A/diffab/utils/train.py:39-60): the host rewrites the buffer before a replay.  Whether clipping
 *   happens at all is still decided by the by-value max_grad_norm > 0 (it changes the launch list).
 * Deterministic (block partials of the norm are summed in a fixed order); capturable into a hipGraph (pointers travel as kernel arguments). */
size_t abopt_adam_ws_floats(int count, const int64_t* numel);
int abopt_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const int64_t* numel, double lr, double beta1, double beta2, double eps, double weight_decay, double max_grad_norm,
                    int64_t* step, float* ws, size_t ws_floats, float* grad_norm_out, const double* hyper_dev, abopt_stream stream);

int abopt_commonness_score(const float* structs, float* score, int B, int n, abopt_stream stream);

/* ---- Measurement hook (bench.py's roofline leg).  When enabled, every launch of the IPA-core kernel is bracketed
 * by hipEvents on the stream it is launched on; abopt_prof_collect synchronises those events and returns the number
 * of launches and their summed duration since the last enable.  Process-global, off by default, not for concurrent
 * use from several host threads (the only global state in the library). */
int abopt_prof_enable(int on);   /* 1: on (forgets earlier pairs), 0: off (forgets), 2: off but keep the recorded pairs;
                                    3: launch spans on (below), 4: stop handing out span slots */
/* Launch spans: the dominant kernel timed INSIDE a replayed hipGraph, where host-recorded event pairs cannot be placed.  While mode 3 is on
 * (e.g. during the capture), every 32-row IPA launch (core, or fused core + tail) is given a slot number; each of its workgroups folds the
 * 100 MHz wall clock of its first / last instruction into {min, max} of the slot.  abopt_prof_spans_reset (stream-ordered) clears the slots --
 * call it before the replay to be read --, abopt_prof_spans synchronises the device and returns the number of launches that ran since and
 * the sum of their (max end - min start) in milliseconds. */
int abopt_prof_spans_reset(abopt_stream stream);
int abopt_prof_spans(int* launches, double* total_ms);
int abopt_prof_collect(int* launches, double* total_ms);
/* Clock probe: shader cycles and 100 MHz wall-clock ticks that wave 0 of workgroup 0 of the most recent 32-row IPA launch (core or fused
 * core + tail) spent from its first to its last instruction: cycles / (10 ns * ticks) = the clock the chip sustained under that kernel.
 * Synchronises the device.  Zeros when no such launch has run. */
int abopt_prof_clock(long long* cycles, long long* wall_ticks_100mhz);
/* The same sum without forgetting the event pairs: records captured into a hipGraph are re-recorded by every replay. */
int abopt_prof_peek(int* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* ABOPT_H */

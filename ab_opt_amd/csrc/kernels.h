// Internal launch interface between the translation units of libabopt_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/abopt.h"

namespace abopt {

// gemm.hip ------------------------------------------------------------------------------------
// ksplit > 1: split-K, slab z of the partial products at Y + z*slab_stride (no bias / activation; the consumer reduces).
int launch_linear(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy,
                  int M, int N, int K, bool relu, hipStream_t st, int ksplit = 1, int64_t slab_stride = 0);

// C[b] = alpha A[b] . B[b]^T, operands k-contiguous or k-strided (training path); ws: split-K scratch (optional)
int launch_gemm_batched(const float* A, int lda, int64_t sa, int a_t, const float* B, int ldb, int64_t sb, int b_t, float* C, int ldc, int64_t sc,
                        int M, int N, int K, int batch, float alpha, float* ws, size_t ws_floats, hipStream_t st, const float* bias = nullptr, int relu = 0);

int launch_colsum(const float* x, int ld, int64_t rows, int cols, float* out, float* ws, size_t ws_floats, hipStream_t st);

// ipa.hip / ipa_core.hip ----------------------------------------------------------------------
// proj [N*L, NP = 2048]: q|k|v|qp|kp|vp (2016 used), points still in the residue frames (the six bias-free projections of ga.py:54-66).
// launch_ipa_frags moves the point sets to the global frame and lays the IPA core's operands out in MFMA fragment order:
// qfrag (queries, pre-scaled, spatial_coef folded in) and kvfrag (keys / values); the core never reads proj.
size_t ipa_kvfrag_floats(int N, int L);
size_t ipa_qfrag_floats(int N, int L);
size_t pair_bias_layer_floats(int N, int L);
// ldp: row stride of proj in floats (NP for the library's own buffer; 2016 reads the projections' output in place: rows stay 16-byte aligned)
int launch_ipa_frags(const float* proj, const float* R, const float* t, const float* spatial_coef, float* qfrag, float* kvfrag, int N, int L, hipStream_t st, int ldp = 2048);
int launch_ipa_core(const float* qfrag, const float* kvfrag, const float* z, const uint8_t* mask, const float* R, const float* t,
                    const float* w_pair_bias, float* feat, float* dbg_logits, float* dbg_alpha, const float* pair_bias_cache,
                    int N, int L, hipStream_t st, int z_shared = 0 /* 1: z and the pair-bias cache hold ONE sample that every batch entry shares */,
                    float* split_ws = nullptr, size_t split_ws_floats = 0 /* scratch of the key-split form (small batches), ipa_split_ws_floats(N, L) */,
                    const float* pair_terms = nullptr /* abopt_pair_terms blob of the same pair_feat: the 32-row kernels then aggregate on the fp16 matrix instructions */);

// ipa_core.hip: core + tail of a block in one launch where the 32-row core applies (sets *fused; otherwise launches nothing)
int launch_ipa_block_fused(const float* qfrag, const float* kvfrag, const float* z, const uint8_t* mask, const float* R, const float* t,
                           const float* pair_bias_cache, int N, int L, hipStream_t st, int z_shared, const float* wot /* W_out as bf16 terms */, const float* wmf, const float* x,
                           const float* ubias, const float* g1, const float* be1, const float* b0, const float* b1, const float* b2, const float* g2,
                           const float* be2, float* out, int* fused, const float* pair_terms = nullptr, float* xt_out = nullptr);

// node_frags.hip: x [N*L,128] -> qfrag / kvfrag directly (projection GEMM + frame transform + fragment layout in one kernel)
size_t node_wfrag_floats();
int launch_node_frags(const float* x, const float* wfrag, const float* R, const float* t, const float* spatial_coef, float* qfrag, float* kvfrag,
                      int N, int L, hipStream_t st,
                      int qk_terms = 0 /* 1: the q / k channel slots as two fp16 terms each (high terms in slot 0, low terms in slot 1): what ipa_core32_kernel<*, true> reads */,
                      const float* x_terms = nullptr /* optional: x as two fp16 terms per value, written by the kernel that produced x (tail / mixer): no split here */);

// rows.hip ------------------------------------------------------------------------------------
int launch_so3_exp(const float* w, float* R, int64_t n, hipStream_t st);
int launch_so3_log(const float* R, float* w, int64_t n, int grad_mode, hipStream_t st);
// mlp.hip: out = LN2(y + MLP(y)), y = LN1(x + mask*(sum of u slabs + ubias))  -- the tail of a GABlock in one launch
int launch_fused_ln_mlp(const float* x, const float* u, int nslab, int64_t slab_stride, const float* ubias, const uint8_t* mask,
                        const float* g1, const float* be1, const float* W0, const float* b0, const float* W1, const float* b1,
                        const float* W2, const float* b2, const float* g2, const float* be2, float* out, int64_t rows, hipStream_t st);
// mlp.hip: the same tail with out_transform fused in; W_out and W_mlp0..2 as bf16 terms in MFMA operand order (abopt.h: w_out_frag, w_mlp_frag)
// dump (optional, training): five [rows,128] slabs for launch_tail_backward
int launch_out_ln_mlp(const float* feat, const float* wof, const float* wmf, const float* x, const float* ubias, const uint8_t* mask,
                      const float* g1, const float* be1, const float* b0, const float* b1, const float* b2, const float* g2, const float* be2,
                      float* out, float* dump, int64_t rows, hipStream_t st, float* xt_out = nullptr /* optional [rows, 128]: the output rows as two fp16 terms for the next block's node_frags */);
// heads.hip: the three denoiser heads (first layers fused, time features as an affine term) -> out3 [rows,32]
size_t heads_wfrag_floats();
size_t mixer_wfrag_floats();
int launch_mixer(const float* res_feat, const int64_t* s_t, const float* wfrag, const float* table, const float* b1, float* x_out, int64_t rows,
                 hipStream_t st, const float* v_t = nullptr, float* R_out = nullptr /* optional: R = exp(v_t) of the same rows, fused */, float* xt_out = nullptr);
// arguments of the heads' geometric epilogue (launch_heads_epilogue) when it runs as the tail of launch_heads_mlp
struct HeadsEpilogue { const float *R = nullptr, *v_t = nullptr; const uint8_t* mask_generate = nullptr; float *v_next = nullptr, *R_next = nullptr, *eps_pos = nullptr, *c_den = nullptr; int grad_mode = 0; unsigned* nonfinite = nullptr; };
// rows.hip: the device word the heads' epilogue raises on a non-finite output (abopt_nonfinite_flag)
unsigned* nonfinite_flag_ptr();
int nonfinite_flag_read(int reset, hipStream_t st, int* flag);
int launch_heads_mlp(const float* xe, const float* beta, const float* wfrag, const float* w1, int ld1, const float* b1, const float* b2c,
                     const float* b2r, const float* b2s, const float* b3c, const float* b3r, const float* b3s, float* out3, int64_t rows, int L,
                     hipStream_t st,
                     const HeadsEpilogue* ep = nullptr /* optional: the geometric epilogue of the same rows in the same launch */);
size_t out_wfrag_floats();
size_t out_wterms_floats();
int launch_out_frag_terms(const float* wof, float* wot, hipStream_t st);
size_t mlp_wfrag_floats();
int launch_pack_tail_weights(const float* w_out, const float* w0, const float* w1, const float* w2, float* wof, float* wmf, float* wmt, hipStream_t st);
int launch_tail_backward(const float* dout, const float* saved, const float* wmt, const uint8_t* mask, const float* g1, const float* g2,
                         float* dpre, float* da1, float* du, float* colpart, int64_t rows, hipStream_t st);
// cat[row] = [res_feat[row] | embed[s_t[row]]], F == 128
int launch_embed_concat(const float* res_feat, const int64_t* s_t, const float* embed, float* cat, int64_t rows, hipStream_t st);
// infeat[row, 0:128] = x, [128:131] = beta, sin beta, cos beta, [131] = 0 ; optional LN'd copy for the prmsd head
int launch_build_infeat(const float* x, const float* beta, float* infeat, const float* ln_gamma, const float* ln_beta,
                        float* infeat_ln, int N, int L, hipStream_t st);
// heads epilogue: dpm_full.py:92-107
int launch_heads_epilogue(const float* R, const float* v_t, const float* eps_crd, const float* eps_rot, const float* seq_logits,
                          int ld3, int ldseq, const uint8_t* mask_generate, float* v_next, float* R_next, float* eps_pos, float* c_den,
                          int64_t rows, int grad_mode, hipStream_t st);
int launch_dpm_losses(const float* R_pred, const float* R_0, const float* p_pred, const float* p_target, const float* c_den, const int64_t* s_t, const int64_t* s_0,
                      const float* abar, const uint8_t* mask_generate, int N, int L, float* part, float* gR, float* gp, float* gc, hipStream_t st);
int launch_abdock_losses(const float* prmsd_logits, const float* p_pred, const float* p0n, const float* coef_a, const float* coef_b, const uint8_t* gen,
                         const uint8_t* mres, const float* offsets, int nb, int N, int L, float scale, int pred_x0, float* part, float* glogit, float* gp,
                         hipStream_t st);
int launch_row_layer_norm(const float* x, const float* gamma, const float* beta, int cols, float eps, int64_t rows, float* y, float* xhat, float* rstd, hipStream_t st);
int launch_row_layer_norm_backward(const float* dy, const float* xhat, const float* rstd, const float* gamma, int cols, int64_t rows, float* dx, float* dyx, hipStream_t st);
int launch_heads_epilogue_backward(const float* R, const float* eps_rot, int ld3, const uint8_t* mask_generate, const float* dR_next, const float* deps_pos,
                                   float* deps_crd, float* deps_rot, int64_t rows, hipStream_t st);
// out[n, b] = mean_l in[n, l, b]
int launch_mean_over_L(const float* in, float* out, int N, int L, int B, hipStream_t st);

// ipa_train.hip: training side of the IPA core -----------------------------------------------------
size_t ipa_train_ws_floats(int N, int L);
int launch_ipa_train_forward(const float* proj_local, const float* R, const float* t, const float* z, const uint8_t* mask,
                             const float* Wb, const float* spatial_coef, const float* pbc, float* feat, float* alpha, int N, int L, float* ws,
                             hipStream_t st);
int launch_ipa_points_backward(const float* dfeat, int ld_dfeat, const float* feat, const float* R, const float* t, float* dout_cat, float* delta,
                               int N, int L, hipStream_t st);
int launch_ipa_backward_operands(const float* proj, const float* R, const float* t, float* Aq, float* Ak, float* Av, int N, int L, hipStream_t st);
int launch_ipa_backward_assemble(const float* P1, const float* P2, const float* P3, const float* Aq, const float* Ak, const float* R,
                                 const float* spatial_coef, float* dproj, float* e, int N, int L, hipStream_t st);
int launch_ipa_pair_backward(const float* z, const float* alpha, const float* dalpha_node, const float* delta, const float* dfeat, int ld_dfeat,
                             const float* Wb, float* g_out, float* dz, float* dwb_part /* [N*L, 12*C] per-row partials of d proj_pair_bias.weight */,
                             int N, int L, hipStream_t st, int dz_accumulate = 0 /* 1: dz += (the blocks of an encoder share one buffer) */);
// d pair_feat of all blocks in one pass from what their backward passes left (alpha, g, d feat, proj_pair_bias.weight), no z read
int launch_ipa_dz_assemble(int nl, const float* const* alpha, const float* const* g, const float* const* dfeat, int ld_dfeat, const float* const* wb,
                           float* dz, int N, int L, hipStream_t st);

// embed.hip: encode() ----------------------------------------------------------------------------
size_t residue_embed_ws_bytes(int N, int L, int A, int hotspot);
int launch_residue_embed(const abopt_encode_inputs* in, const abopt_residue_embed_weights* w, float* res_feat, float* R, float* p,
                         void* ws, size_t ws_bytes, hipStream_t st);
size_t pair_embed_ws_bytes(int N, int L, int A);
int launch_pair_embed(const abopt_encode_inputs* in, const abopt_pair_embed_weights* w, float* pair_feat, float* acts, float* gsave, float* tsave,
                      void* ws, size_t ws_bytes, hipStream_t st);
size_t pair_embed_backward_ws_bytes(int N, int L, int A);
size_t residue_features_ws_bytes(int N, int L);
int launch_residue_features(const abopt_encode_inputs* in, const abopt_residue_embed_weights* w, float* feat, float* R, float* p, void* ws, size_t ws_bytes,
                            hipStream_t st);
int launch_pair_embed_backward(const abopt_encode_inputs* in, const abopt_pair_embed_weights* w, const float* dout, const float* acts, const float* tsave,
                               float* dys, float* ds, float* dy_colsum, void* ws, size_t ws_bytes, hipStream_t st);

int launch_reconstruct_backbone(const float* pos_ctx, const float* R_new, const float* t_new, const int64_t* aa, const int64_t* chain_nb,
                                const int64_t* res_nb, const uint8_t* mask_atoms, const uint8_t* mask_recons, const float* bb_table,
                                const float* o_table, float* pos_new, uint8_t* mask_new, int N, int L, int A, hipStream_t st);

}  // namespace abopt

// C-ABI entry points that orchestrate kernel launches (declared in include/abopt.h).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include "ipa_common.h"
#include "kernels.h"

namespace abopt {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::mutex g_dev_mu;
int device_cu_count(int* cus) {
    static int cached[kMaxDevices] = {};
    int dev = 0;
    ABOPT_HIP(hipGetDevice(&dev));
    ABOPT_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (!cached[dev]) {
        hipDeviceProp_t prop;
        ABOPT_HIP(hipGetDeviceProperties(&prop, dev));
        cached[dev] = prop.multiProcessorCount;
    }
    *cus = cached[dev];
    return ABOPT_OK;
}
int ensure_dynamic_lds(const void* kernel, size_t bytes, LdsConfig& cfg) {
    int dev = 0;
    ABOPT_HIP(hipGetDevice(&dev));
    ABOPT_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (bytes > cfg.bytes[dev]) {
        ABOPT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        cfg.bytes[dev] = bytes;
    }
    return ABOPT_OK;
}

// bump allocator over the caller's workspace, 256-byte aligned carves
struct Carver {
    char* base; size_t cap, off = 0; bool ok = true;
    Carver(void* p, size_t n) : base((char*)p), cap(n) {}
    float* f(size_t nfloat) {
        const size_t bytes = (nfloat * sizeof(float) + 255) & ~(size_t)255;
        if (base && off + bytes > cap) { ok = false; return nullptr; }
        float* r = base ? (float*)(base + off) : nullptr;
        off += bytes;
        return r;
    }
};

constexpr int F = 128, FI = F + 4;
constexpr int OUT_KSPLIT = 2;     // out_transform (K = 1824, N = 128) has only M/64 * 2 tiles: split K so it fills the chip

struct GaScratch { float *proj, *feat, *u, *kvf, *qf, *split; size_t split_floats; float* xt[2]; };
static GaScratch carve_ga(Carver& cv, int64_t M, int N, int L) {
    GaScratch s;
    s.proj = cv.f((size_t)M * NP);
    s.feat = cv.f((size_t)M * ABOPT_IPA_FEAT);
    s.u = cv.f((size_t)M * F * OUT_KSPLIT);
    s.kvf = cv.f(ipa_kvfrag_floats(N, L));
    s.qf = cv.f(ipa_qfrag_floats(N, L));
    s.split_floats = ipa_split_ws_floats(N, L);
    s.split = s.split_floats ? cv.f(s.split_floats) : nullptr;
    s.xt[0] = cv.f((size_t)M * F);          // node features as two fp16 terms, ping-pong between blocks (written by the tail / mixer, read by node_frags)
    s.xt[1] = cv.f((size_t)M * F);
    return s;
}

static int ga_block(const abopt_ga_weights* w, const float* R, const float* t, const float* x, const float* z, const uint8_t* mask,
                    float* x_out, int N, int L, const abopt_ga_debug* dbg, const GaScratch& s, hipStream_t st, const float* pbc = nullptr, int z_shared = 0,
                    const float* pair_terms = nullptr, float* feat_out = nullptr, const float* x_terms = nullptr, float* xt_out = nullptr) {
    const int64_t M = (int64_t)N * L;
    int rc;
    // node projections q|k|v|qp|kp|vp, points to the global frame, MFMA fragment layout: one fused kernel when the packed weights are given
    // the term forms (round 6) go together: fragments with q / k channel terms are written only for the kernels that read them, the 32-row block kernels
    // handed pair terms -- every other core reads fp32 channel slots
    if (!(pbc && pair_terms && !dbg && w->w_node_frag && ipa_core32_applies(N, L, z_shared))) pair_terms = nullptr;
    if (w->w_node_frag) {
        if ((rc = launch_node_frags(x, w->w_node_frag, R, t, w->spatial_coef, s.qf, s.kvf, N, L, st, pair_terms ? 1 : 0, x_terms))) return rc;
    } else {
        if ((rc = launch_linear(x, F, w->w_node, F, nullptr, s.proj, NP, (int)M, ABOPT_NODE_PROJ, F, false, st))) return rc;
        if ((rc = launch_ipa_frags(s.proj, R, t, w->spatial_coef, s.qf, s.kvf, N, L, st))) return rc;
    }
    if (!dbg && !feat_out && pbc && w->w_out_terms && w->w_mlp_frag) {
        // core + tail as one kernel (feat stays on the chip) wherever the 32-row core is the one to run; bit-identical to the two launches below
        int fused = 0;
        if ((rc = launch_ipa_block_fused(s.qf, s.kvf, z, mask, R, t, pbc, N, L, st, z_shared, w->w_out_terms, w->w_mlp_frag, x, w->b_out, w->ln1_gamma,
                                         w->ln1_beta, w->b_mlp0, w->b_mlp1, w->b_mlp2, w->ln2_gamma, w->ln2_beta, x_out, &fused, pair_terms, xt_out))) return rc;
        if (fused) return ABOPT_OK;
    }
    float* feat = (dbg && dbg->feat) ? dbg->feat : (feat_out ? feat_out : s.feat);
    if ((rc = launch_ipa_core(s.qf, s.kvf, z, mask, R, t, w->w_pair_bias, feat,
                              dbg ? dbg->logits : nullptr, dbg ? dbg->alpha : nullptr, pbc, N, L, st, z_shared, s.split, s.split_floats, pair_terms))) return rc;
    // out_transform -> mask -> +x -> LN1 -> MLP -> +res -> LN2
    if (w->w_out_frag && w->w_mlp_frag)
        return launch_out_ln_mlp(feat, w->w_out_frag, w->w_mlp_frag, x, w->b_out, mask, w->ln1_gamma, w->ln1_beta, w->b_mlp0, w->b_mlp1, w->b_mlp2,
                                 w->ln2_gamma, w->ln2_beta, x_out, nullptr, M, st, xt_out);
    if ((rc = launch_linear(feat, ABOPT_IPA_FEAT, w->w_out, ABOPT_IPA_FEAT, nullptr, s.u, F, (int)M, F, ABOPT_IPA_FEAT, false, st,
                            OUT_KSPLIT, M * F))) return rc;
    if ((rc = launch_fused_ln_mlp(x, s.u, OUT_KSPLIT, M * F, w->b_out, mask, w->ln1_gamma, w->ln1_beta, w->w_mlp0, w->b_mlp0, w->w_mlp1, w->b_mlp1,
                                  w->w_mlp2, w->b_mlp2, w->ln2_gamma, w->ln2_beta, x_out, M, st))) return rc;
    return ABOPT_OK;
}

static int check_ga_weights(const abopt_ga_weights* w) {
    ABOPT_CHECK_ARG(w && w->w_node && w->w_pair_bias && w->spatial_coef && w->w_out && w->b_out && w->ln1_gamma && w->ln1_beta &&
                    w->w_mlp0 && w->b_mlp0 && w->w_mlp1 && w->b_mlp1 && w->w_mlp2 && w->b_mlp2 && w->ln2_gamma && w->ln2_beta,
                    "GABlock weights: NULL pointer");
    return ABOPT_OK;
}

}  // namespace abopt

using namespace abopt;

extern "C" int abopt_abi_version(void) { return ABOPT_ABI_VERSION; }

extern "C" size_t abopt_node_frag_floats(void) { return node_wfrag_floats(); }
extern "C" int abopt_node_frag_source_row(int h, int T, int m) {
    if (h < 0 || h >= H || T < 0 || T >= 12 || m < 0 || m >= 16) return -1;
    static const int kSet[12] = {0, 0, 1, 1, 3, 3, 4, 4, 2, 2, 5, 5};   // tile -> projection: 0 q, 1 k, 2 v, 3 q_pts, 4 k_pts, 5 v_pts
    const int set = kSet[T], half = T % 2;
    if (set < 3) return set * (H * D) + h * D + half * 16 + m;
    const int p = half * 4 + m / 4, c = m % 4;
    if (c == 3) return -1;
    return 3 * H * D + (set - 3) * NPT + h * (P * 3) + p * 3 + c;
}
extern "C" const char* abopt_last_error(void) { return g_err; }

extern "C" size_t abopt_heads_frag_floats(void) { return heads_wfrag_floats(); }
extern "C" size_t abopt_mixer_frag_floats(void) { return mixer_wfrag_floats(); }
extern "C" size_t abopt_out_frag_floats(void) { return out_wfrag_floats(); }
extern "C" size_t abopt_mlp_frag_floats(void) { return mlp_wfrag_floats(); }
extern "C" size_t abopt_out_terms_floats(void) { return out_wterms_floats(); }
extern "C" int abopt_out_frag_terms(const float* w_out_frag, float* w_out_terms, abopt_stream stream) {
    ABOPT_CHECK_ARG(w_out_frag && w_out_terms, "out_frag_terms: NULL argument");
    return launch_out_frag_terms(w_out_frag, w_out_terms, (hipStream_t)stream);
}
extern "C" int abopt_pack_tail_weights(const float* w_out, const float* w_mlp0, const float* w_mlp1, const float* w_mlp2, float* w_out_frag,
                                       float* w_mlp_frag, float* w_mlpT_frag, abopt_stream stream) {
    ABOPT_CHECK_ARG(w_out && w_mlp0 && w_mlp1 && w_mlp2 && w_out_frag && w_mlp_frag, "pack_tail_weights: NULL argument");
    return launch_pack_tail_weights(w_out, w_mlp0, w_mlp1, w_mlp2, w_out_frag, w_mlp_frag, w_mlpT_frag, (hipStream_t)stream);
}
extern "C" int abopt_block_tail_forward(const float* feat, const float* w_out_frag, const float* w_mlp_frag, const float* x, const float* b_out,
                                        const uint8_t* mask, const float* ln1_gamma, const float* ln1_beta, const float* b_mlp0, const float* b_mlp1,
                                        const float* b_mlp2, const float* ln2_gamma, const float* ln2_beta, float* out, float* saved, int64_t rows,
                                        abopt_stream stream) {
    ABOPT_CHECK_ARG(rows >= 0, "block_tail_forward: negative row count");
    if (rows == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(feat && w_out_frag && w_mlp_frag && x && ln1_gamma && ln1_beta && b_mlp0 && b_mlp1 && b_mlp2 && ln2_gamma && ln2_beta && out,
                    "block_tail_forward: NULL argument");
    return launch_out_ln_mlp(feat, w_out_frag, w_mlp_frag, x, b_out, mask, ln1_gamma, ln1_beta, b_mlp0, b_mlp1, b_mlp2, ln2_gamma, ln2_beta, out, saved,
                             rows, (hipStream_t)stream);
}
extern "C" int abopt_block_tail_backward(const float* dout, const float* saved, const float* w_mlpT_frag, const uint8_t* mask, const float* ln1_gamma,
                                         const float* ln2_gamma, float* dpre, float* da1, float* du, float* colpart, int64_t rows, abopt_stream stream) {
    ABOPT_CHECK_ARG(rows >= 0, "block_tail_backward: negative row count");
    if (rows == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(dout && saved && w_mlpT_frag && ln1_gamma && ln2_gamma && dpre && da1 && du && colpart, "block_tail_backward: NULL argument");
    return launch_tail_backward(dout, saved, w_mlpT_frag, mask, ln1_gamma, ln2_gamma, dpre, da1, du, colpart, rows, (hipStream_t)stream);
}

extern "C" int abopt_device_info(int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len) {
    int dev = 0;
    ABOPT_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    ABOPT_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return ABOPT_OK;
}

static int check_encode_inputs(const abopt_encode_inputs* in, const char* who) {
    ABOPT_CHECK_ARG(in, "%s: NULL inputs", who);
    ABOPT_CHECK_ARG(in->N >= 0 && in->L >= 0 && (int64_t)in->N * in->L < (1ll << 31) / 64, "%s: bad batch dims N=%d L=%d", who, in->N, in->L);
    if ((int64_t)in->N * in->L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(in->aa && in->res_nb && in->chain_nb && in->pos_atoms && in->mask_atoms, "%s: NULL input tensor", who);
    return ABOPT_OK;
}

extern "C" size_t abopt_residue_embed_workspace_bytes(int N, int L, int atoms, int hotspot) { return residue_embed_ws_bytes(N, L, atoms, hotspot); }
extern "C" size_t abopt_pair_embed_workspace_bytes(int N, int L, int atoms) { return pair_embed_ws_bytes(N, L, atoms); }

extern "C" int abopt_residue_embed_forward(const abopt_encode_inputs* in, const abopt_residue_embed_weights* w, float* res_feat, float* R, float* p,
                                           void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_encode_inputs(in, "residue_embed_forward"))) return rc;
    if ((int64_t)in->N * in->L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(in->fragment_type && w && w->aatype_embed && w->type_embed && w->freq_bands && w->w0 && w->b0 && w->w1 && w->b1 && w->w2 && w->b2 && w->w3 && w->b3 && ws,
                    "residue_embed_forward: NULL argument");
    return launch_residue_embed(in, w, res_feat, R, p, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t abopt_residue_features_workspace_bytes(int N, int L) { return residue_features_ws_bytes(N, L); }

extern "C" int abopt_residue_features(const abopt_encode_inputs* in, const abopt_residue_embed_weights* w, float* features, float* R, float* p,
                                      void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_encode_inputs(in, "residue_features"))) return rc;
    if ((int64_t)in->N * in->L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(in->fragment_type && w && w->aatype_embed && w->type_embed && w->freq_bands && features && R && p && ws, "residue_features: NULL argument");
    return launch_residue_features(in, w, features, R, p, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int abopt_pair_embed_forward(const abopt_encode_inputs* in, const abopt_pair_embed_weights* w, float* pair_feat, float* activations,
                                        float* gauss, float* dgauss, void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_encode_inputs(in, "pair_embed_forward"))) return rc;
    if ((int64_t)in->N * in->L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(w && w->aa_pair_embed && w->relpos_embed && w->aapair_to_distcoef && w->freq_bands && w->wd0 && w->bd0 && w->wd1 && w->bd1 &&
                    w->wo0 && w->bo0 && w->wo1 && w->bo1 && w->wo2 && w->bo2 && pair_feat && ws, "pair_embed_forward: NULL argument");
    ABOPT_CHECK_ARG(gauss || !dgauss, "pair_embed_forward: dgauss needs gauss");
    return launch_pair_embed(in, w, pair_feat, activations, gauss, dgauss, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t abopt_pair_embed_backward_workspace_bytes(int N, int L, int atoms) { return pair_embed_backward_ws_bytes(N, L, atoms); }

extern "C" int abopt_pair_embed_backward(const abopt_encode_inputs* in, const abopt_pair_embed_weights* w, const float* dpair_feat,
                                         const float* activations, const float* dgauss, float* dys, float* dsoftplus, float* dys_colsum,
                                         void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_encode_inputs(in, "pair_embed_backward"))) return rc;
    if ((int64_t)in->N * in->L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(w && w->wd0 && w->wd1 && w->wo0 && w->wo1 && w->wo2 && dpair_feat && activations && dys && dsoftplus && ws && w->aapair_to_distcoef && w->aa_pair_embed && w->relpos_embed,
                    "pair_embed_backward: NULL argument");
    return launch_pair_embed_backward(in, w, dpair_feat, activations, dgauss, dys, dsoftplus, dys_colsum, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int abopt_dpm_losses(const float* R_pred, const float* R_0, const float* p_pred, const float* p_target, const float* c_denoised, const int64_t* s_t,
                                const int64_t* s_0, const float* alpha_bar_t, const uint8_t* mask_generate, int N, int L, float* block_sums, float* dR_pred,
                                float* dp_pred, float* dc_denoised, abopt_stream stream) {
    ABOPT_CHECK_ARG(N >= 0 && L >= 0 && R_pred && R_0 && p_pred && p_target && c_denoised && s_t && s_0 && alpha_bar_t && mask_generate && block_sums && dR_pred &&
                    dp_pred && dc_denoised, "dpm_losses: NULL argument");
    return launch_dpm_losses(R_pred, R_0, p_pred, p_target, c_denoised, s_t, s_0, alpha_bar_t, mask_generate, N, L, block_sums, dR_pred, dp_pred, dc_denoised,
                             (hipStream_t)stream);
}

extern "C" int abopt_abdock_losses(const float* prmsd_logits, const float* p_pred, const float* p0_norm, const float* coef_a, const float* coef_b,
                                   const uint8_t* mask_generate, const uint8_t* mask_res, const float* bin_offsets, int num_bins, int N, int L, float position_scale,
                                   int pred_x0, float* sample_parts, float* dprmsd_logits, float* dp_pred, abopt_stream stream) {
    ABOPT_CHECK_ARG(N >= 0 && L >= 0 && prmsd_logits && p_pred && p0_norm && mask_generate && mask_res && bin_offsets && sample_parts && dprmsd_logits && dp_pred &&
                    (pred_x0 || (coef_a && coef_b)), "abdock_losses: NULL argument");
    return launch_abdock_losses(prmsd_logits, p_pred, p0_norm, coef_a, coef_b, mask_generate, mask_res, bin_offsets, num_bins, N, L, position_scale, pred_x0,
                                sample_parts, dprmsd_logits, dp_pred, (hipStream_t)stream);
}

extern "C" int abopt_layer_norm_forward(const float* x, const float* gamma, const float* beta, int cols, float eps, int64_t rows, float* y, float* xhat, float* rstd,
                                        abopt_stream stream) {
    ABOPT_CHECK_ARG(rows >= 0 && x && gamma && beta && y && (!xhat == !rstd), "layer_norm_forward: NULL argument (xhat and rstd come together)");
    return launch_row_layer_norm(x, gamma, beta, cols, eps, rows, y, xhat, rstd, (hipStream_t)stream);
}
extern "C" int abopt_layer_norm_backward(const float* dy, const float* xhat, const float* rstd, const float* gamma, int cols, int64_t rows, float* dx, float* dy_xhat,
                                         abopt_stream stream) {
    ABOPT_CHECK_ARG(rows >= 0 && dy && xhat && rstd && gamma && dx && dy_xhat, "layer_norm_backward: NULL argument");
    return launch_row_layer_norm_backward(dy, xhat, rstd, gamma, cols, rows, dx, dy_xhat, (hipStream_t)stream);
}

extern "C" int abopt_heads_epilogue_forward(const float* R, const float* v_t, const float* eps_crd, const float* eps_rot, const uint8_t* mask_generate,
                                            float* v_next, float* R_next, float* eps_pos, int64_t rows, int grad_mode, abopt_stream stream) {
    ABOPT_CHECK_ARG(rows >= 0 && R && eps_crd && eps_rot && mask_generate && R_next && eps_pos && (!v_next || v_t), "heads_epilogue_forward: NULL argument");
    return launch_heads_epilogue(R, v_t, eps_crd, eps_rot, nullptr, 3, 0, mask_generate, v_next, R_next, eps_pos, nullptr, rows, grad_mode, (hipStream_t)stream);
}

extern "C" int abopt_heads_epilogue_backward(const float* R, const float* eps_rot, const uint8_t* mask_generate, const float* dR_next, const float* deps_pos,
                                             float* deps_crd, float* deps_rot, int64_t rows, abopt_stream stream) {
    ABOPT_CHECK_ARG(rows >= 0 && R && eps_rot && mask_generate && deps_crd && deps_rot, "heads_epilogue_backward: NULL argument");
    return launch_heads_epilogue_backward(R, eps_rot, 3, mask_generate, dR_next, deps_pos, deps_crd, deps_rot, rows, (hipStream_t)stream);
}

extern "C" int abopt_reconstruct_backbone_partially(const float* pos_ctx, const float* R_new, const float* t_new, const int64_t* aa,
                                                    const int64_t* chain_nb, const int64_t* res_nb, const uint8_t* mask_atoms,
                                                    const uint8_t* mask_recons, const float* bb_table, const float* o_table,
                                                    float* pos_new, uint8_t* mask_new, int N, int L, int A, abopt_stream stream) {
    ABOPT_CHECK_ARG(N >= 0 && L >= 0 && A >= 4, "reconstruct_backbone_partially: bad dims N=%d L=%d A=%d (A >= 4)", N, L, A);
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(pos_ctx && R_new && t_new && aa && chain_nb && res_nb && mask_atoms && mask_recons && bb_table && o_table && pos_new && mask_new,
                    "reconstruct_backbone_partially: NULL argument");
    return launch_reconstruct_backbone(pos_ctx, R_new, t_new, aa, chain_nb, res_nb, mask_atoms, mask_recons, bb_table, o_table, pos_new, mask_new,
                                       N, L, A, (hipStream_t)stream);
}

extern "C" int abopt_so3_exp(const float* w, float* R, int64_t n, abopt_stream stream) {
    ABOPT_CHECK_ARG(w && R && n >= 0, "so3_exp: bad arguments");
    return launch_so3_exp(w, R, n, (hipStream_t)stream);
}
extern "C" int abopt_so3_log(const float* R, float* w, int64_t n, int grad_mode, abopt_stream stream) {
    ABOPT_CHECK_ARG(w && R && n >= 0, "so3_log: bad arguments");
    return launch_so3_log(R, w, n, grad_mode, (hipStream_t)stream);
}

extern "C" size_t abopt_ga_workspace_bytes(int N, int L, int Fd, int Cd) {
    (void)Fd; (void)Cd;
    Carver cv(nullptr, 0);
    const int64_t M = (int64_t)N * L;
    carve_ga(cv, M, N, L);
    cv.f((size_t)M * F);   // ping-pong buffer for the encoder
    return cv.off;
}

static int check_dims(int N, int L, int Fd, int Cd) {
    ABOPT_CHECK_ARG(N >= 0 && L >= 0, "negative batch dims N=%d L=%d", N, L);
    if (Fd != F || Cd != C) { set_error("this build supports res_feat_dim=128, pair_feat_dim=64 only (got %d, %d)", Fd, Cd); return ABOPT_EUNSUPPORTED; }
    ABOPT_CHECK_ARG((int64_t)N * L < (1ll << 31) / 64, "N*L too large");
    return ABOPT_OK;
}

extern "C" size_t abopt_ipa_train_workspace_bytes(int N, int L) { return ipa_train_ws_floats(N, L) * sizeof(float); }

extern "C" int abopt_ipa_core_train_forward(const float* proj_local, const float* R, const float* t, const float* pair_feat, const uint8_t* mask,
                                            const float* w_pair_bias, const float* spatial_coef, const float* pair_bias_cache, float* feat,
                                            float* alpha, int N, int L, int Cd, void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, F, Cd))) return rc;
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(proj_local && R && t && pair_feat && mask && w_pair_bias && spatial_coef && feat && alpha && ws, "ipa_core_train_forward: NULL argument");
    if (ws_bytes < ipa_train_ws_floats(N, L) * sizeof(float)) { set_error("ipa_core_train_forward: workspace too small (%zu bytes given)", ws_bytes); return ABOPT_EWORKSPACE; }
    return launch_ipa_train_forward(proj_local, R, t, pair_feat, mask, w_pair_bias, spatial_coef, pair_bias_cache, feat, alpha, N, L, (float*)ws,
                                    (hipStream_t)stream);
}

extern "C" int abopt_ipa_points_backward(const float* dfeat, int ld_dfeat, const float* feat, const float* R, const float* t,
                                         float* dout_cat, float* delta, int N, int L, abopt_stream stream) {
    ABOPT_CHECK_ARG(N >= 0 && L >= 0, "ipa_points_backward: negative dims");
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(dfeat && feat && R && t && dout_cat && delta && ld_dfeat >= ABOPT_IPA_FEAT, "ipa_points_backward: bad argument");
    return launch_ipa_points_backward(dfeat, ld_dfeat, feat, R, t, dout_cat, delta, N, L, (hipStream_t)stream);
}

extern "C" int abopt_ipa_backward_operands(const float* proj_local, const float* R, const float* t, float* Aq, float* Ak, float* Av,
                                           int N, int L, abopt_stream stream) {
    ABOPT_CHECK_ARG(N >= 0 && L >= 0, "ipa_backward_operands: negative dims");
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(proj_local && R && t && Aq && Ak && Av, "ipa_backward_operands: NULL argument");
    return launch_ipa_backward_operands(proj_local, R, t, Aq, Ak, Av, N, L, (hipStream_t)stream);
}

extern "C" int abopt_ipa_backward_assemble(const float* P1, const float* P2, const float* P3, const float* Aq, const float* Ak, const float* R,
                                           const float* spatial_coef, float* dproj, float* e, int N, int L, abopt_stream stream) {
    ABOPT_CHECK_ARG(N >= 0 && L >= 0, "ipa_backward_assemble: negative dims");
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(P1 && P2 && P3 && Aq && Ak && R && spatial_coef && dproj && e, "ipa_backward_assemble: NULL argument");
    return launch_ipa_backward_assemble(P1, P2, P3, Aq, Ak, R, spatial_coef, dproj, e, N, L, (hipStream_t)stream);
}

extern "C" int abopt_ipa_pair_backward(const float* pair_feat, const float* alpha, const float* dalpha_node, const float* delta,
                                       const float* dfeat, int ld_dfeat, const float* w_pair_bias, float* g, float* dpair_feat,
                                       float* dw_pair_bias_rows, int dpair_feat_accumulate, int N, int L, int Cd, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, F, Cd))) return rc;
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(pair_feat && alpha && dalpha_node && delta && dfeat && w_pair_bias && g && dw_pair_bias_rows && ld_dfeat >= ABOPT_HEADS * 64 && (ld_dfeat % 4) == 0,
                    "ipa_pair_backward: bad argument");          // dpair_feat may be NULL: d pair_feat is then left to abopt_ipa_dz_assemble
    return launch_ipa_pair_backward(pair_feat, alpha, dalpha_node, delta, dfeat, ld_dfeat, w_pair_bias, g, dpair_feat, dw_pair_bias_rows, N, L, (hipStream_t)stream,
                                    dpair_feat_accumulate ? 1 : 0);
}

extern "C" int abopt_ipa_dz_assemble(int num_blocks, const float* const* alpha, const float* const* g, const float* const* dfeat, int ld_dfeat,
                                     const float* const* w_pair_bias, float* dpair_feat, int N, int L, int Cd, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, F, Cd))) return rc;
    ABOPT_CHECK_ARG(alpha && g && dfeat && w_pair_bias && dpair_feat && ld_dfeat >= ABOPT_HEADS * 64 && (ld_dfeat % 4) == 0, "ipa_dz_assemble: bad argument");
    return launch_ipa_dz_assemble(num_blocks, alpha, g, dfeat, ld_dfeat, w_pair_bias, dpair_feat, N, L, (hipStream_t)stream);
}

extern "C" int abopt_ga_block_forward(const abopt_ga_weights* w, const float* R, const float* t, const float* x, const float* z,
                                      const uint8_t* mask, float* x_out, int N, int L, int Fd, int Cd, const abopt_ga_debug* dbg,
                                      void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, Fd, Cd))) return rc;
    if ((rc = check_ga_weights(w))) return rc;
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(R && t && x && z && mask && x_out && ws, "ga_block_forward: NULL argument");
    Carver cv(ws, ws_bytes);
    GaScratch s = carve_ga(cv, (int64_t)N * L, N, L);
    if (!cv.ok) { set_error("ga_block_forward: workspace too small (%zu bytes given)", ws_bytes); return ABOPT_EWORKSPACE; }
    return ga_block(w, R, t, x, z, mask, x_out, N, L, dbg, s, (hipStream_t)stream);
}

extern "C" int abopt_ga_block_forward_cached(const abopt_ga_weights* w, const float* R, const float* t, const float* x, const float* z,
                                             const uint8_t* mask, float* x_out, int N, int L, int Fd, int Cd, const float* pair_bias_cache,
                                             const float* pair_terms, int pair_feat_shared, float* feat_out, void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, Fd, Cd))) return rc;
    if ((rc = check_ga_weights(w))) return rc;
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(R && t && x && z && mask && x_out && ws && pair_bias_cache, "ga_block_forward_cached: NULL argument");
    ABOPT_CHECK_ARG(pair_feat_shared >= 0 && (pair_feat_shared <= 1 || N % pair_feat_shared == 0), "ga_block_forward_cached: pair_feat_shared=%d does not divide N=%d", pair_feat_shared, N);
    Carver cv(ws, ws_bytes);
    GaScratch s = carve_ga(cv, (int64_t)N * L, N, L);
    if (!cv.ok) { set_error("ga_block_forward_cached: workspace too small (%zu bytes given)", ws_bytes); return ABOPT_EWORKSPACE; }
    const int zg = pair_feat_shared == 1 ? N : pair_feat_shared;
    return ga_block(w, R, t, x, z, mask, x_out, N, L, nullptr, s, (hipStream_t)stream, pair_bias_cache, zg, pair_terms, feat_out);
}

static int ga_encoder(const abopt_ga_weights* blocks, int num_layers, const float* R, const float* t, const float* x, const float* z,
                      const uint8_t* mask, float* x_out, int N, int L, const GaScratch& s, float* pong, hipStream_t st,
                      const float* pair_bias_cache = nullptr, int z_shared = 0, const float* pair_terms = nullptr, const float* x_terms = nullptr) {
    // ga.py:190-193: the same R, t, z feed every block.  Ping-pong so the last block writes x_out.
    const float* cur = x;
    // x as fp16 terms travels with x from block to block (round 6): block i reads the terms its producer wrote (the mixer for block 0, `x_terms`; the tail of block
    // i - 1 afterwards) and has its own tail write block i + 1's -- where the packed weights put node_frags and the term-writing tails on the path; ABOPT_X_TERMS=0: never
    const bool xt_off = getenv("ABOPT_X_TERMS") && getenv("ABOPT_X_TERMS")[0] == '0';
    const float* xt_cur = xt_off ? nullptr : x_terms;
    for (int i = 0; i < num_layers; ++i) {
        float* dst = ((num_layers - 1 - i) % 2 == 0) ? x_out : pong;
        const abopt_ga_weights* w = &blocks[i];
        const bool writes = !xt_off && i + 1 < num_layers && w->w_out_frag && w->w_mlp_frag && blocks[i + 1].w_node_frag;
        float* xt_dst = writes ? s.xt[i & 1] : nullptr;
        int rc = ga_block(w, R, t, cur, z, mask, dst, N, L, nullptr, s, st,
                          pair_bias_cache ? pair_bias_cache + (size_t)i * pair_bias_layer_floats(z_shared ? N / z_shared : N, L) : nullptr, z_shared, pair_bias_cache ? pair_terms : nullptr,
                          nullptr, xt_cur, xt_dst);
        if (rc) return rc;
        cur = dst;
        xt_cur = xt_dst;
    }
    if (num_layers == 0) ABOPT_HIP(hipMemcpyAsync(x_out, x, (size_t)N * L * F * sizeof(float), hipMemcpyDeviceToDevice, st));
    return ABOPT_OK;
}

extern "C" int abopt_ga_encoder_forward(const abopt_ga_weights* blocks, int num_layers, const float* R, const float* t, const float* x,
                                        const float* z, const uint8_t* mask, float* x_out, int N, int L, int Fd, int Cd,
                                        void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, Fd, Cd))) return rc;
    ABOPT_CHECK_ARG(blocks && num_layers >= 0, "ga_encoder_forward: bad block list");
    for (int i = 0; i < num_layers; ++i) if ((rc = check_ga_weights(&blocks[i]))) return rc;
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(R && t && x && z && mask && x_out && ws, "ga_encoder_forward: NULL argument");
    ABOPT_CHECK_ARG(x != x_out, "ga_encoder_forward: x_out must not alias x");
    Carver cv(ws, ws_bytes);
    const int64_t M = (int64_t)N * L;
    GaScratch s = carve_ga(cv, M, N, L);
    float* pong = cv.f((size_t)M * F);
    if (!cv.ok) { set_error("ga_encoder_forward: workspace too small (%zu bytes given)", ws_bytes); return ABOPT_EWORKSPACE; }
    return ga_encoder(blocks, num_layers, R, t, x, z, mask, x_out, N, L, s, pong, (hipStream_t)stream);
}

extern "C" size_t abopt_pair_bias_cache_bytes(int N, int L, int num_layers) {
    return (size_t)num_layers * pair_bias_layer_floats(N, L) * sizeof(float);
}

extern "C" int abopt_nonfinite_flag(int reset, abopt_stream stream) {
    int flag = 0;
    const int rc = nonfinite_flag_read(reset, (hipStream_t)stream, &flag);
    return rc ? -1 : flag;
}

extern "C" size_t abopt_pair_terms_bytes(int N, int L) { return pair_terms_blob_floats(N, L) * sizeof(float); }

extern "C" int abopt_pair_terms_used(int N, int L, int pair_feat_shared) {
    if (N <= 0 || L <= 0) return 0;
    return ipa_core32_applies(N, L, pair_feat_shared) ? 1 : 0;
}

extern "C" int abopt_pair_terms(const float* pair_feat, float* terms, int N, int L, int Cd, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, 128, Cd))) return rc;
    ABOPT_CHECK_ARG(pair_feat && terms, "pair_terms: NULL argument");
    ABOPT_CHECK_ARG(L <= 2048, "pair_terms: L=%d (the kernels that read the terms address a sample's slab with 32-bit offsets: L <= 2048)", L);
    return launch_pair_terms(pair_feat, terms, N, L, (hipStream_t)stream);
}

extern "C" int abopt_pair_bias_cache(const abopt_ga_weights* blocks, int num_layers, const float* pair_feat, float* cache,
                                     int N, int L, int Cd, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, 128, Cd))) return rc;
    ABOPT_CHECK_ARG(blocks && pair_feat && cache && num_layers >= 1 && num_layers <= 8, "pair_bias_cache: bad arguments");
    const float* wb[8];
    for (int l = 0; l < num_layers; ++l) { ABOPT_CHECK_ARG(blocks[l].w_pair_bias, "pair_bias_cache: NULL weight"); wb[l] = blocks[l].w_pair_bias; }
    return launch_pair_bias_cache(pair_feat, wb, num_layers, cache, N, L, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------- EpsilonNet
namespace {
struct EpsScratch { GaScratch ga; float *pong, *R, *cat, *x0, *xe, *infeat, *infeat_ln, *hh1, *hh2, *out3, *pr1, *pr2, *pr3; };
EpsScratch carve_eps(Carver& cv, int64_t M, int N, int L, int num_bins) {
    EpsScratch e;
    e.ga = carve_ga(cv, M, N, L);
    e.pong = cv.f((size_t)M * F);
    e.R = cv.f((size_t)M * 9);
    e.cat = cv.f((size_t)M * 2 * F);
    e.x0 = cv.f((size_t)M * F);
    e.xe = cv.f((size_t)M * F);
    e.infeat = cv.f((size_t)M * FI);
    e.infeat_ln = cv.f((size_t)M * FI);
    e.hh1 = cv.f((size_t)M * 3 * F);
    e.hh2 = cv.f((size_t)M * 3 * F);
    e.out3 = cv.f((size_t)M * 32);
    e.pr1 = cv.f((size_t)M * F);
    e.pr2 = cv.f((size_t)M * F);
    e.pr3 = cv.f((size_t)M * (size_t)(num_bins > 0 ? num_bins : 1));
    return e;
}
}  // namespace

extern "C" size_t abopt_eps_workspace_bytes(int N, int L, int Fd, int Cd) {
    (void)Fd; (void)Cd;
    Carver cv(nullptr, 0);
    carve_eps(cv, (int64_t)N * L, N, L, 64);
    return cv.off;
}

extern "C" int abopt_eps_net_forward(const abopt_eps_weights* w, const float* v_t, const float* p_t, const int64_t* s_t,
                                     const float* res_feat, const float* pair_feat, const float* beta,
                                     const uint8_t* mask_generate, const uint8_t* mask_res,
                                     float* v_next, float* R_next, float* eps_pos, float* c_denoised, float* prmsd_logits,
                                     int N, int L, int Fd, int Cd, int grad_mode, const float* pair_bias_cache, int pair_feat_shared,
                                     const float* pair_terms, void* ws, size_t ws_bytes, abopt_stream stream) {
    int rc;
    if ((rc = check_dims(N, L, Fd, Cd))) return rc;
    ABOPT_CHECK_ARG(w && w->seq_embed && w->w_mix0 && w->b_mix0 && w->w_mix1 && w->b_mix1 && w->blocks && w->w_head1 && w->b_head1 &&
                    w->w_crd2 && w->b_crd2 && w->w_crd3 && w->b_crd3 && w->w_rot2 && w->b_rot2 && w->w_rot3 && w->b_rot3 &&
                    w->w_seq2 && w->b_seq2 && w->w_seq3 && w->b_seq3, "eps_net_forward: NULL weight pointer");
    for (int i = 0; i < w->num_layers; ++i) if ((rc = check_ga_weights(&w->blocks[i]))) return rc;
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(v_t && p_t && s_t && res_feat && pair_feat && beta && mask_generate && mask_res && v_next && R_next && eps_pos && c_denoised && ws,
                    "eps_net_forward: NULL argument");
    const bool has_prmsd = w->w_prmsd1 != nullptr;
    ABOPT_CHECK_ARG(!has_prmsd || (prmsd_logits && w->prmsd_ln_gamma && w->prmsd_ln_beta && w->b_prmsd1 && w->w_prmsd2 && w->b_prmsd2 &&
                                   w->w_prmsd3 && w->b_prmsd3 && w->num_bins > 0 && w->num_bins <= 64), "eps_net_forward: incomplete prmsd head");
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)N * L;
    if (M == 0) return ABOPT_OK;
    // pair_feat_shared: 0 distinct | 1 one entry for the whole batch | g > 1 consecutive groups of g samples share an entry
    ABOPT_CHECK_ARG(pair_feat_shared >= 0 && (pair_feat_shared <= 1 || N % pair_feat_shared == 0), "eps_net_forward: pair_feat_shared=%d does not divide N=%d", pair_feat_shared, N);
    ABOPT_CHECK_ARG(!pair_feat_shared || pair_bias_cache, "eps_net_forward: a shared pair_feat comes with its pair-bias cache");
    ABOPT_CHECK_ARG(!pair_terms || pair_bias_cache, "eps_net_forward: pair_terms come with the pair-bias cache of the same pair_feat");
    const int zg = pair_feat_shared == 1 ? N : pair_feat_shared;
    Carver cv(ws, ws_bytes);
    EpsScratch e = carve_eps(cv, M, N, L, 64);
    if (!cv.ok) { set_error("eps_net_forward: workspace too small (%zu bytes given, %zu needed)", ws_bytes, abopt_eps_workspace_bytes(N, L, Fd, Cd)); return ABOPT_EWORKSPACE; }

    // dpm_full.py:86  R = exp(v_t)   and   dpm_full.py:89  res_feat_mixer([res_feat | Embedding(s_t)])
    float* x0_terms = nullptr;
    if (w->w_mix_frag && w->mix_table) {
        const bool xt0 = w->num_layers > 0 && w->blocks[0].w_node_frag && !(getenv("ABOPT_X_TERMS") && getenv("ABOPT_X_TERMS")[0] == '0');
        x0_terms = xt0 ? e.ga.xt[1] : nullptr;                                 // (block 0 writes xt[0], block 1 xt[1], ...: the mixer's copy is read before it is overwritten)
        if ((rc = launch_mixer(res_feat, s_t, w->w_mix_frag, w->mix_table, w->b_mix1, e.cat, M, st, v_t, e.R, x0_terms))) return rc;       // one launch for both
    } else {
        if ((rc = launch_so3_exp(v_t, e.R, M, st))) return rc;
        if ((rc = launch_embed_concat(res_feat, s_t, w->seq_embed, e.cat, M, st))) return rc;
        if ((rc = launch_linear(e.cat, 2 * F, w->w_mix0, 2 * F, w->b_mix0, e.x0, F, (int)M, F, 2 * F, true, st))) return rc;
        if ((rc = launch_linear(e.x0, F, w->w_mix1, F, w->b_mix1, e.cat, F, (int)M, F, F, false, st))) return rc;   // reuse cat[:, :F] as x (ld = F)
    }
    // dpm_full.py:90  encoder
    if ((rc = ga_encoder(w->blocks, w->num_layers, e.R, p_t, e.cat, pair_feat, mask_res, e.xe, N, L, e.ga, e.pong, st, pair_bias_cache, zg, pair_terms, x0_terms))) return rc;
    bool heads_fused = false;
    if (w->w_heads_frag) {
        // dpm_full.py:92-101: time features + the three heads in one launch (heads.hip)
        if (has_prmsd && (rc = launch_build_infeat(e.xe, beta, e.infeat, w->prmsd_ln_gamma, w->prmsd_ln_beta, e.infeat_ln, N, L, st))) return rc;
        // ... and, unless ABOPT_FUSE_HEADS=0 (A/B, tests), their geometric epilogue as the tail of the same kernel
        const char* fh = getenv("ABOPT_FUSE_HEADS");
        heads_fused = !(fh && fh[0] == '0');
        const HeadsEpilogue hep{e.R, v_t, mask_generate, v_next, R_next, eps_pos, c_denoised, grad_mode, nonfinite_flag_ptr()};
        if ((rc = launch_heads_mlp(e.xe, beta, w->w_heads_frag, w->w_head1, FI, w->b_head1, w->b_crd2, w->b_rot2, w->b_seq2, w->b_crd3, w->b_rot3,
                                   w->b_seq3, e.out3, M, L, st, heads_fused ? &hep : nullptr))) return rc;
    } else {
    // dpm_full.py:92-93 time features
    if ((rc = launch_build_infeat(e.xe, beta, e.infeat, w->prmsd_ln_gamma, w->prmsd_ln_beta, has_prmsd ? e.infeat_ln : nullptr, N, L, st))) return rc;
    // three heads, first layers fused (shared input): [M,132] x [384,132]^T
    if ((rc = launch_linear(e.infeat, FI, w->w_head1, FI, w->b_head1, e.hh1, 3 * F, (int)M, 3 * F, FI, true, st))) return rc;
    if ((rc = launch_linear(e.hh1 + 0 * F, 3 * F, w->w_crd2, F, w->b_crd2, e.hh2 + 0 * F, 3 * F, (int)M, F, F, true, st))) return rc;
    if ((rc = launch_linear(e.hh1 + 1 * F, 3 * F, w->w_rot2, F, w->b_rot2, e.hh2 + 1 * F, 3 * F, (int)M, F, F, true, st))) return rc;
    if ((rc = launch_linear(e.hh1 + 2 * F, 3 * F, w->w_seq2, F, w->b_seq2, e.hh2 + 2 * F, 3 * F, (int)M, F, F, true, st))) return rc;
    // third layers into out3 [M,32]: cols 0..2 crd, 4..6 rot, 8..27 seq logits
    if ((rc = launch_linear(e.hh2 + 0 * F, 3 * F, w->w_crd3, F, w->b_crd3, e.out3 + 0, 32, (int)M, 3, F, false, st))) return rc;
    if ((rc = launch_linear(e.hh2 + 1 * F, 3 * F, w->w_rot3, F, w->b_rot3, e.out3 + 4, 32, (int)M, 3, F, false, st))) return rc;
    if ((rc = launch_linear(e.hh2 + 2 * F, 3 * F, w->w_seq3, F, w->b_seq3, e.out3 + 8, 32, (int)M, ABOPT_AA, F, false, st))) return rc;
    }
    if (!heads_fused && (rc = launch_heads_epilogue(e.R, v_t, e.out3 + 0, e.out3 + 4, e.out3 + 8, 32, 32, mask_generate, v_next, R_next, eps_pos, c_denoised,
                                                    M, grad_mode, st))) return rc;
    if (has_prmsd) {
        // PerResiduePredictor (nn.py:179-188) then mean over L (dpm_full.py:109-110)
        if ((rc = launch_linear(e.infeat_ln, FI, w->w_prmsd1, FI, w->b_prmsd1, e.pr1, F, (int)M, F, FI, true, st))) return rc;
        if ((rc = launch_linear(e.pr1, F, w->w_prmsd2, F, w->b_prmsd2, e.pr2, F, (int)M, F, F, true, st))) return rc;
        if ((rc = launch_linear(e.pr2, F, w->w_prmsd3, F, w->b_prmsd3, e.pr3, w->num_bins, (int)M, w->num_bins, F, false, st))) return rc;
        if ((rc = launch_mean_over_L(e.pr3, prmsd_logits, N, L, w->num_bins, st))) return rc;
    }
    return ABOPT_OK;
}

// Per-residue ("row") kernels around the dense projections: SO(3) maps, fused residual+LayerNorm,
// the EpsilonNet prologue (sequence-embedding gather + concat, time features) and its geometric epilogue.
// Reference: AbDock/src/modules/common/{so3.py,layers.py:109-155}, AbDock/src/modules/diffusion/dpm_full.py:85-112.
#include "abopt_common.h"
#include "kernels.h"

namespace abopt {

constexpr int F = 128;

__global__ void so3_exp_kernel(const float* __restrict__ w, float* __restrict__ R, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Mat3 m = so3_exp(w[i * 3], w[i * 3 + 1], w[i * 3 + 2]);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[i * 9 + k] = m.m[k];
}

__global__ void so3_log_kernel(const float* __restrict__ R, float* __restrict__ w, int64_t n, int grad_mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Mat3 m;
#pragma unroll
    for (int k = 0; k < 9; ++k) m.m[k] = R[i * 9 + k];
    const Vec3 v = so3_log(m, grad_mode != 0);
    w[i * 3] = v.x; w[i * 3 + 1] = v.y; w[i * 3 + 2] = v.z;
}

int launch_so3_exp(const float* w, float* R, int64_t n, hipStream_t st) {
    if (n == 0) return ABOPT_OK;
    hipLaunchKernelGGL(so3_exp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, R, n);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}
int launch_so3_log(const float* R, float* w, int64_t n, int grad_mode, hipStream_t st) {
    if (n == 0) return ABOPT_OK;
    hipLaunchKernelGGL(so3_log_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, R, w, n, grad_mode);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// cat = [res_feat | Embedding(s_t)]  (dpm_full.py:89)
__global__ __launch_bounds__(256) void embed_concat_kernel(const float* __restrict__ res_feat, const int64_t* __restrict__ s_t,
                                                           const float* __restrict__ embed, float* __restrict__ cat, int64_t rows) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t s = s_t[row];
    const float2 a = reinterpret_cast<const float2*>(res_feat + row * F)[lane];
    // nn.Embedding(25, F) raises on an index outside [0, 25) (dpm_full.py:89); a kernel cannot raise, so the row is poisoned with
    // NaN instead of reading out of bounds: every output of the step then comes back NaN (never a silently wrong number)
    const bool ok = s >= 0 && s < 25;
    const float qnan = __int_as_float(0x7fc00000);
    const float2 b = ok ? reinterpret_cast<const float2*>(embed + s * F)[lane] : make_float2(qnan, qnan);
    reinterpret_cast<float2*>(cat + row * 2 * F)[lane] = a;
    reinterpret_cast<float2*>(cat + row * 2 * F + F)[lane] = b;
}

int launch_embed_concat(const float* res_feat, const int64_t* s_t, const float* embed, float* cat, int64_t rows, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(embed_concat_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, res_feat, s_t, embed, cat, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// in_feat = [x | beta, sin beta, cos beta | 0] (dpm_full.py:92-93), K padded 131 -> 132; optionally also its
// LayerNorm over the 131 real features for the prmsd head (nn.py:179-180).
constexpr int FI = F + 4;
__global__ __launch_bounds__(256) void build_infeat_kernel(const float* __restrict__ x, const float* __restrict__ beta,
                                                           float* __restrict__ infeat, const float* __restrict__ ln_gamma,
                                                           const float* __restrict__ ln_beta, float* __restrict__ infeat_ln, int N, int L) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)N * L) return;
    const int lane = threadIdx.x & 63;
    const float b = beta[row / L];
    const float2 xv = reinterpret_cast<const float2*>(x + row * F)[lane];
    const float e0 = b, e1 = sinf(b), e2 = cosf(b);
    float* o = infeat + row * FI;
    reinterpret_cast<float2*>(o)[lane] = xv;
    if (lane == 0) { o[F] = e0; o[F + 1] = e1; o[F + 2] = e2; o[F + 3] = 0.f; }
    if (infeat_ln) {
        constexpr float inv = 1.f / (F + 3);
        const float mean = (wave_sum(xv.x + xv.y) + e0 + e1 + e2) * inv;
        const float da = xv.x - mean, db = xv.y - mean, d0 = e0 - mean, d1 = e1 - mean, d2 = e2 - mean;
        const float var = (wave_sum(da * da + db * db) + d0 * d0 + d1 * d1 + d2 * d2) * inv;
        const float sd = sqrtf(var + 1e-10f);
        float* ol = infeat_ln + row * FI;
        const float2 g = reinterpret_cast<const float2*>(ln_gamma)[lane], bt = reinterpret_cast<const float2*>(ln_beta)[lane];
        reinterpret_cast<float2*>(ol)[lane] = make_float2(da / sd * g.x + bt.x, db / sd * g.y + bt.y);
        if (lane == 0) {
            ol[F] = d0 / sd * ln_gamma[F] + ln_beta[F];
            ol[F + 1] = d1 / sd * ln_gamma[F + 1] + ln_beta[F + 1];
            ol[F + 2] = d2 / sd * ln_gamma[F + 2] + ln_beta[F + 2];
            ol[F + 3] = 0.f;
        }
    }
}

int launch_build_infeat(const float* x, const float* beta, float* infeat, const float* ln_gamma, const float* ln_beta,
                        float* infeat_ln, int N, int L, hipStream_t st) {
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(build_infeat_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, beta, infeat, ln_gamma, ln_beta, infeat_ln, N, L);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// Geometric epilogue of the three heads (dpm_full.py:95-107): eps_pos = gen ? R eps_crd : 0;
// R_next = R * U(eps_rot); v_next = gen ? log(R_next) : v_t; c = softmax(seq logits).
__global__ __launch_bounds__(256) void heads_epilogue_kernel(const float* __restrict__ R, const float* __restrict__ v_t,
                                                             const float* __restrict__ eps_crd, const float* __restrict__ eps_rot,
                                                             const float* __restrict__ seq_logits, int ld3, int ldseq,
                                                             const uint8_t* __restrict__ mask_generate, float* __restrict__ v_next,
                                                             float* __restrict__ R_next, float* __restrict__ eps_pos,
                                                             float* __restrict__ c_den, int64_t rows, int grad_mode, unsigned* __restrict__ nonfinite) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    heads_epilogue_row(i, R, v_t, eps_crd + i * ld3, eps_rot + i * ld3, seq_logits ? seq_logits + i * ldseq : nullptr, mask_generate, v_next, R_next, eps_pos, c_den,
                       grad_mode, nonfinite);
}

__device__ unsigned g_nonfinite_flag;
unsigned* nonfinite_flag_ptr() {
    static unsigned* p = nullptr;
    if (!p && hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_nonfinite_flag)) != hipSuccess) p = nullptr;
    return p;
}
int nonfinite_flag_read(int reset, hipStream_t st, int* flag) {
    unsigned* p = nonfinite_flag_ptr();
    ABOPT_CHECK_ARG(p != nullptr, "nonfinite_flag: no device symbol");
    unsigned h = 0;
    ABOPT_HIP(hipMemcpyAsync(&h, p, sizeof(h), hipMemcpyDeviceToHost, st));
    ABOPT_HIP(hipStreamSynchronize(st));
    if (reset && h) ABOPT_HIP(hipMemsetAsync(p, 0, sizeof(h), st));
    *flag = h ? 1 : 0;
    return ABOPT_OK;
}

int launch_heads_epilogue(const float* R, const float* v_t, const float* eps_crd, const float* eps_rot, const float* seq_logits,
                          int ld3, int ldseq, const uint8_t* mask_generate, float* v_next, float* R_next, float* eps_pos, float* c_den,
                          int64_t rows, int grad_mode, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(heads_epilogue_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, R, v_t, eps_crd, eps_rot, seq_logits,
                       ld3, ldseq, mask_generate, v_next, R_next, eps_pos, c_den, rows, grad_mode, seq_logits ? nonfinite_flag_ptr() : nullptr);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// Backward of the geometric epilogue for the training path (dpm_full.py:95-101 under autograd): from d R_next and d eps_pos to the
// gradients of the two 3-vectors the heads produce.  R_next = R U(e), U the rotation of the quaternion (1 + b i + c j + d k) / s:
//   dL/dU = R^T dL/dR_next;  dL/dq_k = sum_ij (dL/dU)_ij dU_ij/dq_k (U is quadratic in the unit quaternion q = (1, b, c, d) / s);
//   dL/d(b, c, d) = J^T dL/dq with dq_0/dx = -x / s^3, dq_x/dx = 1/s - x^2 / s^3, dq_x/dy = -x y / s^3.
// eps_pos = gen ? R eps_crd : 0  ->  d eps_crd = gen ? R^T d eps_pos : 0.
__global__ __launch_bounds__(256) void heads_epilogue_backward_kernel(const float* __restrict__ R, const float* __restrict__ eps_rot, int ld3,
                                                                      const uint8_t* __restrict__ mask_generate, const float* __restrict__ dR_next,
                                                                      const float* __restrict__ deps_pos, float* __restrict__ deps_crd,
                                                                      float* __restrict__ deps_rot, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    float Rm[9], G[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { Rm[k] = R[i * 9 + k]; G[k] = dR_next ? dR_next[i * 9 + k] : 0.f; }
    const bool gen = mask_generate[i] != 0;
    const float px = deps_pos ? deps_pos[i * 3] : 0.f, py = deps_pos ? deps_pos[i * 3 + 1] : 0.f, pz = deps_pos ? deps_pos[i * 3 + 2] : 0.f;
    deps_crd[i * 3 + 0] = gen ? (Rm[0] * px + Rm[3] * py + Rm[6] * pz) : 0.f;
    deps_crd[i * 3 + 1] = gen ? (Rm[1] * px + Rm[4] * py + Rm[7] * pz) : 0.f;
    deps_crd[i * 3 + 2] = gen ? (Rm[2] * px + Rm[5] * py + Rm[8] * pz) : 0.f;
    float U[9];                                                         // dL/dU = R^T G
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) U[r * 3 + c] = Rm[r] * G[c] + Rm[3 + r] * G[3 + c] + Rm[6 + r] * G[6 + c];
    const float qb = eps_rot[i * ld3], qc = eps_rot[i * ld3 + 1], qd = eps_rot[i * ld3 + 2];
    const float s = sqrtf(1.f + qb * qb + qc * qc + qd * qd), is = 1.f / s, is3 = is * is * is;
    const float a = is, b = qb * is, c = qc * is, d = qd * is;
    // dL/dq: the four derivative matrices of quat1ijk_to_rot's entries contracted with dL/dU
    const float ga = 2.f * (a * U[0] - d * U[1] + c * U[2] + d * U[3] + a * U[4] - b * U[5] - c * U[6] + b * U[7] + a * U[8]);
    const float gb = 2.f * (b * U[0] + c * U[1] + d * U[2] + c * U[3] - b * U[4] - a * U[5] + d * U[6] + a * U[7] - b * U[8]);
    const float gc = 2.f * (-c * U[0] + b * U[1] + a * U[2] + b * U[3] + c * U[4] + d * U[5] - a * U[6] + d * U[7] - c * U[8]);
    const float gd = 2.f * (-d * U[0] - a * U[1] + b * U[2] + a * U[3] - d * U[4] + c * U[5] + b * U[6] + c * U[7] + d * U[8]);
    // q = (1, qb, qc, qd) / s:  dL/dx = -x / s^3 (ga + qb gb + qc gc + qd gd) + g_x / s
    const float dot = ga + qb * gb + qc * gc + qd * gd;
    deps_rot[i * 3 + 0] = gb * is - qb * is3 * dot;
    deps_rot[i * 3 + 1] = gc * is - qc * is3 * dot;
    deps_rot[i * 3 + 2] = gd * is - qd * is3 * dot;
}

int launch_heads_epilogue_backward(const float* R, const float* eps_rot, int ld3, const uint8_t* mask_generate, const float* dR_next, const float* deps_pos,
                                   float* deps_crd, float* deps_rot, int64_t rows, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(heads_epilogue_backward_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, R, eps_rot, ld3, mask_generate, dR_next, deps_pos,
                       deps_crd, deps_rot, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// The three per-residue losses of FullDPM.forward (D/modules/diffusion/dpm_full.py:199-231; A: 156-190) and their gradients in one pass:
//   rot  sum over the 3 columns of 1 - cos(R_pred[:, k], R_0[:, k])     (rotation_matrix_cosine_loss: F.cosine_embedding_loss, eps 1e-12 on the squared norms)
//   pos  |p_pred - target|^2                                             (F.mse_loss(...).sum(-1))
//   seq  KL(posterior(s_t, s_0) || posterior(s_t, c_den))                (F.kl_div(log(post_pred + 1e-8), post_true).sum(-1); transition.py:217-229: alpha_bar_t in both factors)
// each summed over the generated residues.  part[block][3] receives the block's sums (added on the host side in a fixed order and divided
// by sum(mask_generate) + 1e-8); gR / gp / gc receive d(sum)/d(input) of the row (zero outside the mask) -- the caller scales them by the
// upstream gradient / denominator.  One thread per residue.
__global__ __launch_bounds__(256) void dpm_losses_kernel(const float* __restrict__ R_pred, const float* __restrict__ R_0, const float* __restrict__ p_pred,
                                                         const float* __restrict__ p_target, const float* __restrict__ c_den, const int64_t* __restrict__ s_t,
                                                         const int64_t* __restrict__ s_0, const float* __restrict__ abar /* [N] alpha_bar_t */,
                                                         const uint8_t* __restrict__ mask_generate, int L, int64_t rows, float* __restrict__ part,
                                                         float* __restrict__ gR, float* __restrict__ gp, float* __restrict__ gc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float l_rot = 0.f, l_pos = 0.f, l_seq = 0.f;
    if (i < rows) {
        const bool gen = mask_generate[i] != 0;
        float a[9], b[9], g[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { a[k] = R_pred[i * 9 + k]; b[k] = R_0[i * 9 + k]; g[k] = 0.f; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {                                           // column c of both matrices
            const float x0 = a[c], x1 = a[3 + c], x2 = a[6 + c], y0 = b[c], y1 = b[3 + c], y2 = b[6 + c];
            const float prod = x0 * y0 + x1 * y1 + x2 * y2;
            const float m1 = x0 * x0 + x1 * x1 + x2 * x2 + 1e-12f, m2 = y0 * y0 + y1 * y1 + y2 * y2 + 1e-12f;
            const float den = sqrtf(m1 * m2), cs = prod / den;
            l_rot += 1.f - cs;
            // d(1 - cos)/dx = -(y / den - cos x / m1)
            g[c] = -(y0 / den - cs * x0 / m1); g[3 + c] = -(y1 / den - cs * x1 / m1); g[6 + c] = -(y2 / den - cs * x2 / m1);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) gR[i * 9 + k] = gen ? g[k] : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = p_pred[i * 3 + k] - p_target[i * 3 + k];
            l_pos += d * d;
            gp[i * 3 + k] = gen ? 2.f * d : 0.f;
        }
        // sequence: th_k = (ab ct_k + u)(ab c0_k + u), u = (1 - ab) / 20; post = th / (sum th + 1e-8)
        const float ab = abar[i / L], u = (1.f - ab) / (float)ABOPT_AA;
        const int st = (int)s_t[i], s0 = (int)s_0[i];
        float fct[ABOPT_AA], thp[ABOPT_AA], tht[ABOPT_AA], sp = 0.f, stt = 0.f;
#pragma unroll
        for (int k = 0; k < ABOPT_AA; ++k) {
            fct[k] = ab * ((k == st) ? 1.f : 0.f) + u;                           // _one_hot20: indices outside 0..19 give a zero row
            tht[k] = fct[k] * (ab * ((k == s0) ? 1.f : 0.f) + u);
            thp[k] = fct[k] * (ab * c_den[i * ABOPT_AA + k] + u);
            stt += tht[k]; sp += thp[k];
        }
        const float zt = stt + 1e-8f, zp = sp + 1e-8f;
        float dth[ABOPT_AA], dot = 0.f;                                          // d kl / d post_pred_k = -post_true_k / (post_pred_k + 1e-8)
#pragma unroll
        for (int k = 0; k < ABOPT_AA; ++k) {
            const float pt = tht[k] / zt, pp = thp[k] / zp;
            l_seq += (pt > 0.f ? pt * logf(pt) : 0.f) - pt * logf(pp + 1e-8f);   // xlogy(t, t) - t * input
            dth[k] = -pt / (pp + 1e-8f);
            dot += dth[k] * pp;
        }
#pragma unroll
        for (int k = 0; k < ABOPT_AA; ++k)                                       // post = th / z: d/d th_k = (dpost_k - sum_j dpost_j post_j) / z;  th_k = fct_k (ab c0_k + u)
            gc[i * ABOPT_AA + k] = gen ? (dth[k] - dot) / zp * fct[k] * ab : 0.f;
        if (!gen) { l_rot = 0.f; l_pos = 0.f; l_seq = 0.f; }
    }
    __shared__ float red[3][4];
    l_rot = wave_sum(l_rot); l_pos = wave_sum(l_pos); l_seq = wave_sum(l_seq);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = l_rot; red[1][threadIdx.x >> 6] = l_pos; red[2][threadIdx.x >> 6] = l_seq; }
    __syncthreads();
    if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

int launch_dpm_losses(const float* R_pred, const float* R_0, const float* p_pred, const float* p_target, const float* c_den, const int64_t* s_t, const int64_t* s_0,
                      const float* abar, const uint8_t* mask_generate, int N, int L, float* part, float* gR, float* gp, float* gc, hipStream_t st) {
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(dpm_losses_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, R_pred, R_0, p_pred, p_target, c_den, s_t, s_0, abar, mask_generate, L,
                       rows, part, gR, gp, gc);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The two losses only the AbDock flavour has (D/modules/diffusion/dpm_full.py:180-198, D/modules/common/prmsd.py:49-70, dpm_full.py:369-378),
// one workgroup per sample:
//   prmsd: rmsd_n = sqrt(sum_gen |scale (pred_p0 - p0)|^2 / #gen)   (detached) -> bin = argmin_b |rmsd - offset_b|,
//          err_n = -log_softmax(prmsd_logits_n)[bin];   loss = sum_n err_n m0_n / (sum_n m0_n + 1e-10),  m0_n = mask_generate[n, 0]
//   dist (obj = pred_x0): smooth_l1(|p_pred_i - p_pred_j| - |p_true_i - p_true_j|) over the pairs gen_i & mres_i & mres_j (i == j included), mean.
// part[n] = {err_n, m0_n, sum of the sample's smooth-l1 terms, their count}; glogit [N, nb] = softmax - onehot (to be scaled by
// m0_n / (sum m0 + 1e-10)); gp [N, L, 3] = d(sum of smooth-l1 terms) / d p_pred (to be scaled by 1 / total count).  pred_x0 = 0: pred_p0 =
// gen ? a_n p0 - b_n p_pred : p0 (transition.py:52-60) and there is no dist loss (gp is zeroed).
__global__ __launch_bounds__(256) void abdock_losses_kernel(const float* __restrict__ prmsd_logits, const float* __restrict__ p_pred, const float* __restrict__ p0n,
                                                            const float* __restrict__ coef_a, const float* __restrict__ coef_b, const uint8_t* __restrict__ gen,
                                                            const uint8_t* __restrict__ mres, const float* __restrict__ offsets, int nb, int L, float scale,
                                                            int pred_x0, float* __restrict__ part, float* __restrict__ glogit, float* __restrict__ gp) {
    extern __shared__ __attribute__((aligned(16))) float al_s[];           // [L][4] p_pred (x, y, z, gen & mres flags) | [L][4] p_true
    float* pp = al_s;
    float* pt = al_s + 4 * L;
    __shared__ float red[2][4];
    __shared__ float s_rmsd;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (int64_t)n * L;
    const float ca = pred_x0 ? 0.f : coef_a[n], cb = pred_x0 ? 0.f : coef_b[n];
    float sq = 0.f, cnt = 0.f;
    for (int l = tid; l < L; l += 256) {
        const bool g = gen[base + l] != 0, m = mres[base + l] != 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = p_pred[(base + l) * 3 + k], b = p0n[(base + l) * 3 + k];
            pp[4 * l + k] = a; pt[4 * l + k] = b;
            const float d = (pred_x0 ? a - b : (ca * b - cb * a) - b) * scale;
            if (g) sq += d * d;
        }
        pp[4 * l + 3] = (g ? 1.f : 0.f) + (m ? 2.f : 0.f);
        if (g) cnt += 1.f;
    }
    sq = wave_sum(sq); cnt = wave_sum(cnt);
    if ((tid & 63) == 0) { red[0][tid >> 6] = sq; red[1][tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) s_rmsd = sqrtf(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])));
    __syncthreads();
    // ---- prmsd cross entropy (wave 0): nb <= 64 bins, one lane per bin
    if (tid < 64) {
        const float rm = s_rmsd;
        const float x = (tid < nb) ? prmsd_logits[(int64_t)n * nb + tid] : -INFINITY;
        const float dist = (tid < nb) ? fabsf(rm - offsets[tid]) : INFINITY;
        // argmin with the lowest index on ties (torch.argmin)
        float best = dist; int bi = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const float mx = wave_max(x);
        const float e = (tid < nb) ? expf(x - mx) : 0.f;
        const float se = wave_sum(e);
        const float lse = mx + logf(se);
        const float xb = __shfl(x, bi);
        if (tid < nb) glogit[(int64_t)n * nb + tid] = e / se - (tid == bi ? 1.f : 0.f);
        if (tid == 0) { part[n * 4 + 0] = -(xb - lse); part[n * 4 + 1] = gen[base] ? 1.f : 0.f; }      // (a NaN rmsd -- no generated residue -- selects bin 0, as argmin does)
    }
    // ---- dist loss: thread k owns residue k: its row terms (i = k) and its column terms (j = k)
    float lsum = 0.f, lcnt = 0.f;
    for (int k = tid; k < L; k += 256) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (pred_x0) {
            const float kx = pp[4 * k], ky = pp[4 * k + 1], kz = pp[4 * k + 2], tx = pt[4 * k], ty = pt[4 * k + 1], tz = pt[4 * k + 2];
            const int fk = (int)pp[4 * k + 3];
            const bool row_k = (fk & 1) && (fk & 2);                          // gen_k & mres_k: k is a row of the selection
            for (int j = 0; j < L; ++j) {
                const int fj = (int)pp[4 * j + 3];
                const bool as_row = row_k && (fj & 2), as_col = (fj & 1) && (fj & 2) && (fk & 2);      // (k, j) selected | (j, k) selected
                if (!as_row && !as_col) continue;
                const float dx = kx - pp[4 * j], dy = ky - pp[4 * j + 1], dz = kz - pp[4 * j + 2];
                const float ex = tx - pt[4 * j], ey = ty - pt[4 * j + 1], ez = tz - pt[4 * j + 2];
                const float dp = sqrtf(dx * dx + dy * dy + dz * dz), dt = sqrtf(ex * ex + ey * ey + ez * ez);
                const float x = dp - dt, ax = fabsf(x);
                const float sl = ax < 1.f ? 0.5f * x * x : ax - 0.5f, ds = ax < 1.f ? x : (x > 0.f ? 1.f : -1.f);
                if (as_row) { lsum += sl; lcnt += 1.f; }
                const float w = (dp > 0.f ? ds / dp : 0.f) * ((as_row ? 1.f : 0.f) + (as_col ? 1.f : 0.f));      // cdist backward: zero at zero distance
                gx += w * dx; gy += w * dy; gz += w * dz;
            }
        }
        gp[(base + k) * 3] = gx; gp[(base + k) * 3 + 1] = gy; gp[(base + k) * 3 + 2] = gz;
    }
    __syncthreads();
    lsum = wave_sum(lsum); lcnt = wave_sum(lcnt);
    if ((tid & 63) == 0) { red[0][tid >> 6] = lsum; red[1][tid >> 6] = lcnt; }
    __syncthreads();
    if (tid == 0) { part[n * 4 + 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]); part[n * 4 + 3] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]); }
}

int launch_abdock_losses(const float* prmsd_logits, const float* p_pred, const float* p0n, const float* coef_a, const float* coef_b, const uint8_t* gen,
                         const uint8_t* mres, const float* offsets, int nb, int N, int L, float scale, int pred_x0, float* part, float* glogit, float* gp,
                         hipStream_t st) {
    if (N == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(nb >= 1 && nb <= 64 && L >= 1 && (size_t)L * 32 <= 160 * 1024, "abdock_losses: num_bins=%d (1..64), L=%d (at most 5120)", nb, L);
    static LdsConfig lds_cfg;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(abdock_losses_kernel), (size_t)L * 32, lds_cfg)) return rc;
    hipLaunchKernelGGL(abdock_losses_kernel, dim3(N), dim3(256), (size_t)L * 32, st, prmsd_logits, p_pred, p0n, coef_a, coef_b, gen, mres, offsets, nb, L, scale, pred_x0,
                       part, glogit, gp);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// LayerNorm of the reference's own definition over rows of `cols` (<= 256) values (D/modules/common/layers.py:146-155: biased variance,
// sqrt(var + eps)); forward keeps xhat = (x - mean) / sd and 1 / sd for the backward:
//   dx = (g dy - mean(g dy) - xhat mean(g dy xhat)) / sd;   d gamma = sum_rows dy xhat, d beta = sum_rows dy (left to abopt_colsum on dyx / dy).
// One wave per row.
__global__ __launch_bounds__(256) void row_layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, int cols,
                                                             float eps, int64_t rows, float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float v[4], s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int c = lane + 64 * q; v[q] = c < cols ? x[r * cols + c] : 0.f; s += v[q]; }
    const float mean = wave_sum(s) / (float)cols;
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int c = lane + 64 * q; v[q] = c < cols ? v[q] - mean : 0.f; s2 += v[q] * v[q]; }
    const float sd = sqrtf(wave_sum(s2) / (float)cols + eps);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        if (c < cols) { const float h = v[q] / sd; y[r * cols + c] = h * gamma[c] + beta[c]; if (xhat) xhat[r * cols + c] = h; }
    }
    if (rstd && lane == 0) rstd[r] = 1.f / sd;
}
__global__ __launch_bounds__(256) void row_layer_norm_backward_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                                      const float* __restrict__ gamma, int cols, int64_t rows, float* __restrict__ dx,
                                                                      float* __restrict__ dyx) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float g[4], h[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        const float d = c < cols ? dy[r * cols + c] : 0.f;
        h[q] = c < cols ? xhat[r * cols + c] : 0.f;
        g[q] = c < cols ? d * gamma[c] : 0.f;
        s1 += g[q]; s2 += g[q] * h[q];
        if (c < cols) dyx[r * cols + c] = d * h[q];
    }
    const float m1 = wave_sum(s1) / (float)cols, m2 = wave_sum(s2) / (float)cols, rs = rstd[r];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int c = lane + 64 * q; if (c < cols) dx[r * cols + c] = (g[q] - m1 - h[q] * m2) * rs; }
}
int launch_row_layer_norm(const float* x, const float* gamma, const float* beta, int cols, float eps, int64_t rows, float* y, float* xhat, float* rstd, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(cols >= 1 && cols <= 256, "layer_norm: %d columns (1..256)", cols);
    hipLaunchKernelGGL(row_layer_norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, gamma, beta, cols, eps, rows, y, xhat, rstd);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}
int launch_row_layer_norm_backward(const float* dy, const float* xhat, const float* rstd, const float* gamma, int cols, int64_t rows, float* dx, float* dyx, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(cols >= 1 && cols <= 256, "layer_norm_backward: %d columns (1..256)", cols);
    hipLaunchKernelGGL(row_layer_norm_backward_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dy, xhat, rstd, gamma, cols, rows, dx, dyx);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// prmsd_logits.mean(dim=1) over ALL L rows incl. padding (dpm_full.py:110).  One workgroup per sample: thread (bin b, slice p of 16)
// sums rows p, p + 16, ... with the loads of four rows in flight, the slices meet in LDS (fixed order: deterministic).
__global__ __launch_bounds__(1024) void mean_over_L_kernel(const float* __restrict__ in, float* __restrict__ out, int L, int B) {
    __shared__ float part[16][64];
    const int n = blockIdx.x, b = threadIdx.x & 63, p = threadIdx.x >> 6;
    const float* src = in + (int64_t)n * L * B + b;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (b < B) {
        int l = p;
        for (; l + 48 < L; l += 64) {
            s0 += src[(int64_t)l * B]; s1 += src[(int64_t)(l + 16) * B]; s2 += src[(int64_t)(l + 32) * B]; s3 += src[(int64_t)(l + 48) * B];
        }
        for (; l < L; l += 16) s0 += src[(int64_t)l * B];
    }
    part[p][b] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (p == 0 && b < B) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += part[q][b];
        out[(int64_t)n * B + b] = s / (float)L;
    }
}

int launch_mean_over_L(const float* in, float* out, int N, int L, int B, hipStream_t st) {
    if (N == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(B <= 64, "mean_over_L: B=%d too large (at most 64 bins)", B);
    hipLaunchKernelGGL(mean_over_L_kernel, dim3(N), dim3(1024), 0, st, in, out, L, B);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

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
                                                             float* __restrict__ c_den, int64_t rows, int grad_mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const bool gen = mask_generate[i] != 0;
    Mat3 Rm;
#pragma unroll
    for (int k = 0; k < 9; ++k) Rm.m[k] = R[i * 9 + k];
    const float cx = eps_crd[i * ld3], cy = eps_crd[i * ld3 + 1], cz = eps_crd[i * ld3 + 2];
    // apply_rotation_to_vector = R p + 0 (geometry.py:116-117)
    eps_pos[i * 3 + 0] = gen ? (Rm.m[0] * cx + Rm.m[1] * cy + Rm.m[2] * cz + 0.f) : 0.f;
    eps_pos[i * 3 + 1] = gen ? (Rm.m[3] * cx + Rm.m[4] * cy + Rm.m[5] * cz + 0.f) : 0.f;
    eps_pos[i * 3 + 2] = gen ? (Rm.m[6] * cx + Rm.m[7] * cy + Rm.m[8] * cz + 0.f) : 0.f;
    const Mat3 U = quat1ijk_to_rot(eps_rot[i * ld3], eps_rot[i * ld3 + 1], eps_rot[i * ld3 + 2]);
    const Mat3 Rn = matmul3(Rm, U);
#pragma unroll
    for (int k = 0; k < 9; ++k) R_next[i * 9 + k] = Rn.m[k];
    if (v_next) {
        const Vec3 w = so3_log(Rn, grad_mode != 0);
        v_next[i * 3 + 0] = gen ? w.x : v_t[i * 3 + 0];
        v_next[i * 3 + 1] = gen ? w.y : v_t[i * 3 + 1];
        v_next[i * 3 + 2] = gen ? w.z : v_t[i * 3 + 2];
    }
    if (!seq_logits) return;                                        // training path: the sequence head's softmax stays in the autograd graph
    float lgt[ABOPT_AA], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < ABOPT_AA; ++k) { lgt[k] = seq_logits[i * ldseq + k]; mx = fmaxf(mx, lgt[k]); }
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < ABOPT_AA; ++k) { lgt[k] = expf(lgt[k] - mx); sm += lgt[k]; }
#pragma unroll
    for (int k = 0; k < ABOPT_AA; ++k) c_den[i * ABOPT_AA + k] = lgt[k] / sm;
}

int launch_heads_epilogue(const float* R, const float* v_t, const float* eps_crd, const float* eps_rot, const float* seq_logits,
                          int ld3, int ldseq, const uint8_t* mask_generate, float* v_next, float* R_next, float* eps_pos, float* c_den,
                          int64_t rows, int grad_mode, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(heads_epilogue_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, R, v_t, eps_crd, eps_rot, seq_logits,
                       ld3, ldseq, mask_generate, v_next, R_next, eps_pos, c_den, rows, grad_mode);
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

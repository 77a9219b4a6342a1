// fp32 linear layers on the CDNA4 matrix cores: Y = act(X W^T + b), torch nn.Linear layout (W is [N,K]).
// Uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, same rate as the fp32 VALU peak) so results
// stay inside the 1e-5 parity budget with no reduced-precision path.
//
// Replaces every dense projection on the path: the six node projections of a GABlock
// (reference ga.py:54-66), out_transform + mlp_transition (ga.py:69-79), res_feat_mixer and the
// eps_* / prmsd heads (dpm_full.py:39-65).
#include "abopt_common.h"
#include "kernels.h"

namespace abopt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 64x64 output tile per 256-thread workgroup (4 waves as 2x2, each wave 32x32 = 2x2 MFMA tiles).
// K is consumed 32 at a time through LDS (one 128-byte line per row per tile).  K-permutation: in MFMA step kk lane group kq supplies
// k = kq*4 + kk for both operands, so a fragment is one ds_read_b128 per 4 MFMA steps.
constexpr int GBM = 64, GBN = 64, GBK = 32, GLD = GBK + 4;

template <bool RELU>
__global__ __launch_bounds__(256) void gemm_xwT_kernel(const float* __restrict__ X, int ldx,
                                                       const float* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ Y, int ldy, int M, int N, int K,
                                                       int kchunk, int64_t slab_stride) {
    __shared__ __attribute__((aligned(16))) float Xs[GBM * GLD];
    __shared__ __attribute__((aligned(16))) float Ws[GBN * GLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    // loader: 8 threads cover the 128 contiguous bytes of one row's K tile (a full cache line), rows lr and lr + 32
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    const int fm = lane & 15, kq = lane >> 4;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool xok0 = (m0 + lr) < M, xok1 = (m0 + lr + 32) < M, wok0 = (n0 + lr) < N, wok1 = (n0 + lr + 32) < N;
    const float* xp0 = X + (size_t)(m0 + lr) * ldx + lc;
    const float* xp1 = xp0 + (size_t)32 * ldx;
    const float* wp0 = W + (size_t)(n0 + lr) * ldw + lc;
    const float* wp1 = wp0 + (size_t)32 * ldw;

    // split-K: blockIdx.z owns K range [z*kchunk, (z+1)*kchunk) and writes its partial product to slab z (the consumer sums
    // the slabs in a fixed order, so results stay deterministic -- no atomics)
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    Y += (int64_t)blockIdx.z * slab_stride;
    // software pipeline: the global loads of K tile i+1 are issued before the MFMAs of tile i
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 x0 = zero4, x1 = zero4, w0 = zero4, w1 = zero4;
    if (kbeg < kend) {
        const bool kok = (kbeg + lc) < kend;
        if (xok0 && kok) x0 = *reinterpret_cast<const f32x4*>(xp0 + kbeg);
        if (xok1 && kok) x1 = *reinterpret_cast<const f32x4*>(xp1 + kbeg);
        if (wok0 && kok) w0 = *reinterpret_cast<const f32x4*>(wp0 + kbeg);
        if (wok1 && kok) w1 = *reinterpret_cast<const f32x4*>(wp1 + kbeg);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&Xs[lr * GLD + lc]) = x0;
        *reinterpret_cast<f32x4*>(&Xs[(lr + 32) * GLD + lc]) = x1;
        *reinterpret_cast<f32x4*>(&Ws[lr * GLD + lc]) = w0;
        *reinterpret_cast<f32x4*>(&Ws[(lr + 32) * GLD + lc]) = w1;
        __syncthreads();
        x0 = x1 = w0 = w1 = zero4;
        if (k0 + GBK < kend) {
            const bool kok = (k0 + GBK + lc) < kend;
            if (xok0 && kok) x0 = *reinterpret_cast<const f32x4*>(xp0 + k0 + GBK);
            if (xok1 && kok) x1 = *reinterpret_cast<const f32x4*>(xp1 + k0 + GBK);
            if (wok0 && kok) w0 = *reinterpret_cast<const f32x4*>(wp0 + k0 + GBK);
            if (wok1 && kok) w1 = *reinterpret_cast<const f32x4*>(wp1 + k0 + GBK);
        }
#pragma unroll
        for (int ks = 0; ks < GBK; ks += 16) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&Xs[(wm + i * 16 + fm) * GLD + ks + kq * 4]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Ws[(wn + j * 16 + fm) * GLD + ks + kq * 4]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        // W as the A operand: the accumulator tile is Y^T, so a lane ends up with 4 consecutive output columns
                        // of one row (one 16-byte store instead of four scattered dwords)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][kk], a[i][kk], acc[i][j], 0, 0, 0);
        }
    }
    // C layout of 16x16x4 with swapped operands: output row = m-tile row (lane & 15), output columns = (lane >> 4) * 4 + r.
    const bool vec_ok = (ldy % 4) == 0 && ((uintptr_t)Y % 16) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = m0 + wm + i * 16 + fm;
            const int col = n0 + wn + j * 16 + kq * 4;
            if (row >= M || col >= N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] + ((bias && col + r < N) ? bias[col + r] : 0.f);
                if (RELU) v[r] = fmaxf(v[r], 0.f);
            }
            float* yp = Y + (size_t)row * ldy + col;
            if (vec_ok && col + 3 < N) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) if (col + r < N) yp[r] = v[r];
        }
}

int launch_linear(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy,
                  int M, int N, int K, bool relu, hipStream_t st, int ksplit, int64_t slab_stride) {
    if (M <= 0 || N <= 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(ksplit >= 1 && (ksplit == 1 || (!relu && !bias)), "linear: split-K partials carry neither bias nor activation");
    const int kchunk = ksplit == 1 ? K : ((K + ksplit * GBK - 1) / (ksplit * GBK)) * GBK;
    ABOPT_CHECK_ARG((K % 4) == 0 && (ldx % 4) == 0 && (ldw % 4) == 0, "linear: K/ldx/ldw must be multiples of 4 (K=%d ldx=%d ldw=%d)", K, ldx, ldw);
    ABOPT_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0, "linear: X/W must be 16-byte aligned");
    dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM, ksplit);
    if (relu) hipLaunchKernelGGL(gemm_xwT_kernel<true>, grid, dim3(256), 0, st, X, ldx, W, ldw, bias, Y, ldy, M, N, K, kchunk, slab_stride);
    else      hipLaunchKernelGGL(gemm_xwT_kernel<false>, grid, dim3(256), 0, st, X, ldx, W, ldw, bias, Y, ldy, M, N, K, kchunk, slab_stride);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

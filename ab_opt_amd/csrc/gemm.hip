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
// K is consumed 16 at a time through LDS.  K-permutation: in MFMA step kk lane group kq supplies
// k = kq*4 + kk for both operands, so a fragment is one ds_read_b128 per 4 MFMA steps.
constexpr int GBM = 64, GBN = 64, GBK = 16, GLD = GBK + 4;

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
    const int lr = tid >> 2, lc = (tid & 3) * 4;
    const int fm = lane & 15, kq = lane >> 4;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool xrow_ok = (m0 + lr) < M, wrow_ok = (n0 + lr) < N;
    const float* xp = X + (size_t)(m0 + lr) * ldx + lc;
    const float* wp = W + (size_t)(n0 + lr) * ldw + lc;

    // split-K: blockIdx.z owns K range [z*kchunk, (z+1)*kchunk) and writes its partial product to slab z (the consumer sums
    // the slabs in a fixed order, so results stay deterministic -- no atomics)
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    Y += (int64_t)blockIdx.z * slab_stride;
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), wv = xv;
        const bool kok = (k0 + lc) < kend;
        if (xrow_ok && kok) xv = *reinterpret_cast<const float4*>(xp + k0);
        if (wrow_ok && kok) wv = *reinterpret_cast<const float4*>(wp + k0);
        __syncthreads();
        *reinterpret_cast<float4*>(&Xs[lr * GLD + lc]) = xv;
        *reinterpret_cast<float4*>(&Ws[lr * GLD + lc]) = wv;
        __syncthreads();
        float4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(&Xs[(wm + i * 16 + fm) * GLD + kq * 4]);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const float4*>(&Ws[(wn + j * 16 + fm) * GLD + kq * 4]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float av = kk == 0 ? a[i].x : kk == 1 ? a[i].y : kk == 2 ? a[i].z : a[i].w;
                    const float bv = kk == 0 ? b[j].x : kk == 1 ? b[j].y : kk == 2 ? b[j].z : b[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
                }
        }
    }
    // C layout of 16x16x4: column = lane & 15, row = (lane >> 4) * 4 + r.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + j * 16 + fm;
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + kq * 4 + r;
                if (row < M) {
                    float v = acc[i][j][r] + bv;
                    if (RELU) v = fmaxf(v, 0.f);
                    Y[(size_t)row * ldy + col] = v;
                }
            }
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

// fp32 linear layers on the CDNA4 matrix cores: Y = act(X W^T + b), torch nn.Linear layout (W is [N,K]).
// Uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, same rate as the fp32 VALU peak) so results
// stay inside the 1e-5 parity budget with no reduced-precision path.
//
// Replaces every dense projection on the path: the six node projections of a GABlock
// (reference ga.py:54-66), out_transform + mlp_transition (ga.py:69-79), res_feat_mixer and the
// eps_* / prmsd heads (dpm_full.py:39-65).
#include "ipa_common.h"
#include <cstdlib>
#include <algorithm>
#include "kernels.h"

namespace abopt {

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
                if (RELU) v[r] = relu_nan(v[r]);
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



// =====================================================================================================================
// General strided-batched fp32 GEMM for the TRAINING path (round 3): C[b] = A[b] . B[b]^T with either operand stored k-contiguous
// ("X W^T" form) or k-strided (transposed), exact fp32 on v_mfma_f32_16x16x4_f32.  It replaces the library GEMMs of the denoiser's
// backward: the projection gradients (ga.py:54-66), the four (N,12,L,L)-sized contractions of the IPA backward (training.IpaCore),
// the out_transform / MLP gradients of the block tail (ga.py:174-177), the heads and the mixer.
//   A(m,k) = AT ? A[k*lda + m] : A[m*lda + k]      B(n,k) = BT ? B[k*ldb + n] : B[n*ldb + k]      C(m,n) = C[m*ldc + n]
// 64x64 tile per 256-thread workgroup, K in tiles of 32 through LDS; a k-strided operand is staged as [k][m] and read with four 4-byte
// LDS loads per fragment instead of one 16-byte load.  Leading dimensions that are not multiples of 4 (the 56/57-wide IPA operands)
// take a scalar load / store path.  blockIdx.z = batch * ksplit + slice: split-K slices write their partial tile to slab `slice` of C
// (stride slab_stride); the caller sums the slabs in a fixed order (deterministic, no atomics).
constexpr int GLT = GBM + 4;          // row stride of a [k][m] staged tile

// one 64 x 64 output tile over K range [kbeg, kend): the body shared by gemm_batched_kernel and gemm_grouped_kernel (A, B, C already point at
// the batch / slab the workgroup owns)
template <bool AT, bool BT>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                          int M, int N, int m0, int n0, int kbeg, int kend, float alpha, const float* __restrict__ bias, int relu,
                                          float* __restrict__ As, float* __restrict__ Bs) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int fm = lane & 15, kq = lane >> 4;
    const bool avec = (lda % 4) == 0 && ((uintptr_t)A % 16) == 0, bvec = (ldb % 4) == 0 && ((uintptr_t)B % 16) == 0;
    // staging registers: two float4 per operand and K tile.  k-contiguous: thread -> (row tid >> 3 (+32), k 4 (tid & 7));
    // k-strided: thread -> (k tid >> 4 (+16), m 4 (tid & 15))
    auto load_op = [&](const float* __restrict__ P, int ld, bool vec, bool T, int r0, int R, int k0, f32x4 (&v)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x4 x = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!T) {
                const int r = r0 + (tid >> 3) + 32 * p, k = k0 + (tid & 7) * 4;
                if (r < R && k < kend) {
                    const float* q = P + (int64_t)r * ld + k;
                    if (vec && k + 3 < kend) x = *reinterpret_cast<const f32x4*>(q);
                    else { for (int i = 0; i < 4; ++i) if (k + i < kend) x[i] = q[i]; }
                }
            } else {
                const int k = k0 + (tid >> 4) + 16 * p, r = r0 + (tid & 15) * 4;
                if (k < kend && r < R) {
                    const float* q = P + (int64_t)k * ld + r;
                    if (vec && r + 3 < R) x = *reinterpret_cast<const f32x4*>(q);
                    else { for (int i = 0; i < 4; ++i) if (r + i < R) x[i] = q[i]; }
                }
            }
            v[p] = x;
        }
    };
    auto store_op = [&](float* S, bool T, const f32x4 (&v)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (!T) *reinterpret_cast<f32x4*>(&S[((tid >> 3) + 32 * p) * GLD + (tid & 7) * 4]) = v[p];
            else    *reinterpret_cast<f32x4*>(&S[((tid >> 4) + 16 * p) * GLT + (tid & 15) * 4]) = v[p];
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 av[2], bv[2];
    if (kbeg < kend) { load_op(A, lda, avec, AT, m0, M, kbeg, av); load_op(B, ldb, bvec, BT, n0, N, kbeg, bv); }
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        __syncthreads();
        store_op(As, AT, av); store_op(Bs, BT, bv);
        __syncthreads();
        if (k0 + GBK < kend) { load_op(A, lda, avec, AT, m0, M, k0 + GBK, av); load_op(B, ldb, bvec, BT, n0, N, k0 + GBK, bv); }
#pragma unroll
        for (int ks = 0; ks < GBK; ks += 16) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!AT) a[i] = *reinterpret_cast<const f32x4*>(&As[(wm + i * 16 + fm) * GLD + ks + kq * 4]);
                else { for (int kk = 0; kk < 4; ++kk) a[i][kk] = As[(ks + kq * 4 + kk) * GLT + wm + i * 16 + fm]; }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!BT) b[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn + j * 16 + fm) * GLD + ks + kq * 4]);
                else { for (int kk = 0; kk < 4; ++kk) b[j][kk] = Bs[(ks + kq * 4 + kk) * GLT + wn + j * 16 + fm]; }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)      // B as the MFMA A operand: a lane ends up with 4 consecutive output columns of one row
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][kk], a[i][kk], acc[i][j], 0, 0, 0);
        }
    }
    const bool cvec = (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = m0 + wm + i * 16 + fm, col = n0 + wn + j * 16 + kq * 4;
            if (row >= M || col >= N) continue;
            f32x4 v = acc[i][j] * alpha;
            if (bias) { for (int r = 0; r < 4; ++r) if (col + r < N) v[r] += bias[col + r]; }       // y = x W^T + b (and ReLU) in the product's epilogue
            if (relu) { for (int r = 0; r < 4; ++r) v[r] = relu_nan(v[r]); }
            float* cp = C + (int64_t)row * ldc + col;
            if (cvec && col + 3 < N) *reinterpret_cast<f32x4*>(cp) = v;
            else { for (int r = 0; r < 4; ++r) if (col + r < N) cp[r] = v[r]; }
        }
}

// register budget: 8 waves per SIMD for the forms with a k-strided operand (52-60 VGPRs: -1.4 us per launch on the weight-gradient and
// d x products, same box, round 5); the nn form needs 70 and spills at 64 (4096 x 256 x 1413: 74 -> 100 us) -- it keeps 4
template <bool AT, bool BT>
__global__ __launch_bounds__(256, (AT || BT) ? 8 : 4) void gemm_batched_kernel(const float* __restrict__ A, int lda, int64_t sa, const float* __restrict__ B, int ldb, int64_t sb,
                                                           float* __restrict__ C, int ldc, int64_t sc, int M, int N, int K, int ksplit, int kchunk,
                                                           int64_t slab_stride, float alpha, const float* __restrict__ bias, int relu) {
    __shared__ __attribute__((aligned(16))) float As[GBM * GLD > GBK * GLT ? GBM * GLD : GBK * GLT];
    __shared__ __attribute__((aligned(16))) float Bs[GBN * GLD > GBK * GLT ? GBN * GLD : GBK * GLT];
    const int batch = blockIdx.z / ksplit, slice = blockIdx.z % ksplit;
    const int kbeg = slice * kchunk;
    gemm_tile<AT, BT>(A + (int64_t)batch * sa, lda, B + (int64_t)batch * sb, ldb, C + (int64_t)batch * sc + (int64_t)slice * slab_stride, ldc, M, N,
                      blockIdx.y * GBM, blockIdx.x * GBN, kbeg, min(K, kbeg + kchunk), alpha, bias, relu, As, Bs);
}

// (Round 4 measured this kernel's bf16-term twin -- operands split into three bf16 terms while a K tile is staged, six
//  v_mfma_f32_16x16x32_bf16 per tile pair instead of 64 v_mfma_f32_16x16x4_f32 -- against it on the training step: 10.94 against 10.25 ms.
//  These products are not matrix-pipe bound at 64 x 64 tiles with one wave per SIMD: the split costs every thread 88 VALU operations and,
//  for a k-strided operand, twelve 4-byte LDS stores per K tile.  Not kept.)

// out[e] = sum over slabs s of in[s * stride + e]  (fixed order)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int slabs, int64_t stride) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = in[e];
    for (int k = 1; k < slabs; ++k) s += in[(int64_t)k * stride + e];
    out[e] = s;
}

// The same for many slabs of few elements (split-K partials of a 64 x 64 weight gradient: 256 slabs of 4096 floats; bucket sums: 2048
// slabs of 4160): a thread per element walks a serial chain of strided loads on a handful of CUs (60 us for 4 MB).  Here 64 elements x G
// slab groups per workgroup: group g adds slabs g, g + G, ... with four loads in flight, the groups are then added in a fixed order.
template <int G>
__global__ __launch_bounds__(64 * G) void slab_sum_groups_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int slabs, int64_t stride) {
    __shared__ float red[G][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int k = g;
        for (; k + 3 * G < slabs; k += 4 * G) {
            s0 += in[(int64_t)k * stride + e]; s1 += in[(int64_t)(k + G) * stride + e];
            s2 += in[(int64_t)(k + 2 * G) * stride + e]; s3 += in[(int64_t)(k + 3 * G) * stride + e];
        }
        for (; k < slabs; k += G) s0 += in[(int64_t)k * stride + e];
    }
    red[g][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < n) {
        float t = red[0][lane];
#pragma unroll
        for (int j = 1; j < G; ++j) t += red[j][lane];
        out[e] = t;
    }
}
static void launch_slab_sum(const float* in, float* out, int64_t n, int slabs, int64_t stride, hipStream_t st) {
    if (slabs >= 128 && n <= (1 << 16)) hipLaunchKernelGGL(slab_sum_groups_kernel<16>, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st, in, out, n, slabs, stride);
    else if (slabs >= 16 && n <= (1 << 18)) hipLaunchKernelGGL(slab_sum_groups_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, in, out, n, slabs, stride);
    else hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n, slabs, stride);
}

// column sums of a row-major [rows, ld] matrix (bias gradients, per-row partials of weight gradients): slice s of the rows -> part[s][cols],
// then slab_sum_kernel adds the slices in a fixed order.  Thread -> (4 consecutive columns, row phase tid / 32 of 8)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ld, int64_t rows, int cols, int64_t rows_per_slice,
                                                            float* __restrict__ part) {
    __shared__ f32x4 red[8][32];
    const int c4 = blockIdx.x * 32 + (threadIdx.x & 31), ph = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool vec = (ld % 4) == 0 && ((uintptr_t)x % 16) == 0;
    if (c4 * 4 < cols)
        for (int64_t r = r0 + ph; r < r1; r += 8) {
            const float* p = x + r * ld + c4 * 4;
            if (vec && c4 * 4 + 3 < cols) s += *reinterpret_cast<const f32x4*>(p);
            else { for (int i = 0; i < 4; ++i) if (c4 * 4 + i < cols) s[i] += p[i]; }
        }
    red[ph][threadIdx.x & 31] = s;
    __syncthreads();
    if (ph == 0 && c4 * 4 < cols) {
        f32x4 t = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][threadIdx.x];
        float* o = part + (int64_t)blockIdx.y * cols + c4 * 4;
        for (int i = 0; i < 4; ++i) if (c4 * 4 + i < cols) o[i] = t[i];
    }
}

int launch_colsum(const float* x, int ld, int64_t rows, int cols, float* out, float* ws, size_t ws_floats, hipStream_t st) {
    if (cols <= 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(rows >= 0 && ld >= cols, "colsum: bad dimensions rows=%lld cols=%d ld=%d", (long long)rows, cols, ld);
    const int cblocks = (cols + 127) / 128;
    int64_t want = rows / 64;
    if (want < 1) want = 1;
    const int cap = 1024 / cblocks > 1 ? 1024 / cblocks : 1;
    int slices = want < cap ? (int)want : cap;
    if (!ws) slices = 1;
    while (slices > 1 && (size_t)slices * cols > ws_floats) --slices;
    const int64_t rps = (rows + slices - 1) / slices;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cblocks, slices), dim3(256), 0, st, x, ld, rows, cols, rps, slices > 1 ? ws : out);
    ABOPT_LAUNCH_CHECK();
    if (slices > 1) {
        launch_slab_sum(ws, out, (int64_t)cols, slices, (int64_t)cols, st);
        ABOPT_LAUNCH_CHECK();
    }
    return ABOPT_OK;
}

// out[b][c] = sum over the rows r with idx[r] == b of x[r*ld + c]  (idx < 0: the row is skipped): the gradient of an embedding table that
// was looked up per ROW of a tall activation matrix (relative-position table of the pair embedding: 65 buckets over N L^2 rows), without
// the [rows, buckets] one-hot matrix.  One wave walks its rows in order and adds each into its own LDS accumulator [nb][64] (lane = column);
// the two waves of a workgroup, then the slices, are summed in a fixed order: deterministic.
constexpr int BKT_MAX = 96;
// Segmented use (abopt_segment_bucket_colsum): a slice IS a segment (its partial sums are the result), and the bucket of row j of segment s is
// idx[(s / idx_div) * rows_per_slice + j] -- one index row shared by idx_div consecutive segments (the residue types of the key residues j for
// every query residue i of a sample).  idx_div = 0: the plain form, idx[r].
__global__ __launch_bounds__(128) void bucket_colsum_kernel(const float* __restrict__ x, int ld, int64_t rows, int cols, const int* __restrict__ idx, int nb,
                                                            int64_t rows_per_slice, float* __restrict__ part, int idx_div) {
    extern __shared__ float bkt_acc_[];                            // [2][nb][64]: sized by the launch (22 buckets: 11 KB, 14 workgroups per CU; a fixed 96-bucket tile held a CU to three)
    float (*acc)[64] = reinterpret_cast<float (*)[64]>(bkt_acc_);
#define BKT_ACC(W_, B_) acc[(W_) * nb + (B_)]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    for (int b = 0; b < nb; ++b) BKT_ACC(w, b)[lane] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
    if (idx_div > 0) idx += ((int64_t)(blockIdx.y / idx_div)) * rows_per_slice - r0;          // idx[r] below = index of row r - r0 of the shared index row
    if (c < cols) {
        // eight rows per trip: their indices and values are requested together (one row at a time the loop is a chain of memory round trips)
        int64_t r = r0 + w;
        for (; r + 14 < r1; r += 16) {
            int bb[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { bb[u] = __builtin_amdgcn_readfirstlane(idx[r + 2 * u]); v[u] = x[(r + 2 * u) * ld + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (bb[u] >= 0 && bb[u] < nb) BKT_ACC(w, bb[u])[lane] += v[u];
        }
        for (; r < r1; r += 2) {
            const int b = __builtin_amdgcn_readfirstlane(idx[r]);
            if (b >= 0 && b < nb) BKT_ACC(w, b)[lane] += x[r * ld + c];
        }
    }
    __syncthreads();
    if (c < cols)
        for (int b = w; b < nb; b += 2) part[((int64_t)blockIdx.y * nb + b) * cols + c] = BKT_ACC(0, b)[lane] + BKT_ACC(1, b)[lane];
#undef BKT_ACC
}

int launch_bucket_colsum(const float* x, int ld, int64_t rows, int cols, const int* idx, int nb, float* out, float* ws, size_t ws_floats, hipStream_t st) {
    if (cols <= 0 || nb <= 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(rows >= 0 && ld >= cols && nb <= BKT_MAX, "bucket_colsum: rows=%lld cols=%d ld=%d buckets=%d (max %d)", (long long)rows, cols, ld, nb, BKT_MAX);
    const int cblocks = (cols + 63) / 64;
    int64_t want = rows / 64;                                   // (4096 rows in 8 slices: 30 us on 16 workgroups; in 64: latency of a 32-row walk)
    if (want < 1) want = 1;
    const int64_t cap = 2048 / cblocks > 1 ? 2048 / cblocks : 1;
    int slices = (int)(want < cap ? want : cap);
    if (!ws) slices = 1;
    while (slices > 1 && (size_t)slices * nb * cols > ws_floats) --slices;
    const int64_t rps = (rows + slices - 1) / slices;
    hipLaunchKernelGGL(bucket_colsum_kernel, dim3(cblocks, slices), dim3(128), (size_t)2 * nb * 64 * sizeof(float), st, x, ld, rows, cols, idx, nb, rps, slices > 1 ? ws : out, 0);
    ABOPT_LAUNCH_CHECK();
    if (slices > 1) {
        const int64_t n = (int64_t)nb * cols;
        launch_slab_sum(ws, out, n, slices, n, st);
        ABOPT_LAUNCH_CHECK();
    }
    return ABOPT_OK;
}

int launch_segment_bucket_colsum(const float* x, int ld, int segments, int rows_per_segment, int cols, const int* idx, int idx_div, int nb, float* out, hipStream_t st) {
    if (cols <= 0 || nb <= 0 || segments <= 0 || rows_per_segment <= 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(ld >= cols && nb <= BKT_MAX && idx_div >= 1 && idx_div <= 65535, "segment_bucket_colsum: segments=%d rows_per_segment=%d cols=%d ld=%d buckets=%d (max %d) idx_div=%d (1..65535)",
                    segments, rows_per_segment, cols, ld, nb, BKT_MAX, idx_div);
    // a segment is a grid row (grid.y <= 65535): more segments go in several launches, cut at multiples of idx_div so that every launch starts on an index row
    const int chunk = (65535 / idx_div) * idx_div;
    for (int s0 = 0; s0 < segments; s0 += chunk) {
        const int ns = min(chunk, segments - s0);
        hipLaunchKernelGGL(bucket_colsum_kernel, dim3((cols + 63) / 64, ns), dim3(128), (size_t)2 * nb * 64 * sizeof(float), st, x + (int64_t)s0 * rows_per_segment * ld, ld,
                           (int64_t)ns * rows_per_segment, cols, idx + (int64_t)(s0 / idx_div) * rows_per_segment, nb, (int64_t)rows_per_segment, out + (int64_t)s0 * nb * cols, idx_div);
        ABOPT_LAUNCH_CHECK();
    }
    return ABOPT_OK;
}

int launch_gemm_batched(const float* A, int lda, int64_t sa, int a_t, const float* B, int ldb, int64_t sb, int b_t, float* C, int ldc, int64_t sc,
                        int M, int N, int K, int batch, float alpha, float* ws, size_t ws_floats, hipStream_t st, const float* bias, int relu) {
    if (M <= 0 || N <= 0 || batch <= 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(K >= 0 && lda >= 1 && ldb >= 1 && ldc >= N, "gemm: bad dimensions M=%d N=%d K=%d lda=%d ldb=%d ldc=%d", M, N, K, lda, ldb, ldc);
    const int tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN) * batch;
    // split K when the output alone cannot fill the chip (weight gradients: K = number of residues, a few dozen output tiles)
    int ksplit = 1;
    // The partial slabs have C's own layout and the slab sum writes every element of it: only for a densely packed C (ldc == N, batches
    // back to back) -- a C that is a column slice of a wider matrix (ldc > N) or has gaps between batches takes the unsplit kernel
    const bool dense_c = ldc == N && (batch == 1 || sc == (int64_t)M * ldc);
    static const int tmax = getenv("ABOPT_GEMM_TMAX") ? atoi(getenv("ABOPT_GEMM_TMAX")) : 256;      // split K below this many output tiles (developer knob; 128 -> 256: the 128-tile d x products of the projections fill the chip, 10.1 -> 9.9 ms per training step)
    if (tiles < tmax && K >= 1024 && ws && !bias && !relu && dense_c) {      // (an epilogue with bias / ReLU needs the whole sum in one workgroup)
        // (up to 1024 slabs: the tall products of the pair embedding -- K = N L^2 rows, one or four output tiles -- are HBM streams, and
        // 256 workgroups of four waves keep too few bytes in flight: 143 us for 2 x 268 MB at 256 slabs)
        static const int kdiv = getenv("ABOPT_GEMM_KDIV") ? atoi(getenv("ABOPT_GEMM_KDIV")) : 256;     // shortest K range of a slab (developer knob; 512 -> 256: 10.8 -> 10.3 ms per training step, more workgroups per weight-gradient product)
        ksplit = min(min(1024 / max(tiles, 1), K / kdiv), 1024);
        while (ksplit > 1 && (size_t)ksplit * batch * M * ldc > ws_floats) --ksplit;
        ksplit = max(ksplit, 1);
    }
    const int kchunk = ksplit == 1 ? max(K, 1) : ((K + ksplit * GBK - 1) / (ksplit * GBK)) * GBK;
    float* out = ksplit == 1 ? C : ws;
    const int64_t slab = (int64_t)batch * M * ldc;
    dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM, batch * ksplit);
#define ABOPT_GEMM(AT_, BT_) hipLaunchKernelGGL((gemm_batched_kernel<AT_, BT_>), grid, dim3(256), 0, st, A, lda, sa, B, ldb, sb, out, ldc, sc, M, N, K, ksplit, \
                                                 kchunk, slab, alpha, bias, relu)
    if (a_t) { if (b_t) ABOPT_GEMM(true, true); else ABOPT_GEMM(true, false); }
    else     { if (b_t) ABOPT_GEMM(false, true); else ABOPT_GEMM(false, false); }
#undef ABOPT_GEMM
    ABOPT_LAUNCH_CHECK();
    if (ksplit > 1) {
        launch_slab_sum(ws, C, slab, ksplit, slab, st);
        ABOPT_LAUNCH_CHECK();
    }
    return ABOPT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped weight-gradient products (round 5): C_p = A_p^T B_p for up to GG_MAX tall operand pairs (K_p rows, both read k-strided in
// place) in ONE launch plus ONE launch for the split-K slab sums of all of them.  A backward pass of the denoiser has ~44 such products
// with 2-64 output tiles each; alone, each is a split-K launch that ramps up and drains in 20-45 us plus a slab-sum launch of 3-7 us.
// Grouped, the workgroups of all problems form one grid (problem p owns workgroups [wg_begin, wg_begin + tiles * ksplit)) and the
// K split is chosen for the group's tile count, not for each product's.  The descriptors travel by value in the kernel arguments: no device
// memory, nothing to keep alive, capturable.  Same tile body, same fixed-order slab sum (four slab groups of a 64-element strip, then the
// groups in order): deterministic; the summation order differs from abopt_gemm's for the same product (another K split).
constexpr int GG_MAX = 24;
struct GroupedProblem {
    const float* A; const float* B; float* C; float* ws;
    int lda, ldb, M, N, K, tiles_n, tiles, ksplit, kchunk, wg_begin, sum_begin, pad_;
};
struct GroupedArgs { int n, pad_; GroupedProblem p[GG_MAX]; };

__global__ __launch_bounds__(256, 8) void gemm_grouped_kernel(const GroupedArgs g) {
    __shared__ __attribute__((aligned(16))) float As[GBK * GLT];
    __shared__ __attribute__((aligned(16))) float Bs[GBK * GLT];
    int pi = 0;
    for (int q = 1; q < g.n; ++q) if ((int)blockIdx.x >= g.p[q].wg_begin) pi = q;
    const GroupedProblem& P = g.p[pi];
    const int local = blockIdx.x - P.wg_begin, slice = local / P.tiles, t = local % P.tiles;
    const int kbeg = slice * P.kchunk;
    float* out = P.ksplit > 1 ? P.ws + (int64_t)slice * P.M * P.N : P.C;
    gemm_tile<true, true>(P.A, P.lda, P.B, P.ldb, out, P.N, P.M, P.N, (t / P.tiles_n) * GBM, (t % P.tiles_n) * GBN, kbeg, min(P.K, kbeg + P.kchunk),
                          1.f, nullptr, 0, As, Bs);
}

// C_p[e] = sum over the slabs of ws_p (slab_sum_groups_kernel<4>'s order) for every problem of the group that was split
__global__ __launch_bounds__(256) void slab_sum_grouped_kernel(const GroupedArgs g) {
    __shared__ float red[4][64];
    int pi = 0;
    for (int q = 1; q < g.n; ++q) if ((int)blockIdx.x >= g.p[q].sum_begin) pi = q;
    const GroupedProblem& P = g.p[pi];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t n = (int64_t)P.M * P.N, e = (int64_t)(blockIdx.x - P.sum_begin) * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int k = grp;
        for (; k + 12 < P.ksplit; k += 16) {
            s0 += P.ws[(int64_t)k * n + e]; s1 += P.ws[(int64_t)(k + 4) * n + e];
            s2 += P.ws[(int64_t)(k + 8) * n + e]; s3 += P.ws[(int64_t)(k + 12) * n + e];
        }
        for (; k < P.ksplit; k += 4) s0 += P.ws[(int64_t)k * n + e];
    }
    red[grp][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && e < n) P.C[e] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

int launch_gemm_tn_grouped(const abopt_gemm_tn_problem* probs, int count, float* ws, size_t ws_floats, hipStream_t st) {
    if (count <= 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(probs && count <= GG_MAX, "gemm_tn_grouped: %d problems (1..%d)", count, GG_MAX);
    GroupedArgs g;
    g.n = count; g.pad_ = 0;
    int total_tiles = 0;
    for (int i = 0; i < count; ++i) {
        const abopt_gemm_tn_problem& q = probs[i];
        ABOPT_CHECK_ARG(q.a && q.b && q.c && q.m >= 1 && q.n >= 1 && q.k >= 0 && q.lda >= q.m && q.ldb >= q.n,
                        "gemm_tn_grouped: problem %d: M=%d N=%d K=%d lda=%d ldb=%d", i, q.m, q.n, q.k, q.lda, q.ldb);
        total_tiles += ((q.m + GBM - 1) / GBM) * ((q.n + GBN - 1) / GBN);
    }
    // K split for the GROUP: enough workgroups to fill the chip about four times over, slabs of at least 256 rows of K (the same rule
    // abopt_gemm applies to a single product), limited by the workspace
    const int want = total_tiles < 256 ? std::max(1024 / total_tiles, 1) : 1;
    size_t ws_used = 0;
    int wg = 0, sum_blocks = 0;
    for (int i = 0; i < count; ++i) {
        const abopt_gemm_tn_problem& q = probs[i];
        GroupedProblem& P = g.p[i];
        P.A = q.a; P.B = q.b; P.C = q.c; P.lda = q.lda; P.ldb = q.ldb; P.M = q.m; P.N = q.n; P.K = q.k; P.pad_ = 0;
        P.tiles_n = (q.n + GBN - 1) / GBN;
        P.tiles = ((q.m + GBM - 1) / GBM) * P.tiles_n;
        int ks = std::max(1, std::min(want, q.k / 256));
        const size_t mn = (size_t)q.m * q.n;
        while (ks > 1 && (!ws || ws_used + (size_t)ks * mn > ws_floats)) --ks;
        P.ksplit = ks;
        P.kchunk = ks == 1 ? std::max(q.k, 1) : ((q.k + ks * GBK - 1) / (ks * GBK)) * GBK;
        P.ws = ks > 1 ? ws + ws_used : nullptr;
        if (ks > 1) ws_used += (size_t)ks * mn;
        P.wg_begin = wg; wg += P.tiles * ks;
        P.sum_begin = sum_blocks; if (ks > 1) sum_blocks += (int)((mn + 63) / 64);
    }
    hipLaunchKernelGGL(gemm_grouped_kernel, dim3((unsigned)wg), dim3(256), 0, st, g);
    ABOPT_LAUNCH_CHECK();
    if (sum_blocks > 0) {
        // problems that were not split own no blocks: give them an empty range that the scan skips (sum_begin of the next split problem)
        hipLaunchKernelGGL(slab_sum_grouped_kernel, dim3((unsigned)sum_blocks), dim3(256), 0, st, g);
        ABOPT_LAUNCH_CHECK();
    }
    return ABOPT_OK;
}

}  // namespace abopt

extern "C" int abopt_gemm_tn_grouped(const abopt_gemm_tn_problem* problems, int count, void* ws, size_t ws_bytes, abopt_stream stream) {
    return abopt::launch_gemm_tn_grouped(problems, count, (float*)ws, ws_bytes / sizeof(float), (hipStream_t)stream);
}

extern "C" int abopt_bucket_colsum(const float* x, int ld, int64_t rows, int cols, const int32_t* idx, int buckets, float* out, void* ws, size_t ws_bytes,
                                   abopt_stream stream) {
    ABOPT_CHECK_ARG(x && out && idx, "bucket_colsum: NULL argument");
    return abopt::launch_bucket_colsum(x, ld, rows, cols, idx, buckets, out, (float*)ws, ws_bytes / sizeof(float), (hipStream_t)stream);
}

extern "C" int abopt_segment_bucket_colsum(const float* x, int ld, int segments, int rows_per_segment, int cols, const int32_t* idx, int idx_div, int buckets,
                                           float* out, abopt_stream stream) {
    ABOPT_CHECK_ARG(x && out && idx, "segment_bucket_colsum: NULL argument");
    return abopt::launch_segment_bucket_colsum(x, ld, segments, rows_per_segment, cols, idx, idx_div, buckets, out, (hipStream_t)stream);
}

extern "C" int abopt_colsum(const float* x, int ld, int64_t rows, int cols, float* out, void* ws, size_t ws_bytes, abopt_stream stream) {
    ABOPT_CHECK_ARG(x && out, "colsum: NULL argument");
    return abopt::launch_colsum(x, ld, rows, cols, out, (float*)ws, ws_bytes / sizeof(float), (hipStream_t)stream);
}

extern "C" int abopt_gemm(const float* A, int lda, int64_t stride_a, int a_transposed, const float* B, int ldb, int64_t stride_b, int b_transposed,
                          float* C, int ldc, int64_t stride_c, int M, int N, int K, int batch, float alpha, const float* bias, int relu,
                          void* ws, size_t ws_bytes, abopt_stream stream) {
    ABOPT_CHECK_ARG(A && B && C, "gemm: NULL operand");
    return abopt::launch_gemm_batched(A, lda, stride_a, a_transposed, B, ldb, stride_b, b_transposed, C, ldc, stride_c, M, N, K, batch, alpha,
                                      (float*)ws, ws_bytes / sizeof(float), (hipStream_t)stream, bias, relu);
}

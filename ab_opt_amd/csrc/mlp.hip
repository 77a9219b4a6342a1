// Fused tail of a GABlock for CDNA4: one launch for
//     y   = LayerNorm1(x + mask * (sum_of_split-K_slabs(out_transform) + b_out))
//     out = LayerNorm2(y + W2 relu(W1 relu(W0 y + b0) + b1) + b2)
// (reference AbDock/src/modules/encoders/ga.py:174-177, LayerNorm: AbDock/src/modules/common/layers.py:146-155).
// Replaces two LayerNorm launches and three 128x128 GEMM launches whose operands (8192 x 128 activations) are tiny: the
// work per row block is latency/launch bound, so a 256-thread workgroup keeps its 32 rows on chip (LDS) through all three
// layers and streams the three 64 KB weight matrices from L2.  fp32 MFMA (v_mfma_f32_16x16x4_f32), exact fp32.
#include "abopt_common.h"
#include "kernels.h"

namespace abopt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int F = 128, XLD = F + 4;

__device__ __forceinline__ f32x4 mfma4m(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// one dense layer, one wave: 16 rows x 64 output columns (n-tiles 4*half .. 4*half+3) = X[16, 128] . W[64 rows of it, 128]^T.
// All weight fragments of the layer are requested up front (they come from L2; 32 independent loads in flight).
__device__ __forceinline__ void wave_linear(const float (*xs)[XLD], const float* __restrict__ W, int half, int fm, int kq, f32x4 (&acc)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wrow = W + (size_t)(half * 64 + fm) * F + kq * 4;      // lane group kq supplies k = 16 kb + 4 kq + s
    float4 b[F / 16][4];                                                // the wave's whole 64 x 128 weight slice: 32 loads in flight at once
#pragma unroll
    for (int kb = 0; kb < F / 16; ++kb)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b[kb][nt] = *reinterpret_cast<const float4*>(wrow + (size_t)nt * 16 * F + kb * 16);
#pragma unroll
    for (int kb = 0; kb < F / 16; ++kb) {
        const float4 a = *reinterpret_cast<const float4*>(&xs[fm][kb * 16 + kq * 4]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            acc[nt] = mfma4m(a.x, b[kb][nt].x, acc[nt]); acc[nt] = mfma4m(a.y, b[kb][nt].y, acc[nt]);
            acc[nt] = mfma4m(a.z, b[kb][nt].z, acc[nt]); acc[nt] = mfma4m(a.w, b[kb][nt].w, acc[nt]);
        }
    }
}

// 128 threads = 2 waves = the two column halves (64 output columns each) of one 16-row group.  Small workgroups on purpose:
// the phases are latency chains (L2 weight loads, LayerNorm reductions), several co-resident workgroups per CU overlap them.
__global__ __launch_bounds__(128) void fused_ln_mlp_kernel(const float* __restrict__ x, const float* __restrict__ u, int nslab, int64_t slab_stride,
                                                           const float* __restrict__ ubias, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ g1, const float* __restrict__ be1,
                                                           const float* __restrict__ W0, const float* __restrict__ b0,
                                                           const float* __restrict__ W1, const float* __restrict__ b1,
                                                           const float* __restrict__ W2, const float* __restrict__ b2,
                                                           const float* __restrict__ g2, const float* __restrict__ be2,
                                                           float* __restrict__ out, int64_t rows) {
    __shared__ __attribute__((aligned(16))) float ys[1][16][XLD];     // LayerNorm1 output (residual of the MLP)
    __shared__ __attribute__((aligned(16))) float ha[1][16][XLD];     // activations, ping
    __shared__ __attribute__((aligned(16))) float hb[1][16][XLD];     // activations, pong
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fm = lane & 15, kq = lane >> 4;
    const int rg = 0, half = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * 16;

    // ---- LayerNorm1: the two waves of a row group take 8 rows each, one row at a time across the wave
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = half * 8 + rr;
        const int64_t row = min(row0 + r, rows - 1);
        const bool keep = mask ? (mask[row] != 0) : true;
        const float2 xv = reinterpret_cast<const float2*>(x + row * F)[lane];
        float2 uv = reinterpret_cast<const float2*>(u + row * F)[lane];
        for (int sl = 1; sl < nslab; ++sl) { const float2 w = reinterpret_cast<const float2*>(u + sl * slab_stride + row * F)[lane]; uv.x += w.x; uv.y += w.y; }
        if (ubias) { const float2 bb = reinterpret_cast<const float2*>(ubias)[lane]; uv.x += bb.x; uv.y += bb.y; }
        if (!keep) uv = make_float2(0.f, 0.f);
        const float a = xv.x + uv.x, b = xv.y + uv.y;
        const float mean = wave_sum(a + b) * (1.f / F);
        const float da = a - mean, db = b - mean;
        const float var = wave_sum(da * da + db * db) * (1.f / F);
        const float sd = sqrtf(var + 1e-10f);
        const float2 g = reinterpret_cast<const float2*>(g1)[lane], bt = reinterpret_cast<const float2*>(be1)[lane];
        *reinterpret_cast<float2*>(&ys[rg][r][2 * lane]) = make_float2(da / sd * g.x + bt.x, db / sd * g.y + bt.y);
    }
    __syncthreads();
    f32x4 acc[4];
    // ---- layer 0: relu(W0 y + b0) -> ha
    wave_linear(ys[rg], W0, half, fm, kq, acc);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = half * 64 + nt * 16 + fm;
        const float bv = b0[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) ha[rg][kq * 4 + r][col] = fmaxf(acc[nt][r] + bv, 0.f);
    }
    __syncthreads();
    // ---- layer 1: relu(W1 h + b1) -> hb
    wave_linear(ha[rg], W1, half, fm, kq, acc);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = half * 64 + nt * 16 + fm;
        const float bv = b1[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[rg][kq * 4 + r][col] = fmaxf(acc[nt][r] + bv, 0.f);
    }
    __syncthreads();
    // ---- layer 2 + residual -> ha, then LayerNorm2
    wave_linear(hb[rg], W2, half, fm, kq, acc);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = half * 64 + nt * 16 + fm;
        const float bv = b2[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) ha[rg][kq * 4 + r][col] = ys[rg][kq * 4 + r][col] + (acc[nt][r] + bv);
    }
    __syncthreads();
    for (int r = half * 8; r < half * 8 + 8; ++r) {
        const int64_t row = row0 + r;
        const float2 v = *reinterpret_cast<const float2*>(&ha[rg][r][2 * lane]);
        const float mean = wave_sum(v.x + v.y) * (1.f / F);
        const float da = v.x - mean, db = v.y - mean;
        const float var = wave_sum(da * da + db * db) * (1.f / F);
        const float sd = sqrtf(var + 1e-10f);
        const float2 g = reinterpret_cast<const float2*>(g2)[lane], bt = reinterpret_cast<const float2*>(be2)[lane];
        if (row < rows) reinterpret_cast<float2*>(out + row * F)[lane] = make_float2(da / sd * g.x + bt.x, db / sd * g.y + bt.y);
    }
}

int launch_fused_ln_mlp(const float* x, const float* u, int nslab, int64_t slab_stride, const float* ubias, const uint8_t* mask,
                        const float* g1, const float* be1, const float* W0, const float* b0, const float* W1, const float* b1,
                        const float* W2, const float* b2, const float* g2, const float* be2, float* out, int64_t rows, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(fused_ln_mlp_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(128), 0, st, x, u, nslab, slab_stride, ubias, mask,
                       g1, be1, W0, b0, W1, b1, W2, b2, g2, be2, out, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

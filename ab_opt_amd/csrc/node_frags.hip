// Fused node projections of a GABlock for CDNA4: the six bias-free nn.Linear of ga.py:54-66 (q | k | v | q_pts | k_pts | v_pts,
// [M,128] x [2016,128]^T), the local -> global map of the three point sets (geometry.py:72-91, ga.py:96-105,129-132), the squared
// point norms, and the re-layout of everything into the MFMA fragment order the IPA core consumes (qfrag / kvfrag, see ipa.hip) --
// in ONE kernel.  The 67 MB projection buffer is never written or re-read.
//
// Weight-stationary: a workgroup owns ONE head.  Its 168 weight rows, permuted and zero-padded at pack time to 12 tiles of 16 rows
//   tile 0,1: q channels 0..15, 16..31   2,3: k   4,5: v   6,7: q_pts points 0..3, 4..7 as (x, y, z, 0) quadruples   8,9: k_pts   10,11: v_pts
// sit in LDS in A-operand fragment order (96 KB, loaded once); residues stream through in 16-row tiles, one tile per wave at a time,
// the x tile in registers as the B operand.  Per (head, 16 residues): 12 x 32 fp32 MFMAs (v_mfma_f32_16x16x4_f32, exact fp32), then a
// register-only epilogue: with this row order an accumulator tile IS a fragment slot -- lane (residue, kq) holds 4 consecutive
// channels, or (x, y, z, pad) of one point, so the frame transform needs no cross-lane traffic.  The value tiles run with the operands
// swapped (accumulator = [residue 4 kq + r][channel fm]), which is the key-major layout of the aggregation operand.
// Traffic per launch at M = 8192: x re-read once per head from L2 (12 x 4 MB), weights 12 x 96 KB, fragments written once (75 MB).
#include "ipa_common.h"
#include "kernels.h"

namespace abopt {

constexpr int NF_TILES = 12, NF_WAVES = 16;                   // 4 waves per SIMD: enough independent MFMA chains to keep the matrix pipe fed
constexpr int NF_TILE_FLOATS = 8 * 64 * 4;                     // one weight tile in fragment order: [j = 0..7][lane][4]

__device__ __forceinline__ float quad_bcast0(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x00, 0xf, 0xf, false)); }
__device__ __forceinline__ float quad_bcast1(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x55, 0xf, 0xf, false)); }
__device__ __forceinline__ float quad_bcast2(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xAA, 0xf, 0xf, false)); }

__global__ __launch_bounds__(NF_WAVES * 64) void node_frags_kernel(const float* __restrict__ x, const float* __restrict__ wfrag, const float* __restrict__ R,
                                                                   const float* __restrict__ t, const float* __restrict__ spatial_coef,
                                                                   float* __restrict__ qfrag, float* __restrict__ kvfrag, int L, int nchunk,
                                                                   int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) char nf_smem[];
    f32x4* wl = reinterpret_cast<f32x4*>(nf_smem);                             // [12 tiles][8][64]
    const int h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const f32x4* wg = reinterpret_cast<const f32x4*>(wfrag) + (int64_t)h * NF_TILES * 8 * 64;
#pragma unroll
        for (int e = 0; e < NF_TILES * 8 * 64 / (NF_WAVES * 64); ++e) wl[e * (NF_WAVES * 64) + tid] = wg[e * (NF_WAVES * 64) + tid];
    }
    const float sc = spatial_coef[h];
    const float gamma = (sc > 20.f) ? sc : log1pf(expf(sc));                     // softplus, ga.py:108
    const float ch_ = (-1.f * gamma * 0.16666666666666666f) / 2.f;               // -gamma sqrt(2/(9*8)) / 2, ga.py:109-110
    const float m2c = -2.f * ch_;
    __syncthreads();

    // tiles are dealt to the (workgroup, wave) slots round-robin; with 4 waves per SIMD the other waves' MFMAs cover a wave's operand loads
    const int stride = gridDim.x * NF_WAVES;
    for (int tile = blockIdx.x * NF_WAVES + wave; tile < total_tiles; tile += stride) {
        const int n = tile / nchunk, cb = tile % nchunk;
        const int64_t rowbase = (int64_t)n * L;
        const int64_t row = rowbase + min(cb * JC + fm, L - 1);                  // rows past the end: clamped copies (finite; the core never stores them)
        f32x4 xf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = *reinterpret_cast<const f32x4*>(x + row * 128 + kq * 32 + 4 * j);   // lane (row fm, kq) holds k = 32 kq + 4 j + i
        f32x4 acc[NF_TILES];
#pragma unroll
        for (int T = 0; T < NF_TILES; ++T) acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // two tiles at a time (two independent accumulator chains hide the 40-cycle dependent-MFMA latency); the weight fragments of
        // step g + 1 are read from LDS before the 8 MFMAs of step g are issued (hipcc left to itself hoists all 96 reads = 384 VGPRs)
        f32x4 wa[2][2];
        wa[0][0] = wl[lane]; wa[0][1] = wl[8 * 64 + lane];
#pragma unroll
        for (int g = 0; g < (NF_TILES / 2) * 8; ++g) {
            const int T = (g >> 3) * 2, j = g & 7;
            const bool swap = (T == 4) || (T == 10);                             // value tiles: x is the A operand -> accumulator [residue 4 kq + r][channel fm]
            if (g + 1 < (NF_TILES / 2) * 8) {
                const int Tn = ((g + 1) >> 3) * 2, jn = (g + 1) & 7;
                wa[(g + 1) & 1][0] = wl[(Tn * 8 + jn) * 64 + lane]; wa[(g + 1) & 1][1] = wl[((Tn + 1) * 8 + jn) * 64 + lane];
            }
            const f32x4 a0 = wa[g & 1][0], a1 = wa[g & 1][1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (swap) { acc[T] = mfma4(xf[j][i], a0[i], acc[T]); acc[T + 1] = mfma4(xf[j][i], a1[i], acc[T + 1]); }
                else      { acc[T] = mfma4(a0[i], xf[j][i], acc[T]); acc[T + 1] = mfma4(a1[i], xf[j][i], acc[T + 1]); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // frames are fetched only now: x fragments and weight registers are dead, so the 128-VGPR budget (4 waves / SIMD) holds
        float Rm[9], tv[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm[k] = R[row * 9 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tv[k] = t[row * 3 + k];
        f32x4* outq = reinterpret_cast<f32x4*>(qfrag) + ((int64_t)tile * H + h) * (4 * 64) + lane;
        f32x4* outk = reinterpret_cast<f32x4*>(kvfrag) + ((int64_t)tile * H + h) * (8 * 64) + lane;
        // ---- q, k: accumulator row 4 kq + r = channel, column fm = residue
        const float s = 0.17677669529663687f;                                    // 1 / sqrt(D), ga.py:84
        outq[0] = acc[0] * s; outq[64] = acc[1] * s;
        outk[0] = acc[2]; outk[64] = acc[3];
        // ---- q_pts, k_pts: (x, y, z, pad) of point kq (tile A) and 4 + kq (tile B) of residue fm; p <- R p + t (geometry.py:72-91)
        auto to_global = [&](const f32x4& p) {
            return (f32x4){Rm[0] * p[0] + Rm[1] * p[1] + Rm[2] * p[2] + tv[0], Rm[3] * p[0] + Rm[4] * p[1] + Rm[5] * p[2] + tv[1],
                           Rm[6] * p[0] + Rm[7] * p[1] + Rm[8] * p[2] + tv[2], 0.f};
        };
        auto sq = [](const f32x4& g) { return fmaf(g[2], g[2], fmaf(g[1], g[1], g[0] * g[0])); };
        {
            f32x4 ga = to_global(acc[6]), gb = to_global(acc[7]);
            const float nq = rows_sum(sq(ga) + sq(gb));                          // |q_pts|^2 over the head's 8 points
            ga *= m2c; gb *= m2c;
            ga[3] = kq == 0 ? ch_ * nq : (kq == 1 ? ch_ : 0.f);                  // norm step, q side
            gb[3] = 0.f;
            outq[128] = ga; outq[192] = gb;
        }
        {
            f32x4 ga = to_global(acc[8]), gb = to_global(acc[9]);
            const float nk = rows_sum(sq(ga) + sq(gb));
            ga[3] = kq == 0 ? 1.f : (kq == 1 ? nk : 0.f);                        // norm step, k side
            outk[128] = ga; outk[192] = gb;
        }
        // ---- v, v_pts: accumulator row 4 kq + r = residue, column fm = channel / (point fm >> 2, coordinate fm & 3)
        {
            const int c = fm & 3, cr = min(c, 2);                                 // row cr of R and t[cr] of residue 4 kq + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t rr = rowbase + min(cb * JC + kq * 4 + r, L - 1);
                const float r0 = R[rr * 9 + cr * 3], r1 = R[rr * 9 + cr * 3 + 1], r2 = R[rr * 9 + cr * 3 + 2], r3 = t[rr * 3 + cr];
                const float xa = quad_bcast0(acc[10][r]), ya = quad_bcast1(acc[10][r]), za = quad_bcast2(acc[10][r]);
                const float xb = quad_bcast0(acc[11][r]), yb = quad_bcast1(acc[11][r]), zb = quad_bcast2(acc[11][r]);
                float g0 = r0 * xa + r1 * ya + r2 * za + r3;
                float g1 = r0 * xb + r1 * yb + r2 * zb + r3;
                if (c == 3) { g0 = 0.f; g1 = 0.f; }
                outk[(4 + r) * 64] = (f32x4){acc[4][r], acc[5][r], g0, g1};
            }
        }
    }
}

size_t node_wfrag_floats() { return (size_t)H * NF_TILES * NF_TILE_FLOATS; }

int launch_node_frags(const float* x, const float* wfrag, const float* R, const float* t, const float* spatial_coef, float* qfrag, float* kvfrag,
                      int N, int L, hipStream_t st) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    const int nchunk = (L + JC - 1) / JC, total = N * nchunk;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        ABOPT_HIP(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        ABOPT_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount;
        ABOPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(node_frags_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NF_TILES * NF_TILE_FLOATS * 4));
    }
    const int groups = max(1, min(cus / H, (total + NF_WAVES - 1) / NF_WAVES));          // one workgroup per CU: 96 KB of LDS each
    hipLaunchKernelGGL(node_frags_kernel, dim3(groups, H), dim3(NF_WAVES * 64), NF_TILES * NF_TILE_FLOATS * 4, st, x, wfrag, R, t, spatial_coef,
                       qfrag, kvfrag, L, nchunk, total);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

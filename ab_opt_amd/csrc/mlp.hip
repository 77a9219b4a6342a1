// Fused tail of a GABlock for CDNA4: one launch for
//     y   = LayerNorm1(x + mask * (sum_of_split-K_slabs(out_transform) + b_out))
//     out = LayerNorm2(y + W2 relu(W1 relu(W0 y + b0) + b1) + b2)
// (reference AbDock/src/modules/encoders/ga.py:174-177, LayerNorm: AbDock/src/modules/common/layers.py:146-155).
// Replaces two LayerNorm launches and three 128x128 GEMM launches whose operands (8192 x 128 activations) are tiny: the
// work per row block is latency/launch bound, so a 256-thread workgroup keeps its 32 rows on chip (LDS) through all three
// layers and stages each 64 KB weight matrix into LDS once (register-prefetched behind the previous phase).
// fp32 MFMA (v_mfma_f32_16x16x4_f32), exact fp32.
#include "tail_common.h"
#include "kernels.h"

namespace abopt {

__device__ __forceinline__ f32x4 mfma4m(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int WPT = F * F / 4 / 256;       // float4 weight loads per thread per layer (16)

// the 64 KB weight matrix of one layer travels global -> registers (issued early, hidden behind the previous phase) -> LDS
struct WRegs { f32x4 v[WPT]; };
__device__ __forceinline__ WRegs mlp_load_w(const float* __restrict__ W) {
    WRegs r;
#pragma unroll
    for (int i = 0; i < WPT; ++i) r.v[i] = reinterpret_cast<const f32x4*>(W)[i * 256 + threadIdx.x];
    return r;
}
__device__ __forceinline__ void mlp_store_w(float (*wl)[XLD], const WRegs& r) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int e = (i * 256 + threadIdx.x) * 4;
        *reinterpret_cast<f32x4*>(&wl[e >> 7][e & 127]) = r.v[i];
    }
}

// one dense layer, one wave: 16 rows x 64 output columns = X[16, 128] . W[64 rows of it, 128]^T, both operands from LDS.
// W is the MFMA A operand, so the accumulator is the transposed tile: a lane holds 4 consecutive output columns of row fm.
__device__ __forceinline__ void wave_linear(const float (*xs)[XLD], const float (*wl)[XLD], int half, int fm, int kq, f32x4 (&acc)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < F / 16; ++kb) {
        const float4 a = *reinterpret_cast<const float4*>(&xs[fm][kb * 16 + kq * 4]);
        float4 b[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b[nt] = *reinterpret_cast<const float4*>(&wl[half * 64 + nt * 16 + fm][kb * 16 + kq * 4]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma4m(b[nt].x, a.x, acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma4m(b[nt].y, a.y, acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma4m(b[nt].z, a.z, acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma4m(b[nt].w, a.w, acc[nt]);
    }
}

// 256 threads = 4 waves = (2 groups of 16 rows) x (2 column halves).  One workgroup per CU (118 KB of LDS): 32 rows stay on
// chip through LayerNorm1, the three layers and LayerNorm2; each layer's weights are staged once per workgroup.
template <int NSLAB>
__global__ __launch_bounds__(256) void fused_ln_mlp_kernel(const float* __restrict__ x, const float* __restrict__ u, int64_t slab_stride,
                                                           const float* __restrict__ ubias, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ g1, const float* __restrict__ be1,
                                                           const float* __restrict__ W0, const float* __restrict__ b0,
                                                           const float* __restrict__ W1, const float* __restrict__ b1,
                                                           const float* __restrict__ W2, const float* __restrict__ b2,
                                                           const float* __restrict__ g2, const float* __restrict__ be2,
                                                           float* __restrict__ out, int64_t rows) {
    __shared__ __attribute__((aligned(16))) float ys[MR][XLD];     // LayerNorm1 output (residual of the MLP)
    __shared__ __attribute__((aligned(16))) float ha[MR][XLD];     // activations, ping
    __shared__ __attribute__((aligned(16))) float hb[MR][XLD];     // activations, pong
    __shared__ __attribute__((aligned(16))) float wl[F][XLD];      // current layer's weights
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fm = lane & 15, kq = lane >> 4;
    const int rg = wave >> 1, half = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * MR;
    WRegs wreg = mlp_load_w(W0);

    // ---- LayerNorm1: each wave takes 8 rows.  All loads of the 8 rows are issued before any arithmetic (a row-at-a-time loop
    // serialised 8 HBM round trips: 27 k of the kernel's 58 k cycles), then the 8 reductions run as independent chains.
    constexpr int RW = MR / 4;
    float a_[RW], b_[RW];
    {
        float2 xv[RW], uv[RW][NSLAB];
        bool keep[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int64_t row = min(row0 + wave * RW + rr, rows - 1);
            keep[rr] = mask ? (mask[row] != 0) : true;
            xv[rr] = reinterpret_cast<const float2*>(x + row * F)[lane];
#pragma unroll
            for (int sl = 0; sl < NSLAB; ++sl) uv[rr][sl] = reinterpret_cast<const float2*>(u + sl * slab_stride + row * F)[lane];
        }
        const float2 bb = ubias ? reinterpret_cast<const float2*>(ubias)[lane] : make_float2(0.f, 0.f);
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            float2 us = uv[rr][0];
#pragma unroll
            for (int sl = 1; sl < NSLAB; ++sl) { us.x += uv[rr][sl].x; us.y += uv[rr][sl].y; }
            us.x += bb.x; us.y += bb.y;
            if (!keep[rr]) us = make_float2(0.f, 0.f);
            a_[rr] = xv[rr].x + us.x; b_[rr] = xv[rr].y + us.y;
        }
    }
    {
        const float2 g = reinterpret_cast<const float2*>(g1)[lane], bt = reinterpret_cast<const float2*>(be1)[lane];
        float mean[RW], var[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) mean[rr] = wave_sum(a_[rr] + b_[rr]) * (1.f / F);
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) { a_[rr] -= mean[rr]; b_[rr] -= mean[rr]; var[rr] = wave_sum(a_[rr] * a_[rr] + b_[rr] * b_[rr]) * (1.f / F); }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const float sd = sqrtf(var[rr] + 1e-10f);
            *reinterpret_cast<float2*>(&ys[wave * RW + rr][2 * lane]) = make_float2(a_[rr] / sd * g.x + bt.x, b_[rr] / sd * g.y + bt.y);
        }
    }
    mlp_store_w(wl, wreg);
    wreg = mlp_load_w(W1);
    __syncthreads();
    f32x4 acc[4];
    // ---- layer 0: relu(W0 y + b0) -> ha
    wave_linear(ys + rg * 16, wl, half, fm, kq, acc);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = half * 64 + nt * 16 + kq * 4;
        const float4 bv = *reinterpret_cast<const float4*>(b0 + col);
        *reinterpret_cast<float4*>(&ha[rg * 16 + fm][col]) = make_float4(relu_nan(acc[nt][0] + bv.x), relu_nan(acc[nt][1] + bv.y),
                                                                          relu_nan(acc[nt][2] + bv.z), relu_nan(acc[nt][3] + bv.w));
    }
    __syncthreads();
    mlp_store_w(wl, wreg);
    wreg = mlp_load_w(W2);
    __syncthreads();
    // ---- layer 1: relu(W1 h + b1) -> hb
    wave_linear(ha + rg * 16, wl, half, fm, kq, acc);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = half * 64 + nt * 16 + kq * 4;
        const float4 bv = *reinterpret_cast<const float4*>(b1 + col);
        *reinterpret_cast<float4*>(&hb[rg * 16 + fm][col]) = make_float4(relu_nan(acc[nt][0] + bv.x), relu_nan(acc[nt][1] + bv.y),
                                                                          relu_nan(acc[nt][2] + bv.z), relu_nan(acc[nt][3] + bv.w));
    }
    __syncthreads();
    mlp_store_w(wl, wreg);
    __syncthreads();
    // ---- layer 2 + residual -> ha, then LayerNorm2
    wave_linear(hb + rg * 16, wl, half, fm, kq, acc);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = half * 64 + nt * 16 + kq * 4;
        const float4 bv = *reinterpret_cast<const float4*>(b2 + col);
        const float4 yv = *reinterpret_cast<const float4*>(&ys[rg * 16 + fm][col]);
        *reinterpret_cast<float4*>(&ha[rg * 16 + fm][col]) = make_float4(yv.x + (acc[nt][0] + bv.x), yv.y + (acc[nt][1] + bv.y),
                                                                          yv.z + (acc[nt][2] + bv.z), yv.w + (acc[nt][3] + bv.w));
    }
    __syncthreads();
    {   // LayerNorm2, the 8 rows of the wave as independent reduction chains
        const float2 g = reinterpret_cast<const float2*>(g2)[lane], bt = reinterpret_cast<const float2*>(be2)[lane];
        float2 v[RW];
        float mean[RW], var[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) { v[rr] = *reinterpret_cast<const float2*>(&ha[wave * RW + rr][2 * lane]); mean[rr] = wave_sum(v[rr].x + v[rr].y) * (1.f / F); }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) { v[rr].x -= mean[rr]; v[rr].y -= mean[rr]; var[rr] = wave_sum(v[rr].x * v[rr].x + v[rr].y * v[rr].y) * (1.f / F); }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int64_t row = row0 + wave * RW + rr;
            const float sd = sqrtf(var[rr] + 1e-10f);
            if (row < rows) reinterpret_cast<float2*>(out + row * F)[lane] = make_float2(v[rr].x / sd * g.x + bt.x, v[rr].y / sd * g.y + bt.y);
        }
    }
}

int launch_fused_ln_mlp(const float* x, const float* u, int nslab, int64_t slab_stride, const float* ubias, const uint8_t* mask,
                        const float* g1, const float* be1, const float* W0, const float* b0, const float* W1, const float* b1,
                        const float* W2, const float* b2, const float* g2, const float* be2, float* out, int64_t rows, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
#define ABOPT_MLP_LAUNCH(NS)                                                                                                  \
    hipLaunchKernelGGL(fused_ln_mlp_kernel<NS>, dim3((unsigned)((rows + MR - 1) / MR)), dim3(256), 0, st, x, u, slab_stride, ubias, mask, \
                       g1, be1, W0, b0, W1, b1, W2, b2, g2, be2, out, rows)
    switch (nslab) {
        case 1: ABOPT_MLP_LAUNCH(1); break;
        case 2: ABOPT_MLP_LAUNCH(2); break;
        case 4: ABOPT_MLP_LAUNCH(4); break;
        case 8: ABOPT_MLP_LAUNCH(8); break;
        default: ABOPT_CHECK_ARG(false, "fused_ln_mlp: unsupported slab count %d", nslab);
    }
#undef ABOPT_MLP_LAUNCH
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}


// =====================================================================================================================
// out_transform + tail in ONE launch:  out = LN2(y + MLP(y)),  y = LN1(x + mask * (feat W_out^T + b_out))      (ga.py:174-177)
// Replaces the split-K out_transform GEMM (two 4 MB partial slabs written and re-read) + fused_ln_mlp.  One 1024-thread workgroup
// (16 waves, 4 per SIMD) owns 32 residues.  Every contraction runs on the bf16 matrix pipe with exact three-term splits of both fp32
// operands (six v_mfma_f32_32x32x16_bf16 per 16 k; node_frags.hip explains the arithmetic); all weights arrive in MFMA operand
// order (packed once), straight from L2 into registers -- no weight staging through LDS.  W_out streams as fp32 and is split in
// registers (2/3 of the bytes of the pre-split form: 41.9 -> 39.6 us on the same box), the small MLP layers arrive pre-split.
// Phase 1: u[32,128] = feat[32,1824] . W_out^T.  Wave (cb, kg) owns output columns 32 cb .. 32 cb + 31 of all 32 rows for k-steps
// 3 kg .. 3 kg + 2 of every 192-column chunk (operands one chunk ahead in registers); the feat chunk is split ONCE by the loader
// threads and staged in LDS as three bf16 planes (double buffered, one barrier per chunk) shared by all waves.
// Phase 2: LayerNorm1, three 128x128 layers, LayerNorm2 with the 32 rows in LDS: layer inputs as bf16 planes, wave (cb, kg) takes
// two of the eight k-steps; the four K groups of a phase meet in LDS and one pass adds bias, applies relu and re-splits.
// [Measured on the way here, per layer of the sampler at M = 8192: split-K GEMM + fused_ln_mlp 66 us; this kernel with fp32 MFMA
//  45 us; as it stands 41 us = 4k + 47k + 22k cycles (prologue, phase 1, phase 2).  Phase 1 without its weight stream runs at the
//  matrix-pipe bound (22k cycles); with it, it is bound by L2 -> CU bandwidth: every CU has to pull all of W_out (1.4 MB as bf16
//  terms) for its 32 rows and sustains ~35 B/clk while all 256 CUs do the same.  Streaming fp32 weights and splitting them in
//  registers moves 1/3 fewer bytes but pays it back in VALU work: 40k cycles, same wall time.]
#ifndef OT_ABL     // developer ablation mask (results are wrong when non-zero): 1 no W stream, 2 no feat staging, 8 no LDS operand reads
#define OT_ABL 0
#endif
#ifdef OT_TIMING   // developer build: section clocks of one workgroup, printed by the launcher
}  // namespace abopt
#include <cstdio>
__device__ long long g_ot_timing[16][8];
namespace abopt {
#endif

// DUMP (training): also writes what the backward needs, five [rows,128] slabs: pre-LayerNorm1 sum | y | h0 | h1 | pre-LayerNorm2 sum
//
// Schedule (round 3; -DOT_TIMING clocks: prologue 3.5k + phase 1 42k + phase 2 21k cycles before, at EVERY batch size -- the kernel was
// bound by its own issue order, not by the W_out stream):
//  phase 1  a wave's k-steps form one software pipeline across the chunk barriers: while the six MFMAs of k-step i issue, the VALU splits
//           the fp32 W_out fragment of k-step i + 1 into its bf16 terms (it used to do that right after every barrier, all 16 waves at
//           once, with the matrix pipe idle) and the loader threads split / stage the next feat chunk; raw W fragments are requested
//           three k-steps ahead.
//  phase 2  each of the 16 waves owns a 16 x 16 output tile of a layer over the full K = 128 (v_mfma_f32_16x16x32_bf16, the layer's
//           weights in 48 registers, requested while the previous layer finishes): no K-group partial sums, one barrier per layer
//           instead of two, activations ping-pong between two sets of bf16 planes.
template <bool DUMP>
__global__ __launch_bounds__(OT_TH) void out_ln_mlp_kernel(const float* __restrict__ feat, const float* __restrict__ wof /* W_out terms, operand order */,
                                                           const float* __restrict__ wmf /* W_mlp0..2 terms, 16x16x32 operand order */,
                                                           const float* __restrict__ x, const float* __restrict__ ubias, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ g1, const float* __restrict__ be1,
                                                           const float* __restrict__ b0, const float* __restrict__ b1, const float* __restrict__ b2,
                                                           const float* __restrict__ g2, const float* __restrict__ be2,
                                                           float* __restrict__ out, float* __restrict__ dump, int64_t rows, unsigned* __restrict__ xt_out) {
    extern __shared__ __attribute__((aligned(16))) char ot_raw[];
    OtSmem& sm = *reinterpret_cast<OtSmem*>(ot_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t slab = rows * F;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * MR;
#ifdef OT_TIMING
    const long long tc0 = clock64();
#endif
    // ---------------------------------------------------------------- phase 1: u = feat . W_out^T
    // The ten 192-column chunks are taken in the order ot_chunk_at(0..9) = 4 5 0 1 2 3 6 7 8 9 (tail_common.h: the order in which the
    // fused core + tail kernel can produce them); position p lives in staging buffer p & 1.
    // feat chunk loader: 32 rows x 24 octets of 8 floats: thread e < 768 -> (row e / 24, octet e % 24); split, then three 16-byte stores
    const int lr = min(tid / (OT_KC / 8), MR - 1), lo = tid % (OT_KC / 8);
    const bool ldr = tid < MR * (OT_KC / 8);
    const float* frow = feat + min(row0 + lr, rows - 1) * OT_K;                                           // staging columns 8 lo .. 8 lo + 7 of a chunk = two quads of feature columns (ot_feat_col)
    auto fload = [&](int c, f32x4& v0, f32x4& v1) {
        v0 = *reinterpret_cast<const f32x4*>(frow + ot_feat_col(c, lo * 8)); v1 = *reinterpret_cast<const f32x4*>(frow + ot_feat_col(c, lo * 8 + 4));
    };
    char* fdst0 = &sm.stage[0][0] + lr * OT_SROW + lo * 16;
    auto chunk_ok = [&](int p) { return ldr && p < OT_NCH && ot_chunk_at(p) * OT_KC + lo * 8 < OT_K; };   // the last chunk is half full
    auto stage_store = [&](int b, const f32x4& v0, const f32x4& v1) {
        const Split2 sp = split2(v0, v1);
        char* d = fdst0 + b * OT_STAGE;
        *reinterpret_cast<u32x4*>(d) = sp.h; *reinterpret_cast<u32x4*>(d + OT_PLANE) = sp.l;
    };
    const int cb = wave & 3, kg = wave >> 2;
    // this wave's W_out stream: the pre-split, scaled terms in operand order, [cb][k-step][term][lane] x 8 fp16, lane (column lane & 31, k half lane >> 5).
    // Its k-steps, in order: position p, slot j -> k-step 12 chunk(p) + 3 kg + j (p < 10, j < 3; past 113: nothing to do -- a clamped reload keeps the code uniform)
    const u32x4* wfr = reinterpret_cast<const u32x4*>(wof) + (int64_t)cb * OT_ST * OT_WVEC + lane;
    auto kstep = [&](int i) { return min(ot_chunk_at(min(i / OT_SPW, OT_NCH - 1)) * OT_SPC + kg * OT_SPW + (i % OT_SPW), OT_ST - 1); };
    constexpr int WRD = 2 * OT_SPW;                                                                       // ring: k-step i lives in slot i % 6, requested six k-steps (two chunks) ahead
    u32x4 wr[WRD][2];
#pragma unroll
    for (int j = 0; j < WRD; ++j) { wr[j][0] = wfr[(int64_t)kstep(j) * OT_WVEC]; wr[j][1] = wfr[(int64_t)kstep(j) * OT_WVEC + 64]; }
    f32x4 fv0 = (f32x4){0.f, 0.f, 0.f, 0.f}, fv1 = fv0;
    if (chunk_ok(0)) {
        fload(ot_chunk_at(0), fv0, fv1);
        stage_store(0, fv0, fv1);
    }
    if (chunk_ok(1)) fload(ot_chunk_at(1), fv0, fv1);
    __syncthreads();
#ifdef OT_TIMING
    const long long tc1 = clock64();
#endif
    f32x16 acc0;                                                                                          // ONE chain per (column block, K group): see ot_kstep3
    acc_zero(acc0);
    const char* xrd = &sm.stage[0][0] + (lane & 31) * OT_SROW + (kg * OT_SPW) * 32 + (lane >> 5) * 16;
    auto chunk_steps = [&](int c, int half) {                                                            // half = c & 1 as a compile-time constant: ring slots 3 half + j
        const int b = c & 1;
        // the last position (chunk 9) has work for K groups 0 and 1 only (and nothing left to stage): the others go straight to the barrier
        if (ot_chunk_at(c) * OT_SPC + kg * OT_SPW < OT_ST) {
#pragma unroll
        for (int j = 0; j < OT_SPW; ++j) {
            const int i = c * OT_SPW + j, slot = half * OT_SPW + j;
            const char* xp = xrd + b * OT_STAGE + j * 32;
            const u32x4 xh = *reinterpret_cast<const u32x4*>(xp), xl = *reinterpret_cast<const u32x4*>(xp + OT_PLANE);
            ot_kstep3(wr[slot][0], wr[slot][1], xh, xl, acc0);
            { const int64_t nst = kstep(i + WRD); wr[slot][0] = wfr[nst * OT_WVEC]; wr[slot][1] = wfr[nst * OT_WVEC + 64]; }
            if (j == 0) {                                                                               // feat staging in the shadow of this chunk's MFMAs
                if (chunk_ok(c + 1)) stage_store(b ^ 1, fv0, fv1);                                      // position c + 1 -> the other buffer
                if (chunk_ok(c + 2)) fload(ot_chunk_at(c + 2), fv0, fv1);
            }
        }
        }
        __syncthreads();
    };
    static_assert(OT_NCH % 2 == 0, "two positions per trip");
    for (int c = 0; c < OT_NCH; c += 2) { chunk_steps(c, 0); chunk_steps(c + 1, 1); }
#ifdef OT_TIMING
    const long long tc2 = clock64();
#endif
    // ---------------------------------------------------------------- phase 2: LayerNorm1 (each wave 2 rows), MLP, LayerNorm2 (tail_common.h)
    TailP2Pre<OT_NW> pre = tail_p2_prefetch<OT_NW>(x, ubias, mask, g1, be1, wmf, row0, rows, wave, lane);
    store_partial1(sm.part[kg], acc0, cb, lane);                                                         // the staging planes are dead: the loop ended on a barrier
    tail_p2_stage_bias(sm.bias, b0, b1, b2, tid);
    __syncthreads();
    char* apA = sm.ap;                                                                                   // planes: LayerNorm1 output, later layer-1 output
    char* apB = reinterpret_cast<char*>(&sm.part[2][0][0]);                                              // second set (layer-0 output): part[2..3] are free after LayerNorm1
    static_assert(OT_NT * AP_PLANE <= (int)(2 * MR * XLD * sizeof(float)), "second activation plane set must fit into two K-group slabs");
    auto get_u = [&](int rl) {
        const float2 u0 = *reinterpret_cast<const float2*>(&sm.part[0][rl][2 * lane]), u1 = *reinterpret_cast<const float2*>(&sm.part[1][rl][2 * lane]);
        const float2 u2 = *reinterpret_cast<const float2*>(&sm.part[2][rl][2 * lane]), u3 = *reinterpret_cast<const float2*>(&sm.part[3][rl][2 * lane]);
        return make_float2((u0.x + u1.x) + (u2.x + u3.x), (u0.y + u1.y) + (u2.y + u3.y));
    };
    tail_p2_run<OT_NW, DUMP>(pre, get_u, sm.ys, sm.bias, apA, apB, wmf, g2, be2, out, dump, slab, row0, rows, wave, lane, xt_out);
#ifdef OT_TIMING
    if (blockIdx.x == 17 && lane == 0) { long long* o = g_ot_timing[wave]; o[0] = tc1 - tc0; o[1] = tc2 - tc1; o[2] = clock64() - tc2; o[3] = 0; o[4] = 0; o[5] = 0; }
#endif
}


size_t out_wfrag_floats() { return (size_t)F * OT_K; }
size_t out_wterms_floats() { return (size_t)F * OT_K; }

// w_out_terms = w_out_frag: since round 5 pack_tail_weights writes the terms the fused core + tail kernel streams itself, and the stand-alone tail
// reads the same layout; the entry point stays (callers keep one buffer per role) and copies.
__global__ __launch_bounds__(256) void out_frag_terms_kernel(const float* __restrict__ wof, float* __restrict__ wot) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id < F * OT_K / 4) reinterpret_cast<f32x4*>(wot)[id] = reinterpret_cast<const f32x4*>(wof)[id];
}
int launch_out_frag_terms(const float* wof, float* wot, hipStream_t st) {
    if (wof == wot) return ABOPT_OK;
    hipLaunchKernelGGL(out_frag_terms_kernel, dim3((F * OT_K / 4 + 255) / 256), dim3(256), 0, st, wof, wot);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}
size_t mlp_wfrag_floats() { return (size_t)3 * F * F * 3 / 2; }

template <bool DUMP>
static int launch_out_ln_mlp_t(const float* feat, const float* wof, const float* wmf, const float* x, const float* ubias, const uint8_t* mask,
                               const float* g1, const float* be1, const float* b0, const float* b1, const float* b2, const float* g2, const float* be2,
                               float* out, float* dump, int64_t rows, hipStream_t st, float* xt_out) {
    static LdsConfig lds_cfg;                                               // per instantiation
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(out_ln_mlp_kernel<DUMP>), sizeof(OtSmem), lds_cfg)) return rc;
    hipLaunchKernelGGL(out_ln_mlp_kernel<DUMP>, dim3((unsigned)((rows + MR - 1) / MR)), dim3(OT_TH), sizeof(OtSmem), st, feat, wof, wmf, x, ubias, mask,
                       g1, be1, b0, b1, b2, g2, be2, out, dump, rows, reinterpret_cast<unsigned*>(xt_out));
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

int launch_out_ln_mlp(const float* feat, const float* wof, const float* wmf, const float* x, const float* ubias, const uint8_t* mask,
                      const float* g1, const float* be1, const float* b0, const float* b1, const float* b2, const float* g2, const float* be2,
                      float* out, float* dump, int64_t rows, hipStream_t st, float* xt_out) {
    if (rows == 0) return ABOPT_OK;
    const int rc = dump ? launch_out_ln_mlp_t<true>(feat, wof, wmf, x, ubias, mask, g1, be1, b0, b1, b2, g2, be2, out, dump, rows, st, xt_out)
                        : launch_out_ln_mlp_t<false>(feat, wof, wmf, x, ubias, mask, g1, be1, b0, b1, b2, g2, be2, out, dump, rows, st, xt_out);
#ifdef OT_TIMING
    {
        long long hh[16][8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_ot_timing), sizeof(hh));
        static int calls = 0;
        if (++calls == 8)
            for (int w = 0; w < 16; w += 5) fprintf(stderr, "[ot timing WG 17 wave %d] prologue %lld | phase 1 loop %lld | phase 2 %lld (partials + LayerNorm1 %lld, layer 0 %lld, layers 1 + 2 %lld)\n", w, hh[w][0], hh[w][1], hh[w][2], hh[w][3], hh[w][4], hh[w][5]);
    }
#endif
    return rc;
}

// =====================================================================================================================
// Weights of the tail in MFMA operand order, packed on the device (two launches per block).  Forward operands are TWO fp16 terms of S w, S a power
// of two per matrix with max |w| S in [2^14, 2^15) (ipa_common.h: split_pair2):
//   W_out     -> wof [4 cb][114 k-steps][term h | l][64 lanes] x 8 fp16 (32x32x16 operand order, K in staging order ot_feat_col),
//   W_mlp0..2 -> wmf [3][8 ct][4 k-steps][term][64 lanes] x 8 fp16 (16x16x32 operand order), then at float OT_SCALE_OFF the four S and the four 1 / S
//                (then 4 x 64 slice maxima, scratch of the two packing kernels; the rest of the buffer, which keeps the size of wmt, is zeroed),
// and the TRANSPOSED MLP weights -> wmt [3][4][8][3][64] x 8 bf16 (32x32x16 operand order, three bf16 terms: operands of the backward chain).
// One thread per (matrix, block, step, lane): 8 values -> two or three 16-byte vectors.  Layout: include/abopt.h (w_out_frag, w_mlp_frag).
// Stage 1 of the scales: |w| maxima of 64 slices per matrix -> scales[8 + 64 mtx + slice] (scratch behind the eight scale slots; one workgroup per
// matrix took 107 us for W_out -- 0.64 ms of a training step, which packs six blocks per step).  The pack kernel folds them.
constexpr int OT_NSL = 64;
__global__ __launch_bounds__(256) void tail_weight_absmax_kernel(const float* __restrict__ w_out, const float* __restrict__ w0, const float* __restrict__ w1,
                                                                 const float* __restrict__ w2, float* __restrict__ scales, int tail_floats) {
    __shared__ float red[4];
    const int mtx = blockIdx.y, slice = blockIdx.x;
    for (int e = 8 + 4 * OT_NSL + (mtx * OT_NSL + slice) * 256 + threadIdx.x; e < tail_floats; e += 4 * OT_NSL * 256) scales[e] = 0.f;      // the unused rest of w_mlp_frag
    const float* src = mtx == 0 ? w_out : (mtx == 1 ? w0 : (mtx == 2 ? w1 : w2));
    const int n4 = (mtx == 0 ? F * OT_K : F * F) / 4, per = (n4 + OT_NSL - 1) / OT_NSL;
    float m = 0.f;
    for (int e = slice * per + threadIdx.x; e < min(n4, (slice + 1) * per); e += 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[e];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float a = fabsf(v[i]); m = (a <= 3.0e38f) ? fmaxf(m, a) : m; }      // NaN / inf do not set the scale
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) scales[8 + mtx * OT_NSL + slice] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// max |w| -> the exponent e of S = 2^e: m = f 2^(15 - e), f in [0.5, 1)  ->  m 2^e in [2^14, 2^15); 0 for an all-zero matrix
__device__ __forceinline__ int tail_scale_exp(float m) {
    int e = 0;
    if (m > 0.f) { (void)frexpf(m, &e); e = 15 - e; }
    return max(-100, min(100, e));
}

__global__ __launch_bounds__(256) void pack_tail_weights_kernel(const float* __restrict__ w_out, const float* __restrict__ w0, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, float* __restrict__ wof, float* __restrict__ wmf,
                                                                float* __restrict__ wmt) {
    constexpr int NOUT = 4 * OT_ST * 64, NMLP = 4 * OT_MS * 64;          // NMLP = 2048 = 8 ct * 4 k-steps * 64 lanes as well
    int id = blockIdx.x * 256 + threadIdx.x;
    __shared__ float scales[4];
    {   // wave w folds the 64 slice maxima of matrix w; workgroup 0 publishes S and 1 / S
        const int w = threadIdx.x >> 6;
        const float m = wave_max(wmf[OT_SCALE_OFF + 8 + w * OT_NSL + (threadIdx.x & 63)]);
        if ((threadIdx.x & 63) == 0) {
            const int e = tail_scale_exp(m);
            scales[w] = ldexpf(1.f, e);
            if (blockIdx.x == 0) { wmf[OT_SCALE_OFF + w] = ldexpf(1.f, e); wmf[OT_SCALE_OFF + 4 + w] = ldexpf(1.f, -e); }
        }
    }
    __syncthreads();
    if (id < NOUT) {
        const int lane = id & 63, st = (id >> 6) % OT_ST, cb = (id >> 6) / OT_ST;
        const float* p = w_out + (int64_t)(cb * 32 + (lane & 31)) * OT_K;
        const int kk = st * 16 + (lane >> 5) * 8, c = kk / OT_KC, j = kk % OT_KC;      // K index in staging order -> feature columns (two quads)
        const float S = scales[0];
        const Split2 sp = split2(*reinterpret_cast<const f32x4*>(p + ot_feat_col(c, j)) * S, *reinterpret_cast<const f32x4*>(p + ot_feat_col(c, j + 4)) * S);
        u32x4* d = reinterpret_cast<u32x4*>(wof) + (int64_t)(cb * OT_ST + st) * OT_WVEC + lane;
        d[0] = sp.h; d[64] = sp.l;
        return;
    }
    id -= NOUT;
    if (id >= 6 * NMLP) return;
    const int m = id / NMLP; id %= NMLP;
    const int layer = m % 3; const bool tr = m >= 3;
    if (tr && !wmt) return;
    const float* src = layer == 0 ? w0 : (layer == 1 ? w1 : w2);
    const int lane = id & 63;
    if (!tr) {                                                           // forward: [layer][ct][s][term][lane (m, kq)] x 8 = terms of S W[16 ct + m][32 s + 8 kq + i]
        const int st = (id >> 6) % 4, ct = (id >> 6) / 4;
        const float* p = src + (int64_t)(ct * 16 + (lane & 15)) * F + st * 32 + (lane >> 4) * 8;
        const float S = scales[1 + layer];
        const Split2 sp = split2(*reinterpret_cast<const f32x4*>(p) * S, *reinterpret_cast<const f32x4*>(p + 4) * S);
        u32x4* d = reinterpret_cast<u32x4*>(wmf) + (int64_t)((layer * 8 + ct) * 4 + st) * OT_WVEC + lane;
        d[0] = sp.h; d[64] = sp.l;
        return;
    }
    // backward: [cb][s][term][lane (c, kh)] = W[16 s + 8 kh + i][32 cb + c]
    const int st = (id >> 6) % OT_MS, cb = (id >> 6) / OT_MS;
    const float* p = src + (int64_t)(st * 16 + (lane >> 5) * 8) * F + cb * 32 + (lane & 31);
    f32x4 lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[i] = p[(int64_t)i * F]; hi[i] = p[(int64_t)(i + 4) * F]; }
    u32x4* d = reinterpret_cast<u32x4*>(wmt) + layer * (NMLP * 3) + ((int64_t)(cb * OT_MS + st) * 3) * 64 + lane;
    const Split3 sp = split3(lo, hi);
    d[0] = sp.h; d[64] = sp.m; d[128] = sp.l;
}

int launch_pack_tail_weights(const float* w_out, const float* w0, const float* w1, const float* w2, float* wof, float* wmf, float* wmt, hipStream_t st) {
    // slice maxima first (the pack kernel folds them into the scales)
    hipLaunchKernelGGL(tail_weight_absmax_kernel, dim3(OT_NSL, 4), dim3(256), 0, st, w_out, w0, w1, w2, wmf + OT_SCALE_OFF, (int)(mlp_wfrag_floats() - OT_SCALE_OFF));
    ABOPT_LAUNCH_CHECK();
    const int total = 4 * OT_ST * 64 + 6 * 4 * OT_MS * 64;
    hipLaunchKernelGGL(pack_tail_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w_out, w0, w1, w2, wof, wmf, wmt);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// =====================================================================================================================
// Backward of the tail for 32 rows per workgroup (training): everything that is row-local in
//     out = LN2(r),  r = y + W2 h1 + b2,  h1 = relu(W1 h0 + b1),  h0 = relu(W0 y + b0),  y = LN1(a1),  a1 = x + mask * u
// from d out and the five slabs the forward dumped (a1 | y | h0 | h1 | r).  Writes d pre-activations of the three layers (the weight
// gradients are tall GEMMs on them, left to the caller), d a1 (= d x through the residual), d u = mask * d a1 (operand of the two
// out_transform GEMMs) and the column sums of this tile for the eight vector gradients
//     0 d beta2 | 1 d gamma2 | 2 d b2 | 3 d b1 | 4 d b0 | 5 d beta1 | 6 d gamma1 | 7 d b_out
// as per-workgroup partials [tiles][8][128] (summed by the caller: deterministic, no atomics).  The three chain products
// d h = d pre . W run on the bf16 matrix pipe like the forward, with the transposed weights packed by pack_tail_weights.
namespace {
struct TbSmem {
    char ap[3 * AP_PLANE];
    float part[4][MR][XLD];                   // K-group partials; also scratch for the column sums of the row-wise phases
    float drs[MR][XLD];                       // d r, later d y
    float scr[MR][XLD];
};
// column sums of up to three [32][XLD] arrays: thread t < 128 nq -> (array t >> 7, column t & 127); quantity ids q0, q0 + 1, ...
__device__ __forceinline__ void column_sums(const float (*a0)[XLD], const float (*a1)[XLD], const float (*a2)[XLD], int q0, int nq,
                                            float* __restrict__ colpart, int tid) {
    if (tid < 128 * nq) {
        const int q = tid >> 7, col = tid & 127;
        const float (*a)[XLD] = q == 0 ? a0 : (q == 1 ? a1 : a2);
        float sacc = 0.f;
#pragma unroll 8
        for (int r = 0; r < MR; ++r) sacc += a[r][col];
        colpart[(q0 + q) * F + col] = sacc;
    }
}
}  // namespace

__global__ __launch_bounds__(OT_TH) void tail_backward_kernel(const float* __restrict__ dout, const float* __restrict__ saved, const float* __restrict__ wmt,
                                                              const uint8_t* __restrict__ mask, const float* __restrict__ g1, const float* __restrict__ g2,
                                                              float* __restrict__ dpre, float* __restrict__ da1, float* __restrict__ du,
                                                              float* __restrict__ colpart, int64_t rows) {
    extern __shared__ __attribute__((aligned(16))) char tb_raw[];
    TbSmem& sm = *reinterpret_cast<TbSmem*>(tb_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * MR, slab = rows * F;
    const int cb = wave & 3, kg = wave >> 2;
    float* cp = colpart + (int64_t)blockIdx.x * 8 * F;
    constexpr int RW = MR / OT_NW;
    MlpW mw = load_mlp_w(wmt, 2, cb, kg, lane);
    // ---- LayerNorm2 backward (each wave 2 rows): d r = (g2 d out - mean(.) - xhat mean(. xhat)) / sigma
    {
        const float2 g = reinterpret_cast<const float2*>(g2)[lane];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int rl = wave * RW + rr;
            const bool live = row0 + rl < rows;
            const int64_t row = min(row0 + rl, rows - 1);
            float2 rv = reinterpret_cast<const float2*>(saved + 4 * slab + row * F)[lane];
            float2 dv = reinterpret_cast<const float2*>(dout + row * F)[lane];
            if (!live) dv = make_float2(0.f, 0.f);
            const float mean = wave_sum(rv.x + rv.y) * (1.f / F);
            rv.x -= mean; rv.y -= mean;
            const float rsd = 1.f / sqrtf(wave_sum(rv.x * rv.x + rv.y * rv.y) * (1.f / F) + 1e-10f);
            const float2 xh = make_float2(rv.x * rsd, rv.y * rsd);
            const float2 dh = make_float2(dv.x * g.x, dv.y * g.y);
            const float m1 = wave_sum(dh.x + dh.y) * (1.f / F), m2 = wave_sum(dh.x * xh.x + dh.y * xh.y) * (1.f / F);
            const float2 dr = make_float2((dh.x - m1 - xh.x * m2) * rsd, (dh.y - m1 - xh.y * m2) * rsd);
            *reinterpret_cast<float2*>(&sm.drs[rl][2 * lane]) = dr;
            *reinterpret_cast<float2*>(&sm.part[0][rl][2 * lane]) = dv;
            *reinterpret_cast<float2*>(&sm.part[1][rl][2 * lane]) = make_float2(dv.x * xh.x, dv.y * xh.y);
            store_terms2(sm.ap, rl * AP_ROW + lane * 4, dr.x, dr.y);
            if (live) reinterpret_cast<float2*>(dpre + 2 * slab + row * F)[lane] = dr;          // d pre-activation of layer 2 (no relu)
        }
    }
    __syncthreads();
    column_sums(sm.part[0], sm.part[1], sm.drs, 0, 3, cp, tid);
    __syncthreads();
    const int er = tid >> 5, ec = (tid & 31) * 4;
    const bool elive = row0 + er < rows;
    const int64_t erow = min(row0 + er, rows - 1);
    auto gather = [&]() {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(&sm.part[0][er][ec]), p1 = *reinterpret_cast<const f32x4*>(&sm.part[1][er][ec]);
        const f32x4 p2 = *reinterpret_cast<const f32x4*>(&sm.part[2][er][ec]), p3 = *reinterpret_cast<const f32x4*>(&sm.part[3][er][ec]);
        return (p0 + p1) + (p2 + p3);
    };
    // ---- d h1 = d pre2 . W2, relu mask -> d pre1 ; d h0 = d pre1 . W1, relu mask -> d pre0
#pragma unroll
    for (int layer = 1; layer >= 0; --layer) {
        mlp_partial(sm.ap, mw, sm.part[kg], cb, kg, lane);
        mw = load_mlp_w(wmt, layer, cb, kg, lane);
        __syncthreads();
        {
            f32x4 v = gather();
            const f32x4 hv = *reinterpret_cast<const f32x4*>(saved + (2 + layer) * slab + erow * F + ec);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = hv[i] > 0.f ? v[i] : 0.f;
            if (!elive) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            store_terms2(sm.ap, er * AP_ROW + ec * 2, v[0], v[1]);
            store_terms2(sm.ap, er * AP_ROW + ec * 2 + 4, v[2], v[3]);
            *reinterpret_cast<f32x4*>(&sm.scr[er][ec]) = v;
            if (elive) *reinterpret_cast<f32x4*>(dpre + layer * slab + erow * F + ec) = v;
        }
        __syncthreads();
        column_sums(sm.scr, sm.scr, sm.scr, layer == 1 ? 3 : 4, 1, cp, tid);
    }
    // ---- d y = d r + d pre0 . W0
    mlp_partial(sm.ap, mw, sm.part[kg], cb, kg, lane);
    __syncthreads();
    {
        const f32x4 v = gather();
        f32x4 d = *reinterpret_cast<const f32x4*>(&sm.drs[er][ec]);
        d += v;
        *reinterpret_cast<f32x4*>(&sm.drs[er][ec]) = d;
    }
    __syncthreads();
    // ---- LayerNorm1 backward (each wave 2 rows) -> d a1, d u
    {
        const float2 g = reinterpret_cast<const float2*>(g1)[lane];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int rl = wave * RW + rr;
            const bool live = row0 + rl < rows;
            const int64_t row = min(row0 + rl, rows - 1);
            float2 av = reinterpret_cast<const float2*>(saved + row * F)[lane];
            const float2 dv = *reinterpret_cast<const float2*>(&sm.drs[rl][2 * lane]);
            const float mean = wave_sum(av.x + av.y) * (1.f / F);
            av.x -= mean; av.y -= mean;
            const float rsd = 1.f / sqrtf(wave_sum(av.x * av.x + av.y * av.y) * (1.f / F) + 1e-10f);
            const float2 xh = make_float2(av.x * rsd, av.y * rsd);
            const float2 dh = make_float2(dv.x * g.x, dv.y * g.y);
            const float m1 = wave_sum(dh.x + dh.y) * (1.f / F), m2 = wave_sum(dh.x * xh.x + dh.y * xh.y) * (1.f / F);
            const float2 da = make_float2((dh.x - m1 - xh.x * m2) * rsd, (dh.y - m1 - xh.y * m2) * rsd);
            const bool keep = mask ? (mask[row] != 0) : true;
            const float2 duv = keep ? da : make_float2(0.f, 0.f);
            *reinterpret_cast<float2*>(&sm.part[0][rl][2 * lane]) = dv;                           // d beta1 (rows past the end hold zeros: d r and d pre0 were zeroed)
            *reinterpret_cast<float2*>(&sm.part[1][rl][2 * lane]) = make_float2(dv.x * xh.x, dv.y * xh.y);
            *reinterpret_cast<float2*>(&sm.part[2][rl][2 * lane]) = live ? duv : make_float2(0.f, 0.f);
            if (live) {
                reinterpret_cast<float2*>(da1 + row * F)[lane] = da;
                reinterpret_cast<float2*>(du + row * F)[lane] = duv;
            }
        }
    }
    __syncthreads();
    column_sums(sm.part[0], sm.part[1], sm.part[2], 5, 3, cp, tid);
}

int launch_tail_backward(const float* dout, const float* saved, const float* wmt, const uint8_t* mask, const float* g1, const float* g2,
                         float* dpre, float* da1, float* du, float* colpart, int64_t rows, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    static LdsConfig lds_cfg;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(tail_backward_kernel), sizeof(TbSmem), lds_cfg)) return rc;
    hipLaunchKernelGGL(tail_backward_kernel, dim3((unsigned)((rows + MR - 1) / MR)), dim3(OT_TH), sizeof(TbSmem), st, dout, saved, wmt, mask, g1, g2,
                       dpre, da1, du, colpart, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

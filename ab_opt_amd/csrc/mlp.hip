// Fused tail of a GABlock for CDNA4: one launch for
//     y   = LayerNorm1(x + mask * (sum_of_split-K_slabs(out_transform) + b_out))
//     out = LayerNorm2(y + W2 relu(W1 relu(W0 y + b0) + b1) + b2)
// (reference AbDock/src/modules/encoders/ga.py:174-177, LayerNorm: AbDock/src/modules/common/layers.py:146-155).
// Replaces two LayerNorm launches and three 128x128 GEMM launches whose operands (8192 x 128 activations) are tiny: the
// work per row block is latency/launch bound, so a 256-thread workgroup keeps its 32 rows on chip (LDS) through all three
// layers and stages each 64 KB weight matrix into LDS once (register-prefetched behind the previous phase).
// fp32 MFMA (v_mfma_f32_16x16x4_f32), exact fp32.
#include "abopt_common.h"
#include "kernels.h"

namespace abopt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int F = 128, XLD = F + 4;

__device__ __forceinline__ f32x4 mfma4m(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int MR = 32;                     // rows per workgroup
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
        *reinterpret_cast<float4*>(&ha[rg * 16 + fm][col]) = make_float4(fmaxf(acc[nt][0] + bv.x, 0.f), fmaxf(acc[nt][1] + bv.y, 0.f),
                                                                          fmaxf(acc[nt][2] + bv.z, 0.f), fmaxf(acc[nt][3] + bv.w, 0.f));
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
        *reinterpret_cast<float4*>(&hb[rg * 16 + fm][col]) = make_float4(fmaxf(acc[nt][0] + bv.x, 0.f), fmaxf(acc[nt][1] + bv.y, 0.f),
                                                                          fmaxf(acc[nt][2] + bv.z, 0.f), fmaxf(acc[nt][3] + bv.w, 0.f));
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
// Replaces the split-K out_transform GEMM (two 4 MB partial slabs written and re-read) + fused_ln_mlp.  One 1024-thread
// workgroup (16 waves, 4 per SIMD) owns 32 residues.  Phase 1: u[32,128] = feat[32,1824] . W_out^T on the matrix cores -- wave
// (ct, kh) owns output columns 16 ct .. 16 ct + 15 of both 16-row tiles for one half of every K chunk; its W_out fragments stream
// from L2 in fragment order (packed once, 1 KB per wave load, one chunk ahead in registers), the feat tile goes through LDS in
// 96-column chunks (double buffered, one barrier per chunk) shared by all waves; the two K halves are summed through LDS.
// Phase 2: the 32 rows stay in LDS through LayerNorm1, the three 128x128 layers and LayerNorm2.
namespace {
constexpr int OT_K = ABOPT_IPA_FEAT;          // 1824
constexpr int OT_KC = 96, OT_NCH = OT_K / OT_KC, OT_LD = OT_KC + 4;      // 19 chunks of 96 columns; +4: rows 4 banks apart
constexpr int OT_G = OT_K / 16;               // 114 groups of 16 k: lane group kq holds k = 16 g + 4 kq + i
constexpr int OT_GPW = OT_KC / 16 / 2;        // 3 groups per wave per chunk
constexpr int OT_TH = 1024;
static_assert(OT_K % OT_KC == 0 && OT_KC % 32 == 0, "out_transform K tiling");

struct OtSmem {
    float fs[2][MR][OT_LD];                   // feat chunks
    float ys[MR][XLD], ha[MR][XLD], hb[MR][XLD];
    float wl[F][XLD];
};

// one dense layer of the tail for 16 waves: wave = (row group rg, column tile c8): 16 rows x 16 output columns
__device__ __forceinline__ f32x4 wave_linear16(const float (*xs)[XLD], const float (*wl)[XLD], int c8, int fm, int kq) {
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;                   // two chains (dependent-MFMA latency)
#pragma unroll
    for (int kb = 0; kb < F / 16; kb += 2) {
        const float4 a0 = *reinterpret_cast<const float4*>(&xs[fm][kb * 16 + kq * 4]);
        const float4 b0 = *reinterpret_cast<const float4*>(&wl[c8 * 16 + fm][kb * 16 + kq * 4]);
        const float4 a1 = *reinterpret_cast<const float4*>(&xs[fm][kb * 16 + 16 + kq * 4]);
        const float4 b1 = *reinterpret_cast<const float4*>(&wl[c8 * 16 + fm][kb * 16 + 16 + kq * 4]);
        acc0 = mfma4m(b0.x, a0.x, acc0); acc1 = mfma4m(b1.x, a1.x, acc1);
        acc0 = mfma4m(b0.y, a0.y, acc0); acc1 = mfma4m(b1.y, a1.y, acc1);
        acc0 = mfma4m(b0.z, a0.z, acc0); acc1 = mfma4m(b1.z, a1.z, acc1);
        acc0 = mfma4m(b0.w, a0.w, acc0); acc1 = mfma4m(b1.w, a1.w, acc1);
    }
    return acc0 + acc1;
}
constexpr int OT_WPT = F * F / 4 / OT_TH;      // float4 weight loads per thread per layer (4)
struct WRegs16 { f32x4 v[OT_WPT]; };
__device__ __forceinline__ WRegs16 mlp_load_w16(const float* __restrict__ W) {
    WRegs16 r;
#pragma unroll
    for (int i = 0; i < OT_WPT; ++i) r.v[i] = reinterpret_cast<const f32x4*>(W)[i * OT_TH + threadIdx.x];
    return r;
}
__device__ __forceinline__ void mlp_store_w16(float (*wl)[XLD], const WRegs16& r) {
#pragma unroll
    for (int i = 0; i < OT_WPT; ++i) {
        const int e = (i * OT_TH + threadIdx.x) * 4;
        *reinterpret_cast<f32x4*>(&wl[e >> 7][e & 127]) = r.v[i];
    }
}
}  // namespace

__global__ __launch_bounds__(OT_TH) void out_ln_mlp_kernel(const float* __restrict__ feat, const float* __restrict__ wof /* W_out in fragment order */,
                                                           const float* __restrict__ x, const float* __restrict__ ubias, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ g1, const float* __restrict__ be1,
                                                           const float* __restrict__ W0, const float* __restrict__ b0,
                                                           const float* __restrict__ W1, const float* __restrict__ b1,
                                                           const float* __restrict__ W2, const float* __restrict__ b2,
                                                           const float* __restrict__ g2, const float* __restrict__ be2,
                                                           float* __restrict__ out, int64_t rows) {
    extern __shared__ __attribute__((aligned(16))) char ot_raw[];
    OtSmem& sm = *reinterpret_cast<OtSmem*>(ot_raw);
    const int tid = threadIdx.x, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * MR;
    // ---------------------------------------------------------------- phase 1: u = feat . W_out^T
    // feat chunk loader: 32 rows x 96 floats = 768 float4: thread e < 768 -> (row e / 24, float4 e % 24)
    const bool ldr = tid < MR * (OT_KC / 4);
    const float* fsrc = feat + min(row0 + tid / (OT_KC / 4), rows - 1) * OT_K + (tid % (OT_KC / 4)) * 4;
    float* fdst0 = &sm.fs[0][min(tid / (OT_KC / 4), MR - 1)][(tid % (OT_KC / 4)) * 4];
    const int ct = wave & 7, kh = wave >> 3;
    const f32x4* wfr = reinterpret_cast<const f32x4*>(wof) + (int64_t)ct * OT_G * 64 + lane;         // this wave's column tile: [g][lane]
    f32x4 wq[OT_GPW];
#pragma unroll
    for (int gl = 0; gl < OT_GPW; ++gl) wq[gl] = wfr[(kh * OT_GPW + gl) * 64];
    f32x4 fv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ldr) { fv = *reinterpret_cast<const f32x4*>(fsrc); *reinterpret_cast<f32x4*>(fdst0) = fv; fv = *reinterpret_cast<const f32x4*>(fsrc + OT_KC); }
    WRegs16 wreg = mlp_load_w16(W0);
    __syncthreads();
    f32x4 accu[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int c = 0; c < OT_NCH; ++c) {
        const int b = c & 1;
#pragma unroll
        for (int gl = 0; gl < OT_GPW; ++gl) {
            const f32x4 a = wq[gl];
            const int gloc = kh * OT_GPW + gl;                                                        // group within the chunk
            wq[gl] = wfr[(int64_t)min((c + 1) * (OT_KC / 16) + gloc, OT_G - 1) * 64];                  // next chunk's fragment of this slot (clamped past the end)
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(&sm.fs[b][fm][gloc * 16 + kq * 4]);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(&sm.fs[b][16 + fm][gloc * 16 + kq * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i) { accu[0] = mfma4m(a[i], x0[i], accu[0]); accu[1] = mfma4m(a[i], x1[i], accu[1]); }
        }
        if (ldr) {
            if (c + 1 < OT_NCH) *reinterpret_cast<f32x4*>(fdst0 + (b ^ 1) * (MR * OT_LD)) = fv;       // chunk c + 1 -> the other buffer
            if (c + 2 < OT_NCH) fv = *reinterpret_cast<const f32x4*>(fsrc + (c + 2) * OT_KC);
        }
        __syncthreads();
    }
    // accumulator row 4 kq + r = output column 16 ct + 4 kq + r, column fm = residue; the two K halves land in ha / hb
    {
        float (*dst)[XLD] = kh ? sm.hb : sm.ha;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) *reinterpret_cast<f32x4*>(&dst[rt * 16 + fm][ct * 16 + kq * 4]) = accu[rt];
    }
    __syncthreads();
    // ---------------------------------------------------------------- phase 2: LayerNorm1 (each wave 2 rows), MLP, LayerNorm2
    constexpr int RW = MR / 16;
    {
        const float2 bb = ubias ? reinterpret_cast<const float2*>(ubias)[lane] : make_float2(0.f, 0.f);
        const float2 g = reinterpret_cast<const float2*>(g1)[lane], bt = reinterpret_cast<const float2*>(be1)[lane];
        float a_[RW], b_[RW], mean[RW], var[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int rl = wave * RW + rr;
            const int64_t row = min(row0 + rl, rows - 1);
            const bool keep = mask ? (mask[row] != 0) : true;
            const float2 xv = reinterpret_cast<const float2*>(x + row * F)[lane];
            const float2 u0 = *reinterpret_cast<const float2*>(&sm.ha[rl][2 * lane]), u1 = *reinterpret_cast<const float2*>(&sm.hb[rl][2 * lane]);
            float2 us = make_float2((u0.x + u1.x) + bb.x, (u0.y + u1.y) + bb.y);
            if (!keep) us = make_float2(0.f, 0.f);
            a_[rr] = xv.x + us.x; b_[rr] = xv.y + us.y;
        }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) mean[rr] = wave_sum(a_[rr] + b_[rr]) * (1.f / F);
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) { a_[rr] -= mean[rr]; b_[rr] -= mean[rr]; var[rr] = wave_sum(a_[rr] * a_[rr] + b_[rr] * b_[rr]) * (1.f / F); }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const float sd = sqrtf(var[rr] + 1e-10f);
            *reinterpret_cast<float2*>(&sm.ys[wave * RW + rr][2 * lane]) = make_float2(a_[rr] / sd * g.x + bt.x, b_[rr] / sd * g.y + bt.y);
        }
    }
    mlp_store_w16(sm.wl, wreg);
    wreg = mlp_load_w16(W1);
    __syncthreads();
    const int rg = wave >> 3, c8 = wave & 7;
    const int col = c8 * 16 + kq * 4;
    // ---- layer 0: relu(W0 y + b0) -> ha   (u in ha / hb was consumed by LayerNorm1 before the barrier above)
    {
        const f32x4 acc = wave_linear16(sm.ys + rg * 16, sm.wl, c8, fm, kq);
        const float4 bv = *reinterpret_cast<const float4*>(b0 + col);
        *reinterpret_cast<float4*>(&sm.ha[rg * 16 + fm][col]) = make_float4(fmaxf(acc[0] + bv.x, 0.f), fmaxf(acc[1] + bv.y, 0.f), fmaxf(acc[2] + bv.z, 0.f), fmaxf(acc[3] + bv.w, 0.f));
    }
    __syncthreads();
    mlp_store_w16(sm.wl, wreg);
    wreg = mlp_load_w16(W2);
    __syncthreads();
    // ---- layer 1: relu(W1 h + b1) -> hb
    {
        const f32x4 acc = wave_linear16(sm.ha + rg * 16, sm.wl, c8, fm, kq);
        const float4 bv = *reinterpret_cast<const float4*>(b1 + col);
        *reinterpret_cast<float4*>(&sm.hb[rg * 16 + fm][col]) = make_float4(fmaxf(acc[0] + bv.x, 0.f), fmaxf(acc[1] + bv.y, 0.f), fmaxf(acc[2] + bv.z, 0.f), fmaxf(acc[3] + bv.w, 0.f));
    }
    __syncthreads();
    mlp_store_w16(sm.wl, wreg);
    __syncthreads();
    // ---- layer 2 + residual -> ha, then LayerNorm2
    {
        const f32x4 acc = wave_linear16(sm.hb + rg * 16, sm.wl, c8, fm, kq);
        const float4 bv = *reinterpret_cast<const float4*>(b2 + col);
        const float4 yv = *reinterpret_cast<const float4*>(&sm.ys[rg * 16 + fm][col]);
        *reinterpret_cast<float4*>(&sm.ha[rg * 16 + fm][col]) = make_float4(yv.x + (acc[0] + bv.x), yv.y + (acc[1] + bv.y), yv.z + (acc[2] + bv.z), yv.w + (acc[3] + bv.w));
    }
    __syncthreads();
    {
        const float2 g = reinterpret_cast<const float2*>(g2)[lane], bt = reinterpret_cast<const float2*>(be2)[lane];
        float2 v[RW];
        float mean[RW], var[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) { v[rr] = *reinterpret_cast<const float2*>(&sm.ha[wave * RW + rr][2 * lane]); mean[rr] = wave_sum(v[rr].x + v[rr].y) * (1.f / F); }
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

size_t out_wfrag_floats() { return (size_t)F * OT_K; }

int launch_out_ln_mlp(const float* feat, const float* wof, const float* x, const float* ubias, const uint8_t* mask,
                      const float* g1, const float* be1, const float* W0, const float* b0, const float* W1, const float* b1,
                      const float* W2, const float* b2, const float* g2, const float* be2, float* out, int64_t rows, hipStream_t st) {
    if (rows == 0) return ABOPT_OK;
    static bool configured = false;
    if (!configured) {
        ABOPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(out_ln_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(OtSmem)));
        configured = true;
    }
    hipLaunchKernelGGL(out_ln_mlp_kernel, dim3((unsigned)((rows + MR - 1) / MR)), dim3(OT_TH), sizeof(OtSmem), st, feat, wof, x, ubias, mask,
                       g1, be1, W0, b0, W1, b1, W2, b2, g2, be2, out, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

// Invariant-point-attention core for CDNA4 (gfx950).
//
// Replaces, fused in one pass over the pair features z (reference AbDock/src/modules/encoders/ga.py):
//   _node_logits :81-86, _pair_logits :88-90, _spatial_logits :92-112, _alpha_from_logits :11-26,
//   _pair_aggregation :114-118, _node_aggregation :120-125, _spatial_aggregation :127-147.
// The reference materialises (N,L,L,12,{24,32,64}) temporaries; here z[n,i,:,:] is read from HBM
// exactly once per (n,i) and everything else stays on chip.
#include "abopt_common.h"
#include "kernels.h"
#include <vector>
#include <utility>

namespace abopt {

constexpr int H = ABOPT_HEADS, D = ABOPT_QK_DIM, P = ABOPT_POINTS, C = 64;
constexpr int NP = ABOPT_NODE_PROJ;          // 2016 floats per residue
constexpr int OFF_Q = 0, OFF_K = H * D, OFF_V = 2 * H * D, OFF_QP = 3 * H * D, OFF_KP = OFF_QP + H * P * 3, OFF_VP = OFF_KP + H * P * 3;
constexpr int FEAT = ABOPT_IPA_FEAT;         // 1824

// ------------------------------------------------------------------ local -> global of the point sets
// geometry.py:72-91 applied to proj_{query,key,value}_point outputs (ga.py:96-105,129-132): p <- R p + t, in place.
__global__ __launch_bounds__(256) void points_to_global_kernel(float* __restrict__ proj, const float* __restrict__ R,
                                                               const float* __restrict__ t, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (row, point)
    const int npts = 3 * H * P;                                              // 288 points per residue
    if (idx >= rows * npts) return;
    const int64_t row = idx / npts;
    const int pt = (int)(idx % npts);
    float* p = proj + row * NP + OFF_QP + pt * 3;
    const float* Rr = R + row * 9;
    const float* tr = t + row * 3;
    const float x = p[0], y = p[1], z = p[2];
    p[0] = Rr[0] * x + Rr[1] * y + Rr[2] * z + tr[0];
    p[1] = Rr[3] * x + Rr[4] * y + Rr[5] * z + tr[1];
    p[2] = Rr[6] * x + Rr[7] * y + Rr[8] * z + tr[2];
}

int launch_points_to_global(float* proj, const float* R, const float* t, int64_t rows, hipStream_t st) {
    const int64_t total = rows * 3 * H * P;
    if (total == 0) return ABOPT_OK;
    hipLaunchKernelGGL(points_to_global_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, proj, R, t, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// ------------------------------------------------------------------ IPA core, version 0 (row-per-workgroup, VALU)
// One 256-thread workgroup per query residue (n, i).  z[n,i,:,:] is staged once in LDS (row stride 65
// floats: conflict-free both for "thread owns row j" and "thread owns channel c" access), logits for the
// whole row live in LDS, softmax is a wave-per-3-heads reduction, and the three aggregations re-read the
// LDS copy of z.  Needs (65 + 12) * 4 * L bytes of LDS => L <= 480.
constexpr int ZLD = C + 1;

__global__ __launch_bounds__(256) void ipa_core_v0_kernel(const float* __restrict__ proj, const float* __restrict__ z,
                                                          const uint8_t* __restrict__ mask, const float* __restrict__ R,
                                                          const float* __restrict__ t, const float* __restrict__ Wb,
                                                          const float* __restrict__ spatial_coef, float* __restrict__ feat,
                                                          float* __restrict__ dbg_logits, float* __restrict__ dbg_alpha, int L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* zs = smem;                       // [L][65]
    float* lg = zs + (size_t)L * ZLD;       // [L][12]   logits, then alpha
    float* agg = lg + (size_t)L * H;        // [1440]    fp | fn | ag

    const int i = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int64_t row_i = (int64_t)n * L + i;
    const float* zi = z + row_i * (int64_t)L * C;
    const float* pi = proj + row_i * NP;
    const bool mi = mask[row_i] != 0;

    // stage z[n,i] : coalesced float4 loads, scalar LDS stores (odd stride)
    for (int e = tid; e < L * (C / 4); e += 256) {
        const float4 v = reinterpret_cast<const float4*>(zi)[e];
        const int j = e / (C / 4), c = (e % (C / 4)) * 4;
        float* d = zs + j * ZLD + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();

    // logits: thread owns key residue j
    for (int j = tid; j < L; j += 256) {
        const float* pj = proj + ((int64_t)n * L + j) * NP;
        const bool mj = mask[(int64_t)n * L + j] != 0;
        const float* zr = zs + j * ZLD;
#pragma unroll 1
        for (int h = 0; h < H; ++h) {
            float lp = 0.f;
#pragma unroll 16
            for (int c = 0; c < C; ++c) lp = fmaf(zr[c], Wb[h * C + c], lp);
            float ln = 0.f;
#pragma unroll 8
            for (int d = 0; d < D; ++d) ln = fmaf(pi[OFF_Q + h * D + d] * pj[OFF_K + h * D + d], 0.17677669529663687f, ln);
            float d2 = 0.f;
#pragma unroll 8
            for (int e = 0; e < P * 3; ++e) {
                const float df = pi[OFF_QP + h * P * 3 + e] - pj[OFF_KP + h * P * 3 + e];
                d2 = fmaf(df, df, d2);
            }
            // gamma = softplus(spatial_coef); ga.py:108-111
            const float sc = spatial_coef[h];
            const float gamma = (sc > 20.f) ? sc : log1pf(expf(sc));
            const float ls = d2 * ((-1.f * gamma * 0.16666666666666666f) / 2.f);      // sqrt(2/(9*8)) = 1/6
            float lt = ((ln + lp) + ls) * 0.5773502691896258f;                         // sqrt(1/3)
            if (dbg_logits) dbg_logits[(row_i * L + j) * H + h] = lt;
            if (!(mi && mj)) lt -= 1e5f;
            lg[j * H + h] = lt;
        }
    }
    __syncthreads();

    // softmax over j, three heads per wave (ga.py:24-25)
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int h = wave * 3; h < wave * 3 + 3; ++h) {
            float mx = -INFINITY;
            for (int j = lane; j < L; j += 64) mx = fmaxf(mx, lg[j * H + h]);
            mx = wave_max(mx);
            float sm = 0.f;
            for (int j = lane; j < L; j += 64) { const float e = expf(lg[j * H + h] - mx); lg[j * H + h] = e; sm += e; }
            sm = wave_sum(sm);
            const float inv = mi ? 1.f / sm : 0.f;
            for (int j = lane; j < L; j += 64) {
                const float a = mi ? lg[j * H + h] / sm : 0.f;
                (void)inv;
                lg[j * H + h] = a;
                if (dbg_alpha) dbg_alpha[(row_i * L + j) * H + h] = a;
            }
        }
    }
    __syncthreads();

    // aggregations: thread owns an output channel
    constexpr int NOUT = H * C + H * D + H * P * 3;     // 768 + 384 + 288 = 1440
    for (int o = tid; o < NOUT; o += 256) {
        float acc = 0.f;
        if (o < H * C) {
            const int h = o / C, c = o % C;
            for (int j = 0; j < L; ++j) acc = fmaf(lg[j * H + h], zs[j * ZLD + c], acc);
        } else if (o < H * C + H * D) {
            const int oo = o - H * C, h = oo / D;
            const float* vj = proj + (int64_t)n * L * NP + OFF_V + oo;
            for (int j = 0; j < L; ++j) acc = fmaf(lg[j * H + h], vj[(int64_t)j * NP], acc);
        } else {
            const int oo = o - H * C - H * D, h = oo / (P * 3);
            const float* vj = proj + (int64_t)n * L * NP + OFF_VP + oo;
            for (int j = 0; j < L; ++j) acc = fmaf(lg[j * H + h], vj[(int64_t)j * NP], acc);
        }
        agg[o] = acc;
    }
    __syncthreads();

    // epilogue: feat = [fp | fn | local points | distance | direction]  (ga.py:133-145)
    float* fo = feat + row_i * FEAT;
    for (int o = tid; o < H * C + H * D; o += 256) fo[o] = agg[o];
    if (tid < H * P) {
        const float* Rr = R + row_i * 9;
        const float* tr = t + row_i * 3;
        const float* a = agg + H * C + H * D + tid * 3;
        const float dx = a[0] - tr[0], dy = a[1] - tr[1], dz = a[2] - tr[2];
        // R^T (a - t)
        const float lx = Rr[0] * dx + Rr[3] * dy + Rr[6] * dz;
        const float ly = Rr[1] * dx + Rr[4] * dy + Rr[7] * dz;
        const float lz = Rr[2] * dx + Rr[5] * dy + Rr[8] * dz;
        const float dist = sqrtf(lx * lx + ly * ly + lz * lz);
        const float inv = 1.f / (dist + 1e-4f);
        float* fpnt = fo + H * C + H * D;
        fpnt[tid * 3 + 0] = lx; fpnt[tid * 3 + 1] = ly; fpnt[tid * 3 + 2] = lz;
        fpnt[H * P * 3 + tid] = dist;
        float* fdir = fpnt + H * P * 3 + H * P;
        fdir[tid * 3 + 0] = lx * inv; fdir[tid * 3 + 1] = ly * inv; fdir[tid * 3 + 2] = lz * inv;
    }
}

// ------------------------------------------------------------------ measurement hook (see abopt_prof_enable)
namespace prof {
static bool g_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
static size_t g_used = 0;
static void begin(hipStream_t st) {
    if (!g_on) return;
    if (g_used == g_pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { g_on = false; return; }
        g_pool.emplace_back(a, b);
    }
    hipEventRecord(g_pool[g_used].first, st);
}
static void end(hipStream_t st) {
    if (!g_on) return;
    hipEventRecord(g_pool[g_used].second, st);
    ++g_used;
}
}  // namespace prof

int launch_ipa_core(const float* proj, const float* z, const uint8_t* mask, const float* R, const float* t,
                    const float* w_pair_bias, const float* spatial_coef, float* feat,
                    float* dbg_logits, float* dbg_alpha, int N, int L, hipStream_t st) {
    if (N == 0 || L == 0) return ABOPT_OK;
    const size_t lds = ((size_t)L * (ZLD + H) + 1440) * sizeof(float);
    ABOPT_CHECK_ARG(lds <= 160 * 1024, "ipa_core v0: L=%d needs %zu bytes of LDS (max 163840)", L, lds);
    ABOPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_core_v0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof::begin(st);
    hipLaunchKernelGGL(ipa_core_v0_kernel, dim3(L, N), dim3(256), lds, st, proj, z, mask, R, t, w_pair_bias, spatial_coef,
                       feat, dbg_logits, dbg_alpha, L);
    prof::end(st);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

extern "C" int abopt_prof_enable(int on) {
    abopt::prof::g_on = on != 0;
    abopt::prof::g_used = 0;
    return ABOPT_OK;
}

extern "C" int abopt_prof_collect(int* launches, double* total_ms) {
    double tot = 0.0;
    for (size_t i = 0; i < abopt::prof::g_used; ++i) {
        float ms = 0.f;
        ABOPT_HIP(hipEventSynchronize(abopt::prof::g_pool[i].second));
        ABOPT_HIP(hipEventElapsedTime(&ms, abopt::prof::g_pool[i].first, abopt::prof::g_pool[i].second));
        tot += ms;
    }
    if (launches) *launches = (int)abopt::prof::g_used;
    if (total_ms) *total_ms = tot;
    abopt::prof::g_used = 0;
    return ABOPT_OK;
}

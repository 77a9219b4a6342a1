// Training side of the IPA core (FullDPM.forward, AbDock/src/modules/diffusion/dpm_full.py:156-234, config 5).
//
// Forward: the inference kernel (ipa_ws.hip) run with its logits dump, followed by the masked softmax, so autograd can keep
// alpha (N, L, L, 12) instead of the (N, L, L, 12, 64) products the reference's broadcast formulation materialises
// (ga.py:114-118: 805 MB per sample and layer).
//
// Backward, pair side: everything that touches z[n,i,j,:] in one streaming pass (read z once, write dz once):
//     dalpha_ijh = dalpha_node_ijh + sum_c dfp_ihc z_ijc                       (d/d alpha of ga.py:116-118; the node/point terms are (N,L,L,12) GEMMs done by the host)
//     g_ijh      = alpha_ijh (dalpha_ijh - delta_ih) / sqrt(3)                 (softmax backward of ga.py:11-26 and the logit scale, ga.py:165)
//     dz_ijc     = sum_h alpha_ijh dfp_ihc + g_ijh Wb_hc                       (pair aggregation + proj_pair_bias, ga.py:88-90)
// delta_ih = sum_j alpha_ijh dalpha_ijh is supplied by the host as <d feat, feat> of the aggregated outputs (the
// flash-attention identity).  g is written for the (N,L,L,12)-sized q/k/point/Wb gradient GEMMs.
#include "ipa_common.h"
#include "kernels.h"

namespace abopt {

// One workgroup per query row (n, i): 4 waves, each handles 4 keys per iteration (lane = (jl, x)): x = head in phase 1
// (dalpha, g), x = group of 4 channels in phase 2 (dz).  The two phases exchange (alpha, g) through a wave-private LDS tile.
__global__ __launch_bounds__(256) void ipa_pair_backward_kernel(const float* __restrict__ z, const float* __restrict__ alpha,
                                                                const float* __restrict__ dalpha_node, const float* __restrict__ delta,
                                                                const float* __restrict__ dfeat, int ld_dfeat, const float* __restrict__ Wb,
                                                                float* __restrict__ g_out, float* __restrict__ dz, int L) {
    __shared__ __attribute__((aligned(16))) float zs[4][4][C + 4];       // per wave: 4 key rows of z
    __shared__ __attribute__((aligned(16))) float ag[4][4][2][16];        // per wave: alpha, g of 4 keys x 12 heads
    const int64_t row = blockIdx.x;                                        // n * L + i
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, jl = lane >> 4, x = lane & 15;
    const float* dfp = dfeat + row * ld_dfeat;                             // [12][64]
    // phase-1 operand: dfp[h = x][0:64];  phase-2 operands: dfp[0:12][4x .. 4x+3], Wb[0:12][4x .. 4x+3]
    float4 d1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) d1[q] = (x < H) ? reinterpret_cast<const float4*>(dfp + x * C)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 d2[H], w2[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { d2[h] = reinterpret_cast<const float4*>(dfp + h * C)[x]; w2[h] = reinterpret_cast<const float4*>(Wb + h * C)[x]; }
    const float del = (x < H) ? delta[row * H + x] : 0.f;
    const float* zrow = z + row * (int64_t)L * C;
    const float* arow = alpha + row * (int64_t)L * H;
    const float* nrow = dalpha_node + row * (int64_t)L * H;
    float* grow = g_out + row * (int64_t)L * H;
    float* dzrow = dz + row * (int64_t)L * C;
    for (int j0 = wave * 4; j0 < L; j0 += 16) {
        const int j = j0 + jl;
        const bool ok = j < L;
        const int jc = ok ? j : L - 1;
        // stage the wave's 4 z rows: lane (jl, x) loads channels 4x..4x+3 of key j (1 KB per wave, coalesced)
        const float4 zv = reinterpret_cast<const float4*>(zrow + (int64_t)jc * C)[x];
        wave_lds_sync();
        *reinterpret_cast<float4*>(&zs[wave][jl][x * 4]) = zv;
        float a = 0.f, dn = 0.f;
        if (x < H) { a = arow[(int64_t)jc * H + x]; dn = nrow[(int64_t)jc * H + x]; }
        wave_lds_sync();
        // phase 1: dalpha for (key jl, head x)
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 zz = *reinterpret_cast<const float4*>(&zs[wave][jl][q * 4]);
            acc = fmaf(d1[q].x, zz.x, acc); acc = fmaf(d1[q].y, zz.y, acc); acc = fmaf(d1[q].z, zz.z, acc); acc = fmaf(d1[q].w, zz.w, acc);
        }
        const float gv = a * ((dn + acc) - del) * 0.5773502691896258f;
        if (x < H && ok) grow[(int64_t)j * H + x] = gv;
        ag[wave][jl][0][x] = a; ag[wave][jl][1][x] = gv;
        wave_lds_sync();
        // phase 2: dz for (key jl, channels 4x..4x+3)
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float ah = ag[wave][jl][0][h], gh = ag[wave][jl][1][h];
            o.x = fmaf(ah, d2[h].x, o.x); o.y = fmaf(ah, d2[h].y, o.y); o.z = fmaf(ah, d2[h].z, o.z); o.w = fmaf(ah, d2[h].w, o.w);
            o.x = fmaf(gh, w2[h].x, o.x); o.y = fmaf(gh, w2[h].y, o.y); o.z = fmaf(gh, w2[h].z, o.z); o.w = fmaf(gh, w2[h].w, o.w);
        }
        if (ok) reinterpret_cast<float4*>(dzrow + (int64_t)j * C)[x] = o;
    }
}

int launch_ipa_pair_backward(const float* z, const float* alpha, const float* dalpha_node, const float* delta, const float* dfeat, int ld_dfeat,
                             const float* Wb, float* g_out, float* dz, int N, int L, hipStream_t st) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    hipLaunchKernelGGL(ipa_pair_backward_kernel, dim3((unsigned)((int64_t)N * L)), dim3(256), 0, st, z, alpha, dalpha_node, delta, dfeat, ld_dfeat, Wb, g_out, dz, L);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

size_t ipa_train_ws_floats(int N, int L) { return (size_t)N * L * NP + ipa_kvfrag_floats(N, L) + 64; }

// proj_local [N*L, 2016] (points in the residue frames, as the six projections produce them) -> feat [N*L, 1824], alpha [N, L, L, 12]
int launch_ipa_train_forward(const float* proj_local, const float* R, const float* t, const float* z, const uint8_t* mask,
                             const float* Wb, const float* spatial_coef, float* feat, float* alpha, int N, int L, float* ws, hipStream_t st) {
    const int64_t M = (int64_t)N * L;
    if (M == 0) return ABOPT_OK;
    float* proj = ws;
    float* kvf = ws + (size_t)M * NP;
    ABOPT_HIP(hipMemcpy2DAsync(proj, (size_t)NP * sizeof(float), proj_local, (size_t)ABOPT_NODE_PROJ * sizeof(float),
                               (size_t)ABOPT_NODE_PROJ * sizeof(float), (size_t)M, hipMemcpyDeviceToDevice, st));
    int rc;
    if ((rc = launch_points_to_global(proj, R, t, M, st, kvf, N, L))) return rc;
    // alpha doubles as the logits dump: alpha_from_logits rewrites it in place
    return launch_ipa_core(proj, z, mask, R, t, Wb, spatial_coef, feat, alpha, alpha, nullptr, kvf, N, L, st);
}

}  // namespace abopt

// Training side of the IPA core (FullDPM.forward, AbDock/src/modules/diffusion/dpm_full.py:156-234, config 5).
//
// Forward: the inference kernel (ipa_core.hip) with its head-major dump of the scaled logits and the row statistics, then one
// elementwise pass that turns the dump into alpha in place (alpha_finalize), so autograd keeps alpha (N, 12, L, L) instead of the
// (N, L, L, 12, 64) products the reference's broadcast formulation materialises (ga.py:114-118: 805 MB per sample and layer).
//
// Backward, pair side: everything that touches z[n,i,j,:] in one streaming pass (read z once, write dz once):
//     dalpha_ijh = dalpha_node_ijh + sum_c dfp_ihc z_ijc                       (d/d alpha of ga.py:116-118; the node/point terms are (N,L,L,12) GEMMs done by the host)
//     g_ijh      = alpha_ijh (dalpha_ijh - delta_ih) / sqrt(3)                 (softmax backward of ga.py:11-26 and the logit scale, ga.py:165)
//     dz_ijc     = sum_h alpha_ijh dfp_ihc + g_ijh Wb_hc                       (pair aggregation + proj_pair_bias, ga.py:88-90)
// delta_ih = sum_j alpha_ijh dalpha_ijh is supplied by the host as <d feat, feat> of the aggregated outputs (the
// flash-attention identity).  g is written for the q/k/point/Wb gradient GEMMs, which are plain batched L x L products.
#include "ipa_common.h"
#include "kernels.h"

namespace abopt {

// alpha, dalpha_node and g use the head-major layout (N, 12, L, L): every (n, h) slice is then a plain row-major L x L
// matrix for the batched library GEMMs the host runs on them.
// One workgroup per query row (n, i), four waves; wave w takes the 16-key chunks w, w+4, ...  Same tiling as the pair waves of
// the forward kernel (ipa_core.hip), all three contractions on the matrix cores:
//   dalpha_pair[j, h] = z[j, :] . dfp[h, :]          A = z chunk transposed through a wave-private LDS tile, B = dfp (LDS)
//   dz[j, c]          = sum_h alpha[h,j] dfp[h,c] + g[h,j] Wb[h,c]     A = [dfp^T | Wb^T] (LDS), B = [alpha ; g] transposed through LDS
// g is produced in lanes (h = lane & 15, keys 4 (lane >> 4) + r): the accumulator layout of the first product and the
// 16-byte store layout of the head-major output.
struct PairBwdSmem {
    float dfp[16][C + 4];            // rows h (12..15 zero): B operand of the first product
    float at[C][36];                 // [c][0:16] = dfp^T, [16:32] = Wb^T (heads 12..15 zero): A operand of the dz product
    float zst[4][JC][ZSLD];          // per wave: z chunk, [key][channel]
    float ag[4][JC][36];             // per wave: [key][0:16] = alpha over heads, [16:32] = g over heads
};

__global__ __launch_bounds__(256) void ipa_pair_backward_kernel(const float* __restrict__ z, const float* __restrict__ alpha,
                                                                const float* __restrict__ dalpha_node, const float* __restrict__ delta,
                                                                const float* __restrict__ dfeat, int ld_dfeat, const float* __restrict__ Wb,
                                                                float* __restrict__ g_out, float* __restrict__ dz, float* __restrict__ dwb_part, int L, int dz_accumulate) {
    __shared__ __attribute__((aligned(16))) PairBwdSmem sm;
    const int64_t row = blockIdx.x;                                        // n * L + i
    const int64_t n = row / L;
    const int i = (int)(row % L);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const float* dfp = dfeat + row * ld_dfeat;                             // [12][64]
    for (int e = tid; e < 16 * C; e += 256) {
        const int h = e / C, c = e % C;
        const float d = (h < H) ? dfp[h * C + c] : 0.f, w = (h < H) ? Wb[h * C + c] : 0.f;
        sm.dfp[h][c] = d; sm.at[c][h] = d; sm.at[c][16 + h] = w;
    }
    __syncthreads();
    const float del = (fm < H) ? delta[row * H + fm] : 0.f;
    const float* zrow = z + row * (int64_t)L * C;
    float* dzrow = dz ? dz + row * (int64_t)L * C : nullptr;
    const int64_t hm = ((n * H + fm) * (int64_t)L + i) * L;                // (n, h = fm, i, j = 0) of the head-major arrays
    const int nchunk = (L + JC - 1) / JC;
    f32x4 zr[4], zc[4], accW[4];                                          // accW: this wave's share of sum_j g[h,j] z[j,c] (d proj_pair_bias.weight)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) accW[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) zr[r] = *(reinterpret_cast<const f32x4*>(zrow + (int64_t)min(wave * JC + kq * 4 + r, L - 1) * C) + fm);
    for (int ch = wave; ch < nchunk; ch += 4) {
        const int j0 = ch * JC;
        // ---- z chunk -> LDS (transpose), next chunk's rows requested
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) { zc[r] = zr[r]; *reinterpret_cast<f32x4*>(&sm.zst[wave][kq * 4 + r][fm * 4]) = zr[r]; }
        if (ch + 4 < nchunk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) zr[r] = *(reinterpret_cast<const f32x4*>(zrow + (int64_t)min((ch + 4) * JC + kq * 4 + r, L - 1) * C) + fm);
        }
        // alpha / dalpha_node of (head fm, keys j0 + 4 kq ..): 16 B per lane from the head-major rows
        f32x4 a4 = (f32x4){0.f, 0.f, 0.f, 0.f}, n4 = a4;
        const int jq = j0 + kq * 4;
        if (fm < H) {
            if (jq + 3 < L && (L % 4) == 0) {
                a4 = *reinterpret_cast<const f32x4*>(alpha + hm + jq);
                n4 = *reinterpret_cast<const f32x4*>(dalpha_node + hm + jq);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (jq + r < L) { a4[r] = alpha[hm + jq + r]; n4[r] = dalpha_node[hm + jq + r]; }
            }
        }
        wave_lds_sync();
        // ---- dalpha_pair = z . dfp^T on the matrix cores, then g
        f32x4 acc4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 za = *reinterpret_cast<const float4*>(&sm.zst[wave][fm][kq * 16 + q * 4]);
            const float4 dv = *reinterpret_cast<const float4*>(&sm.dfp[fm][kq * 16 + q * 4]);
            acc4[q] = mfma4(za.x, dv.x, (f32x4){0.f, 0.f, 0.f, 0.f});
            acc4[q] = mfma4(za.y, dv.y, acc4[q]); acc4[q] = mfma4(za.z, dv.z, acc4[q]); acc4[q] = mfma4(za.w, dv.w, acc4[q]);
        }
        const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        f32x4 g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) g4[r] = a4[r] * ((n4[r] + acc[r]) - del) * 0.5773502691896258f;
        if (fm < H) {
            if (jq + 3 < L && (L % 4) == 0) *reinterpret_cast<f32x4*>(g_out + hm + jq) = g4;
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) if (jq + r < L) g_out[hm + jq + r] = g4[r];
        }
        // ---- d Wb[h, c] += sum_j g[h, j] z[j, c]: z rows are already the A operand, g the B operand (keys past L carry g = 0)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) accW[mt] = mfma4(zc[r][mt], g4[r], accW[mt]);
        if (!dz) continue;                                                 // (wave-uniform) dz is assembled for all blocks at once: ipa_dz_assemble_kernel
        // ---- [alpha ; g] -> LDS transposed ([key][head]) for the dz product
#pragma unroll
        for (int r = 0; r < 4; ++r) { sm.ag[wave][kq * 4 + r][fm] = a4[r]; sm.ag[wave][kq * 4 + r][16 + fm] = g4[r]; }
        wave_lds_sync();
        // ---- dz[j, c]: rows = channels of tile ct, cols = keys
        const bool jok = (j0 + fm) < L;
        float* dzj = dzrow + (int64_t)(j0 + fm) * C + kq * 4;
        f32x4 bq[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) bq[blk] = *reinterpret_cast<const f32x4*>(&sm.ag[wave][fm][blk * 16 + kq * 4]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const f32x4 aq = *reinterpret_cast<const f32x4*>(&sm.at[ct * 16 + fm][blk * 16 + kq * 4]);
#pragma unroll
                for (int s = 0; s < 4; ++s) o = mfma4(aq[s], bq[blk][s], o);
            }
            if (jok) {                                                             // dz_accumulate: the blocks of an encoder share one d pair_feat buffer
                if (dz_accumulate) o += *reinterpret_cast<const f32x4*>(dzj + ct * 16);
                *reinterpret_cast<f32x4*>(dzj + ct * 16) = o;
            }
        }
    }
    // ---- the row's d Wb partial: accumulator tile mt holds channels 4 fm' + mt (rows 4 kq + r = fm') of head fm
    __syncthreads();
    float (*red)[16][C + 4] = reinterpret_cast<float (*)[16][C + 4]>(&sm.zst[0][0][0]);          // 4 x [16 heads][64 channels], reuses the z tiles
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][fm][(kq * 4 + r) * 4 + mt] = accW[mt][r];
    __syncthreads();
    for (int e = tid; e < H * C; e += 256) {
        const int h = e / C, c = e % C;
        dwb_part[row * (H * C) + e] = (red[0][h][c] + red[1][h][c]) + (red[2][h][c] + red[3][h][c]);
    }
}

// d pair_feat of ALL blocks of an encoder in one pass (round 4):
//     dz_ijc = sum over blocks l, heads h of  alpha_l,ijh dfp_l,ihc + g_l,ijh Wb_l,hc
// Each block's ipa_pair_backward_kernel used to add its term into the shared d pair_feat buffer: a read-modify-write of N L^2 C floats
// per block (2.95 GB per step at config 5).  The terms need no z, only what the blocks' backward passes leave behind anyway (alpha, g,
// d feat, the weights), so they are summed here with ONE write of dz: a K = 32 nl contraction per (row, key chunk), same tiling as above.
struct DzLayers { const float* alpha[8]; const float* g[8]; const float* dfeat[8]; const float* wb[8]; int nl; int ld_dfeat; };
// K = 24 per block (12 heads of alpha . dfp + 12 of g . Wb) as six 16x16x4 steps: lane group kq supplies slot kq * 4 + m in steps m = 0..3 (one 16-byte
// LDS read) and slot 16 + kq * 2 + (m - 4) in steps 4, 5 (one 8-byte read); slots 0..11 = heads of the first term, 12..23 = heads of the second, the same
// for both operands.  (Until the end of round 5 the heads were padded to 16 per term: 8 steps, a quarter of them on zeros, and 64.5 KB of LDS -- two
// workgroups per CU; 50 KB now: three.)
constexpr int DZK = 24, DZLD = 28;
struct DzSmem {
    float at[6][C][DZLD];            // per block: [c][0:12] = dfp^T of this row, [12:24] = Wb^T
    float ag[4][JC][DZLD];           // per wave: [key][0:12] = alpha over heads, [12:24] = g over heads (one block at a time)
};
__global__ __launch_bounds__(256, 3) void ipa_dz_assemble_kernel(DzLayers a, float* __restrict__ dz, int L) {
    extern __shared__ __attribute__((aligned(16))) char dz_raw[];
    DzSmem& sm = *reinterpret_cast<DzSmem*>(dz_raw);
    const int64_t row = blockIdx.x;
    const int64_t n = row / L;
    const int i = (int)(row % L);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    for (int l = 0; l < a.nl; ++l) {
        const float* dfp = a.dfeat[l] + row * a.ld_dfeat;
        for (int e = tid; e < 16 * C; e += 256) {
            const int h = e / C, c = e % C;
            if (h < H) { sm.at[l][c][h] = dfp[h * C + c]; sm.at[l][c][H + h] = a.wb[l][h * C + c]; }
        }
    }
    __syncthreads();
    float* dzrow = dz + row * (int64_t)L * C;
    const int64_t hm = ((n * H + fm) * (int64_t)L + i) * L;
    const int nchunk = (L + JC - 1) / JC;
    // the (alpha, g) quads of ALL blocks for a chunk are requested together, one chunk ahead of their use (requested block by block next
    // to the LDS transposes the kernel ran at 1.9 TB/s: every block's round trip was exposed)
    f32x4 an[6], gn[6];
    auto fetch = [&](int ch, f32x4 (&av)[6], f32x4 (&gv)[6]) {
        const int jq = ch * JC + kq * 4;
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            av[l] = (f32x4){0.f, 0.f, 0.f, 0.f}; gv[l] = av[l];
            if (l < a.nl && fm < H && ch < nchunk) {
                if (jq + 3 < L && (L % 4) == 0) {
                    av[l] = *reinterpret_cast<const f32x4*>(a.alpha[l] + hm + jq);
                    gv[l] = *reinterpret_cast<const f32x4*>(a.g[l] + hm + jq);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (jq + r < L) { av[l][r] = a.alpha[l][hm + jq + r]; gv[l][r] = a.g[l][hm + jq + r]; }
                }
            }
        }
    };
    fetch(wave, an, gn);
    for (int ch = wave; ch < nchunk; ch += 4) {
        const int j0 = ch * JC;
        f32x4 ac[6], gc[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) { ac[l] = an[l]; gc[l] = gn[l]; }
        fetch(ch + 4, an, gn);
        f32x4 o[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            if (l >= a.nl) break;
            wave_lds_sync();
            if (fm < H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { sm.ag[wave][kq * 4 + r][fm] = ac[l][r]; sm.ag[wave][kq * 4 + r][H + fm] = gc[l][r]; }
            }
            wave_lds_sync();
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sm.ag[wave][fm][kq * 4]);
            const float2 b2 = *reinterpret_cast<const float2*>(&sm.ag[wave][fm][16 + kq * 2]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sm.at[l][ct * 16 + fm][kq * 4]);
                const float2 a2 = *reinterpret_cast<const float2*>(&sm.at[l][ct * 16 + fm][16 + kq * 2]);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) o[ct] = mfma4(a4[s_], b4[s_], o[ct]);
                o[ct] = mfma4(a2.x, b2.x, o[ct]);
                o[ct] = mfma4(a2.y, b2.y, o[ct]);
            }
        }
        if (j0 + fm < L) {
            float* dzj = dzrow + (int64_t)(j0 + fm) * C + kq * 4;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<f32x4*>(dzj + ct * 16) = o[ct];
        }
    }
}

int launch_ipa_dz_assemble(int nl, const float* const* alpha, const float* const* g, const float* const* dfeat, int ld_dfeat, const float* const* wb,
                           float* dz, int N, int L, hipStream_t st) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(nl >= 1 && nl <= 6, "ipa_dz_assemble: %d blocks (1..6)", nl);
    DzLayers a;
    a.nl = nl; a.ld_dfeat = ld_dfeat;
    for (int l = 0; l < nl; ++l) {
        ABOPT_CHECK_ARG(alpha[l] && g[l] && dfeat[l] && wb[l], "ipa_dz_assemble: NULL pointer for block %d", l);
        a.alpha[l] = alpha[l]; a.g[l] = g[l]; a.dfeat[l] = dfeat[l]; a.wb[l] = wb[l];
    }
    static LdsConfig lds_cfg;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_dz_assemble_kernel), sizeof(DzSmem), lds_cfg)) return rc;
    hipLaunchKernelGGL(ipa_dz_assemble_kernel, dim3((unsigned)((int64_t)N * L)), dim3(256), sizeof(DzSmem), st, a, dz, L);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// alpha (ga.py:11-26,166) in place over the core's head-major dump x [N,12,L,L]: alpha = mask_i mask_j ? exp2(x - m_ih) / l_ih : 0
// with the row statistics the core kept; one thread per 4 keys (scalar tail when L is not a multiple of 4)
__global__ __launch_bounds__(256) void alpha_finalize_kernel(float* __restrict__ xa, const float* __restrict__ stats, const uint8_t* __restrict__ mask,
                                                             int L, int64_t quads) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;       // ((n * H + h) * L + i) * (L / 4) + j / 4
    if (q >= quads) return;
    const int l4 = L >> 2;
    const int jq = (int)(q % l4);
    const int64_t nhi = q / l4;
    const int i = (int)(nhi % L);
    const int64_t nh = nhi / L, n = nh / H;
    const int h = (int)(nh % H);
    const int64_t row = n * L + i;
    const float2 st = *reinterpret_cast<const float2*>(stats + (row * H + h) * 2);
    const bool mi = mask[row] != 0;
    const float inv = 1.f / st.y;
    f32x4* p = reinterpret_cast<f32x4*>(xa) + q;
    f32x4 v = *p;
    const uint32_t mk = *reinterpret_cast<const uint32_t*>(mask + n * L + jq * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (mi && ((mk >> (8 * r)) & 0xffu)) ? __builtin_amdgcn_exp2f(v[r] - st.x) * inv : 0.f;
    *p = v;
}
__global__ __launch_bounds__(256) void alpha_finalize_scalar_kernel(float* __restrict__ xa, const float* __restrict__ stats, const uint8_t* __restrict__ mask,
                                                                    int L, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;       // ((n * H + h) * L + i) * L + j
    if (e >= total) return;
    const int j = (int)(e % L);
    const int64_t nhi = e / L;
    const int i = (int)(nhi % L);
    const int64_t nh = nhi / L, n = nh / H;
    const int h = (int)(nh % H);
    const int64_t row = n * L + i;
    const bool live = mask[row] != 0 && mask[n * L + j] != 0;
    xa[e] = live ? __builtin_amdgcn_exp2f(xa[e] - stats[(row * H + h) * 2]) / stats[(row * H + h) * 2 + 1] : 0.f;
}

// Backward of the points epilogue (ga.py:133-139: loc = R^T (agg - t), dist = |loc|, dir = loc / (dist + 1e-4)) and the
// flash-attention delta, one workgroup per query row, thread = (head, point):
//   dout_cat[n, h, i, 0:32]  = d feat_node_ih                      (head-major copy for the batched GEMMs)
//   dout_cat[n, h, i, 32:56] = d agg_pts_ih = R_i . d loc
//   delta[n, i, h]           = <d feat_p2n, feat_p2n>_h + <d feat_node, feat_node>_h + <d agg_pts, agg_pts>_h,  agg_pts = R loc + t
__global__ __launch_bounds__(128) void ipa_points_backward_kernel(const float* __restrict__ dfeat, int ld_dfeat, const float* __restrict__ feat,
                                                                  const float* __restrict__ R, const float* __restrict__ t,
                                                                  float* __restrict__ dout_cat, float* __restrict__ delta, int L) {
    const int64_t row = blockIdx.x;
    const int64_t n = row / L;
    const int i = (int)(row % L), tid = threadIdx.x;
    if (tid >= H * P) return;
    const int h = tid / P, p = tid % P;
    const float* df = dfeat + row * ld_dfeat;
    const float* ff = feat + row * FEAT;
    constexpr int o = H * C + H * D;                             // start of the point features
    const float* Rr = R + row * 9;
    const float* tr = t + row * 3;
    const int hp = h * P + p;
    const float lx = ff[o + hp * 3], ly = ff[o + hp * 3 + 1], lz = ff[o + hp * 3 + 2];
    const float nrm = sqrtf(lx * lx + ly * ly + lz * lz);
    const float ux = nrm > 0.f ? lx / nrm : 0.f, uy = nrm > 0.f ? ly / nrm : 0.f, uz = nrm > 0.f ? lz / nrm : 0.f;
    const float inv = 1.f / (nrm + 1e-4f);
    const float dd = df[o + NPT + hp];                           // d dist
    const float* ddir = df + o + NPT + H * P + hp * 3;
    const float* dloc = df + o + hp * 3;
    const float k = dd - (ddir[0] * lx + ddir[1] * ly + ddir[2] * lz) * inv * inv;
    const float dlx = dloc[0] + ddir[0] * inv + k * ux, dly = dloc[1] + ddir[1] * inv + k * uy, dlz = dloc[2] + ddir[2] * inv + k * uz;
    const float gx = Rr[0] * dlx + Rr[1] * dly + Rr[2] * dlz, gy = Rr[3] * dlx + Rr[4] * dly + Rr[5] * dlz, gz = Rr[6] * dlx + Rr[7] * dly + Rr[8] * dlz;
    const float ax = Rr[0] * lx + Rr[1] * ly + Rr[2] * lz + tr[0], ay = Rr[3] * lx + Rr[4] * ly + Rr[5] * lz + tr[1], az = Rr[6] * lx + Rr[7] * ly + Rr[8] * lz + tr[2];
    float* oc = dout_cat + ((n * H + h) * (int64_t)L + i) * (D + P * 3);
    oc[D + p * 3] = gx; oc[D + p * 3 + 1] = gy; oc[D + p * 3 + 2] = gz;
    float part = gx * ax + gy * ay + gz * az;
    // this thread's slice of the head's pair (8 of 64 channels) and node (4 of 32 channels) dot products; d feat_node copied head-major
#pragma unroll
    for (int c = 0; c < 8; ++c) part = fmaf(df[h * C + p * 8 + c], ff[h * C + p * 8 + c], part);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float v = df[H * C + h * D + p * 4 + d];
        oc[p * 4 + d] = v;
        part = fmaf(v, ff[H * C + h * D + p * 4 + d], part);
    }
    part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);      // the 8 point-threads of a head are adjacent lanes
    if (p == 0) delta[row * H + h] = part;
}

int launch_ipa_points_backward(const float* dfeat, int ld_dfeat, const float* feat, const float* R, const float* t, float* dout_cat, float* delta,
                               int N, int L, hipStream_t st) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    hipLaunchKernelGGL(ipa_points_backward_kernel, dim3((unsigned)((int64_t)N * L)), dim3(128), 0, st, dfeat, ld_dfeat, feat, R, t, dout_cat, delta, L);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// Head-major GEMM operands of the backward from the (local-frame) projections: one workgroup per residue, thread = (head, point)
//   Aq[n,h,i,:] = [q_ih (32) | q_pts global (24) | 1],  Ak likewise from k,  Av[n,h,i,:] = [v_ih (32) | v_pts global (24)]
__global__ __launch_bounds__(128) void ipa_backward_operands_kernel(const float* __restrict__ proj, const float* __restrict__ R, const float* __restrict__ t,
                                                                    float* __restrict__ Aq, float* __restrict__ Ak, float* __restrict__ Av, int L) {
    const int64_t row = blockIdx.x;
    const int64_t n = row / L;
    const int i = (int)(row % L), tid = threadIdx.x;
    if (tid >= H * P) return;
    const int h = tid / P, p = tid % P;
    const float* pr = proj + row * ABOPT_NODE_PROJ;
    const float* Rr = R + row * 9;
    const float* tr = t + row * 3;
    const int64_t hb = (n * H + h) * (int64_t)L + i;
    float* outs[3] = {Aq + hb * 57, Ak + hb * 57, Av + hb * 56};
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        float* o = outs[s3];
#pragma unroll
        for (int d = 0; d < 4; ++d) o[p * 4 + d] = pr[s3 * H * D + h * D + p * 4 + d];
        const float* lp = pr + 3 * H * D + s3 * NPT + (h * P + p) * 3;
        o[D + p * 3 + 0] = Rr[0] * lp[0] + Rr[1] * lp[1] + Rr[2] * lp[2] + tr[0];
        o[D + p * 3 + 1] = Rr[3] * lp[0] + Rr[4] * lp[1] + Rr[5] * lp[2] + tr[1];
        o[D + p * 3 + 2] = Rr[6] * lp[0] + Rr[7] * lp[1] + Rr[8] * lp[2] + tr[2];
        if (s3 < 2 && p == 0) o[D + P * 3] = 1.f;
    }
}

// d proj [N,L,2016] (gradients wrt q|k|v and the LOCAL-frame points) from the three batched products
//   P1 = g Ak = [sum_j g k_j | sum_j g kg_j | sum_j g],  P2 = g^T Aq,  P3 = alpha^T [d feat_node | d agg_pts]      (head-major, 57/57/56 wide)
// and e[n,i,h] = (|qg|^2 rowsum + |kg|^2 colsum - 2 <qg, sum_j g kg_j>) x d coef_h / d spatial_coef_h, whose sum over (n, i) is
// d loss / d spatial_coef_h (ga.py:108-111)
__global__ __launch_bounds__(128) void ipa_backward_assemble_kernel(const float* __restrict__ P1, const float* __restrict__ P2, const float* __restrict__ P3,
                                                                    const float* __restrict__ Aq, const float* __restrict__ Ak, const float* __restrict__ R,
                                                                    const float* __restrict__ spatial_coef, float* __restrict__ dproj, float* __restrict__ e, int L) {
    const int64_t row = blockIdx.x;
    const int64_t n = row / L;
    const int i = (int)(row % L), tid = threadIdx.x;
    if (tid >= H * P) return;
    const int h = tid / P, p = tid % P;
    const int64_t hb = (n * H + h) * (int64_t)L + i;
    const float* p1 = P1 + hb * 57; const float* p2 = P2 + hb * 57; const float* p3 = P3 + hb * 56;
    const float* aq = Aq + hb * 57; const float* ak = Ak + hb * 57;
    const float* Rr = R + row * 9;
    float* dp = dproj + row * ABOPT_NODE_PROJ;
    const float sc = 0.17677669529663687f;                       // 1 / sqrt(32)
    const float scv = spatial_coef[h];
    const float gam = (scv > 20.f) ? scv : log1pf(expf(scv));
    const float c2 = 2.f * ((-1.f * gam * 0.16666666666666666f) / 2.f);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        dp[h * D + p * 4 + d] = p1[p * 4 + d] * sc;
        dp[H * D + h * D + p * 4 + d] = p2[p * 4 + d] * sc;
        dp[2 * H * D + h * D + p * 4 + d] = p3[p * 4 + d];
    }
    const float grow = p1[D + P * 3], gcol = p2[D + P * 3];
    float gq[3], gk[3], gv[3], part = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float qg = aq[D + p * 3 + a], kg = ak[D + p * 3 + a], gkg = p1[D + p * 3 + a], gqg = p2[D + p * 3 + a];
        gq[a] = c2 * (qg * grow - gkg);
        gk[a] = c2 * (kg * gcol - gqg);
        gv[a] = p3[D + p * 3 + a];
        part += qg * qg * grow + kg * kg * gcol - 2.f * qg * gkg;
    }
    float* outs[3] = {dp + 3 * H * D + (h * P + p) * 3, dp + 3 * H * D + NPT + (h * P + p) * 3, dp + 3 * H * D + 2 * NPT + (h * P + p) * 3};
    const float* gs[3] = {gq, gk, gv};
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {                             // R^T d
        const float* gvec = gs[s3];
        outs[s3][0] = Rr[0] * gvec[0] + Rr[3] * gvec[1] + Rr[6] * gvec[2];
        outs[s3][1] = Rr[1] * gvec[0] + Rr[4] * gvec[1] + Rr[7] * gvec[2];
        outs[s3][2] = Rr[2] * gvec[0] + Rr[5] * gvec[1] + Rr[8] * gvec[2];
    }
    part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
    // d softplus(x)/dx = sigmoid(x); the logit coefficient is -softplus(coef) sqrt(2 / (9 P)) / 2 (ga.py:108-111): e sums to d loss / d spatial_coef
    if (p == 0) e[row * H + h] = part * (-(1.f / (1.f + expf(-scv))) * 0.08333333333333333f);
}

int launch_ipa_backward_operands(const float* proj, const float* R, const float* t, float* Aq, float* Ak, float* Av, int N, int L, hipStream_t st) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    hipLaunchKernelGGL(ipa_backward_operands_kernel, dim3((unsigned)((int64_t)N * L)), dim3(128), 0, st, proj, R, t, Aq, Ak, Av, L);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

int launch_ipa_backward_assemble(const float* P1, const float* P2, const float* P3, const float* Aq, const float* Ak, const float* R,
                                 const float* spatial_coef, float* dproj, float* e, int N, int L, hipStream_t st) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    hipLaunchKernelGGL(ipa_backward_assemble_kernel, dim3((unsigned)((int64_t)N * L)), dim3(128), 0, st, P1, P2, P3, Aq, Ak, R, spatial_coef, dproj, e, L);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

int launch_ipa_pair_backward(const float* z, const float* alpha, const float* dalpha_node, const float* delta, const float* dfeat, int ld_dfeat,
                             const float* Wb, float* g_out, float* dz, float* dwb_part, int N, int L, hipStream_t st, int dz_accumulate) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    hipLaunchKernelGGL(ipa_pair_backward_kernel, dim3((unsigned)((int64_t)N * L)), dim3(256), 0, st, z, alpha, dalpha_node, delta, dfeat, ld_dfeat, Wb, g_out, dz, dwb_part, L, dz_accumulate);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

size_t ipa_train_ws_floats(int N, int L) { return (size_t)N * L * NP + ipa_kvfrag_floats(N, L) + ipa_qfrag_floats(N, L) + (size_t)N * L * H * 2 + 64; }

// proj_local [N*L, 2016] (points in the residue frames, as the six projections produce them) -> feat [N*L, 1824], alpha [N, 12, L, L];
// pbc (optional): this layer's slice of a pair-bias cache built from the same z and weights (launch_pair_bias_cache)
int launch_ipa_train_forward(const float* proj_local, const float* R, const float* t, const float* z, const uint8_t* mask,
                             const float* Wb, const float* spatial_coef, const float* pbc, float* feat, float* alpha, int N, int L, float* ws,
                             hipStream_t st) {
    const int64_t M = (int64_t)N * L;
    if (M == 0) return ABOPT_OK;
    float* kvf = ws + (size_t)M * NP;
    int rc;
    float* qf = kvf + ipa_kvfrag_floats(N, L);
    // the fragment kernel reads the projections' output in place (row stride 2016; until round 5 a 2-D copy re-strode it to 2048 first: 10.5 us per block)
    ABOPT_CHECK_ARG(((uintptr_t)proj_local % 16) == 0, "ipa_core_train_forward: proj must be 16-byte aligned");
    if ((rc = launch_ipa_frags(proj_local, R, t, spatial_coef, qf, kvf, N, L, st, ABOPT_NODE_PROJ))) return rc;
    float* stats = qf + ipa_qfrag_floats(N, L);
    // the core writes its scaled logits head-major into the alpha buffer; one elementwise pass turns them into alpha in place
    if ((rc = launch_ipa_core_kernel(qf, kvf, z, mask, R, t, Wb, feat, alpha, stats, pbc, N, L, st, 0))) return rc;
    if ((L & 3) == 0) {
        const int64_t quads = (int64_t)N * H * L * (L / 4);
        hipLaunchKernelGGL(alpha_finalize_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, alpha, stats, mask, L, quads);
    } else {
        const int64_t total = (int64_t)N * H * L * L;
        hipLaunchKernelGGL(alpha_finalize_scalar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, alpha, stats, mask, L, total);
    }
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

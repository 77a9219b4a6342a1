// Fused node projections of a GABlock for CDNA4: the six bias-free nn.Linear of ga.py:54-66 (q | k | v | q_pts | k_pts | v_pts,
// [M,128] x [2016,128]^T), the local -> global map of the three point sets (geometry.py:72-91, ga.py:96-105,129-132), the squared
// point norms, and the re-layout of everything into the MFMA fragment order the IPA core consumes (qfrag / kvfrag, see ipa.hip) --
// in ONE kernel.  The 67 MB projection buffer is never written or re-read.
//
// Weight-stationary: a workgroup owns HALF a head (since round 5; a whole head before).  A head's 168 weight rows, permuted and zero-padded at pack time to 12 tiles of 16 rows
//   tile 0,1: q channels 0..15, 16..31   2,3: k   4,5: q_pts points 0..3, 4..7 as (x, y, z, 0) quadruples  |  6,7: k_pts   8,9: v   10,11: v_pts
// sit in LDS in MFMA operand order (tiles 0..5 or 6..11: 48 KB, loaded once; three 4-wave workgroups per CU -- at the bench shape every wave of the chip
// runs exactly two tasks, where one 12-wave workgroup per CU and head left 24.4 tasks to 12 waves); residues stream through in pairs of 16-row tiles.
//
// Arithmetic (round 5): fp32 x fp32 products on the fp16 matrix pipe with TWO terms per operand (ipa_common.h: split_pair2): h = fp16(x),
// l = fp16(x - h), |x - h - l| <= 2^-22 |x|; the weights are multiplied by a power of two S (max |w| S in [2^14, 2^15), so their low terms
// stay normal), split once at pack time (hip.pack_node_weights) and the sums multiplied by 1 / S (exact) in the epilogue:
//     x w S = h_x l_w + l_x h_w + h_x h_w   [+ l_x l_w, <= 2^-22 relative, dropped]
// three v_mfma_f32_16x16x32_f16 (16 cycles each, K = 32) where the exact-fp32 path needs eight v_mfma_f32_16x16x4_f32 (32 cycles each).
// Rounds 2-4 used three bf16 terms and six products (exact up to 2^-25); the two schemes differ from fp64 by the same amount, the fp32
// accumulation error they share (ipa_common.h).  x is split in registers (6 VALU ops per pair of values).
//
// A task is (32 residues, half of the head's tiles): 6 tiles x 4 k-steps x 2 row tiles x 3 products = 144 MFMAs on 48 accumulator
// registers, with each weight fragment read from LDS once per TWO row tiles (at one row tile per read the kernel would sit exactly on
// the 128 B/clk LDS limit).  Register-only epilogue: with this row order an accumulator tile IS a fragment slot -- lane (residue, kq)
// holds 4 consecutive channels, or (x, y, z, pad) of one point, so the frame transform needs no cross-lane traffic.  The value tiles
// run with the operands swapped (accumulator = [residue 4 kq + r][channel fm]), which is the key-major layout of the aggregation
// operand.
// Traffic per launch at M = 8192: x re-read from L2 (24 x 4 MB), weights 12 x 96 KB, fragments written once (75 MB).
#include "ipa_common.h"
#include "kernels.h"

#ifdef NF_TIMING   // developer build: clocks of one workgroup
#include <cstdio>
__device__ long long g_nf_timing[16][8];
#endif

namespace abopt {

constexpr int NF_F = 128;                                       // node feature width (ga.py:54-66 with node_feat_dim = 128)
#ifndef NF_SPLIT
#define NF_SPLIT 1       // 1 (round 5): a workgroup owns HALF a head (six tiles, 48 KB of LDS), three 4-wave workgroups per CU | 0: a whole head (96 KB), one 12-wave workgroup per CU
#endif
#ifndef NF_WAVES_
#define NF_WAVES_ (NF_SPLIT ? 4 : 12)
#endif
#ifndef NF_WGPC
#define NF_WGPC 3        // NF_SPLIT: workgroups per CU the grid is sized for
#endif
constexpr int NF_TILES = 12, NF_HT = NF_TILES / 2, NF_WAVES = NF_WAVES_;       // 12 waves = 3 per SIMD (152 VGPRs): the task epilogues of one wave hide behind the MFMAs of two others (8 -> 12 waves: 35.8 -> 34.9 us at M = 8192, 205 -> 188 us at M = 48000, same box)
constexpr int NF_KS = NF_F / 32, NF_SPL = 2;                // k-steps of 32, fp16 terms per fp32 value
constexpr int NF_HEAD_VEC = NF_TILES * NF_KS * NF_SPL * 64;    // 16-byte vectors (8 fp16) per head: [tile][k-step][term][lane]
constexpr int NF_LDS_VEC = NF_SPLIT ? NF_HEAD_VEC / 2 : NF_HEAD_VEC;      // what a workgroup keeps in LDS


__device__ __forceinline__ float quad_bcast0(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x00, 0xf, 0xf, false)); }
__device__ __forceinline__ float quad_bcast1(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x55, 0xf, 0xf, false)); }
__device__ __forceinline__ float quad_bcast2(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xAA, 0xf, 0xf, false)); }

// One task: tiles [HALF * 6, HALF * 6 + 6) of head h for the row tiles tile0, tile0 + 1.
template <int HALF, bool XT>
__device__ __forceinline__ void nf_task(const float* __restrict__ x, const unsigned* __restrict__ xt, const u32x4* wl, const float* __restrict__ R, const float* __restrict__ t,
                                        float* __restrict__ qfrag, float* __restrict__ kvfrag, int L, int nchunk, int total_tiles, int tile0, int h,
                                        float ch_, float m2c, float winv, int lane, int fm, int kq, int qk_terms) {
    int64_t rowbase[2], row[2];
    int cbs[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int tile = min(tile0 + rt, total_tiles - 1);                       // odd tile count: the last task computes its last tile twice, stores once
        const int n = tile / nchunk;
        cbs[rt] = tile % nchunk;
        rowbase[rt] = (int64_t)n * L;
        row[rt] = rowbase[rt] + min(cbs[rt] * JC + fm, L - 1);                   // rows past the end: clamped copies (finite; the core never stores them)
    }
    // lane (row fm, kq) holds k = 32 s + 8 kq + i of its row for k-step s -- as fp32 (split here) or, when the kernel that produced x also wrote its terms
    // (xt: [row][64 words of high terms | 64 words of low terms]), as the two 16-byte term vectors themselves
    f32x4 xa[2][2];
    Split2 xn[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if constexpr (XT) {
            xn[rt].h = *reinterpret_cast<const u32x4*>(xt + row[rt] * NF_F + kq * 4);
            xn[rt].l = *reinterpret_cast<const u32x4*>(xt + row[rt] * NF_F + 64 + kq * 4);
        } else {
            xa[rt][0] = *reinterpret_cast<const f32x4*>(x + row[rt] * NF_F + kq * 8);
            xa[rt][1] = *reinterpret_cast<const f32x4*>(x + row[rt] * NF_F + kq * 8 + 4);
        }
    }
    f32x4 acc[2][NF_HT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int T = 0; T < NF_HT; ++T) acc[rt][T] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4* wh = wl + (NF_SPLIT ? 0 : (HALF * NF_HT) * (NF_KS * NF_SPL * 64)) + lane;
    // weight fragments of step g + 1 are read from LDS before the 12 MFMAs of step g are issued (hipcc left to itself hoists every read)
    u32x4 wa[2][NF_SPL];
#pragma unroll
    for (int sp = 0; sp < NF_SPL; ++sp) wa[0][sp] = wh[sp * 64];
    Split2 xs[2];
#pragma unroll
    for (int g = 0; g < NF_KS * NF_HT; ++g) {
        const int s = g / NF_HT, T = g % NF_HT;
        if (T == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if constexpr (XT) xs[rt] = xn[rt];
                else xs[rt] = split2(xa[rt][0], xa[rt][1]);
            }
            if (s + 1 < NF_KS) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    if constexpr (XT) {
                        xn[rt].h = *reinterpret_cast<const u32x4*>(xt + row[rt] * NF_F + (s + 1) * 16 + kq * 4);
                        xn[rt].l = *reinterpret_cast<const u32x4*>(xt + row[rt] * NF_F + 64 + (s + 1) * 16 + kq * 4);
                    } else {
                        xa[rt][0] = *reinterpret_cast<const f32x4*>(x + row[rt] * NF_F + (s + 1) * 32 + kq * 8);
                        xa[rt][1] = *reinterpret_cast<const f32x4*>(x + row[rt] * NF_F + (s + 1) * 32 + kq * 8 + 4);
                    }
                }
            }
        }
#if defined(NF_ABL) && (NF_ABL & 8)
        if (g == 0) {
#else
        if (g + 1 < NF_KS * NF_HT) {
#endif
            const int sn = (g + 1) / NF_HT, Tn = (g + 1) % NF_HT;
#pragma unroll
            for (int sp = 0; sp < NF_SPL; ++sp) wa[(g + 1) & 1][sp] = wh[((Tn * NF_KS + sn) * NF_SPL + sp) * 64];
        }
#if defined(NF_ABL) && (NF_ABL & 8)
        const u32x4 wH = wa[g ? 1 : 0][0], wL = wa[g ? 1 : 0][1];
#else
        const u32x4 wH = wa[g & 1][0], wL = wa[g & 1][1];
#endif
        const bool swap = (HALF == 1) && (T >= 2);                               // value tiles: x is the A operand -> accumulator [residue 4 kq + r][channel fm]
        // smallest terms first; the two row tiles alternate so consecutive MFMAs never depend on each other
#define NF_PROD(XT, WT)                                                                                                         \
        if (swap) { acc[0][T] = mfma_h(xs[0].XT, WT, acc[0][T]); acc[1][T] = mfma_h(xs[1].XT, WT, acc[1][T]); }               \
        else      { acc[0][T] = mfma_h(WT, xs[0].XT, acc[0][T]); acc[1][T] = mfma_h(WT, xs[1].XT, acc[1][T]); }
        NF_PROD(l, wH) NF_PROD(h, wL) NF_PROD(h, wH)
#undef NF_PROD
        __builtin_amdgcn_sched_barrier(0);
    }
    auto sq = [](const f32x4& g) { return fmaf(g[2], g[2], fmaf(g[1], g[1], g[0] * g[0])); };
#if defined(NF_ABL) && (NF_ABL & 4)
    {
        float sum = 0.f;
        for (int rt = 0; rt < 2; ++rt) for (int T = 0; T < NF_HT; ++T) for (int r = 0; r < 4; ++r) sum += acc[rt][T][r];
        if (sum != 1.2345e-30f) return;
    }
#endif
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (tile0 + rt >= total_tiles) break;
        const int tile = tile0 + rt;
#pragma unroll
        for (int T = 0; T < NF_HT; ++T) acc[rt][T] *= winv;                      // sums of S w x -> w x (exact)
        // frames are fetched only now: x fragments and weight registers are dead
        float Rm[9], tv[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm[k] = R[row[rt] * 9 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tv[k] = t[row[rt] * 3 + k];
        // p <- R p + t (geometry.py:72-91) on (x, y, z, pad) of one point of residue fm
        auto to_global = [&](const f32x4& p) {
            return (f32x4){Rm[0] * p[0] + Rm[1] * p[1] + Rm[2] * p[2] + tv[0], Rm[3] * p[0] + Rm[4] * p[1] + Rm[5] * p[2] + tv[1],
                           Rm[6] * p[0] + Rm[7] * p[1] + Rm[8] * p[2] + tv[2], 0.f};
        };
        f32x4* outq = reinterpret_cast<f32x4*>(qfrag) + ((int64_t)tile * H + h) * (4 * 64) + lane;
        f32x4* outk = reinterpret_cast<f32x4*>(kvfrag) + ((int64_t)tile * H + h) * (8 * 64) + lane;
        if (HALF == 0) {
            // ---- q, k: accumulator row 4 kq + r = channel, column fm = residue
            const float s = 0.17677669529663687f;                                // 1 / sqrt(D), ga.py:84
            if (qk_terms) {
                // round 6: the consumer (ipa_core32_kernel<*, true>) multiplies the 32 channels of q / sqrt(D) and k as two fp16 terms each -- slot 0 holds the high
                // terms, slot 1 the low terms; K slot e of lane (residue fm, kq) is channel 4 kq + e (e < 4) or 16 + 4 kq + e - 4: the same map on both sides
                const Split2 tq = split2(acc[rt][0] * s, acc[rt][1] * s), tk = split2(acc[rt][2], acc[rt][3]);
                outq[0] = __builtin_bit_cast(f32x4, tq.h); outq[64] = __builtin_bit_cast(f32x4, tq.l);
                outk[0] = __builtin_bit_cast(f32x4, tk.h); outk[64] = __builtin_bit_cast(f32x4, tk.l);
            } else {
            outq[0] = acc[rt][0] * s; outq[64] = acc[rt][1] * s;
            outk[0] = acc[rt][2]; outk[64] = acc[rt][3];
            }
            // ---- q_pts: point kq (tile A) and 4 + kq (tile B) of residue fm
            f32x4 ga = to_global(acc[rt][4]), gb = to_global(acc[rt][5]);
            const float nq = rows_sum(sq(ga) + sq(gb));                          // |q_pts|^2 over the head's 8 points
            ga *= m2c; gb *= m2c;
            ga[3] = kq == 0 ? ch_ * nq : (kq == 1 ? ch_ : 0.f);                  // norm step, q side
            gb[3] = 0.f;
            outq[128] = ga; outq[192] = gb;
        } else {
            // ---- k_pts
            f32x4 ga = to_global(acc[rt][0]), gb = to_global(acc[rt][1]);
            const float nk = rows_sum(sq(ga) + sq(gb));
            ga[3] = kq == 0 ? 1.f : (kq == 1 ? nk : 0.f);                        // norm step, k side
            outk[128] = ga; outk[192] = gb;
            // ---- v, v_pts: accumulator row 4 kq + r = residue, column fm = channel / (point fm >> 2, coordinate fm & 3)
            const int c = fm & 3, cr = min(c, 2);                                // row cr of R and t[cr] of residue 4 kq + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t rr = rowbase[rt] + min(cbs[rt] * JC + kq * 4 + r, L - 1);
                const float r0 = R[rr * 9 + cr * 3], r1 = R[rr * 9 + cr * 3 + 1], r2 = R[rr * 9 + cr * 3 + 2], r3 = t[rr * 3 + cr];
                const float xa_ = quad_bcast0(acc[rt][4][r]), ya = quad_bcast1(acc[rt][4][r]), za = quad_bcast2(acc[rt][4][r]);
                const float xb = quad_bcast0(acc[rt][5][r]), yb = quad_bcast1(acc[rt][5][r]), zb = quad_bcast2(acc[rt][5][r]);
                float g0 = r0 * xa_ + r1 * ya + r2 * za + r3;
                float g1 = r0 * xb + r1 * yb + r2 * zb + r3;
                if (c == 3) { g0 = 0.f; g1 = 0.f; }
                outk[(4 + r) * 64] = (f32x4){acc[rt][2][r], acc[rt][3][r], g0, g1};
            }
        }
    }
}

// XT: x arrives as fp16 terms (xt) -- its own instantiation, so that neither path carries the other's registers (one kernel with a run-time switch: 172 instead of
// 144 registers, two waves per SIMD instead of three, 24.4 -> 30.7 us at the bench shape)
template <bool XT>
__global__ __launch_bounds__(NF_WAVES * 64) void node_frags_kernel(const float* __restrict__ x, const unsigned* __restrict__ xt, const float* __restrict__ wfrag, const float* __restrict__ R,
                                                                   const float* __restrict__ t, const float* __restrict__ spatial_coef,
                                                                   float* __restrict__ qfrag, float* __restrict__ kvfrag, int L, int nchunk,
                                                                   int total_tiles, int qk_terms) {
    extern __shared__ __attribute__((aligned(16))) char nf_smem[];
    u32x4* wl = reinterpret_cast<u32x4*>(nf_smem);                             // [12 tiles][4 k-steps][2 terms][64]
    const int h = NF_SPLIT ? blockIdx.y >> 1 : blockIdx.y, tid = threadIdx.x, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const int half = blockIdx.y & 1;                                           // NF_SPLIT: which six tiles this workgroup owns
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef NF_TIMING
    const long long c0 = clock64(), w0 = wall_clock64();
#endif
    {
        const u32x4* wg = reinterpret_cast<const u32x4*>(wfrag) + (int64_t)h * NF_HEAD_VEC + (NF_SPLIT ? half * NF_LDS_VEC : 0);
        static_assert(NF_LDS_VEC % (NF_WAVES * 64) == 0, "weight load loop");
#pragma unroll
        for (int e = 0; e < NF_LDS_VEC / (NF_WAVES * 64); ++e) wl[e * (NF_WAVES * 64) + tid] = wg[e * (NF_WAVES * 64) + tid];
    }
    const float sc = spatial_coef[h];
    const float winv = wfrag[(int64_t)H * NF_HEAD_VEC * 4 + 1];                  // 1 / S behind the packed weights
    const float gamma = (sc > 20.f) ? sc : log1pf(expf(sc));                     // softplus, ga.py:108
    const float ch_ = (-1.f * gamma * 0.16666666666666666f) / 2.f;               // -gamma sqrt(2/(9*8)) / 2, ga.py:109-110
    const float m2c = -2.f * ch_;
    __syncthreads();
#ifdef NF_TIMING
    const long long c1 = clock64();
#endif
    // every workgroup owns a contiguous, equal (+-1) share of the head's tasks (pair of row tiles, half of the tiles); its waves take
    // them round-robin, so both halves of a row-tile pair run on neighbouring waves and share the x rows in L1
    // (NF_SPLIT: a task index is a pair of row tiles, all of this workgroup's tasks are of its own half)
    const int ntask = NF_SPLIT ? (total_tiles + 1) / 2 : 2 * ((total_tiles + 1) / 2);
    const int t_lo = (int)((int64_t)ntask * blockIdx.x / gridDim.x), t_hi = (int)((int64_t)ntask * (blockIdx.x + 1) / gridDim.x);
#ifdef NF_TIMING
    long long te[3] = {0, 0, 0};
    int ti = 0;
#endif
    for (int task = t_lo + wave; task < t_hi; task += NF_WAVES) {
        const int tile0 = NF_SPLIT ? task * 2 : (task >> 1) * 2;
        if (NF_SPLIT ? half : (task & 1)) nf_task<1, XT>(x, xt, wl, R, t, qfrag, kvfrag, L, nchunk, total_tiles, tile0, h, ch_, m2c, winv, lane, fm, kq, qk_terms);
        else          nf_task<0, XT>(x, xt, wl, R, t, qfrag, kvfrag, L, nchunk, total_tiles, tile0, h, ch_, m2c, winv, lane, fm, kq, qk_terms);
#ifdef NF_TIMING
        if (ti < 3) te[ti++] = clock64() - c0;
#endif
    }
#ifdef NF_TIMING
    if (blockIdx.x == NF_TIMING && blockIdx.y == 3 && lane == 0) {        // -DNF_TIMING=<row group>: 5 owns 25 tasks at the bench shape, 0 owns 24
        long long* o = g_nf_timing[wave];
        o[0] = c1 - c0; o[1] = clock64() - c0; o[2] = wall_clock64() - w0; o[3] = te[0]; o[4] = te[1]; o[5] = (t_hi - t_lo); o[6] = te[2];
    }
#endif
}

size_t node_wfrag_floats() { return (size_t)H * NF_HEAD_VEC * 4 + 4; }      // + {S, 1 / S, 0, 0}

int launch_node_frags(const float* x, const float* wfrag, const float* R, const float* t, const float* spatial_coef, float* qfrag, float* kvfrag,
                      int N, int L, hipStream_t st, int qk_terms, const float* x_terms) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    const int nchunk = (L + JC - 1) / JC, total = N * nchunk;
    int cus = 0, rc;
    if ((rc = device_cu_count(&cus))) return rc;
    static LdsConfig lds_cfg[2];
    const bool xt = x_terms != nullptr;
    if ((rc = ensure_dynamic_lds(xt ? reinterpret_cast<const void*>(node_frags_kernel<true>) : reinterpret_cast<const void*>(node_frags_kernel<false>), NF_LDS_VEC * 16,
                                 lds_cfg[xt])))
        return rc;
#if NF_SPLIT
    // 24 (head, half) columns of workgroups x `groups` shares of the row-tile pairs; 48 KB of LDS each: NF_WGPC = 3 per CU.  At the bench shape
    // (256 pairs, 256 CUs): 32 groups of 8 pairs, two tasks for each of the four waves -- every wave of the chip does the same amount of work.
    const int ntask = (total + 1) / 2;
    const int groups = max(1, min(cus * NF_WGPC / (2 * H), (ntask + NF_WAVES - 1) / NF_WAVES));
    const dim3 grid(groups, 2 * H);
#else
    const int ntask = 2 * ((total + 1) / 2);
    const int groups = max(1, min(cus / H, (ntask + NF_WAVES - 1) / NF_WAVES));          // one workgroup per CU: 96 KB of LDS each
    const dim3 grid(groups, H);
#endif
    if (xt)
        hipLaunchKernelGGL(node_frags_kernel<true>, grid, dim3(NF_WAVES * 64), NF_LDS_VEC * 16, st, x, reinterpret_cast<const unsigned*>(x_terms), wfrag, R, t,
                           spatial_coef, qfrag, kvfrag, L, nchunk, total, qk_terms);
    else
        hipLaunchKernelGGL(node_frags_kernel<false>, grid, dim3(NF_WAVES * 64), NF_LDS_VEC * 16, st, x, nullptr, wfrag, R, t, spatial_coef, qfrag, kvfrag, L, nchunk,
                           total, qk_terms);
    ABOPT_LAUNCH_CHECK();
#ifdef NF_TIMING
    {
        long long hh[16][8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_nf_timing), sizeof(hh));
        static int calls = 0;
        if (++calls == 8)
            for (int w = 0; w < NF_WAVES; ++w)
                fprintf(stderr, "[nf timing WG(%d,3) wave %d] W load %lld | total %lld shader clk = %lld x 10 ns wall | task ends %lld %lld %lld | tasks of WG %lld\n",
                        (int)NF_TIMING, w, hh[w][0], hh[w][1], hh[w][2], hh[w][3], hh[w][4], hh[w][6], hh[w][5]);
    }
#endif
    return ABOPT_OK;
}

}  // namespace abopt

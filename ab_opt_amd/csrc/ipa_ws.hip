// IPA core, wave-specialised version ("ws") for CDNA4: same maths and tiling as ipa_core_v1 (ipa.hip: 16 query rows
// per workgroup, keys in chunks of 16, online softmax, fp32 MFMA everywhere), but the two kinds of work run in
// DIFFERENT waves of a 512-thread workgroup, pipelined one chunk apart:
//
//   waves 0-3  "pair waves":  phase B(t)  -- stream z, pair-bias MFMA, softmax, pair aggregation        (4 query rows each)
//   waves 4-7  "node waves":  phase C(t-1) then phase A(t+1) -- node/point aggregation, q.k and point distances (3 heads each)
//
// Each SIMD hosts one pair wave and one node wave, so the matrix pipe always has two independent instruction streams
// with complementary mixes (HBM streaming + MFMA vs L2 gathers + VALU + MFMA), one barrier per chunk instead of two, and
// every global operand is fetched a whole pipeline stage before it is used (node waves hold the k/kg and v/vp fragments
// of the next stage in registers; pair waves prefetch the next z row).  The S/P tile and the rescale factors are
// double-buffered in LDS so the stages never touch the same buffer between two barriers.
//
// Reference semantics: AbDock/src/modules/encoders/ga.py:11-26,81-147 (see ipa.hip for the line-by-line mapping).
#include "ipa_common.h"
#include "kernels.h"
#include <cstdlib>
#include <cstdio>

namespace abopt {

constexpr int WS_MAX_L = 2048;       // longest complex this kernel takes (key mask staged in LDS); longer ones use ipa_core_v1

struct WsSmem {
    float sp[2][BI][16 * PLD + 4];   // S then P per chunk parity, [i][h*PLD + j]
    float zst[8][JC][ZSLD];          // per-pair-wave z staging (transposes the chunk for the pair-bias MFMA)
    float nq[BI][16];                // |q_pts|^2 per (query row, head)
    float scl[2][BI][16];            // rescale factor of chunk parity
    float lsum[BI][16];              // softmax denominators
    float wbs[16][C + 4];            // pair-bias weights, rows 12..15 zero
    float coef[16];                  // -softplus(spatial_coef) sqrt(2/(9 P)) / 2
    float q[BI][H * D + 4];          // queries of the block (phase A operand A)
    uint8_t mk[WS_MAX_L + 64];           // key mask of the sample (no global loads besides the z ring inside the pair-wave loop)
};

struct KFrag { float4 k0, k1; float2 g[3]; float nk; }; // phase A operands of one head: 8 key channels, 6 key-point coords, |k_pts|^2
struct VFrag { float2 v[4], p[4]; };                   // phase C operands of one head: 4 keys x (2 value channels, 2 point coords)

// operands of (chunk, head) from the fragment-order copy written by points_to_global_frags_kernel (ipa.hip): 4 + 4 fully
// coalesced 1 KB loads instead of 14 row gathers
__device__ __forceinline__ void load_kfrag(KFrag& f, const f32x4* __restrict__ fr, int lane) {
    const f32x4 a = fr[lane], b = fr[64 + lane], c = fr[128 + lane], d = fr[192 + lane];
    f.k0 = make_float4(a[0], a[1], a[2], a[3]); f.k1 = make_float4(b[0], b[1], b[2], b[3]);
    f.g[0] = make_float2(c[0], c[1]); f.g[1] = make_float2(c[2], c[3]); f.g[2] = make_float2(d[0], d[1]);
    f.nk = d[2];
}

__device__ __forceinline__ void load_vfrag(VFrag& f, const f32x4* __restrict__ fr, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f32x4 a = fr[(4 + s) * 64 + lane];
        f.v[s] = make_float2(a[0], a[1]); f.p[s] = make_float2(a[2], a[3]);
    }
}

#ifdef WS_TIMING   // developer build (-DWS_TIMING): s_memtime section timers of one workgroup
#define TSTAMP(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
__device__ long long g_ws_timing[2][8];
#else
#define TSTAMP(k)
#endif

// NPW = number of pair waves; each owns RPW = 16 / NPW query rows.  Only NPW = 4 is instantiated: 8 pair waves (3 waves per SIMD,
// <= 168 VGPRs) spilled and measured 349 us against 239 us.
template <bool DBG, int NPW, bool CACHED>
__global__ __launch_bounds__((NPW + 4) * 64, (NPW + 4) / 4) void ipa_core_ws_kernel(const float* __restrict__ proj, const float* __restrict__ z,
                                                             const uint8_t* __restrict__ mask, const float* __restrict__ R,
                                                             const float* __restrict__ t, const float* __restrict__ Wb,
                                                             const float* __restrict__ spatial_coef, float* __restrict__ feat,
                                                             float* __restrict__ dbg_logits, const float* __restrict__ pbc, const float* __restrict__ kvfrag, int N, int L, int nib, int xcd_remap, int z_shared) {
    __shared__ __attribute__((aligned(16))) WsSmem sm;
    int n, ib;
    {   // all i-blocks of a sample on one XCD when N % 8 == 0 (L2 locality of its k/v tiles; speed only)
        const int b = blockIdx.x;
        if (xcd_remap) { const int xcd = b & 7, k = b >> 3; n = xcd + 8 * (k / nib); ib = k % nib; }
        else { n = b / nib; ib = b % nib; }
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const int nchunk = (L + JC - 1) / JC;
#ifdef WS_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#endif
    constexpr int RPW = BI / NPW, UC = 4 / RPW, NT = (NPW + 4) * 64;      // rows per pair wave, chunks per ring revolution, threads
    const bool pair_wave = wave < NPW;
    const int w4 = pair_wave ? wave : wave - NPW;                           // index within the role
    const int nchunk2 = ((nchunk + UC - 1) / UC) * UC;                     // both roles run the same (possibly padded) number of chunks
    const int i0 = ib * BI;
    const int64_t rowbase = (int64_t)n * L;
    const float* projn = proj + rowbase * NP;

    // ---- prologue (all 8 waves), run by each role AFTER it has put its first global loads in flight
#define WS_PROLOGUE_FILL() {                                                                                            \
    if (tid < BI * 16) {                                                                                               \
        const int il = tid >> 4, h = tid & 15;                                                                         \
        sm.nq[il][h] = (h < H) ? projn[(int64_t)min(i0 + il, L - 1) * NP + OFF_NQ + h] : 0.f;                          \
    }                                                                                                                  \
    for (int e = tid; e < BI * (H * D / 4); e += NT) {                                                                 \
        const int il = e / (H * D / 4), c4 = e % (H * D / 4);                                                          \
        *reinterpret_cast<float4*>(&sm.q[il][c4 * 4]) = reinterpret_cast<const float4*>(projn + (int64_t)min(i0 + il, L - 1) * NP + OFF_Q)[c4]; \
    }                                                                                                                  \
    for (int e = tid; e < 16 * (C / 4); e += NT) {                                                                     \
        const int h = e / (C / 4), c4 = e % (C / 4);                                                                   \
        float4 w4v = make_float4(0.f, 0.f, 0.f, 0.f);                                                                  \
        if (h < H) w4v = reinterpret_cast<const float4*>(Wb + h * C)[c4];                                              \
        *reinterpret_cast<float4*>(&sm.wbs[h][c4 * 4]) = w4v;                                                          \
    }                                                                                                                  \
    for (int e = tid; e < nchunk2 * JC; e += NT) sm.mk[e] = (e < L) ? mask[rowbase + e] : 0;                           \
    if (tid < H) {                                                                                                     \
        const float sc = spatial_coef[tid];                                                                            \
        const float gamma = (sc > 20.f) ? sc : log1pf(expf(sc));                                                       \
        sm.coef[tid] = (-1.f * gamma * 0.16666666666666666f) / 2.f;                                                    \
    }                                                                                                                  \
    }

    if (pair_wave) {
        // =========================================================================== pair waves: phase B
        bool mi_b[RPW];
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii) { const int i = i0 + w4 * RPW + ii; mi_b[ii] = (i < L) && mask[rowbase + i] != 0; }
        float m_run[RPW], l_run[RPW];
        f32x4 accP[RPW][4];
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii) {
            m_run[ii] = -INFINITY; l_run[ii] = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) accP[ii][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // z ring: 4 register slots of one row chunk each (4 x 16 B per lane); rows are requested 3 ahead of their use so
        // that ~12 KB per pair wave (48 KB per CU) is always in flight: HBM latency under load is ~2 us.
        f32x4 ring[4][4];
        f32x4 ringb[4];                                                     // CACHED: the row chunk's precomputed pair bias, lane (head fm, keys 4 kq ..)
        // wave-uniform row base in SGPRs + 32-bit per-lane byte offsets: three VALU per load instead of five 64-bit ones
        const int w4u = __builtin_amdgcn_readfirstlane(w4);
        const unsigned lane_b = (unsigned)fm * 16u;
#define WS_ISSUE_Z(SLOT, ROW, CH)                                                                                        \
    {                                                                                                                    \
        const int64_t zrow_ = (z_shared ? 0 : rowbase) + min(i0 + w4u * RPW + (ROW), L - 1);   /* z_shared: one pair_feat for the whole batch */ \
        const char* zi_ = reinterpret_cast<const char*>(z + (zrow_ * (int64_t)L) * C);                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                              \
            ring[SLOT][r_] = *reinterpret_cast<const f32x4*>(zi_ + ((unsigned)min((CH) * JC + kq * 4 + r_, L - 1) * (unsigned)(C * 4) + lane_b)); \
        if (CACHED) ringb[SLOT] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(pbc + (zrow_ * nchunk + min((CH), nchunk - 1)) * 256) + (unsigned)lane * 16u);  \
    }
        // ring position p = c * RPW + ii (c = chunk within the revolution) is also the slot; requests run 3 positions ahead
        WS_ISSUE_Z(0, 0 % RPW, 0 / RPW) WS_ISSUE_Z(1, 1 % RPW, 1 / RPW) WS_ISSUE_Z(2, 2 % RPW, 2 / RPW)
        WS_PROLOGUE_FILL()
        __syncthreads();                                                    // prologue tile visible
        __syncthreads();                                                    // barrier #0: S(0) ready
        TSTAMP(0)
        for (int ch0 = 0; ch0 < nchunk2; ch0 += UC) {
#pragma unroll
          for (int c = 0; c < UC; ++c) {
            const int ch = ch0 + c;
            const int jc0 = ch * JC, buf = ch & 1;
            bool mj[4], jv[4];
            const uint32_t mk4 = *reinterpret_cast<const uint32_t*>(&sm.mk[jc0 + kq * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int j = jc0 + kq * 4 + r; jv[r] = j < L; mj[r] = ((mk4 >> (8 * r)) & 0xffu) != 0; }
#pragma unroll
            for (int ii = 0; ii < RPW; ++ii) {
                const int il = w4 * RPW + ii;
                constexpr int dummy_ = 0; (void)dummy_;
                const int pos = c * RPW + ii;                               // compile-time after unrolling
                WS_ISSUE_Z((pos + 3) & 3, (pos + 3) % RPW, ch0 + (pos + 3) / RPW)       // 3 positions ahead (clamped past the end: harmless re-read)
                f32x4 zr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) zr[r] = ring[pos][r];
                TSTAMP(1)
                f32x4 acc;
                if (CACHED) {
                    acc = ringb[pos];                                       // pair bias of this (row, chunk) from the per-call cache
                } else {
                    wave_lds_sync();                                        // previous row's transposed reads are done
#pragma unroll
                    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(&sm.zst[w4][kq * 4 + r][fm * 4]) = zr[r];
                    wave_lds_sync();                                        // cross-lane transpose through LDS
                    f32x4 acc4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 za = *reinterpret_cast<const float4*>(&sm.zst[w4][fm][kq * 16 + q * 4]);
                        const float4 wv = *reinterpret_cast<const float4*>(&sm.wbs[fm][kq * 16 + q * 4]);
                        acc4[q] = mfma4(za.x, wv.x, (f32x4){0.f, 0.f, 0.f, 0.f});
                        acc4[q] = mfma4(za.y, wv.y, acc4[q]); acc4[q] = mfma4(za.z, wv.z, acc4[q]); acc4[q] = mfma4(za.w, wv.w, acc4[q]);
                    }
                    acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
                }
                TSTAMP(2)
                const float4 tns = *reinterpret_cast<const float4*>(&sm.sp[buf][il][fm * PLD + kq * 4]);
                float sv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float lt = (f4get(tns, r) + acc[r]) * 0.5773502691896258f;
                    if (DBG && jv[r] && fm < H && (i0 + il) < L) dbg_logits[((rowbase + i0 + il) * L + jc0 + kq * 4 + r) * H + fm] = lt;
                    if (!(mi_b[ii] && mj[r])) lt -= 1e5f;                   // ga.py:20-23
                    sv[r] = (fm < H) ? lt : 0.f;
                    sv[r] = jv[r] ? sv[r] : -INFINITY;
                }
                const float mx = rows_max(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
                const float m_new = fmaxf(m_run[ii], mx);
                const float sc = __expf(m_run[ii] - m_new);
                float pv[4], ps = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pv[r] = __expf(sv[r] - m_new); ps += pv[r]; }
                ps = rows_sum(ps);
                l_run[ii] = l_run[ii] * sc + ps;
                m_run[ii] = m_new;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) accP[ii][mt] *= sc;
                TSTAMP(3)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) accP[ii][mt] = mfma4(zr[r][mt], pv[r], accP[ii][mt]);
                *reinterpret_cast<float4*>(&sm.sp[buf][il][fm * PLD + kq * 4]) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                if (kq == 0) sm.scl[buf][il][fm] = sc;
                TSTAMP(4)
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                                                // barrier #(ch+1)
            TSTAMP(5)
          }
        }
        // alpha = P / l, zero for masked queries (ga.py:24-25); pair features out
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii) {
            const int il = w4 * RPW + ii, i = i0 + il;
            if (kq == 0) sm.lsum[il][fm] = l_run[ii];
            if (i < L && fm < H) {
                const float inv = mi_b[ii] ? 1.f / l_run[ii] : 0.f;
                float* fo = feat + (rowbase + i) * FEAT + fm * C + kq * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    reinterpret_cast<float4*>(fo)[r] = make_float4(accP[ii][0][r] * inv, accP[ii][1][r] * inv, accP[ii][2][r] * inv, accP[ii][3][r] * inv);
            }
        }
        TSTAMP(6)
        __syncthreads();                                                    // F1: lsum visible, node waves done with C(last)
        __syncthreads();                                                    // F2: aggregated points in LDS
    } else {
        // =========================================================================== node waves: phases A and C
        bool mi_a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = i0 + kq * 4 + r; mi_a[r] = (i < L) && mask[rowbase + i] != 0; }
        f32x4 accV[3][2], accT[3][2];
#pragma unroll
        for (int hh = 0; hh < 3; ++hh)
#pragma unroll
            for (int k = 0; k < 2; ++k) { accV[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; accT[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        KFrag kf[3];
        VFrag vf[3];
        const f32x4* kvn = reinterpret_cast<const f32x4*>(kvfrag) + (int64_t)n * nchunk * H * 512;   // this sample's fragments
        float2 qgf[3][3];                                                   // query-point fragments (A operand: row = query fm, same K permutation as KFrag::g)
        {
            const float* qprow = projn + (int64_t)min(i0 + fm, L - 1) * NP + OFF_QP + 2 * kq;
#pragma unroll
            for (int hh = 0; hh < 3; ++hh)
#pragma unroll
                for (int u = 0; u < 3; ++u) qgf[hh][u] = *reinterpret_cast<const float2*>(qprow + (w4 * 3 + hh) * (P * 3) + 8 * u);
        }

        auto phase_a = [&](int ch) {                                        // S(ch) -> sp[ch & 1], consuming kf
            const int buf = ch & 1;
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                const int h = w4 * 3 + hh;
                const float coefh = sm.coef[h];
                // A operand: row = query fm, K-permuted: step s <-> channel 8 kq + s (same permutation on the key side)
                const float4 q0 = *reinterpret_cast<const float4*>(&sm.q[fm][h * D + kq * 8]);
                const float4 q1 = *reinterpret_cast<const float4*>(&sm.q[fm][h * D + kq * 8 + 4]);
                f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;                 // two chains (dependent-MFMA latency)
#pragma unroll
                for (int s = 0; s < 4; ++s) { acc0 = mfma4(f4get(q0, s), f4get(kf[hh].k0, s), acc0); acc1 = mfma4(f4get(q1, s), f4get(kf[hh].k1, s), acc1); }
                const f32x4 acc = acc0 + acc1;
                // squared point distances: |q|^2 + |k|^2 - 2 q.k, the cross term on the matrix cores (K = 24 coordinates)
                f32x4 accp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 3; ++u) { accp = mfma4(qgf[hh][u].x, kf[hh].g[u].x, accp); accp = mfma4(qgf[hh][u].y, kf[hh].g[u].y, accp); }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                               // accumulator row = query 4 kq + r, column = key fm
                    const float d2 = (sm.nq[kq * 4 + r][h] + kf[hh].nk) - 2.f * accp[r];
                    sm.sp[buf][kq * 4 + r][h * PLD + fm] = acc[r] * 0.17677669529663687f + d2 * coefh;
                }
            }
        };
        auto issue_k = [&](int ch) {                                        // fetch phase-A operands of chunk ch
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) load_kfrag(kf[hh], kvn + ((int64_t)min(ch, nchunk - 1) * H + w4 * 3 + hh) * 512, lane);
        };
        auto issue_v = [&](int ch) {                                        // fetch phase-C operands of chunk ch
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) load_vfrag(vf[hh], kvn + ((int64_t)min(ch, nchunk - 1) * H + w4 * 3 + hh) * 512, lane);
        };
        auto phase_c = [&](int ch) {                                        // consume P(ch) from sp[ch & 1] and vf
            const int buf = ch & 1;
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                const int h = w4 * 3 + hh;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = sm.scl[buf][kq * 4 + r][h];
                    accV[hh][0][r] *= sc; accV[hh][1][r] *= sc; accT[hh][0][r] *= sc; accT[hh][1][r] *= sc;
                }
                const VFrag& vfh = vf[hh];
                const float4 pa = *reinterpret_cast<const float4*>(&sm.sp[buf][fm][h * PLD + kq * 4]);   // A: row = query fm, step s <-> key 4 kq + s
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float a = f4get(pa, s);
                    accV[hh][0] = mfma4(a, vfh.v[s].x, accV[hh][0]);
                    accV[hh][1] = mfma4(a, vfh.v[s].y, accV[hh][1]);
                    accT[hh][0] = mfma4(a, vfh.p[s].x, accT[hh][0]);
                    accT[hh][1] = mfma4(a, vfh.p[s].y, accT[hh][1]);
                }
            }
        };

        issue_k(0);
        WS_PROLOGUE_FILL()
        __syncthreads();                                                    // prologue tile visible
        phase_a(0);
        if (nchunk2 > 1) issue_k(1);
        issue_v(0);
        __syncthreads();                                                    // barrier #0
        TSTAMP(0)
        for (int ch = 0; ch < nchunk2; ++ch) {
            if (ch >= 1) {
                phase_c(ch - 1);
                TSTAMP(1)
                issue_v(ch);
                TSTAMP(2)
            }
            if (ch + 1 < nchunk2) {
                phase_a(ch + 1);
                TSTAMP(3)
                if (ch + 2 < nchunk2) issue_k(ch + 2);
                TSTAMP(4)
            }
            __syncthreads();                                                // barrier #(ch+1)
            TSTAMP(5)
        }
        phase_c(nchunk2 - 1);
        TSTAMP(6)
        __syncthreads();                                                    // F1
        float* pts = &sm.sp[0][0][0];                                       // [BI][H][24]; both sp buffers are free now
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) {
            const int h = w4 * 3 + hh;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = kq * 4 + r, i = i0 + il;
                const float inv = mi_a[r] ? 1.f / sm.lsum[il][h] : 0.f;
                if (i < L)
                    reinterpret_cast<float2*>(feat + (rowbase + i) * FEAT + H * C + h * D)[fm] = make_float2(accV[hh][0][r] * inv, accV[hh][1][r] * inv);
                if (fm < 12) *reinterpret_cast<float2*>(&pts[(il * H + h) * (P * 3) + 2 * fm]) = make_float2(accT[hh][0][r] * inv, accT[hh][1][r] * inv);
            }
        }
        __syncthreads();                                                    // F2
    }

    // ---- all waves: local frame, norm, direction of the aggregated points (ga.py:136-139)
    const float* pts = &sm.sp[0][0][0];
    for (int e = tid; e < BI * H * P; e += NT) {
        const int il = e / (H * P), hp = e % (H * P), i = i0 + il;
        if (i >= L) continue;
        const float* Rr = R + (rowbase + i) * 9;
        const float* tr = t + (rowbase + i) * 3;
        const float* a = pts + (il * H * P + hp) * 3;
        const float dx = a[0] - tr[0], dy = a[1] - tr[1], dz = a[2] - tr[2];
        const float lx = Rr[0] * dx + Rr[3] * dy + Rr[6] * dz;
        const float ly = Rr[1] * dx + Rr[4] * dy + Rr[7] * dz;
        const float lz = Rr[2] * dx + Rr[5] * dy + Rr[8] * dz;
        const float dist = sqrtf(lx * lx + ly * ly + lz * lz);
        const float inv = 1.f / (dist + 1e-4f);
        float* fpnt = feat + (rowbase + i) * FEAT + H * C + H * D;
        fpnt[hp * 3 + 0] = lx; fpnt[hp * 3 + 1] = ly; fpnt[hp * 3 + 2] = lz;
        fpnt[H * P * 3 + hp] = dist;
        float* fdir = fpnt + H * P * 3 + H * P;
        fdir[hp * 3 + 0] = lx * inv; fdir[hp * 3 + 1] = ly * inv; fdir[hp * 3 + 2] = lz * inv;
    }
#ifdef WS_TIMING
    TSTAMP(7)
    if (blockIdx.x == 17 && lane == 0 && (wave == 0 || wave == NPW))
        for (int k = 0; k < 8; ++k) g_ws_timing[wave == 0 ? 0 : 1][k] = tacc[k];
#endif
}

// Pair-bias cache: lp[l][n,i,j,h] = z[n,i,j,:] . Wb_l[h,:] for every layer l in ONE pass over z (ga.py:88-90).  z and the
// weights do not change during the 100 steps of FullDPM.sample, so the sampler builds this once per call and the per-step
// kernel reads 48 useful bytes per (i,j) instead of spending 64 of its 218 MFMAs per chunk (and an LDS transpose) on it.
// Layout per layer: [N*L (query row)][nchunk][16 (head, 12 used)][16 (key in chunk)] -- exactly the per-lane float4 the pair
// waves consume, 1 KB contiguous per (row, chunk).  Same MFMA chain order as the fused path => bit-identical logits.
struct WbList { const float* w[8]; };

__global__ __launch_bounds__(256) void pair_bias_cache_kernel(const float* __restrict__ z, WbList wl, int num_layers, float* __restrict__ cache,
                                                              int64_t rows, int L, int nchunk) {
    __shared__ __attribute__((aligned(16))) float zst[4][JC][ZSLD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fm = lane & 15, kq = lane >> 4;
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;                  // (query row, chunk)
    if (unit >= rows * nchunk) return;
    const int64_t row = unit / nchunk;
    const int ch = (int)(unit % nchunk);
    const float* zi = z + (row * (int64_t)L) * C;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x4*>(&zst[wave][kq * 4 + r][fm * 4]) = *(reinterpret_cast<const f32x4*>(zi + (int64_t)min(ch * JC + kq * 4 + r, L - 1) * C) + fm);
    wave_lds_sync();                                                      // cross-lane transpose through LDS
    float4 za[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) za[q] = *reinterpret_cast<const float4*>(&zst[wave][fm][kq * 16 + q * 4]);
    for (int l = 0; l < num_layers; ++l) {
        f32x4 acc4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fm < H) wv = reinterpret_cast<const float4*>(wl.w[l] + fm * C + kq * 16)[q];
            acc4[q] = mfma4(za[q].x, wv.x, (f32x4){0.f, 0.f, 0.f, 0.f});
            acc4[q] = mfma4(za[q].y, wv.y, acc4[q]); acc4[q] = mfma4(za[q].z, wv.z, acc4[q]); acc4[q] = mfma4(za[q].w, wv.w, acc4[q]);
        }
        const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        *(reinterpret_cast<f32x4*>(cache + ((int64_t)l * rows * nchunk + unit) * 256) + lane) = acc;
    }
}

int launch_pair_bias_cache(const float* z, const float* const* wb, int num_layers, float* cache, int N, int L, hipStream_t st) {
    ABOPT_CHECK_ARG(num_layers >= 1 && num_layers <= 8, "pair_bias_cache: 1..8 layers supported (got %d)", num_layers);
    WbList wl;
    for (int l = 0; l < 8; ++l) wl.w[l] = l < num_layers ? wb[l] : nullptr;
    const int nchunk = (L + JC - 1) / JC;
    const int64_t units = (int64_t)N * L * nchunk;
    if (units == 0) return ABOPT_OK;
    hipLaunchKernelGGL(pair_bias_cache_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, z, wl, num_layers, cache, (int64_t)N * L, L, nchunk);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

int launch_ipa_core_ws(const float* proj, const float* z, const uint8_t* mask, const float* R, const float* t,
                       const float* w_pair_bias, const float* spatial_coef, float* feat, float* dbg_logits,
                       const float* pair_bias_cache, const float* kvfrag, int N, int L, hipStream_t st, int z_shared) {
    const int nib = (L + BI - 1) / BI;
    const int remap = (N % 8 == 0) ? 1 : 0;
    prof::begin(st);
#define WS_LAUNCH(DBGV, NPWV, CV) hipLaunchKernelGGL((ipa_core_ws_kernel<DBGV, NPWV, CV>), dim3((unsigned)(N * nib)), dim3((NPWV + 4) * 64), 0, st, proj, z, \
                                                     mask, R, t, w_pair_bias, spatial_coef, feat, dbg_logits, pair_bias_cache, kvfrag, N, L, nib, remap, z_shared)
    if (pair_bias_cache) { if (dbg_logits) WS_LAUNCH(true, 4, true); else WS_LAUNCH(false, 4, true); }
    else if (dbg_logits) WS_LAUNCH(true, 4, false);
    else                 WS_LAUNCH(false, 4, false);
    prof::end(st);
    ABOPT_LAUNCH_CHECK();
#ifdef WS_TIMING
    {
        long long h[2][8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ws_timing), sizeof(h));
        static int calls = 0;
        if (++calls == 8) {
            fprintf(stderr, "[ws timing, cycles of WG 17] pair: pre %lld | wait-z(+stage) %lld | pairbias %lld | softmax %lld | agg+Pwrite %lld | barrier %lld | final %lld | epilogue %lld\n",
                    h[0][0], h[0][1], h[0][2], h[0][3], h[0][4], h[0][5], h[0][6], h[0][7]);
            fprintf(stderr, "[ws timing, cycles of WG 17] node: pre %lld | phase_c %lld | issue_v %lld | phase_a %lld | issue_k %lld | barrier %lld | last_c %lld | epilogue %lld\n",
                    h[1][0], h[1][1], h[1][2], h[1][3], h[1][4], h[1][5], h[1][6], h[1][7]);
        }
    }
#endif
    return ABOPT_OK;
}

}  // namespace abopt

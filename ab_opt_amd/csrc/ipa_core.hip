// IPA core for CDNA4 (gfx950): invariant point attention between the node projections and out_transform, fused in ONE pass
// over the pair features z.  Reference semantics: AbDock/src/modules/encoders/ga.py
//   _node_logits :81-86   _pair_logits :88-90   _spatial_logits :92-112   _alpha_from_logits :11-26
//   _pair_aggregation :114-118   _node_aggregation :120-125   _spatial_aggregation :127-147
// The reference materialises (N,L,L,12,{24,32,64}) temporaries; here z[n,i,:,:] is read from HBM exactly once per (n,i) and
// everything else stays on chip.  All contractions are v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).
//
// One 1024-thread workgroup = 16 query residues of one sample; keys are consumed 16 at a time with an online softmax (any L).
// The 16 waves (4 per SIMD, <= 128 VGPRs each) have three roles, pipelined one key chunk apart through a triple-buffered
// S/P tile in LDS, one barrier per chunk:
//
//   8 "pair" waves  (2 query rows each)   chunk t  : stream z[i, chunk] (+ the cached pair bias row) through a 3-deep register ring, logits = (S + pair bias)
//                                                    sqrt(1/3), mask, online softmax, pair aggregation  fp[c,h] += z[j,c] P[j,h]
//                                                    (M = channel, N = head, K = key); P back to LDS in place of S
//   4 "A" waves     (3 heads each)        chunk t+1: S^T[j,i] = k'_j . q'_i  -- ONE 15-step MFMA chain per head: the operands are
//                                                    augmented so that q.k/sqrt(D), the point cross term and both squared norms
//                                                    of  -gamma sqrt(2/(9P))/2 |q_pts - k_pts|^2  are K-slices of the same product
//                                                    (no epilogue arithmetic); writes S as 16-byte rows
//   4 "C" waves     (3 heads each)        chunk t-1: fn[d,i] += v[j,d] P[i,j], pts[e,i] += v_pts[j,e] P[i,j]  (M = channel, N = query)
//
// Each SIMD hosts 2 pair waves + 1 A + 1 C wave: four independent instruction streams share its matrix pipe (157 MFMAs per
// chunk per SIMD), so the pipe stays fed while individual waves wait for HBM (pair waves) or L2 (A / C waves).
// Operands arrive pre-arranged in MFMA fragment order (ipa.hip: ipa_frags_kernel): every global load of the A / C waves and
// every LDS access of the S/P tile is a conflict-free, fully coalesced 16 bytes per lane.
#include <cstdlib>
#include "tail_common.h"
#include "kernels.h"

// Developer ablations (never in the shipped build): make CXXEXTRA=-DCORE_ABL=<bits>.  bit 0: pair waves load z only once;
// bit 1: A/C waves load their fragments only once; bit 2: pair waves skip the aggregation MFMAs; bit 3: A waves skip their MFMAs;
// bit 4: C waves skip their MFMAs; bit 5: pair waves skip the softmax arithmetic.  Results are wrong by construction.
#ifndef CORE_ABL
#define CORE_ABL 0
#endif

#ifdef CORE_TIMING   // developer build (make CXXEXTRA=-DCORE_TIMING): s_memtime section timers of one workgroup, printed by the launcher
#include <cstdio>
#define TSTAMP(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
#define TSYNC(kw, kb) { TSTAMP(kw) __syncthreads(); TSTAMP(kb) }
__device__ long long g_core_timing[3][8];
#else
#define TSTAMP(k)
#define TSYNC(kw, kb) __syncthreads();
#if defined(PERSIST_TIMING) || defined(C32_TIMING) || defined(C32_COUNT)   // developer builds: per-role clocks of one workgroup
#include <cstdio>
__device__ long long g_core_timing[3][8];
#endif
#endif

#ifndef PBC_CHUNK_MAJOR
#define PBC_CHUNK_MAJOR 1 // layout of a layer's slab of the pair-bias cache: 1 [chunk][row of the batch][12 x 16] (the same argument as ZT_CHUNK_MAJOR below) | 0 [row][chunk][12 x 16]
#endif
namespace abopt {

constexpr int NPW = 8, RPW = BI / NPW;          // pair waves, query rows per pair wave
constexpr int NTH = 1024;
constexpr int SROW = 16 * JC + 4;               // one query row of the S/P tile: [16 head slots][16 keys] + 4 (rows 4 banks apart)
constexpr int SCLD = 17;                        // row stride of the per-(row, head) scalars
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kSqrt13 = 0.5773502691896258f;   // sqrt(1/3), ga.py:165
constexpr float kScale2 = kSqrt13 * kLog2e;            // logits are kept in base-2 units: p = exp2(l2 - m2)
constexpr float kMask2 = 1e5f * kLog2e;           // the reference's additive -1e5 on masked pairs (ga.py:20-23), same units
constexpr int SPLIT_ROW = H * C + H * D + H * P * 3;   // 1440 unnormalised accumulators per row and key slice

// position of key group kq (keys 4 kq .. 4 kq + 3) inside head h's 16-key row of the S/P tile.  The rotation by h >> 1 makes the
// pair waves' 16-byte reads AND writes (lane = (head, key group)) bank-conflict free; see DESIGN.md section 3.1.
__device__ __forceinline__ int sp_off(int h, int kq) { return h * JC + 4 * ((kq + (h >> 1)) & 3); }

// Aggregated point back to the residue frame: l = R^T (a - t) (geometry.py:94-117), its norm and 1 / (norm + 1e-4) (ga.py:138-139), with the
// operation order written out -- every kernel that produces these feature columns (one-block, persistent, 32-row, key-split merge, fused
// core + tail) calls this, so their results agree bit for bit whatever the compiler would contract elsewhere.
__device__ __forceinline__ void point_local(float dx, float dy, float dz, float r0, float r1, float r2, float r3, float r4, float r5, float r6, float r7,
                                            float r8, float& lx, float& ly, float& lz, float& d, float& inv) {
    lx = __builtin_fmaf(r6, dz, __builtin_fmaf(r3, dy, r0 * dx));
    ly = __builtin_fmaf(r7, dz, __builtin_fmaf(r4, dy, r1 * dx));
    lz = __builtin_fmaf(r8, dz, __builtin_fmaf(r5, dy, r2 * dx));
    d = sqrtf(__builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx)));
    inv = 1.f / (d + 1e-4f);
}

struct CoreLds {
    float* sp;      // [3][BI][SROW]
    f32x4* qf;      // [H][4][64]        query-side operands of the block, fragment order
    float* scl;     // [2][BI][SCLD]     rescale factor of the chunk (by chunk parity)
    float* lsum;    // [BI][SCLD]        softmax denominators
    float* zst;     // [NPW][JC][ZSLD]   in-kernel pair bias only: per-wave transpose tile
    float* wbs;     // [16][C + 4]       in-kernel pair bias only: proj_pair_bias weights, rows 12..15 zero
    uint8_t* mk;    // [nchunk * JC]     key mask of the sample, 0 past the end
};
template <bool CACHED>
__host__ __device__ constexpr size_t core_lds_fixed_bytes() {
    return sizeof(float) * (3 * BI * SROW + H * 4 * 64 * 4 + 2 * BI * SCLD + BI * SCLD + (CACHED ? 0 : NPW * JC * ZSLD + 16 * (C + 4)));
}

// DUMP (training / parity tests): the pair waves also write the scaled, UNMASKED logits x = logit * sqrt(1/3) * log2(e) head-major
// (dump [N,12,L,L], one 16-byte store per lane and row chunk) and the final running maximum / sum of every (row, head)
// (dump_stats [N*L,12,2]); alpha = mask ? exp2(x - m) / l : 0 is one elementwise pass later (ipa_train.hip: alpha_finalize).
// SPLIT (small batches: fewer query blocks than half the CUs): blockIdx.x = block * nsplit + slice, a workgroup walks only the key chunks
// [slice * nchunk / nsplit, (slice + 1) * nchunk / nsplit) and leaves its UNNORMALISED accumulators ([rows][1440]: 768 pair | 384 node |
// 288 aggregated points) and the running maximum / sum of every (row, head) in `part` / `pstats`; ipa_split_merge_kernel combines the
// slices (softmax merge) and applies the point epilogue.  A step's core is a 16-chunk serial loop (63 us) however few samples there are;
// with 2..4 slices the loop is 8..4 chunks on twice / four times as many CUs.
template <bool DUMP, bool CACHED, bool SPLIT = false>
__global__ __launch_bounds__(NTH) void ipa_core_kernel(const float* __restrict__ qfrag, const float* __restrict__ kvfrag, const float* __restrict__ z,
                                                       const uint8_t* __restrict__ mask, const float* __restrict__ R, const float* __restrict__ t,
                                                       const float* __restrict__ Wb, float* __restrict__ feat, float* __restrict__ dump,
                                                       float* __restrict__ dump_stats,
                                                       const float* __restrict__ pbc, int N, int L, int nib, int xcd_remap, int z_shared,
                                                       float* __restrict__ part = nullptr, float* __restrict__ pstats = nullptr, int nsplit = 1) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    CoreLds sm;
    {
        float* p = reinterpret_cast<float*>(smem_raw);
        sm.sp = p; p += 3 * BI * SROW;
        sm.qf = reinterpret_cast<f32x4*>(p); p += H * 4 * 64 * 4;
        sm.scl = p; p += 2 * BI * SCLD;
        sm.lsum = p; p += BI * SCLD;
        sm.zst = p; if (!CACHED) p += NPW * JC * ZSLD;
        sm.wbs = p; if (!CACHED) p += 16 * (C + 4);
        sm.mk = reinterpret_cast<uint8_t*>(p);
    }
    int n, ib;
    {   // all i-blocks of a sample on one XCD when N % 8 == 0 (blocks are dealt round-robin to the 8 XCDs): its key/value fragments stay
        // in that XCD's L2.  Speed only, never correctness.
        const int b = SPLIT ? blockIdx.x / nsplit : blockIdx.x;
        if (xcd_remap) { const int xcd = b & 7, k = b >> 3; n = xcd + 8 * (k / nib); ib = k % nib; }
        else { n = b / nib; ib = b % nib; }
    }
    const int tid = threadIdx.x, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = (L + JC - 1) / JC;
    const int i0 = ib * BI;
    const int64_t rowbase = (int64_t)n * L;
    const int64_t zbase = z_shared ? (int64_t)(n / z_shared) * L : rowbase;   // z_shared = g > 0: consecutive groups of g samples share one pair_feat (and bias cache) entry
    // Key chunks are visited in natural order by every query block of a sample: the 16 blocks then read the same 96 KB of key/value
    // fragments at about the same time and all but the first hit in the XCD's L2.  (A per-block rotated order was measured: L2 hit
    // rate of the fragments fell from ~90 % to ~25 %, FETCH_SIZE 0.75 -> 1.16 GB per launch, kernel 185 -> 199 us.)
    const int slice = SPLIT ? blockIdx.x % nsplit : 0;
    const int c_lo = SPLIT ? slice * nchunk / nsplit : 0, c_hi = SPLIT ? (slice + 1) * nchunk / nsplit : nchunk;
    const int ncl = c_hi - c_lo;                                        // key chunks of this workgroup (all of them unless SPLIT)
    const int c0 = SPLIT ? c_lo : ((CORE_ABL & 512) ? ib % nchunk : 0);
    auto chunk_of = [&](int it) { const int c = it + c0; return c < nchunk ? c : c - nchunk; };      // iteration -> key chunk
#ifdef CORE_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();   // 0 prologue | 1 work | 2 barrier wait | 3 epilogue own | 4 F1/F2 waits | 5 common epilogue
#endif

    // ---------------------------------------------------------------- prologue (every role first puts its own global loads in flight)
    auto fill_lds = [&]() {
        const f32x4* qg = reinterpret_cast<const f32x4*>(qfrag) + ((int64_t)n * nib + ib) * (H * 4 * 64);
#pragma unroll
        for (int e = 0; e < H * 4 * 64 / NTH; ++e) sm.qf[e * NTH + tid] = qg[e * NTH + tid];
        // (head slots 12..15 of the S/P tile are never written by the A waves: whatever they hold stays in MFMA columns / lanes
        //  12..15 of the pair waves, which are never stored)
        for (int e = tid; e < nchunk * JC; e += NTH) sm.mk[e] = (e < L) ? mask[rowbase + e] : 0;
        if (!CACHED) {
            for (int e = tid; e < 16 * (C / 4); e += NTH) {
                const int h = e / (C / 4), c4 = e % (C / 4);
                f32x4 w4v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (h < H) w4v = reinterpret_cast<const f32x4*>(Wb + h * C)[c4];
                *reinterpret_cast<f32x4*>(&sm.wbs[h * (C + 4) + c4 * 4]) = w4v;
            }
        }
    };

    if (wave < NPW) {
        // =========================================================================================== pair waves
        const int il0 = wave * RPW;
        const char* zrow[RPW];
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii)
            zrow[ii] = reinterpret_cast<const char*>(z + ((zbase + min(i0 + il0 + ii, L - 1)) * (int64_t)L) * C);
        const unsigned lane_b = (unsigned)fm * 16u;
        f32x4 ring[3][4];                                                // z ring: one row chunk per slot (4 x 16 B per lane), requests run 2 positions ahead
        f32x4 ringb[3];                                                  // CACHED: the row chunk's pair bias, lane = (head fm, keys 4 kq ..): 768 contiguous bytes per wave load
        const char* pbrow[RPW];                                          // wave-uniform row bases (SGPRs) + one per-lane byte offset
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii)
            pbrow[ii] = CACHED ? reinterpret_cast<const char*>(pbc + ((zbase + min(i0 + il0 + ii, L - 1)) * (int64_t)(PBC_CHUNK_MAJOR ? 1 : nchunk)) * (H * JC)) : nullptr;
        const unsigned pb_chunk = PBC_CHUNK_MAJOR ? (unsigned)((N / (z_shared ? z_shared : 1)) * L) * (unsigned)(H * JC * 4) : (unsigned)(H * JC * 4);      // bytes from a row's chunk to its next one
        const unsigned pb_lane = (unsigned)(min(fm, H - 1) * JC + kq * 4) * 4u;
        // the dump goes out through a buffer descriptor: base = this sample's [12,L,L] slab (SGPRs), one lane-constant byte offset
        // (head, key group) and a wave-uniform row/chunk offset -- no 64-bit per-lane addresses in the hot loop
        // (the descriptor covers exactly this sample's slab, so an offset past it is discarded by the hardware)
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(DUMP ? dump + (int64_t)n * H * L * L : nullptr, 0, H * L * L * 4, 0x00020000);
        const unsigned dvoff = fm < H ? (unsigned)((fm * L) * L + kq * 4) * 4u : 0x7ffffff0u;
        const bool dvec = (L & 3) == 0;                                  // rows of the dump are 16-byte aligned
        const bool dfull = (L & 15) == 0;                                // no partial key chunk: every lane of heads 0..11 stores
#define PW_ISSUE(SLOT, II, CH)                                                                                          \
    {                                                                                                                    \
        const int ch_ = chunk_of(min((CH), ncl - 1));                          /* past the end: harmless re-read */      \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                                \
            ring[SLOT][r_] = ZLOAD(reinterpret_cast<const f32x4*>(zrow[II] + ((unsigned)min(ch_ * JC + kq * 4 + r_, L - 1) * (unsigned)(C * 4) + lane_b))); \
        if (CACHED && !(CORE_ABL & 64)) ringb[SLOT] = ZLOAD(reinterpret_cast<const f32x4*>(pbrow[II] + ((unsigned)ch_ * pb_chunk + pb_lane))); \
    }
// z and its bias cache are read once per launch: non-temporal loads (measured 182 -> 174 us per launch at N=32, L=256)
#ifdef CORE_NO_NT
#define ZLOAD(p) (*(p))
#else
#define ZLOAD(p) __builtin_nontemporal_load(p)
#endif
        PW_ISSUE(0, 0, 0) PW_ISSUE(1, 1, 0)
        fill_lds();
        float m_run[RPW], l_run[RPW];
        f32x4 accP[RPW][4];
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii) {
            m_run[ii] = -INFINITY; l_run[ii] = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) accP[ii][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int spo = sp_off(fm, kq);
        __syncthreads();                                                    // LDS tile visible
        bool mi_b[RPW];
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii) mi_b[ii] = (i0 + il0 + ii < L) && sm.mk[min(i0 + il0 + ii, L - 1)] != 0;
        __syncthreads();                                                    // barrier #0: S(0) ready
        TSTAMP(0)

        // one (query row, chunk) position: ring slot SLOT holds its z; BUF = chunk % 3
#define PW_POS(SLOT, II, CH, BUF)                                                                                        \
    {                                                                                                                    \
        if (!(CORE_ABL & 1)) PW_ISSUE(((SLOT) + 2) % 3, II, (CH) + 1)       /* two positions ahead = same row, next chunk */ \
        const int il_ = il0 + (II);                                                                                      \
        float* spp_ = sm.sp + ((BUF) * BI + il_) * SROW + spo;                                                           \
        f32x4 sv_ = *reinterpret_cast<const f32x4*>(spp_);                                                               \
        if (CACHED) sv_ += ringb[SLOT];                                                                                  \
        if (!CACHED) {                                                      /* pair bias in place: z chunk transposed through a wave-private LDS tile */ \
            float* zt_ = sm.zst + wave * (JC * ZSLD);                                                                    \
            wave_lds_sync();                                                                                             \
            _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) *reinterpret_cast<f32x4*>(&zt_[(kq * 4 + r_) * ZSLD + fm * 4]) = ring[SLOT][r_]; \
            wave_lds_sync();                                                                                             \
            f32x4 a4_[4];                                                                                                \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                           \
                const f32x4 za_ = *reinterpret_cast<const f32x4*>(&zt_[fm * ZSLD + kq * 16 + q_ * 4]);                   \
                const f32x4 wv_ = *reinterpret_cast<const f32x4*>(&sm.wbs[fm * (C + 4) + kq * 16 + q_ * 4]);            \
                a4_[q_] = mfma4(za_[0], wv_[0], (f32x4){0.f, 0.f, 0.f, 0.f});                                            \
                a4_[q_] = mfma4(za_[1], wv_[1], a4_[q_]); a4_[q_] = mfma4(za_[2], wv_[2], a4_[q_]); a4_[q_] = mfma4(za_[3], wv_[3], a4_[q_]); \
            }                                                                                                            \
            sv_ += (a4_[0] + a4_[1]) + (a4_[2] + a4_[3]);                                                                \
        }                                                                                                                \
        float l2_[4];                                                                                                    \
        sv_ *= kScale2;                                                                                                  \
        if (DUMP && (i0 + il_) < L) {                                           /* wave-uniform */                        \
            const int so_ = ((i0 + il_) * L + chunk_of(CH) * JC) * 4;                                                    \
            if (dfull) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv_), drsrc, dvoff, so_, 0);     /* lanes of heads 12..15 carry an out-of-range offset: dropped by the buffer bounds check */ \
            else if (fm < H) {                                                                                           \
                const int j0_ = chunk_of(CH) * JC + kq * 4;                                                              \
                if (dvec) { if (j0_ < L) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv_), drsrc, dvoff, so_, 0); } \
                else { _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) if (j0_ + r_ < L) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv_[r_]), drsrc, dvoff, so_ + 4 * r_, 0); } \
            }                                                                                                            \
        }                                                                                                                \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                               \
            const float x_ = sv_[r_];                                                                                    \
            l2_[r_] = ((mk4_ >> (8 * r_)) & 0xffu) ? x_ : x_ - kMask2;      /* ga.py:20-23 (masked QUERY rows are zeroed at the end instead) */ \
        }                                                                                                                \
        f32x4 pv_; float sc_;                                                                                            \
        if (CORE_ABL & 32) { sc_ = 1.f; pv_ = (f32x4){l2_[0], l2_[1], l2_[2], l2_[3]}; l_run[II] += l2_[0]; }           \
        else {                                                                                                           \
        const float mx_ = rows_max(fmaxf(fmaxf(l2_[0], l2_[1]), fmaxf(l2_[2], l2_[3])));                                 \
        const float mn_ = fmaxf(m_run[II], mx_);                                                                         \
        sc_ = __builtin_amdgcn_exp2f(m_run[II] - mn_);                                                                   \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) pv_[r_] = __builtin_amdgcn_exp2f(l2_[r_] - mn_);                \
        const float ps_ = rows_sum((pv_[0] + pv_[1]) + (pv_[2] + pv_[3]));                                               \
        l_run[II] = l_run[II] * sc_ + ps_;                                                                               \
        m_run[II] = mn_;                                                                                                 \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) accP[II][mt_] *= sc_;                                        \
        }                                                                                                                \
        if (CORE_ABL & 4) { _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) accP[II][r_] += ring[SLOT][r_] * pv_[r_]; } \
        else                                                                                                             \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                                 \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) accP[II][mt_] = mfma4(ring[SLOT][r_][mt_], pv_[r_], accP[II][mt_]); \
        *reinterpret_cast<f32x4*>(spp_) = pv_;                                                                           \
        if (kq == 0) sm.scl[(((CH) & 1) * BI + il_) * SCLD + fm] = sc_;                                                  \
    }
        // one chunk = positions (row 0, row 1); K = chunk index within the 3-chunk revolution of the ring
#define PW_CHUNK(K, CH)                                                                                                  \
    {                                                                                                                    \
        const uint32_t mk4_ = *reinterpret_cast<const uint32_t*>(&sm.mk[chunk_of(CH) * JC + kq * 4]);                    \
        PW_POS((2 * (K)) % 3, 0, CH, K)                                                                                  \
        PW_POS((2 * (K) + 1) % 3, 1, CH, K)                                                                              \
        TSYNC(1, 2)                                                         /* barrier #(CH + 1) */                       \
    }
        int ch = 0;
        for (; ch + 3 <= ncl; ch += 3) { PW_CHUNK(0, ch) PW_CHUNK(1, ch + 1) PW_CHUNK(2, ch + 2) }
        if (ch < ncl) {
            PW_CHUNK(0, ch)
            if (ch + 1 < ncl) PW_CHUNK(1, ch + 1)
        }
        // alpha = P / l, zero for masked queries (ga.py:24-25); pair features out
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii) {
            const int il = il0 + ii, i = i0 + il;
            if (kq == 0) sm.lsum[il * SCLD + fm] = l_run[ii];
            if (DUMP && kq == 0 && i < L && fm < H) *reinterpret_cast<float2*>(dump_stats + ((rowbase + i) * H + fm) * 2) = make_float2(m_run[ii], l_run[ii]);
            if (SPLIT && kq == 0 && i < L && fm < H)
                *reinterpret_cast<float2*>(pstats + (((int64_t)slice * N * L + rowbase + i) * H + fm) * 2) = make_float2(m_run[ii], l_run[ii]);
            if (i < L && fm < H) {
                const float inv = SPLIT ? 1.f : (mi_b[ii] ? 1.f / l_run[ii] : 0.f);
                float* fo = SPLIT ? part + ((int64_t)slice * N * L + rowbase + i) * SPLIT_ROW + fm * C + kq * 16
                                  : feat + (rowbase + i) * FEAT + fm * C + kq * 16;        // accumulator row 4 kq + r of tile mt <-> channel 16 kq + 4 r + mt
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    reinterpret_cast<f32x4*>(fo)[r] = (f32x4){accP[ii][0][r] * inv, accP[ii][1][r] * inv, accP[ii][2][r] * inv, accP[ii][3][r] * inv};
            }
        }
        TSYNC(3, 4)                                                         // F1: lsum visible, C waves done with the last chunk
        TSYNC(3, 4)                                                         // F2: aggregated points in LDS
    } else if (wave < NPW + 4) {
        // =========================================================================================== A waves: S(t + 1)
        const int h0 = (wave - NPW) * 3;
        const f32x4* kvn = reinterpret_cast<const f32x4*>(kvfrag) + (int64_t)n * nchunk * H * 512;
        f32x4 kf[3][4];
        // per head: operands of the NEXT chunk are requested as soon as this chunk's MFMAs have consumed the registers, so every
        // load has a whole chunk period to arrive (no conditional loads: the compiler keeps exact vmcnt counts)
#define AW_ISSUE(HH, CH)                                                                                                 \
    {                                                                                                                    \
        const int c_ = chunk_of(min((CH), ncl - 1));                                                                     \
        const f32x4* fr_ = kvn + ((int64_t)c_ * H + h0 + (HH)) * 512 + lane;                                             \
        if (!(CORE_ABL & 128)) { _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) kf[HH][s_] = fr_[s_ * 64]; }           \
    }
#define AW_ISSUE0(HH) { const f32x4* fr_ = kvn + ((int64_t)chunk_of(0) * H + h0 + (HH)) * 512 + lane;                   \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) kf[HH][s_] = fr_[s_ * 64]; }
        auto produce = [&](int c, int buf) {                                // S(c) -> sp[buf], then request chunk c + 1
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                const int h = h0 + hh;
                const f32x4* qh = sm.qf + (h * 4) * 64 + lane;
                const f32x4 q0 = qh[0], q1 = qh[64], q2 = qh[128], q3 = qh[192];
                // rows = keys (A operand k'), columns = queries (B operand q'); two chains hide the 40-cycle dependent-MFMA latency
                f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                if (CORE_ABL & 8) { acc0 = kf[hh][0] * q0 + kf[hh][1] * q1; acc1 = kf[hh][2] * q2 + kf[hh][3] * q3; }
                else {
#pragma unroll
                for (int s = 0; s < 4; ++s) { acc0 = mfma4(kf[hh][0][s], q0[s], acc0); acc1 = mfma4(kf[hh][2][s], q2[s], acc1); }
#pragma unroll
                for (int s = 0; s < 4; ++s) { acc0 = mfma4(kf[hh][1][s], q1[s], acc0); if (s < 3) acc1 = mfma4(kf[hh][3][s], q3[s], acc1); }
                }
                const f32x4 sres = acc0 + acc1;
                if (!(CORE_ABL & 2) && (!(CORE_ABL & 1024) || (c & 1))) { if (hh == 0) AW_ISSUE(0, c + 1) else if (hh == 1) AW_ISSUE(1, c + 1) else AW_ISSUE(2, c + 1) }
                *reinterpret_cast<f32x4*>(sm.sp + (buf * BI + fm) * SROW + sp_off(h, kq)) = sres;      // accumulator row 4 kq + r = key, column fm = query
                __builtin_amdgcn_sched_barrier(0);                          // keep the per-head order: hipcc otherwise sinks all reloads to the end of the iteration
            }
        };
        AW_ISSUE0(0) AW_ISSUE0(1) AW_ISSUE0(2)
        fill_lds();
        __syncthreads();
        produce(0, 0);
        __syncthreads();                                                    // barrier #0
        TSTAMP(0)
        int buf = 1;
        for (int c = 1; c < ncl; ++c) {
            produce(c, buf);
            buf = (buf == 2) ? 0 : buf + 1;
            TSYNC(1, 2)                                                     // barrier #c
        }
        TSYNC(1, 2)                                                         // barrier #nchunk
        TSYNC(3, 4)                                                         // F1
        TSYNC(3, 4)                                                         // F2
    } else {
        // =========================================================================================== C waves: node / point aggregation of chunk t - 1
        const int h0 = (wave - NPW - 4) * 3;
        const f32x4* kvn = reinterpret_cast<const f32x4*>(kvfrag) + (int64_t)n * nchunk * H * 512;
        f32x4 vf[3][4];
        f32x4 accV[3][2], accT[3][2];
#pragma unroll
        for (int hh = 0; hh < 3; ++hh)
#pragma unroll
            for (int k = 0; k < 2; ++k) { accV[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; accT[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#define CW_ISSUE(HH, CH)                                                                                                 \
    {                                                                                                                    \
        const f32x4* fr_ = kvn + ((int64_t)chunk_of(min((CH), ncl - 1)) * H + h0 + (HH)) * 512 + 4 * 64 + lane;          \
        if (!(CORE_ABL & 256)) { _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) vf[HH][s_] = fr_[s_ * 64]; }           \
    }
#define CW_ISSUE0(HH) { const f32x4* fr_ = kvn + ((int64_t)chunk_of(0) * H + h0 + (HH)) * 512 + 4 * 64 + lane;          \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) vf[HH][s_] = fr_[s_ * 64]; }
        auto consume = [&](int c, int buf) {                                // P(c) in sp[buf]; then request chunk c + 1 (same per-head pipeline as the A waves)
            const int par = c & 1;
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                const int h = h0 + hh;
                const float sc = sm.scl[(par * BI + fm) * SCLD + h];                                   // rescale of (query fm, head h)
                const f32x4 pa = *reinterpret_cast<const f32x4*>(sm.sp + (buf * BI + fm) * SROW + sp_off(h, kq));   // B: column = query fm, step s <-> key 4 kq + s
                accV[hh][0] *= sc; accV[hh][1] *= sc; accT[hh][0] *= sc; accT[hh][1] *= sc;
                if (CORE_ABL & 16) { accV[hh][0] += vf[hh][0] * pa; accV[hh][1] += vf[hh][1] * pa; accT[hh][0] += vf[hh][2] * pa; accT[hh][1] += vf[hh][3] * pa; }
                else
#pragma unroll
                for (int s = 0; s < 4; ++s) {                                                          // A: row fm <-> value channels 2 fm (+1) / point coordinates 2 fm (+1)
                    accV[hh][0] = mfma4(vf[hh][s][0], pa[s], accV[hh][0]);
                    accV[hh][1] = mfma4(vf[hh][s][1], pa[s], accV[hh][1]);
                    accT[hh][0] = mfma4(vf[hh][s][2], pa[s], accT[hh][0]);
                    accT[hh][1] = mfma4(vf[hh][s][3], pa[s], accT[hh][1]);
                }
                if (!(CORE_ABL & 2) && (!(CORE_ABL & 1024) || (c & 1))) { if (hh == 0) CW_ISSUE(0, c + 1) else if (hh == 1) CW_ISSUE(1, c + 1) else CW_ISSUE(2, c + 1) }
                __builtin_amdgcn_sched_barrier(0);                          // keep the per-head order (see the A waves)
            }
        };
        CW_ISSUE0(0) CW_ISSUE0(1) CW_ISSUE0(2)
        fill_lds();
        __syncthreads();
        const bool mi_c = (i0 + fm < L) && sm.mk[min(i0 + fm, L - 1)] != 0;
        __syncthreads();                                                    // barrier #0
        TSTAMP(0)
        TSYNC(1, 2)                                                         // barrier #1: P(0) ready
        int buf = 0;                                                        // buffer of the chunk being consumed
        for (int c = 1; c < ncl; ++c) {
            consume(c - 1, buf);
            buf = (buf == 2) ? 0 : buf + 1;
            TSYNC(1, 2)                                                     // barrier #(c + 1)
        }
        consume(ncl - 1, buf);
        TSYNC(3, 4)                                                         // F1
        float* pts = sm.sp;                                                 // [BI][H][24] aggregated global-frame points; the S/P tile is free now
        const int i = i0 + fm;
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) {
            const int h = h0 + hh;
            const float inv = SPLIT ? 1.f : (mi_c ? 1.f / sm.lsum[fm * SCLD + h] : 0.f);
            // accumulator row 4 kq + r of tile k <-> value channel 16 k + 4 kq + r   /   coordinate r of point 4 k + kq (r = 3: padding)
            float* prow = SPLIT ? part + ((int64_t)slice * N * L + rowbase + min(i, L - 1)) * SPLIT_ROW : nullptr;
            if (i < L) {
                float* fo = SPLIT ? prow + H * C + h * D + kq * 4 : feat + (rowbase + i) * FEAT + H * C + h * D + kq * 4;
                *reinterpret_cast<f32x4*>(fo) = accV[hh][0] * inv;
                *reinterpret_cast<f32x4*>(fo + 16) = accV[hh][1] * inv;
            }
            float* po = (SPLIT ? prow + H * C + H * D + h * (P * 3) : pts + (fm * H + h) * (P * 3)) + kq * 3;
            if (SPLIT && i >= L) continue;
#pragma unroll
            for (int r = 0; r < 3; ++r) { po[r] = accT[hh][0][r] * inv; po[12 + r] = accT[hh][1][r] * inv; }
        }
        TSYNC(3, 4)                                                         // F2
    }

    if (SPLIT) return;                                                      // the merge kernel finishes the rows
    // ---------------------------------------------------------------- all waves: local frame, norm, direction of the aggregated points (ga.py:136-139)
    // one thread per 4 consecutive points of a residue: 16-byte LDS reads and global stores
    const float* pts = sm.sp;
    for (int e = tid; e < BI * (H * P / 4); e += NTH) {
        const int il = e / (H * P / 4), g = e % (H * P / 4), i = i0 + il;
        if (i >= L) continue;
        const float* Rr = R + (rowbase + i) * 9;
        const float* tr = t + (rowbase + i) * 3;
        const float r0 = Rr[0], r1 = Rr[1], r2 = Rr[2], r3 = Rr[3], r4 = Rr[4], r5 = Rr[5], r6 = Rr[6], r7 = Rr[7], r8 = Rr[8];
        const float t0 = tr[0], t1 = tr[1], t2 = tr[2];
        f32x4 a[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const f32x4*>(pts + (il * H * P + 4 * g) * 3 + 4 * q);
        float loc[12], dir[12];
        f32x4 dist;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dx = a[(3 * k) >> 2][(3 * k) & 3] - t0, dy = a[(3 * k + 1) >> 2][(3 * k + 1) & 3] - t1, dz = a[(3 * k + 2) >> 2][(3 * k + 2) & 3] - t2;
            float lx, ly, lz, d, inv;
            point_local(dx, dy, dz, r0, r1, r2, r3, r4, r5, r6, r7, r8, lx, ly, lz, d, inv);
            loc[3 * k] = lx; loc[3 * k + 1] = ly; loc[3 * k + 2] = lz;
            dir[3 * k] = lx * inv; dir[3 * k + 1] = ly * inv; dir[3 * k + 2] = lz * inv;
            dist[k] = d;
        }
        float* fpnt = feat + (rowbase + i) * FEAT + H * C + H * D;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            reinterpret_cast<f32x4*>(fpnt + 12 * g)[q] = (f32x4){loc[4 * q], loc[4 * q + 1], loc[4 * q + 2], loc[4 * q + 3]};
            reinterpret_cast<f32x4*>(fpnt + H * P * 3 + H * P + 12 * g)[q] = (f32x4){dir[4 * q], dir[4 * q + 1], dir[4 * q + 2], dir[4 * q + 3]};
        }
        *reinterpret_cast<f32x4*>(fpnt + H * P * 3 + 4 * g) = dist;
    }
#ifdef CORE_TIMING
    TSTAMP(5)
    if (blockIdx.x == 17 && lane == 0 && (wave == 0 || wave == NPW || wave == NPW + 4))
        for (int k = 0; k < 8; ++k) g_core_timing[wave == 0 ? 0 : (wave == NPW ? 1 : 2)][k] = tacc[k];
#endif
}

// =====================================================================================================================
// Persistent form of the cached core (the sampler's variant: no dump, pair bias from the per-call cache).  At the bench shape a CU
// runs two workgroups back to back, and each pays a prologue of three dependent memory round trips (q' -> LDS, first key fragments,
// first z rows: ~15k of its ~185k cycles) and an epilogue in which most waves idle (~13k).  Here ONE workgroup per CU walks its
// query blocks b, b + G, b + 2G, ... as a single stream of (block, chunk) positions g: the roles keep their one-position offsets
// ACROSS block boundaries, so while the pair waves finish a block the A waves already fetch the next block's q' (each A wave only
// ever reads its own three heads' q' slots: a wave-private swap, no extra barrier) and produce its S(0), the z ring and the C
// waves' value fragments roll on into the next block, and a block's epilogue is spread over the next block's first two intervals
// (C waves: node features and aggregated points after their last chunk; A waves: the per-point epilogue one interval later, out of
// line -- the same call from the pair waves costs their hot loop 90 spilled registers, 173 -> 200 us).
// Same arithmetic in the same order as ipa_core_kernel<false, true>: results are bit-identical (test_persistent_core_is_bit_identical).
struct PBlk { int n, i0; int64_t rowbase, zbase; };
__device__ __forceinline__ PBlk pblk_of(int b, int nib, int L, int xcd_remap, int z_shared) {
    int n, ib;
    if (xcd_remap) { const int xcd = b & 7, k = b >> 3; n = xcd + 8 * (k / nib); ib = k % nib; }
    else { n = b / nib; ib = b % nib; }
    PBlk r;
    r.n = n; r.i0 = ib * BI; r.rowbase = (int64_t)n * L; r.zbase = z_shared ? (int64_t)(n / z_shared) * L : r.rowbase;
    return r;
}

// per-point epilogue of a finished block (ga.py:136-139): local frame, norm, direction of the aggregated points; one thread per 4
// consecutive points of a residue.  Out of line on purpose: it runs once per block on waves whose hot loop must keep its registers.
__device__ __attribute__((noinline)) void persist_point_epilogue(const float* __restrict__ ptsb, const float* __restrict__ R, const float* __restrict__ t,
                                                                 float* __restrict__ feat, int64_t rowbase, int i0, int L, int th, int nth, int nrows = BI) {
    for (int e = th; e < nrows * (H * P / 4); e += nth) {
        const int il = e / (H * P / 4), g4 = e % (H * P / 4), i = i0 + il;
        if (i >= L) continue;
        const float* Rr = R + (rowbase + i) * 9;
        const float* tr = t + (rowbase + i) * 3;
        const float r0 = Rr[0], r1 = Rr[1], r2 = Rr[2], r3 = Rr[3], r4 = Rr[4], r5 = Rr[5], r6 = Rr[6], r7 = Rr[7], r8 = Rr[8];
        const float t0 = tr[0], t1 = tr[1], t2 = tr[2];
        f32x4 a[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const f32x4*>(ptsb + (il * H * P + 4 * g4) * 3 + 4 * q);
        float loc[12], dir[12];
        f32x4 dist;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dx = a[(3 * k) >> 2][(3 * k) & 3] - t0, dy = a[(3 * k + 1) >> 2][(3 * k + 1) & 3] - t1, dz = a[(3 * k + 2) >> 2][(3 * k + 2) & 3] - t2;
            float lx, ly, lz, d, inv;
            point_local(dx, dy, dz, r0, r1, r2, r3, r4, r5, r6, r7, r8, lx, ly, lz, d, inv);
            loc[3 * k] = lx; loc[3 * k + 1] = ly; loc[3 * k + 2] = lz;
            dir[3 * k] = lx * inv; dir[3 * k + 1] = ly * inv; dir[3 * k + 2] = lz * inv;
            dist[k] = d;
        }
        float* fpnt = feat + (rowbase + i) * FEAT + H * C + H * D;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            reinterpret_cast<f32x4*>(fpnt + 12 * g4)[q] = (f32x4){loc[4 * q], loc[4 * q + 1], loc[4 * q + 2], loc[4 * q + 3]};
            reinterpret_cast<f32x4*>(fpnt + H * P * 3 + H * P + 12 * g4)[q] = (f32x4){dir[4 * q], dir[4 * q + 1], dir[4 * q + 2], dir[4 * q + 3]};
        }
        *reinterpret_cast<f32x4*>(fpnt + H * P * 3 + 4 * g4) = dist;
    }
}

// Short crops (L <= 64, the reference's CDR + 20 antigen residues: a block is over after 2..4 positions and the per-block pieces set the
// pace -- at L = 48, by -DPERSIST_TIMING barrier waits per interval class: the A waves' point epilogue ~10k cycles, their q' swap ~6k,
// the C waves' block epilogue ~7k on top of 3 x 9k of positions, the pair waves waiting at 60 % of the barriers) were tried in a
// specialised form in round 3: point epilogue on the pair waves (208 bytes of scratch per lane: 4.36 -> 4.87 ms per 1000-pose step)
// and z requests after the position's MFMAs instead of before (neutral).  Neither is kept.
__global__ __launch_bounds__(NTH) void ipa_core_persist_kernel(const float* __restrict__ qfrag, const float* __restrict__ kvfrag, const float* __restrict__ z,
                                                               const uint8_t* __restrict__ mask, const float* __restrict__ R, const float* __restrict__ t,
                                                               float* __restrict__ feat, const float* __restrict__ pbc, int L, int nib, int total_blocks,
                                                               int xcd_remap, int z_shared) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nchunk = (L + JC - 1) / JC;
    float* sp = reinterpret_cast<float*>(smem_raw);                     // [3][BI][SROW]
    f32x4* qf = reinterpret_cast<f32x4*>(sp + 3 * BI * SROW);           // [H][4][64]
    float* scl = reinterpret_cast<float*>(qf + H * 4 * 64);             // [2][BI][SCLD]  by position parity
    float* lsum = scl + 2 * BI * SCLD;                                  // [BI][SCLD]
    float* ptsb = lsum + BI * SCLD;                                     // [BI][H][24]   aggregated global-frame points of the block being finished
    uint8_t* mk = reinterpret_cast<uint8_t*>(ptsb + BI * H * P * 3);     // [2][nchunk * JC]  key masks of the current / next sample
    const int mkld = nchunk * JC;
    const int tid = threadIdx.x, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
#ifdef PERSIST_TIMING
    long long pt_wait = 0, pt_mem = 0, pt_lds = 0, pt_mf = 0, pt_start = clock64(), pt_wk[3] = {0, 0, 0};
    int pt_cls = 0;                                                     // class of the current interval: 0 first chunk of a block, 2 last, 1 between
#define PSYNC() { const long long a_ = clock64(); __syncthreads(); const long long d_ = clock64() - a_; pt_wait += d_; pt_wk[pt_cls] += d_; }
#define PCLS(G) { const int c_ = (G) % nchunk; pt_cls = c_ == 0 ? 0 : (c_ == nchunk - 1 ? 2 : 1); }
#else
#define PSYNC() __syncthreads();
#define PCLS(G)
#endif
    const int nb = (total_blocks - (int)blockIdx.x + G - 1) / G;        // blocks of this workgroup (>= 1)
    const int gtot = nb * nchunk;                                       // positions
    auto blk = [&](int j) { return pblk_of((int)blockIdx.x + min(j, nb - 1) * G, nib, L, xcd_remap, z_shared); };

    // first block: q' of all heads and the key mask by all threads (later blocks: by the A waves)
    auto fill_first = [&]() {
        const PBlk b0 = blk(0);
        const f32x4* qg = reinterpret_cast<const f32x4*>(qfrag) + ((int64_t)b0.n * nib + b0.i0 / BI) * (H * 4 * 64);
#pragma unroll
        for (int e = 0; e < H * 4 * 64 / NTH; ++e) qf[e * NTH + tid] = qg[e * NTH + tid];
        for (int e = tid; e < mkld; e += NTH) mk[e] = (e < L) ? mask[b0.rowbase + e] : 0;
    };
    if (wave < NPW) {
        // =========================================================================================== pair waves: position g in interval g
        const int il0 = wave * RPW;
        const unsigned lane_b = (unsigned)fm * 16u;
        const unsigned pb_lane = (unsigned)(min(fm, H - 1) * JC + kq * 4) * 4u;
        const char *zrow[RPW], *pbrow[RPW], *zrow_n[RPW], *pbrow_n[RPW];     // wave-uniform row bases of the current and the next block
        const unsigned pb_chunk = PBC_CHUNK_MAJOR ? (unsigned)(((total_blocks / nib) / (z_shared ? z_shared : 1)) * L) * (unsigned)(H * JC * 4) : (unsigned)(H * JC * 4);
        auto rows_of = [&](const PBlk& b, const char** zr, const char** pr) {
#pragma unroll
            for (int ii = 0; ii < RPW; ++ii) {
                const int64_t row = b.zbase + min(b.i0 + il0 + ii, L - 1);
                zr[ii] = reinterpret_cast<const char*>(z + (row * (int64_t)L) * C);
                pr[ii] = reinterpret_cast<const char*>(pbc + (row * (int64_t)(PBC_CHUNK_MAJOR ? 1 : nchunk)) * (H * JC));
            }
        };
        f32x4 ring[3][4], ringb[3];
        bool has_next = nb > 1;
        // CH may be nchunk (= chunk 0 of the next block); past the very last position: a harmless re-read
#define PP_ISSUE(SLOT, II, CH)                                                                                           \
    {                                                                                                                    \
        const bool nx_ = (CH) >= nchunk;                                                                                 \
        const int ch_ = nx_ ? (has_next ? 0 : nchunk - 1) : (CH);                                                        \
        const char* zr_ = (nx_ && has_next) ? zrow_n[II] : zrow[II];                                                     \
        const char* pr_ = (nx_ && has_next) ? pbrow_n[II] : pbrow[II];                                                   \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                                \
            ring[SLOT][r_] = ZLOAD(reinterpret_cast<const f32x4*>(zr_ + ((unsigned)min(ch_ * JC + kq * 4 + r_, L - 1) * (unsigned)(C * 4) + lane_b))); \
        ringb[SLOT] = ZLOAD(reinterpret_cast<const f32x4*>(pr_ + ((unsigned)ch_ * pb_chunk + pb_lane)));   \
    }
        PBlk bk = blk(0);
        rows_of(bk, zrow, pbrow);
        { const PBlk bn = blk(1); rows_of(bn, zrow_n, pbrow_n); }
        PP_ISSUE(0, 0, 0) PP_ISSUE(1, 1, 0)
        fill_first();
        float m_run[RPW], l_run[RPW];
        f32x4 accP[RPW][4];
        const int spo = sp_off(fm, kq);
        PSYNC()                                                    // LDS tile visible
        PSYNC()                                                    // S(0) ready
        // one (query row, position): ring slot SLOT holds its z; BUF = position % 3; PAR = position parity
#define PP_POS(SLOT, II, CH, BUF, PAR)                                                                                   \
    {                                                                                                                    \
        PP_ISSUE(((SLOT) + 2) % 3, II, (CH) + 1)                            /* two ring positions ahead = same row, next chunk */ \
        const int il_ = il0 + (II);                                                                                      \
        float* spp_ = sp + ((BUF) * BI + il_) * SROW + spo;                                                              \
        f32x4 sv_ = *reinterpret_cast<const f32x4*>(spp_);                                                               \
        sv_ += ringb[SLOT];                                                                                              \
        sv_ *= kScale2;                                                                                                  \
        float l2_[4];                                                                                                    \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                               \
            const float x_ = sv_[r_];                                                                                    \
            l2_[r_] = ((mk4_ >> (8 * r_)) & 0xffu) ? x_ : x_ - kMask2;                                                   \
        }                                                                                                                \
        const float mx_ = rows_max(fmaxf(fmaxf(l2_[0], l2_[1]), fmaxf(l2_[2], l2_[3])));                                 \
        const float mn_ = fmaxf(m_run[II], mx_);                                                                         \
        const float sc_ = __builtin_amdgcn_exp2f(m_run[II] - mn_);                                                       \
        f32x4 pv_;                                                                                                       \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) pv_[r_] = __builtin_amdgcn_exp2f(l2_[r_] - mn_);                \
        const float ps_ = rows_sum((pv_[0] + pv_[1]) + (pv_[2] + pv_[3]));                                               \
        l_run[II] = l_run[II] * sc_ + ps_;                                                                               \
        m_run[II] = mn_;                                                                                                 \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) accP[II][mt_] *= sc_;                                        \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                                 \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) accP[II][mt_] = mfma4(ring[SLOT][r_][mt_], pv_[r_], accP[II][mt_]); \
        *reinterpret_cast<f32x4*>(spp_) = pv_;                                                                           \
        if (kq == 0) scl[((PAR) * BI + il_) * SCLD + fm] = sc_;                                                          \
    }
#define PP_CHUNK(K, CH, PAR)                                                                                             \
    {                                                                                                                    \
        const uint32_t mk4_ = *reinterpret_cast<const uint32_t*>(&mkc[(CH) * JC + kq * 4]);                              \
        PP_POS((2 * (K)) % 3, 0, CH, K, PAR)                                                                             \
        PP_POS((2 * (K) + 1) % 3, 1, CH, K, PAR)                                                                         \
    }
        // The position stream is unrolled by 3 so that every ring slot is a compile-time register (hipcc turns a run-time choice between
        // slot patterns into hundreds of spills); block boundaries are run-time events inside each copy.
        int c = 0, j = 0;                                                   // chunk within the block, block of the current position
        const uint8_t* mkc = mk;
        bool mi_b[RPW];
        auto block_begin = [&]() {
            if (j > 0) {
                bk = blk(j);
#pragma unroll
                for (int ii = 0; ii < RPW; ++ii) { zrow[ii] = zrow_n[ii]; pbrow[ii] = pbrow_n[ii]; }
            }
            has_next = j + 1 < nb;
            { const PBlk bn = blk(j + 1); rows_of(bn, zrow_n, pbrow_n); }
            mkc = mk + (j & 1) * mkld;
#pragma unroll
            for (int ii = 0; ii < RPW; ++ii) {
                mi_b[ii] = (bk.i0 + il0 + ii < L) && mkc[min(bk.i0 + il0 + ii, L - 1)] != 0;
                m_run[ii] = -INFINITY; l_run[ii] = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) accP[ii][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        };
        // alpha = P / l, zero for masked queries (ga.py:24-25): pair features out, denominators for the C waves
        auto block_end = [&]() {
#pragma unroll
            for (int ii = 0; ii < RPW; ++ii) {
                const int il = il0 + ii, i = bk.i0 + il;
                if (kq == 0) lsum[il * SCLD + fm] = l_run[ii];
                if (i < L && fm < H) {
                    const float inv = mi_b[ii] ? 1.f / l_run[ii] : 0.f;
                    float* fo = feat + (bk.rowbase + i) * FEAT + fm * C + kq * 16;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        reinterpret_cast<f32x4*>(fo)[r] = (f32x4){accP[ii][0][r] * inv, accP[ii][1][r] * inv, accP[ii][2][r] * inv, accP[ii][3][r] * inv};
                }
            }
        };
#define PP_STEP(K, G)                                                                                                    \
    {                                                                                                                    \
        PCLS(G)                                                                                                          \
        if (c == 0) block_begin();                                                                                       \
        PP_CHUNK(K, c, (G) & 1)                                                                                          \
        if (c == nchunk - 1) { block_end(); c = 0; ++j; } else ++c;                                                      \
        PSYNC()                                                    /* B_g */                                     \
    }
        int g = 0;
        for (; g + 3 <= gtot; g += 3) { PP_STEP(0, g) PP_STEP(1, g + 1) PP_STEP(2, g + 2) }
        if (g < gtot) {
            PP_STEP(0, g)
            if (g + 1 < gtot) PP_STEP(1, g + 1)
        }
#undef PP_STEP
        PSYNC()                                                    // T1: the C waves finished the last block
    } else if (wave < NPW + 4) {
        // =========================================================================================== A waves: S(g + 1) in interval g
        const int h0 = (wave - NPW) * 3;
        const int atid = tid - NPW * 64;
        f32x4 kf[3][4];
        auto kv_of = [&](const PBlk& b) { return reinterpret_cast<const f32x4*>(kvfrag) + (int64_t)b.n * nchunk * H * 512; };
#define PA_ISSUE(HH, KV, CH) { const f32x4* fr_ = (KV) + ((int64_t)(CH) * H + h0 + (HH)) * 512 + lane;                   \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) kf[HH][s_] = fr_[s_ * 64]; }
        // S(position) -> sp[buf]; after each head's MFMAs its registers are refilled with the NEXT position's key fragments
        auto produce = [&](int buf, const f32x4* kv_next, int ch_next) {
#ifdef PERSIST_TIMING
            { const long long a_ = clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pt_mem += clock64() - a_; }
#endif
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                const int h = h0 + hh;
                const f32x4* qh = qf + (h * 4) * 64 + lane;
                const f32x4 q0 = qh[0], q1 = qh[64], q2 = qh[128], q3 = qh[192];
                f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
                for (int s = 0; s < 4; ++s) { acc0 = mfma4(kf[hh][0][s], q0[s], acc0); acc1 = mfma4(kf[hh][2][s], q2[s], acc1); }
#pragma unroll
                for (int s = 0; s < 4; ++s) { acc0 = mfma4(kf[hh][1][s], q1[s], acc0); if (s < 3) acc1 = mfma4(kf[hh][3][s], q3[s], acc1); }
                const f32x4 sres = acc0 + acc1;
                if (hh == 0) PA_ISSUE(0, kv_next, ch_next) else if (hh == 1) PA_ISSUE(1, kv_next, ch_next) else PA_ISSUE(2, kv_next, ch_next)
                *reinterpret_cast<f32x4*>(sp + (buf * BI + fm) * SROW + sp_off(h, kq)) = sres;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // position p -> (key/value base of its block, chunk); past the last position: re-read the last chunk
        auto kv_pos = [&](int p, const f32x4*& kv, int& ch) {
            const int pc = min(p, gtot - 1);
            const PBlk b = blk(pc / nchunk);
            kv = kv_of(b); ch = pc % nchunk;
        };
        {
            const f32x4* kv0; int ch0;
            kv_pos(0, kv0, ch0);
            PA_ISSUE(0, kv0, ch0) PA_ISSUE(1, kv0, ch0) PA_ISSUE(2, kv0, ch0)
        }
        fill_first();
        PSYNC()
        { const f32x4* kv1; int ch1; kv_pos(1, kv1, ch1); produce(0, kv1, ch1); }
        PSYNC()                                                    // S(0) ready
        int bufn = 1;                                                       // buffer of position g + 1
        for (int g = 0; g < gtot; ++g) {
            PCLS(g)
            const int gn = g + 1;
            if (gn < gtot) {
                if (gn % nchunk == 0) {
                    // block switch: this wave's three heads of q' (nobody else reads these LDS slots) and, shared by the four A waves, the next key mask
                    const int jn = gn / nchunk;
                    const PBlk bnx = blk(jn);
                    const f32x4* qg = reinterpret_cast<const f32x4*>(qfrag) + ((int64_t)bnx.n * nib + bnx.i0 / BI) * (H * 4 * 64);
#pragma unroll
                    for (int hh = 0; hh < 3; ++hh) {
                        f32x4 tq[4];
#pragma unroll
                        for (int s_ = 0; s_ < 4; ++s_) tq[s_] = qg[((h0 + hh) * 4 + s_) * 64 + lane];
#pragma unroll
                        for (int s_ = 0; s_ < 4; ++s_) qf[((h0 + hh) * 4 + s_) * 64 + lane] = tq[s_];
                    }
                    uint8_t* mkn = mk + (jn & 1) * mkld;
                    for (int e = atid; e < mkld; e += 4 * 64) mkn[e] = (e < L) ? mask[bnx.rowbase + e] : 0;
                    wave_lds_sync();
                }
                const f32x4* kvx; int chx;
                kv_pos(gn + 1, kvx, chx);
                produce(bufn, kvx, chx);
                bufn = (bufn == 2) ? 0 : bufn + 1;
            }
            if (g >= nchunk && g % nchunk == 1) {                // second interval of a block: the C waves wrote the previous block's points last interval
                const PBlk bp = blk(g / nchunk - 1);
                persist_point_epilogue(ptsb, R, t, feat, bp.rowbase, bp.i0, L, atid, 4 * 64);
            }
            PSYNC()                                                // B_g
        }
        PSYNC()                                                    // T1: the C waves finished the last block
        { const PBlk bl = blk(nb - 1); persist_point_epilogue(ptsb, R, t, feat, bl.rowbase, bl.i0, L, atid, 4 * 64); }
    } else {
        // =========================================================================================== C waves: position g - 1 in interval g
        const int h0 = (wave - NPW - 4) * 3;
        f32x4 vf[3][4];
        f32x4 accV[3][2], accT[3][2];
#pragma unroll
        for (int hh = 0; hh < 3; ++hh)
#pragma unroll
            for (int k = 0; k < 2; ++k) { accV[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; accT[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        auto kv_of = [&](const PBlk& b) { return reinterpret_cast<const f32x4*>(kvfrag) + (int64_t)b.n * nchunk * H * 512; };
#define PC_ISSUE(HH, KV, CH) { const f32x4* fr_ = (KV) + ((int64_t)(CH) * H + h0 + (HH)) * 512 + 4 * 64 + lane;         \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) vf[HH][s_] = fr_[s_ * 64]; }
        auto kv_pos = [&](int p, const f32x4*& kv, int& ch) {
            const int pc = min(p, gtot - 1);
            const PBlk b = blk(pc / nchunk);
            kv = kv_of(b); ch = pc % nchunk;
        };
        auto consume = [&](int buf, int par, const f32x4* kv_next, int ch_next) {
#ifdef PERSIST_TIMING
            { const long long a_ = clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pt_mem += clock64() - a_; }
#endif
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                const int h = h0 + hh;
#ifdef PERSIST_TIMING
                const long long ta_ = clock64();
#endif
                const float sc = scl[(par * BI + fm) * SCLD + h];
                const f32x4 pa = *reinterpret_cast<const f32x4*>(sp + (buf * BI + fm) * SROW + sp_off(h, kq));
#ifdef PERSIST_TIMING
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const long long tb_ = clock64();
#endif
                accV[hh][0] *= sc; accV[hh][1] *= sc; accT[hh][0] *= sc; accT[hh][1] *= sc;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    accV[hh][0] = mfma4(vf[hh][s][0], pa[s], accV[hh][0]);
                    accV[hh][1] = mfma4(vf[hh][s][1], pa[s], accV[hh][1]);
                    accT[hh][0] = mfma4(vf[hh][s][2], pa[s], accT[hh][0]);
                    accT[hh][1] = mfma4(vf[hh][s][3], pa[s], accT[hh][1]);
                }
#ifdef PERSIST_TIMING
                __builtin_amdgcn_sched_barrier(0);
                { const long long tc_ = clock64(); pt_lds += tb_ - ta_; pt_mf += tc_ - tb_; }
#endif
                if (hh == 0) PC_ISSUE(0, kv_next, ch_next) else if (hh == 1) PC_ISSUE(1, kv_next, ch_next) else PC_ISSUE(2, kv_next, ch_next)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // after a block's last chunk: node features to HBM, aggregated points to LDS, accumulators back to zero
        auto block_epilogue = [&](int j) {
            const PBlk b = blk(j);
            const uint8_t* mkc = mk + (j & 1) * mkld;
            const bool mi_c = (b.i0 + fm < L) && mkc[min(b.i0 + fm, L - 1)] != 0;
            const int i = b.i0 + fm;
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                const int h = h0 + hh;
                const float inv = mi_c ? 1.f / lsum[fm * SCLD + h] : 0.f;
                if (i < L) {
                    float* fo = feat + (b.rowbase + i) * FEAT + H * C + h * D + kq * 4;
                    *reinterpret_cast<f32x4*>(fo) = accV[hh][0] * inv;
                    *reinterpret_cast<f32x4*>(fo + 16) = accV[hh][1] * inv;
                }
                float* po = ptsb + (fm * H + h) * (P * 3) + kq * 3;
#pragma unroll
                for (int r = 0; r < 3; ++r) { po[r] = accT[hh][0][r] * inv; po[12 + r] = accT[hh][1][r] * inv; }
#pragma unroll
                for (int k = 0; k < 2; ++k) { accV[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; accT[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            }
        };
        {
            const f32x4* kv0; int ch0;
            kv_pos(0, kv0, ch0);
            PC_ISSUE(0, kv0, ch0) PC_ISSUE(1, kv0, ch0) PC_ISSUE(2, kv0, ch0)
        }
        fill_first();
        PSYNC()
        PSYNC()                                                    // S(0) ready
        PSYNC()                                                    // B_0: P(0) ready
        int buf = 0;                                                        // buffer of position g - 1
        for (int g = 1; g <= gtot; ++g) {
            PCLS(g)
            const int p = g - 1;
            const f32x4* kvx; int chx;
            kv_pos(g, kvx, chx);
            consume(buf, p & 1, kvx, chx);
            buf = (buf == 2) ? 0 : buf + 1;
            if (p % nchunk == nchunk - 1) block_epilogue(p / nchunk);
            PSYNC()                                                // B_g (g < gtot), T1 (g == gtot)
        }
    }
#undef PP_ISSUE
#undef PP_POS
#undef PP_CHUNK
#undef PA_ISSUE
#ifdef PERSIST_TIMING
    if (blockIdx.x == 17 && lane == 0 && (wave == 0 || wave == NPW || wave == NPW + 4)) {
        long long* o = g_core_timing[wave == 0 ? 0 : (wave == NPW ? 1 : 2)];
        o[0] = clock64() - pt_start; o[1] = pt_wait; o[2] = pt_mem; o[3] = pt_lds; o[4] = pt_mf; o[5] = pt_wk[0]; o[6] = pt_wk[1]; o[7] = pt_wk[2];
    }
#endif
#undef PSYNC
#undef PCLS
#undef PC_ISSUE
}

// =====================================================================================================================
// 32-row form of the cached core (round 3).  What bounds the 16-row kernels is the number of bytes a CU pulls through its L1 per query
// row: a 16-row block reads its sample's key/value fragments (96 KB per key chunk) next to 80 KB of z and pair bias, 805 MB of 1.44 GB
// per launch at the bench shape.  Here ONE workgroup owns 32 query rows, so a fragment read serves twice the rows (256 KB instead of
// 352 KB per chunk and 32 rows).  The state of 32 rows does not fit 16 waves x 128 registers; it does fit
// 8 waves x 256 (two per SIMD, the unified VGPR/AGPR file of gfx950):
//   4 pair waves (8 query rows each)   128 accumulator registers + the 3-slot z ring (60)
//   2 "A" waves (6 heads each)         q' of their heads for both row tiles LIVES IN REGISTERS (192; 96 KB of LDS otherwise), key fragments
//                                      double buffered, one head ahead
//   2 "C" waves (6 heads each)         192 accumulator registers (value / point aggregation of 32 rows), value fragments double buffered
// Wave w and w + 4 share a SIMD: every SIMD hosts one pair wave and one A or C wave (308 / 320 MFMAs per chunk).  Same per-row
// arithmetic in the same order as the 16-row kernels: results are bit-identical to them (test_persistent_core_is_bit_identical runs this
// kernel for its large batches).  One block per workgroup; the sampler's bench shape (N = 32, L = 256) is exactly 256 blocks.
#ifndef C32_RING
#define C32_RING 3       // slots of the pair waves' z ring (3 or 4)
#endif
#ifndef C32_CPIPE
#define C32_CPIPE 0
#endif
#ifndef C32_LAZY
#define C32_LAZY 2       // exact lazy rescale: skip the accumulator rescale where every factor is exactly 1 (wave-uniform branch; bit-identical).  bit 1 = pair waves (a row),
                         // bit 2 = C waves (a (head, row tile)), in the fused kernel only.  79 % of the rescales qualify at the bench shape.  Round 5, same-box A/Bs of the fused
                         // kernel on five boxes: C waves -4.5, -0.4 / -2.6, -1.4 / -0.7, -1.5 / -0.2 % (and +4 % once with an earlier build); pair waves too: 0 on average; the
                         // two-launch core +4.5 % (hence FUSE only).  Shipped for the C waves: a per cent on average, never a different bit.
#endif
#ifdef C32_COUNT      // developer build: how often a lazy rescale could skip (printed by the launcher): {pair row-chunks, of them unchanged, C (head, tile, chunk), unchanged, C chunks (6 heads x 32 rows), unchanged}
__device__ unsigned long long g_c32_count[8];
#define C32_CNT(k, v) if (lane == 0) atomicAdd(&g_c32_count[k], (unsigned long long)(v));
#else
#define C32_CNT(k, v)
#endif
#ifndef C32_ABL
#define C32_ABL 0        // developer ablations (timing only, results wrong): 1 no z / bias loads in the loop | 32 z / bias loads all from one L1-resident address | 64 no softmax arithmetic in rows 1..7 (P = S) | 128 no pair MFMAs | 256 / 512 no A / C MFMAs (nothing instead) | 2 no fragment loads in the loop | 4 / 8 / 16 pair / A / C MFMAs off | 1024 no accumulator rescale (pair and C waves) | 2048 no v_exp_f32 in rows 1..7
#endif
#ifndef ZT_CHUNK_MAJOR
#define ZT_CHUNK_MAJOR 2  // layout of the pair terms: 2 [chunk][row of the batch][4 KB] -- every workgroup of a launch reads the SAME chunk of its rows at the same time, so the live
                          // set of a chunk interval is ONE dense plane (33 MB at the bench shape) instead of 4 KB out of every 64 KB row | 1 chunk-major inside a sample's slab | 0 [row][chunk][4 KB]
#endif
#ifndef C32_ZAUX
#define C32_ZAUX 2       // cache policy bits of the z / bias stream's buffer loads (2 = nt)
#endif
constexpr int BI2 = 32, NPW2 = 4, RPW2 = BI2 / NPW2, NTH2 = 512, HPW = 6;

#define C32_EXP2(x) ((C32_ABL & 2048) ? (x) : __builtin_amdgcn_exp2f(x))      // (ablation 2048: no v_exp_f32 in rows 1..7)
#ifdef C32_OLDMASK
#define C32_L2(x, r) (((mk4_ >> (8 * (r))) & 0xffu) ? (x) * kScale2 : (x) * kScale2 - kMask2)
#else
#define C32_L2(x, r) __builtin_fmaf((x), kScale2, mterm_[r])
#endif
// FUSE (round 4): the block's tail -- out_transform, mask, residual, LayerNorm, mlp_transition, LayerNorm (ga.py:174-177) -- runs as the
// EPILOGUE of this kernel on the 32 rows the workgroup owns; `feat` (7.3 KB per row: 60 MB written here and read back by the tail kernel
// at the bench shape) never leaves the chip and the block is two launches instead of three.  After the key loop:
//   C waves   normalise their accumulators; node features -> fp16-term planes of staging buffers 0 / 1 (chunks 4 / 5 of the 1824 feature
//             columns), aggregated points -> LDS.  From then on A and C waves are the four CONSUMERS (one per SIMD): wave 4 + cb owns the
//             32 output columns 32 cb .. 32 cb + 31 of u = feat . W_out^T for all 32 rows and every k-step -- the four accumulator chains
//             (one per K group) the stand-alone tail kernel spreads over four waves, in the same order (tail_common.h: ot_chunk_at,
//             ot_kstep3); W_out streams from L2 as pre-split fp16 terms, as there.
//   pair waves are the PRODUCERS: they split their 32 x 768 normalised pair features once (192 registers of packed terms) and, one
//             192-column chunk per interval, write them -- then what the point epilogue derives from the aggregated points (local
//             coordinates, distances, directions: chunks 6..9) -- into the staging buffer the consumers read next.  One barrier per chunk.
//   all waves the stand-alone kernel's phase 2 (tail_common.h: tail_p2_run) on 8 waves.
// Same arithmetic in the same order as ipa_core32_kernel<false, ZT> followed by out_ln_mlp_kernel: bit-identical
// (tests/test_hip_parity.py::test_fused_block_is_bit_identical).
// Clock probe (abopt_prof_clock): wave 0 of workgroup 0 of the last 32-row launch leaves {shader cycles, 100 MHz wall ticks} from its first to
// its last instruction -- the clock the chip sustained under THAT kernel (DVFS moves it between 1.7 and 2.2 GHz), so a
// reader of the bench line can tell a slow box from a slow kernel.  Two scalar timer reads and one store per launch.
__device__ long long g_clock_probe[2];
// Launch spans (abopt_prof_enable(3) / abopt_prof_spans): with a slot number in its arguments, every workgroup of a 32-row launch folds the
// 100 MHz wall clock of its first and last instruction into {min start, max end} of that slot -- the launch's duration as rocprofv3 sees it,
// but available INSIDE a replayed hipGraph, where host-recorded event pairs cannot be placed (the slot is baked into the captured node; the
// host resets the slots before the replay it wants to read).  Two atomics per workgroup.
constexpr int PROF_SLOTS = 2048;
__device__ unsigned long long g_prof_span[PROF_SLOTS][2];
__global__ void prof_span_reset_kernel() {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < PROF_SLOTS) { g_prof_span[i][0] = ~0ull; g_prof_span[i][1] = 0ull; }
}
#ifndef C32F_ABL
#define C32F_ABL 0       // developer ablations of the fused epilogue (timing only, results wrong): 1 no W_out refills | 2 no MFMAs | 4 producers write nothing | 8 / 16 C waves: no node-feature staging / no point writes
#endif
#ifdef C32F_TIMING   // developer build: clock stamps of every wave of workgroup 17 through the fused epilogue, printed by the launcher
__device__ long long g_c32f_timing[8][16];
#define C32F_STAMP(k) if (blockIdx.x == 17 && lane == 0) g_c32f_timing[wave][k] = clock64() - t32f_begin;
#else
#define C32F_STAMP(k)
#endif
struct TailArgs {
    const float *wot, *wmf, *x, *ubias, *g1, *be1, *b0, *b1, *b2, *g2, *be2;
    float* out;
    unsigned* xt;           // optional: the output rows as two fp16 terms as well (tail_common.h: tail_p2_run)
};
constexpr int C32_LOOP_LDS_FLOATS = 3 * 32 * SROW + 2 * 32 * SCLD + 32 * SCLD + 32 * 32;      // sp | scl | lsum | mlr (then the key mask)
constexpr int C32F_PTS_OFF = ((C32_LOOP_LDS_FLOATS * 4 + 2048 + 255) / 256) * 256;               // aggregated points [32][12][24] fp32, behind the key mask of L <= 2048
constexpr int C32F_PTSLD = H * P * 3 + 4;                                                          // row stride of the aggregated points (292 floats: rows 36 banks apart)
constexpr int C32F_LDS_BYTES = C32F_PTS_OFF + 32 * C32F_PTSLD * 4;
constexpr int C32F_U_OFF = 0, C32F_YS_OFF = MR * XLD * 4, C32F_APA_OFF = 2 * MR * XLD * 4;
constexpr int C32F_BIAS_OFF = 2 * OT_STAGE;                                                     // the three MLP biases: free LDS behind the staging buffers, filled during the dump
static_assert(2 * OT_STAGE <= 3 * 32 * SROW * 4, "staging buffers must fit into the S/P tile");
static_assert(C32F_BIAS_OFF + 3 * F * 4 <= 3 * 32 * SROW * 4 && C32F_APA_OFF + OT_NT * AP_PLANE <= 2 * OT_STAGE && OT_NT * AP_PLANE <= 32 * C32F_PTSLD * 4, "phase-2 buffers of the fused tail");
static_assert(C32F_LDS_BYTES <= 160 * 1024, "LDS of the fused block kernel");

// a product rounded to fp32 HERE: the values below go straight into split_pair2's `e - h`, which the compiler would otherwise contract
// with the multiplication into one fma (an exact product minus h) -- the two-launch form rounds the product when it stores feat
// (a multiplication the compiler may not contract -- not an asm statement: hipcc pads no MFMA hazard for the result register of one, see pk_bf16)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
// four consecutive feature columns of one row -> their two fp16 planes of a staging buffer
__device__ __forceinline__ void stage_put4(char* buf, int row, int col, float v0, float v1, float v2, float v3) {
    unsigned h0, l0, h1, l1;
    split_pair2(v0, v1, h0, l0);
    split_pair2(v2, v3, h1, l1);
    char* d = buf + row * OT_SROW + col * 2;
    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(d + OT_PLANE) = make_uint2(l0, l1);
}

// ZT (round 6): the pair aggregation sum_j P z on the fp16 matrix instructions.  z arrives PRE-SPLIT (`zt`, pair_terms_kernel: once per sample() call,
// like the bias cache) as two fp16 terms of S_ic z packed along K -- per (query row, chunk, channel tile) a lane's 16 bytes are {h(keys 4 kq .. + 3),
// l(same keys)} of channel 4 fm + tile: exactly one A operand of v_mfma_f32_16x16x32_f16, and the same 4 bytes per value the fp32 stream has.  The lane's own
// four probabilities (times 2^14, so that their low terms are normal fp16 numbers) are split in registers (16 VALU operations per row-chunk) and form the
// B operand {P_h, P_h}: one 16-cycle MFMA per channel tile gives P_h z_h + P_h z_l, a K = 16 MFMA P_l z_h -- 8 MFMAs of 16 cycles per row-chunk instead
// of 16 of 32 (P_l z_l <= 2^-22 relative is dropped, as in ipa_common.h).  S_ic is a power of two per (query row, channel) (max_j |z[i,j,c]| S_ic in [2^13, 2^14)),
// 2^-14 / S_ic (`zsc`, [rows][64]) leaves through the final normalisation: exact.  P itself, its row sums and everything the C waves compute are those of the fp32 form, bit for bit.  Measured (profiles/r06_a_*): -7 % of the replayed step.
template <bool FUSE, bool ZT>
__global__ __launch_bounds__(NTH2) void ipa_core32_kernel(const float* __restrict__ qfrag, const float* __restrict__ kvfrag, const float* __restrict__ z,
                                                          const uint8_t* __restrict__ mask, const float* __restrict__ R, const float* __restrict__ t,
                                                          float* __restrict__ feat, const float* __restrict__ pbc, int L, int nib2, int xcd_remap, int z_shared,
                                                          TailArgs ta, int prof_slot, const float* __restrict__ zt, const float* __restrict__ zsc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sp = reinterpret_cast<float*>(smem_raw);                     // [3][BI2][SROW]
    float* scl = sp + 3 * BI2 * SROW;                                   // [2][BI2][SCLD]
    float* lsum = scl + 2 * BI2 * SCLD;                                 // [BI2][SCLD]
    float* mlr = lsum + BI2 * SCLD;                                     // [BI2][16][2]  running maximum / sum of every (row, head): the pair waves have no registers left for them
    uint8_t* mk = reinterpret_cast<uint8_t*>(mlr + BI2 * 32);           // [nchunk * JC]
    int n, ib;
    {
        const int b = blockIdx.x;
        if (xcd_remap == 2) {
            // groups of g samples share a complex (config 4: 8 complexes x 16 samples in one launch): a complex stays on ONE XCD, and the
            // workgroups that are resident there together work on the same query block of its g samples -- the 2 MB of z (and the bias
            // cache rows) of that block are fetched from HBM once and served to the other g - 1 samples by the XCD's L2
            const int xcd = b & 7, k = b >> 3, per = nib2 * z_shared, r = k % per;
            ib = r / z_shared; n = (xcd + 8 * (k / per)) * z_shared + r % z_shared;
        } else if (xcd_remap) { const int xcd = b & 7, k = b >> 3; n = xcd + 8 * (k / nib2); ib = k % nib2; }
        else { n = b / nib2; ib = b % nib2; }
    }
    const int tid = threadIdx.x, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = (L + JC - 1) / JC, nib16 = (L + BI - 1) / BI;
    const long long probe_c0 = clock64(), probe_w0 = wall_clock64();
    if (prof_slot >= 0 && threadIdx.x == 0) atomicMin(&g_prof_span[prof_slot][0], (unsigned long long)probe_w0);
#ifdef C32F_TIMING
    const long long t32f_begin = clock64();
#endif
#ifdef C32_TIMING   // developer build: per-role clocks of workgroup 17: total | barrier waits in the chunk loop | wall clock (100 MHz ticks)
    const long long t_begin = clock64(), w_begin = wall_clock64();
    long long t_wait = 0;
    if (blockIdx.x == 17 && lane == 0) g_core_timing[2][7 - 0] = 0;
    if (blockIdx.x == 17 && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&g_core_timing[0][4 + (wave >> 2)]) + 0, (unsigned long long)__builtin_amdgcn_s_getreg(2308) << (8 * (wave & 3)));   // SIMD of every wave
#define C32_SYNC() { const long long t0_ = clock64(); __syncthreads(); t_wait += clock64() - t0_; }
#define C32_REPORT(ROLE) if (blockIdx.x == 17 && lane == 0 && (wave == 0 || wave == NPW2 || wave == NPW2 + 2)) { g_core_timing[ROLE][0] = clock64() - t_begin; g_core_timing[ROLE][1] = t_wait; g_core_timing[ROLE][2] = wall_clock64() - w_begin; }
#else
#define C32_SYNC() __syncthreads();
#define C32_REPORT(ROLE) {}
#endif
    const int i0 = ib * BI2;
    const int64_t rowbase = (int64_t)n * L;
    const int64_t zbase = z_shared ? (int64_t)(n / z_shared) * L : rowbase;
    auto fill_mask = [&]() { for (int e = tid; e < nchunk * JC; e += NTH2) mk[e] = (e < L) ? mask[rowbase + e] : 0; };
    // ---- fused tail (FUSE): LDS views valid after barrier F1, and the consumer role (A and C waves after their loops)
    char* const stage = smem_raw;                                               // [2][3][32][OT_SROW]: aliases the S/P tile
    float* const ptsf = reinterpret_cast<float*>(smem_raw + C32F_PTS_OFF);      // [32][H][24]
    const int64_t row0f = rowbase + i0, row_endf = rowbase + L;               // the 32 rows of this block in the flattened [N L] order
    TailP2Pre<NTH2 / 64> pre;                                                   // what phase 2 needs from global memory: requested while phase 1 finishes
#ifndef C32F_RD
#define C32F_RD 12
#endif
    constexpr int RD = C32F_RD;                                                 // W_out k-steps in flight per consumer wave: 12 x 2 KB of terms, 96 KB per CU
    // W_out arrives PRE-SPLIT (ta.wot: the scaled fp16 terms pack_tail_weights_kernel makes once per weight version; out_ln_mlp_kernel streams
    // the same layout): one consumer wave per SIMD has nobody to hide a split's VALU operations behind -- with fp32 fragments split in
    // registers (round 4, first version) the phase was bound by that wave's instruction issue (466 cycles per k-step, 64k cycles in all).
    struct WT { u32x4 h, l; };
    auto consumer = [&](int cb, WT (&wt)[RD]) {
        // k-step number i of the sequence = position i / 12, (K group, slot) i % 12  ->  k-step 12 chunk(position) + i % 12 of W_out
        const u32x4* wfr = reinterpret_cast<const u32x4*>(ta.wot) + (int64_t)cb * OT_ST * OT_WVEC + lane;
        auto kstep = [&](int i) { return min(ot_chunk_at(min(i / OT_SPC, OT_NCH - 1)) * OT_SPC + (i % OT_SPC), OT_ST - 1); };
        f32x16 acc[4];                                                          // one chain per K group (ot_kstep3)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc_zero(acc[g]);
        C32F_STAMP(1)
        __syncthreads();                                                        // E0: staging buffers 0 / 1 and the aggregated points are in LDS
        C32F_STAMP(2)
        const char* xrd = stage + (lane & 31) * OT_SROW + (lane >> 5) * 16;
        // One k-step: the three products of ot_kstep3 on operands already in registers; the feat terms of the next k-step are requested from
        // LDS before the first product (this wave is alone on its SIMD's matrix pipe: nobody else hides an LDS round trip) and the ring
        // slot is refilled with the W terms of k-step i + RD.  Straight-line code per position (a branch per k-step makes the compiler
        // wait for ALL outstanding requests at every block entry).
#define C32F_KSTEP(P_, KK, NK)                                                                                           \
        {                                                                                                                \
            const int i_ = (P_) * OT_SPC + (KK);                                                                         \
            const u32x4 wH = wt[(KK) % RD].h, wL = wt[(KK) % RD].l;                                                      \
            const u32x4 xh = xnh, xl = xnl;                                                                              \
            if ((KK) + 1 < (NK)) {                                                                                       \
                xnh = *reinterpret_cast<const u32x4*>(xb + ((KK) % OT_SPC + 1) * 32); xnl = *reinterpret_cast<const u32x4*>(xb + ((KK) % OT_SPC + 1) * 32 + OT_PLANE); \
            }                                                                                                            \
            if (C32F_ABL & 2) acc[((KK) % OT_SPC) / OT_SPW][0] += __uint_as_float((wH[0] ^ xh[0]) + (wL[2] ^ xl[2]));    \
            else ot_kstep3(wH, wL, xh, xl, acc[((KK) % OT_SPC) / OT_SPW]);                                               \
            if (!(C32F_ABL & 1) && i_ + RD < OT_ST) { const u32x4* nf_ = wfr + (int64_t)kstep(i_ + RD) * OT_WVEC; wt[(KK) % RD].h = nf_[0]; wt[(KK) % RD].l = nf_[64]; } \
        }
#define C32F_XFIRST() { xnh = *reinterpret_cast<const u32x4*>(xb); xnl = *reinterpret_cast<const u32x4*>(xb + OT_PLANE); }
        u32x4 xnh, xnl;
        static_assert(OT_SPC == 12 && OT_ST - (OT_NCH - 1) * OT_SPC == 6 && ot_chunk_at(OT_NCH - 1) == OT_NCH - 1, "positions 0..8 are full chunks, the last one holds six k-steps");
        static_assert((2 * OT_SPC) % RD == 0, "ring slots must repeat every two positions");
        for (int p = 0; p < OT_NCH - 2; p += 2) {                               // two positions per trip: the ring slot of a k-step is a compile-time constant
            {
                const char* xb = xrd;
                C32F_XFIRST()
                C32F_KSTEP(p, 0, 12) C32F_KSTEP(p, 1, 12) C32F_KSTEP(p, 2, 12) C32F_KSTEP(p, 3, 12) C32F_KSTEP(p, 4, 12) C32F_KSTEP(p, 5, 12)
                C32F_KSTEP(p, 6, 12) C32F_KSTEP(p, 7, 12) C32F_KSTEP(p, 8, 12) C32F_KSTEP(p, 9, 12) C32F_KSTEP(p, 10, 12) C32F_KSTEP(p, 11, 12)
                C32F_STAMP(3 + p)
                __syncthreads();                                                // E(p + 1)
            }
            {
                const char* xb = xrd + OT_STAGE;
                C32F_XFIRST()
                C32F_KSTEP(p, 12, 24) C32F_KSTEP(p, 13, 24) C32F_KSTEP(p, 14, 24) C32F_KSTEP(p, 15, 24) C32F_KSTEP(p, 16, 24) C32F_KSTEP(p, 17, 24)
                C32F_KSTEP(p, 18, 24) C32F_KSTEP(p, 19, 24) C32F_KSTEP(p, 20, 24) C32F_KSTEP(p, 21, 24) C32F_KSTEP(p, 22, 24) C32F_KSTEP(p, 23, 24)
                C32F_STAMP(4 + p)
                __syncthreads();                                                // E(p + 2)
            }
        }
        {
            const char* xb = xrd;                                               // position 8
            C32F_XFIRST()
            C32F_KSTEP(8, 0, 12) C32F_KSTEP(8, 1, 12) C32F_KSTEP(8, 2, 12) C32F_KSTEP(8, 3, 12) C32F_KSTEP(8, 4, 12) C32F_KSTEP(8, 5, 12)
            C32F_KSTEP(8, 6, 12) C32F_KSTEP(8, 7, 12) C32F_KSTEP(8, 8, 12) C32F_KSTEP(8, 9, 12) C32F_KSTEP(8, 10, 12) C32F_KSTEP(8, 11, 12)
            C32F_STAMP(3 + 8)
            __syncthreads();                                                    // E9
        }
        {
            const char* xb = xrd + OT_STAGE;                                    // position 9: six k-steps
            C32F_XFIRST()

            C32F_KSTEP(8, 12, 18) C32F_KSTEP(8, 13, 18) C32F_KSTEP(8, 14, 18) C32F_KSTEP(8, 15, 18) C32F_KSTEP(8, 16, 18) C32F_KSTEP(8, 17, 18)
            C32F_STAMP(3 + 9)
            __syncthreads();                                                    // E10
        }
#undef C32F_KSTEP
#undef C32F_XFIRST
        C32F_STAMP(13)
        // S_out u (without the bias) = ((p0 + p1) + (p2 + p3)): what out_ln_mlp_kernel forms from its four K-group slabs
        float (*us)[XLD] = reinterpret_cast<float (*)[XLD]>(smem_raw + C32F_U_OFF);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = 4 * g + e;
                v[e] = (acc[0][q] + acc[1][q]) + (acc[2][q] + acc[3][q]);
            }
            *reinterpret_cast<f32x4*>(&us[lane & 31][cb * 32 + g * 8 + (lane >> 5) * 4]) = v;
        }
    };
    auto consumer_prefetch = [&](int cb, WT (&wt)[RD]) {
        const u32x4* wfr = reinterpret_cast<const u32x4*>(ta.wot) + (int64_t)cb * OT_ST * OT_WVEC + lane;
#pragma unroll
        for (int j = 0; j < RD; ++j) { const u32x4* f = wfr + (int64_t)(ot_chunk_at(0) * OT_SPC + j) * OT_WVEC; wt[j].h = f[0]; wt[j].l = f[64]; }
        static_assert(RD <= OT_SPC, "the first fragments belong to position 0");
    };

    if (wave < NPW2) {
        // =========================================================================================== pair waves: 8 rows each
        const int il0 = wave * RPW2;
        // z and the bias cache are read through buffer descriptors of this sample's slabs: a request is ONE instruction -- descriptor (SGPRs),
        // the row's byte offset (an SGPR, added by the hardware) and the lane's offset inside a row (a VGPR that lives for a whole chunk)
        // (ZT: the same bytes per row-chunk, 4 KB, in the term layout [chunk][channel tile][lane]: every request is 1 KB contiguous, chunks past L are zero-padded)
        const int rows_tot_ = ((int)gridDim.x / nib2 / (z_shared ? z_shared : 1)) * L;          // rows of the terms buffer (distinct samples x L): ZT_CHUNK_MAJOR == 2
        const __amdgpu_buffer_rsrc_t zrs = ZT ? (ZT_CHUNK_MAJOR == 2 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zt), 0, (unsigned)rows_tot_ * (unsigned)nchunk * (JC * C * 4), 0x00020000)
                                                                     : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zt + zbase * (int64_t)nchunk * (JC * C)), 0, L * nchunk * (JC * C * 4), 0x00020000))
                                              : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(z + zbase * (int64_t)L * C), 0, L * L * C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t brs = PBC_CHUNK_MAJOR ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pbc), 0, (unsigned)rows_tot_ * (unsigned)nchunk * (H * JC * 4), 0x00020000)
                                                           : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pbc + zbase * (int64_t)nchunk * (H * JC)), 0, L * nchunk * (H * JC) * 4, 0x00020000);
        int zrow[RPW2], pbrow[RPW2];
#pragma unroll
        for (int ii = 0; ii < RPW2; ++ii) {
            const int row = min(i0 + il0 + ii, L - 1);
            zrow[ii] = ZT ? (ZT_CHUNK_MAJOR == 2 ? ((int)zbase + row) * (JC * C * 4) : (ZT_CHUNK_MAJOR ? row * (JC * C * 4) : row * nchunk * (JC * C * 4))) : row * L * (C * 4);
            pbrow[ii] = PBC_CHUNK_MAJOR ? ((int)zbase + row) * (H * JC * 4) : row * nchunk * (H * JC * 4);
        }
        const unsigned lane_b = (unsigned)fm * 16u;
        const unsigned pb_lane = (unsigned)(min(fm, H - 1) * JC + kq * 4) * 4u;
        f32x4 ring[C32_RING][4], ringb[4];                                  // z ring: requests run C32_RING - 1 positions ahead; the bias (needed one position earlier, see below) 3 ahead
        // byte offsets of the lane's four key rows / of its bias quad inside a row, for the chunk the requests currently go to: computed once
        // per chunk, so a request is one instruction (SGPR row base + VGPR offset) and no address arithmetic rides in the hot loop
        unsigned koff_[4], boff_;
#define P2_KOFF(CH)                                                                                                      \
    {                                                                                                                    \
        const int ch_ = (C32_ABL & 32) ? 0 : min((CH), nchunk - 1);             /* past the end: harmless re-read */      \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) koff_[r_] = ZT ? ((unsigned)ch_ * (unsigned)(ZT_CHUNK_MAJOR == 2 ? rows_tot_ * (JC * C * 4) : (ZT_CHUNK_MAJOR ? L * (JC * C * 4) : (JC * C * 4))) + (unsigned)(r_ * 1024 + lane * 16)) : (unsigned)min(ch_ * JC + kq * 4 + r_, L - 1) * (unsigned)(C * 4) + lane_b; \
    }
#define P2_BOFF(CH) boff_ = (unsigned)((C32_ABL & 32) ? 0 : min((CH), nchunk - 1)) * (PBC_CHUNK_MAJOR ? (unsigned)rows_tot_ * (unsigned)(H * JC * 4) : (unsigned)(H * JC * 4)) + pb_lane;
#ifdef C32_OLDLOAD
        const char* zslab = reinterpret_cast<const char*>(z + zbase * (int64_t)L * C);
        const char* bslab = reinterpret_cast<const char*>(pbc + (PBC_CHUNK_MAJOR ? 0 : zbase * (int64_t)nchunk * (H * JC)));
#define P2_ISSUE_Z(SLOT, II)                                                                                            \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                                \
            ring[SLOT][r_] = ZLOAD(reinterpret_cast<const f32x4*>(zslab + zrow[II] + koff_[r_]));
#define P2_ISSUE_B(SLOT, II) ringb[SLOT] = ZLOAD(reinterpret_cast<const f32x4*>(bslab + pbrow[II] + boff_));
#else
#define P2_ISSUE_Z(SLOT, II)                                                                                            \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                                \
            ring[SLOT][r_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, koff_[r_], zrow[(C32_ABL & 32) ? 0 : II], C32_ZAUX));   /* aux 2 = nt, as ZLOAD */
#define P2_ISSUE_B(SLOT, II) ringb[SLOT] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, boff_, pbrow[(C32_ABL & 32) ? 0 : II], C32_ZAUX));
#endif
        P2_KOFF(0) P2_BOFF(0)
        P2_ISSUE_B(0, 0) P2_ISSUE_Z(0, 0) P2_ISSUE_B(1, 1) P2_ISSUE_Z(1, 1) P2_ISSUE_B(2, 2)
        if (C32_RING == 4) P2_ISSUE_Z(2, 2)
        fill_mask();
        f32x4 accP[RPW2][4];
        float* mlw = mlr + (il0 * 16 + fm) * 2;                             // this wave's rows; all four key groups of a lane column keep the same value
#pragma unroll
        for (int ii = 0; ii < RPW2; ++ii) {
            if (kq == 0) *reinterpret_cast<float2*>(mlw + ii * 32) = make_float2(-INFINITY, 0.f);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) accP[ii][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int spo = sp_off(fm, kq);
        __syncthreads();                                                    // key mask visible
        __syncthreads();                                                    // barrier #0: S(0) ready
        // Position (chunk CH, row II) is number 8 CH + II; its z sits in ring slot (8 CH + II) % C32_RING, its bias in slot II % 4.
        // With two waves on a SIMD nobody else fills the ~40 dependent VALU / LDS steps of a row's softmax, so a row's softmax runs
        // in the shadow of the PREVIOUS row's 16 MFMAs (software pipeline inside a chunk; a chunk's first row waits for its barrier).
        // softmax of row II of the chunk in sp[BUF] -> pvn_, scn_ (P, the rescale factor and the running maximum / sum go to LDS)
#define P2_SM(II, BUF)                                                                                                   \
    {                                                                                                                    \
        const int il_ = il0 + (II);                                                                                      \
        float* spp_ = sp + ((BUF) * BI2 + il_) * SROW + spo;                                                             \
        f32x4 sv_ = *reinterpret_cast<const f32x4*>(spp_);                                                               \
        const float2 ml_ = *reinterpret_cast<const float2*>(mlw + (II) * 32);                                            \
        sv_ += ringb[(II) & 3];                                                                                          \
        float l2_[4];                                                                                                    \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) l2_[r_] = C32_L2(sv_[r_], r_);          \
        const float mx_ = rows_max(fmaxf(fmaxf(l2_[0], l2_[1]), fmaxf(l2_[2], l2_[3])));                                 \
        const float mn_ = fmaxf(ml_.x, mx_);                                                                             \
        scn_ = __builtin_amdgcn_exp2f(ml_.x - mn_);                                                                      \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) pvn_[r_] = __builtin_amdgcn_exp2f(l2_[r_] - mn_);               \
        const float ps_ = rows_sum((pvn_[0] + pvn_[1]) + (pvn_[2] + pvn_[3]));                                           \
        const float ln_ = ml_.y * scn_ + ps_;                                                                            \
        *reinterpret_cast<f32x4*>(spp_) = pvn_;                                                                          \
        if (kq == 0) { scl[((CHPAR) * BI2 + il_) * SCLD + fm] = scn_; *reinterpret_cast<float2*>(mlw + (II) * 32) = make_float2(mn_, ln_); } \
    }
        // one MFMA of row II (k-step K_ >> 2, channel tile K_ & 3) and a fence: the source order below IS the issue order
#define P2_MF(SLOT, II, K_)                                                                                              \
        if (C32_ABL & 4096) accP[II][(K_) & 3][(K_) >> 2] += ring[SLOT][(K_) >> 2][(K_) & 3] * pvc_[(K_) >> 2]; else     /* (ablation 4096: one VALU fma per operand instead of the MFMA: the loads stay alive) */ \
        if (!(C32_ABL & 128)) accP[II][(K_) & 3] = mfma4(ring[SLOT][(K_) >> 2][(K_) & 3], pvc_[(K_) >> 2], accP[II][(K_) & 3]);                \
        __builtin_amdgcn_sched_barrier(0);
        // rows 0..6 of a chunk: the 16 MFMAs of row II, each followed by a piece of row II + 1's softmax and accumulator rescale
        // (an MFMA holds the pipe for 32 cycles; 3..8 dependent VALU steps ride in its shadow)
#define P2_POS(SLOT, II, CH, BUF)                                                                                        \
    {                                                                                                                    \
        const f32x4 pvc_ = pvn_;                                                                                         \
        if ((II) + 3 == 8) P2_BOFF((CH) + 1)                                /* the requests move on to the next chunk */ \
        if ((II) + C32_RING - 1 == 8) P2_KOFF((CH) + 1)                                                                  \
        if (!(C32_ABL & 1)) {                                                                                            \
            P2_ISSUE_B(((II) + 3) & 3, ((II) + 3) & 7)                                                                   \
            P2_ISSUE_Z(((SLOT) + C32_RING - 1) % C32_RING, ((II) + C32_RING - 1) & 7)                                    \
        }                                                                                                                \
        const int il_ = il0 + (II) + 1;                                                                                  \
        float* spp_ = sp + ((BUF) * BI2 + il_) * SROW + spo;                                                             \
        f32x4 sv_ = *reinterpret_cast<const f32x4*>(spp_);                                                               \
        const float2 ml_ = *reinterpret_cast<const float2*>(mlw + ((II) + 1) * 32);                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        P2_MF(SLOT, II, 0) P2_MF(SLOT, II, 1)                                                                            \
        if (C32_ABL & 64) { pvn_ = sv_; scn_ = 1.f; *reinterpret_cast<f32x4*>(spp_) = pvn_;                              \
            P2_MF(SLOT, II, 2) P2_MF(SLOT, II, 3) P2_MF(SLOT, II, 4) P2_MF(SLOT, II, 5) P2_MF(SLOT, II, 6) P2_MF(SLOT, II, 7) P2_MF(SLOT, II, 8) P2_MF(SLOT, II, 9) \
            P2_MF(SLOT, II, 10) P2_MF(SLOT, II, 11) P2_MF(SLOT, II, 12) P2_MF(SLOT, II, 13) P2_MF(SLOT, II, 14) P2_MF(SLOT, II, 15) } else {   \
        sv_ += ringb[((II) + 1) & 3];                                                                                    \
        P2_MF(SLOT, II, 2)                                                                                               \
        float l2_[4];                                                                                                    \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) l2_[r_] = C32_L2(sv_[r_], r_);          \
        P2_MF(SLOT, II, 3)                                                                                               \
        float mx_ = fmaxf(fmaxf(l2_[0], l2_[1]), fmaxf(l2_[2], l2_[3]));                                                 \
        { auto a_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx_), __float_as_uint(mx_), false, false);          \
          mx_ = fmaxf(__uint_as_float(a_[0]), __uint_as_float(a_[1])); }                                                 \
        P2_MF(SLOT, II, 4)                                                                                               \
        { auto b_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx_), __float_as_uint(mx_), false, false);          \
          mx_ = fmaxf(__uint_as_float(b_[0]), __uint_as_float(b_[1])); }                                                 \
        const float mn_ = fmaxf(ml_.x, mx_);                                                                             \
        scn_ = __builtin_amdgcn_exp2f(ml_.x - mn_);                                                                      \
        P2_MF(SLOT, II, 5)                                                                                               \
        pvn_[0] = C32_EXP2(l2_[0] - mn_); pvn_[1] = C32_EXP2(l2_[1] - mn_);                                              \
        P2_MF(SLOT, II, 6)                                                                                               \
        pvn_[2] = C32_EXP2(l2_[2] - mn_); pvn_[3] = C32_EXP2(l2_[3] - mn_);                                              \
        float ps_ = (pvn_[0] + pvn_[1]) + (pvn_[2] + pvn_[3]);                                                           \
        P2_MF(SLOT, II, 7)                                                                                               \
        { auto a_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(ps_), __float_as_uint(ps_), false, false);          \
          ps_ = __uint_as_float(a_[0]) + __uint_as_float(a_[1]); }                                                       \
        P2_MF(SLOT, II, 8)                                                                                               \
        { auto b_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps_), __float_as_uint(ps_), false, false);          \
          ps_ = __uint_as_float(b_[0]) + __uint_as_float(b_[1]); }                                                       \
        const float ln_ = ml_.y * scn_ + ps_;                                                                            \
        P2_MF(SLOT, II, 9)                                                                                               \
        *reinterpret_cast<f32x4*>(spp_) = pvn_;                                                                          \
        P2_MF(SLOT, II, 10)                                                                                              \
        if (kq == 0) { scl[((CHPAR) * BI2 + il_) * SCLD + fm] = scn_; *reinterpret_cast<float2*>(mlw + ((II) + 1) * 32) = make_float2(mn_, ln_); } \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        /* C32_LAZY & 1: no running maximum of this row moved in this chunk (every factor exactly 1) -> the 16 multiplications change */ \
        /* nothing and are skipped (wave-uniform); they then run as one group behind the row's last MFMAs instead of between them     */ \
        if (C32_LAZY & 1) {                                                                                              \
        P2_MF(SLOT, II, 11) P2_MF(SLOT, II, 12) P2_MF(SLOT, II, 13) P2_MF(SLOT, II, 14) P2_MF(SLOT, II, 15)              \
        if ((__builtin_amdgcn_ballot_w64(scn_ != 1.f) & 0x0fff0fff0fff0fffull) != 0ull) {                                \
            accP[(II) + 1][0] *= scn_; accP[(II) + 1][1] *= scn_; accP[(II) + 1][2] *= scn_; accP[(II) + 1][3] *= scn_; } \
        } else {                                                                                                         \
        P2_MF(SLOT, II, 11)                                                                                              \
        if (!(C32_ABL & 1024)) accP[(II) + 1][0] *= scn_;                                                                \
        P2_MF(SLOT, II, 12)                                                                                              \
        if (!(C32_ABL & 1024)) accP[(II) + 1][1] *= scn_;                                                                \
        P2_MF(SLOT, II, 13)                                                                                              \
        if (!(C32_ABL & 1024)) accP[(II) + 1][2] *= scn_;                                                                \
        P2_MF(SLOT, II, 14)                                                                                              \
        if (!(C32_ABL & 1024)) accP[(II) + 1][3] *= scn_;                                                                \
        P2_MF(SLOT, II, 15) } }                                                                                          \
    }
        // the last row of a chunk: nothing to overlap with (the next chunk's logits are behind the barrier)
#define P2_POS_LAST(SLOT, II, CH)                                                                                        \
    {                                                                                                                    \
        const f32x4 pvc_ = pvn_;                                                                                         \
        if ((II) + 3 == 8) P2_BOFF((CH) + 1)                                /* the requests move on to the next chunk */ \
        if ((II) + C32_RING - 1 == 8) P2_KOFF((CH) + 1)                                                                  \
        if (!(C32_ABL & 1)) {                                                                                            \
            P2_ISSUE_B(((II) + 3) & 3, ((II) + 3) & 7)                                                                   \
            P2_ISSUE_Z(((SLOT) + C32_RING - 1) % C32_RING, ((II) + C32_RING - 1) & 7)                                    \
        }                                                                                                                \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                                 \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) { if (C32_ABL & 4096) accP[II][mt_][r_] += ring[SLOT][r_][mt_] * pvc_[r_]; else accP[II][mt_] = mfma4(ring[SLOT][r_][mt_], pvc_[r_], accP[II][mt_]); } \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
        // ---- ZT (round 6): z arrives as K-packed two-term fp16 (pair_terms_kernel): the ring slot r of a row IS the A operand of channel tile r -- 8 halves {z_h(keys 4 kq .. + 3), z_l(same keys)} of
        // channel 4 fm + r -- and the lane's own four probabilities are the B operand: {P_h, P_h} against {z_h, z_l}, then {P_l, 0} against {z_h, -}
        // the lane's four probabilities times 2^14 (exact; their low fp16 terms are then normal numbers down to P = 2^-28) -> {P_h(k0, k1), P_h(k2, k3), P_l(..), P_l(..)}
#define P2H_SPLIT()                                                                                                      \
        { unsigned h01_, l01_, h23_, l23_;                                                                               \
          split_pair2(pvn_[0] * 16384.f, pvn_[1] * 16384.f, h01_, l01_); split_pair2(pvn_[2] * 16384.f, pvn_[3] * 16384.f, h23_, l23_); \
          pkn_ = (u32x4){h01_, h23_, l01_, l23_}; }
#define P2H_SM(II, BUF)                                                                                                   \
    {                                                                                                                    \
        const int il_ = il0 + (II);                                                                                      \
        float* spp_ = sp + ((BUF) * BI2 + il_) * SROW + spo;                                                             \
        f32x4 sv_ = *reinterpret_cast<const f32x4*>(spp_);                                                               \
        const float2 ml_ = *reinterpret_cast<const float2*>(mlw + (II) * 32);                                            \
        sv_ += ringb[(II) & 3];                                                                                          \
        float l2_[4];                                                                                                    \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) l2_[r_] = C32_L2(sv_[r_], r_);                                  \
        const float mx_ = rows_max(fmaxf(fmaxf(l2_[0], l2_[1]), fmaxf(l2_[2], l2_[3])));                                 \
        const float mn_ = fmaxf(ml_.x, mx_);                                                                             \
        scn_ = __builtin_amdgcn_exp2f(ml_.x - mn_);                                                                      \
        f32x4 pvn_;                                                                                                      \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) pvn_[r_] = __builtin_amdgcn_exp2f(l2_[r_] - mn_);               \
        const float ps_ = rows_sum((pvn_[0] + pvn_[1]) + (pvn_[2] + pvn_[3]));                                           \
        const float ln_ = ml_.y * scn_ + ps_;                                                                            \
        P2H_SPLIT()                                                                                                      \
        *reinterpret_cast<f32x4*>(spp_) = pvn_;                              /* the C waves read P as fp32, as before */ \
        if (kq == 0) { scl[((CHPAR) * BI2 + il_) * SCLD + fm] = scn_; *reinterpret_cast<float2*>(mlw + (II) * 32) = make_float2(mn_, ln_); } \
    }
        // MFMA number K_ of row II: channel tile K_ & 3, term K_ >> 2
#define P2H_MF(SLOT, II, K_)                                                                                              \
        { const u32x4 za_ = __builtin_bit_cast(u32x4, ring[SLOT][(K_) & 3]);                                              \
          if ((K_) >> 2) accP[II][(K_) & 3] = mfma_h16((u32x2){za_[0], za_[1]}, bl_, accP[II][(K_) & 3]);                 \
          else accP[II][(K_) & 3] = mfma_h(za_, bh_, accP[II][(K_) & 3]); }                                              \
        __builtin_amdgcn_sched_barrier(0);
#define P2H_POS(SLOT, II, CH, BUF)                                                                                        \
    {                                                                                                                    \
        const u32x4 bh_ = (u32x4){pkn_[0], pkn_[1], pkn_[0], pkn_[1]}; const u32x2 bl_ = (u32x2){pkn_[2], pkn_[3]};          \
        if ((II) + 3 == 8) P2_BOFF((CH) + 1)                                /* the requests move on to the next chunk */ \
        if ((II) + C32_RING - 1 == 8) P2_KOFF((CH) + 1)                                                                  \
        if (!(C32_ABL & 1)) {                                                                                            \
            P2_ISSUE_B(((II) + 3) & 3, ((II) + 3) & 7)                                                                   \
            P2_ISSUE_Z(((SLOT) + C32_RING - 1) % C32_RING, ((II) + C32_RING - 1) & 7)                                    \
        }                                                                                                                \
        const int il_ = il0 + (II) + 1;                                                                                  \
        float* spp_ = sp + ((BUF) * BI2 + il_) * SROW + spo;                                                             \
        f32x4 sv_ = *reinterpret_cast<const f32x4*>(spp_);                                                               \
        const float2 ml_ = *reinterpret_cast<const float2*>(mlw + ((II) + 1) * 32);                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        P2H_MF(SLOT, II, 0) P2H_MF(SLOT, II, 1)                                                                            \
        sv_ += ringb[((II) + 1) & 3];                                                                                    \
        float l2_[4];                                                                                                    \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) l2_[r_] = C32_L2(sv_[r_], r_);                                  \
        P2H_MF(SLOT, II, 2)                                                                                               \
        float mx_ = fmaxf(fmaxf(l2_[0], l2_[1]), fmaxf(l2_[2], l2_[3]));                                                 \
        { auto a_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx_), __float_as_uint(mx_), false, false);          \
          mx_ = fmaxf(__uint_as_float(a_[0]), __uint_as_float(a_[1])); }                                                 \
        P2H_MF(SLOT, II, 3)                                                                                               \
        { auto b_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx_), __float_as_uint(mx_), false, false);          \
          mx_ = fmaxf(__uint_as_float(b_[0]), __uint_as_float(b_[1])); }                                                 \
        const float mn_ = fmaxf(ml_.x, mx_);                                                                             \
        scn_ = __builtin_amdgcn_exp2f(ml_.x - mn_);                                                                      \
        P2H_MF(SLOT, II, 4)                                                                                               \
        f32x4 pvn_;                                                                                                      \
        pvn_[0] = C32_EXP2(l2_[0] - mn_); pvn_[1] = C32_EXP2(l2_[1] - mn_);                                              \
        pvn_[2] = C32_EXP2(l2_[2] - mn_); pvn_[3] = C32_EXP2(l2_[3] - mn_);                                              \
        P2H_MF(SLOT, II, 5)                                                                                               \
        float ps_ = (pvn_[0] + pvn_[1]) + (pvn_[2] + pvn_[3]);                                                           \
        { auto a_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(ps_), __float_as_uint(ps_), false, false);          \
          ps_ = __uint_as_float(a_[0]) + __uint_as_float(a_[1]); }                                                       \
        P2H_SPLIT()                                                                                                      \
        P2H_MF(SLOT, II, 6)                                                                                               \
        { auto b_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps_), __float_as_uint(ps_), false, false);          \
          ps_ = __uint_as_float(b_[0]) + __uint_as_float(b_[1]); }                                                       \
        const float ln_ = ml_.y * scn_ + ps_;                                                                            \
        *reinterpret_cast<f32x4*>(spp_) = pvn_;                              /* the C waves read P as fp32, as before */ \
        P2H_MF(SLOT, II, 7)                                                                                               \
        if (kq == 0) { scl[((CHPAR) * BI2 + il_) * SCLD + fm] = scn_; *reinterpret_cast<float2*>(mlw + ((II) + 1) * 32) = make_float2(mn_, ln_); } \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if ((__builtin_amdgcn_ballot_w64(scn_ != 1.f) & 0x0fff0fff0fff0fffull) != 0ull) {                                \
            accP[(II) + 1][0] *= scn_; accP[(II) + 1][1] *= scn_; accP[(II) + 1][2] *= scn_; accP[(II) + 1][3] *= scn_; } \
    }
#define P2H_POS_LAST(SLOT, II, CH)                                                                                        \
    {                                                                                                                    \
        const u32x4 bh_ = (u32x4){pkn_[0], pkn_[1], pkn_[0], pkn_[1]}; const u32x2 bl_ = (u32x2){pkn_[2], pkn_[3]};          \
        if ((II) + 3 == 8) P2_BOFF((CH) + 1)                                                                             \
        if ((II) + C32_RING - 1 == 8) P2_KOFF((CH) + 1)                                                                  \
        if (!(C32_ABL & 1)) {                                                                                            \
            P2_ISSUE_B(((II) + 3) & 3, ((II) + 3) & 7)                                                                   \
            P2_ISSUE_Z(((SLOT) + C32_RING - 1) % C32_RING, ((II) + C32_RING - 1) & 7)                                    \
        }                                                                                                                \
        P2H_MF(SLOT, II, 0) P2H_MF(SLOT, II, 1) P2H_MF(SLOT, II, 2) P2H_MF(SLOT, II, 3) P2H_MF(SLOT, II, 4) P2H_MF(SLOT, II, 5) P2H_MF(SLOT, II, 6) P2H_MF(SLOT, II, 7) \
    }
#define P2_SLOT(K, II) ((8 * (K) + (II)) % C32_RING)
#define P2_CHUNK(K, CH)                                                                                                  \
    {                                                                                                                    \
        const uint32_t mk4_ = *reinterpret_cast<const uint32_t*>(&mk[(CH) * JC + kq * 4]);                               \
        float mterm_[4];                                                    /* x k + (0 | -1e5 log2 e) in one fma: the values of the 16-row kernels' (mask ? x k : x k - 1e5 log2 e), whose second form the compiler contracts to the same fma */ \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) mterm_[r_] = ((mk4_ >> (8 * r_)) & 0xffu) ? 0.f : -kMask2;      \
        const int CHPAR = (CH) & 1;                                                                                      \
        f32x4 pvn_; float scn_;                                                                                                  \
        P2_SM(0, K)                                                                                                      \
        if (!(C32_ABL & 1024) && !((C32_LAZY & 1) && (__builtin_amdgcn_ballot_w64(scn_ != 1.f) & 0x0fff0fff0fff0fffull) == 0ull)) { _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) accP[0][mt_] *= scn_; } \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        P2_POS(P2_SLOT(K, 0), 0, CH, K) P2_POS(P2_SLOT(K, 1), 1, CH, K) P2_POS(P2_SLOT(K, 2), 2, CH, K) P2_POS(P2_SLOT(K, 3), 3, CH, K) \
        P2_POS(P2_SLOT(K, 4), 4, CH, K) P2_POS(P2_SLOT(K, 5), 5, CH, K) P2_POS(P2_SLOT(K, 6), 6, CH, K) P2_POS_LAST(P2_SLOT(K, 7), 7, CH) \
        C32_SYNC()                                                          /* barrier #(CH + 1) */                      \
    }
#define P2H_CHUNK(K, CH)                                                                                                  \
    {                                                                                                                    \
        const uint32_t mk4_ = *reinterpret_cast<const uint32_t*>(&mk[(CH) * JC + kq * 4]);                               \
        float mterm_[4];                                                    /* x k + (0 | -1e5 log2 e) in one fma: the values of the 16-row kernels' (mask ? x k : x k - 1e5 log2 e), whose second form the compiler contracts to the same fma */ \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) mterm_[r_] = ((mk4_ >> (8 * r_)) & 0xffu) ? 0.f : -kMask2;      \
        const int CHPAR = (CH) & 1;                                                                                      \
        u32x4 pkn_; float scn_;                                                                                                  \
        P2H_SM(0, K)                                                                                                      \
        if (!(C32_ABL & 1024) && !((C32_LAZY & 1) && (__builtin_amdgcn_ballot_w64(scn_ != 1.f) & 0x0fff0fff0fff0fffull) == 0ull)) { _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) accP[0][mt_] *= scn_; } \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        P2H_POS(P2_SLOT(K, 0), 0, CH, K) P2H_POS(P2_SLOT(K, 1), 1, CH, K) P2H_POS(P2_SLOT(K, 2), 2, CH, K) P2H_POS(P2_SLOT(K, 3), 3, CH, K) \
        P2H_POS(P2_SLOT(K, 4), 4, CH, K) P2H_POS(P2_SLOT(K, 5), 5, CH, K) P2H_POS(P2_SLOT(K, 6), 6, CH, K) P2H_POS_LAST(P2_SLOT(K, 7), 7, CH) \
        C32_SYNC()                                                          /* barrier #(CH + 1) */                      \
    }
        int ch = 0;
        if constexpr (ZT) {
        for (; ch + 3 <= nchunk; ch += 3) { P2H_CHUNK(0, ch) P2H_CHUNK(1, ch + 1) P2H_CHUNK(2, ch + 2) }
        if (ch < nchunk) {
            P2H_CHUNK(0, ch)
            if (ch + 1 < nchunk) P2H_CHUNK(1, ch + 1)
        }
        } else {
        for (; ch + 3 <= nchunk; ch += 3) { P2_CHUNK(0, ch) P2_CHUNK(1, ch + 1) P2_CHUNK(2, ch + 2) }
        if (ch < nchunk) {
            P2_CHUNK(0, ch)
            if (ch + 1 < nchunk) P2_CHUNK(1, ch + 1)
        }
        }
#undef P2H_CHUNK
#undef P2H_POS
#undef P2H_POS_LAST
#undef P2H_MF
#undef P2H_SM
#undef P2H_SPLIT
#undef P2_CHUNK
#undef P2_SLOT
#undef P2_POS
#undef P2_POS_LAST
#undef P2_MF
#undef P2_SM
#undef P2_ISSUE_Z
#undef P2_ISSUE_B
#undef P2_KOFF
#undef P2_BOFF
        // alpha = P / l, zero for masked queries (ga.py:24-25); pair features out
        if constexpr (!FUSE) {
#pragma unroll
        for (int ii = 0; ii < RPW2; ++ii) {
            const int il = il0 + ii, i = i0 + il;
            const float l_fin = mlw[ii * 32 + 1];                           // written by this lane column's kq == 0 lane of this very wave
            if (kq == 0) lsum[il * SCLD + fm] = l_fin;
            if (i < L && fm < H) {
                const bool mi = mk[i] != 0;
                const float inv = mi ? 1.f / l_fin : 0.f;
                float* fo = feat + (rowbase + i) * FEAT + fm * C + kq * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (ZT) {                                     // 1 / S_ic of channels 16 kq + 4 r .. + 3 (powers of two: the products with inv are exact)
                        const f32x4 zs = *reinterpret_cast<const f32x4*>(zsc + (zbase + i) * C + kq * 16 + r * 4);
                        reinterpret_cast<f32x4*>(fo)[r] = (f32x4){mul_rn(accP[ii][0][r], inv * zs[0]), mul_rn(accP[ii][1][r], inv * zs[1]), mul_rn(accP[ii][2][r], inv * zs[2]), mul_rn(accP[ii][3][r], inv * zs[3])};
                    } else
                    reinterpret_cast<f32x4*>(fo)[r] = (f32x4){accP[ii][0][r] * inv, accP[ii][1][r] * inv, accP[ii][2][r] * inv, accP[ii][3][r] * inv};
                }
            }
        }
        C32_REPORT(0)
        __syncthreads();                                                    // F1: lsum visible, C waves done with the last chunk
        __syncthreads();                                                    // F2: aggregated points in LDS
        } else {
        // ---- fused tail, producer role.  Lane (head fm, key group kq) holds, for row ii and r = 0..3, the four consecutive feature columns
        // 64 fm + 16 kq + 4 r .. + 3 (accP[ii][0..3][r]): chunk r of the tail's column order (tail_common.h: ot_feat_col) takes exactly that
        // entry from every lane.  Normalisation and the split into bf16 terms happen chunk by chunk, in the interval before the consumers
        // need it (splitting everything up front put 800 VALU operations of this wave next to the C waves' dump on the same SIMD).
#pragma unroll
        for (int ii = 0; ii < RPW2; ++ii)
            if (kq == 0) lsum[(il0 + ii) * SCLD + fm] = mlw[ii * 32 + 1];
        C32_REPORT(0)
        C32F_STAMP(0)
        __syncthreads();                                                    // F1: lsum visible, every wave is done with the S/P tile (mlr and the key mask are not aliased)
        float rinv[RPW2];
#pragma unroll
        for (int ii = 0; ii < RPW2; ++ii) {
            const int i = i0 + il0 + ii;
            const bool mi = (i < L) && mk[min(i, L - 1)] != 0;
            rinv[ii] = mi ? 1.f / mlw[ii * 32 + 1] : 0.f;
        }
        // this thread's share of the point chunks: row prow, eighth psub of the columns
        const int ptid = wave * 64 + lane, prow = ptid >> 3, psub = ptid & 7;
        float rr_[9], tt_[3];
        {
            const int64_t grow = rowbase + min(i0 + prow, L - 1);
#pragma unroll
            for (int q = 0; q < 9; ++q) rr_[q] = R[grow * 9 + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) tt_[q] = t[grow * 3 + q];
        }
        C32F_STAMP(1)
        __syncthreads();                                                    // E0
        C32F_STAMP(2)
        // local coordinates / distance / direction of aggregated point number pt of row prow (point_local: the arithmetic of the unfused epilogues)
        auto local_of = [&](int pt, float& lx, float& ly, float& lz, float& d, float& inv) {
            const float* a = ptsf + prow * C32F_PTSLD + pt * 3;
            point_local(a[0] - tt_[0], a[1] - tt_[1], a[2] - tt_[2], rr_[0], rr_[1], rr_[2], rr_[3], rr_[4], rr_[5], rr_[6], rr_[7], rr_[8], lx, ly, lz, d, inv);
        };
        // npt consecutive points starting at pt0 -> 3 npt consecutive columns starting at col0 of staging buffer b: coordinates (DIR = false) or directions
        auto put_points = [&](char* buf, int col0, int pt0, int npt, bool dir) {
            float v[24];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < npt) {
                    float lx, ly, lz, d, inv;
                    local_of(pt0 + k, lx, ly, lz, d, inv);
                    v[3 * k] = dir ? mul_rn(lx, inv) : lx; v[3 * k + 1] = dir ? mul_rn(ly, inv) : ly; v[3 * k + 2] = dir ? mul_rn(lz, inv) : lz;
                }
            }
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (4 * q < 3 * npt) stage_put4(buf, prow, col0 + 4 * q, v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        };
#pragma unroll
        for (int p = 0; p < OT_NCH; ++p) {
            // interval p: the consumers read buffer p & 1 (position p); this role fills buffer (p + 1) & 1 with position p + 1
            // (positions 0 and 1 -- the node features -- were written by the C waves before E0)
            if (p >= 1 && p + 1 < OT_NCH && !(C32F_ABL & 4)) {
                char* buf = stage + ((p + 1) & 1) * OT_STAGE;
                const int c = ot_chunk_at(p + 1);
                if (c < 4) {                                                // pair features, channels 16 kq + 4 c + 0..3 of every head (ot_feat_col)
                    if (fm < H) {
#pragma unroll
                        for (int ii = 0; ii < RPW2; ++ii) {
                            if constexpr (ZT) {                                         // 1 / S_ic of channels 16 kq + 4 c .. + 3 of row ii (powers of two: the products with rinv are exact)
                                const f32x4 zs = *reinterpret_cast<const f32x4*>(zsc + (zbase + min(i0 + il0 + ii, L - 1)) * C + kq * 16 + c * 4);
                                stage_put4(buf, il0 + ii, fm * 16 + kq * 4, mul_rn(accP[ii][0][c], rinv[ii] * zs[0]), mul_rn(accP[ii][1][c], rinv[ii] * zs[1]),
                                           mul_rn(accP[ii][2][c], rinv[ii] * zs[2]), mul_rn(accP[ii][3][c], rinv[ii] * zs[3]));
                            } else
                            stage_put4(buf, il0 + ii, fm * 16 + kq * 4, mul_rn(accP[ii][0][c], rinv[ii]), mul_rn(accP[ii][1][c], rinv[ii]),
                                       mul_rn(accP[ii][2][c], rinv[ii]), mul_rn(accP[ii][3][c], rinv[ii]));
                        }
                    }
                } else if (c == 6) {                                        // columns 1152..1343: coordinates of points 0..63
                    put_points(buf, psub * 24, psub * 8, 8, false);
                } else if (c == 7) {                                        // 1344..1439: coordinates of points 64..95 | 1440..1535: the 96 distances
                    put_points(buf, psub * 12, 64 + psub * 4, 4, false);
                    float dd[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k) { float lx, ly, lz, inv; local_of(psub * 12 + k, lx, ly, lz, dd[k], inv); }
#pragma unroll
                    for (int q = 0; q < 3; ++q) stage_put4(buf, prow, 96 + psub * 12 + 4 * q, dd[4 * q], dd[4 * q + 1], dd[4 * q + 2], dd[4 * q + 3]);
                } else if (c == 8) {                                        // 1536..1727: directions of points 0..63
                    put_points(buf, psub * 24, psub * 8, 8, true);
                } else {                                                    // 1728..1823: directions of points 64..95
                    put_points(buf, psub * 12, 64 + psub * 4, 4, true);
                }
            }
            C32F_STAMP(3 + p)
            __syncthreads();                                                // E(p + 1)
        }
        C32F_STAMP(13)
        }
    } else if (wave < NPW2 + 2) {
        // =========================================================================================== A waves: S(t + 1), 6 heads x 2 row tiles
        const int h0 = (wave - NPW2) * HPW;
        const f32x4* kvn = reinterpret_cast<const f32x4*>(kvfrag) + (int64_t)n * nchunk * H * 512;
        f32x4 qr[HPW][2][4];                                                // q' of this wave's heads, both row tiles: stays in registers
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4* qg = reinterpret_cast<const f32x4*>(qfrag) + (((int64_t)n * nib16 + min(2 * ib + rt, nib16 - 1)) * H + h0 + hh) * (4 * 64) + lane;
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) qr[hh][rt][s_] = qg[s_ * 64];
            }
        f32x4 kf[2][4];
#define A2_ISSUE(B, HH, CH) { const f32x4* fr_ = kvn + ((int64_t)min((CH), nchunk - 1) * H + h0 + (HH)) * 512 + lane;   \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) kf[B][s_] = fr_[s_ * 64]; }
        auto produce = [&](int c, int buf) {                                // S(c) -> sp[buf]; the key fragments of the NEXT head are requested first
#pragma unroll
            for (int hh = 0; hh < HPW; ++hh) {
                const int h = h0 + hh;
                if (!(C32_ABL & 2)) { if (hh + 1 < HPW) { if (hh & 1) A2_ISSUE(0, hh + 1, c) else A2_ISSUE(1, hh + 1, c) } else A2_ISSUE(0, 0, c + 1) }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const f32x4 q0 = qr[hh][rt][0], q1 = qr[hh][rt][1], q2 = qr[hh][rt][2], q3 = qr[hh][rt][3];
                    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                    if (C32_ABL & 8) { acc0 = kf[hh & 1][0] * q0 + kf[hh & 1][1] * q1; acc1 = kf[hh & 1][2] * q2 + kf[hh & 1][3] * q3; }
                    else if (C32_ABL & 256) { acc0 = kf[hh & 1][0]; acc1 = q0; }
                    else if constexpr (ZT) {
                        // the 32 channels of q / sqrt(D) and k arrive as two fp16 terms each (node_frags, qk_terms: slot 0 = high, slot 1 = low terms): three 16-cycle
                        // products, smallest first, instead of eight 32-cycle fp32 steps; the point part + norm step (squared distances of global coordinates
                        // cancel there: 24 bits needed) stays on the fp32 chain
                        const u32x4 kh = __builtin_bit_cast(u32x4, kf[hh & 1][0]), kl = __builtin_bit_cast(u32x4, kf[hh & 1][1]);
                        const u32x4 qh = __builtin_bit_cast(u32x4, q0), ql = __builtin_bit_cast(u32x4, q1);
                        acc0 = mfma_h(kl, qh, acc0); acc0 = mfma_h(kh, ql, acc0);
#pragma unroll
                        for (int s = 0; s < 4; ++s) { acc1 = mfma4(kf[hh & 1][2][s], q2[s], acc1); if (s == 1) acc0 = mfma_h(kh, qh, acc0); }
#pragma unroll
                        for (int s = 0; s < 3; ++s) acc1 = mfma4(kf[hh & 1][3][s], q3[s], acc1);
                    }
                    else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) { acc0 = mfma4(kf[hh & 1][0][s], q0[s], acc0); acc1 = mfma4(kf[hh & 1][2][s], q2[s], acc1); }
#pragma unroll
                    for (int s = 0; s < 4; ++s) { acc0 = mfma4(kf[hh & 1][1][s], q1[s], acc0); if (s < 3) acc1 = mfma4(kf[hh & 1][3][s], q3[s], acc1); }
                    }
                    const f32x4 sres = acc0 + acc1;
                    *reinterpret_cast<f32x4*>(sp + (buf * BI2 + rt * 16 + fm) * SROW + sp_off(h, kq)) = sres;
                }
                __builtin_amdgcn_sched_barrier(0);                          // keep the per-head order
            }
        };
        A2_ISSUE(0, 0, 0)
        fill_mask();
        __syncthreads();
        produce(0, 0);
        __syncthreads();                                                    // barrier #0
        int buf = 1;
        for (int c = 1; c < nchunk; ++c) {
            produce(c, buf);
            buf = (buf == 2) ? 0 : buf + 1;
            C32_SYNC()                                                      // barrier #c
        }
        C32_SYNC()                                                          // barrier #nchunk
        C32_REPORT(1)
        if constexpr (!FUSE) {
        __syncthreads();                                                    // F1
        __syncthreads();                                                    // F2
        } else {
        WT raw[RD];
        consumer_prefetch(wave - NPW2, raw);                                // the first W_out fragments travel while the other roles finish
        C32F_STAMP(0)
        __syncthreads();                                                    // F1
        tail_p2_stage_bias(reinterpret_cast<float (*)[F]>(smem_raw + C32F_BIAS_OFF), ta.b0, ta.b1, ta.b2, tid - NPW2 * 64);   // these two waves idle until E0
        consumer(wave - NPW2, raw);                                         // E0 .. E10 inside
        }
#undef A2_ISSUE
    } else {
        // =========================================================================================== C waves: aggregation of chunk t - 1, 6 heads x 2 row tiles
        const int h0 = (wave - NPW2 - 2) * HPW;
        const f32x4* kvn = reinterpret_cast<const f32x4*>(kvfrag) + (int64_t)n * nchunk * H * 512;
        f32x4 vf[2][4];
        f32x4 accV[HPW][2][2], accT[HPW][2][2];
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int k = 0; k < 2; ++k) { accV[hh][rt][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; accT[hh][rt][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#define C2_ISSUE(B, HH, CH) { const f32x4* fr_ = kvn + ((int64_t)min((CH), nchunk - 1) * H + h0 + (HH)) * 512 + 4 * 64 + lane; \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) vf[B][s_] = fr_[s_ * 64]; }
        // C32_CPIPE (developer switch, round 5): the P rows and rescale factors of head hh + 1 are requested from LDS before the products of head hh
        // (the chunk's tile is complete behind the barrier), so a head no longer opens with an exposed LDS round trip
        auto consume = [&](int c, int buf) {
            const int par = c & 1;
            unsigned long long anych_ = 0ull;
            float scn_[2]; f32x4 pan_[2];
#define C2_LDS(HH) { _Pragma("unroll") for (int rt_ = 0; rt_ < (C32_CPIPE == 2 ? 1 : 2); ++rt_) {                       \
                scn_[rt_] = scl[(par * BI2 + rt_ * 16 + fm) * SCLD + h0 + (HH)];                                         \
                pan_[rt_] = *reinterpret_cast<const f32x4*>(sp + (buf * BI2 + rt_ * 16 + fm) * SROW + sp_off(h0 + (HH), kq)); } }
            if (C32_CPIPE) C2_LDS(0)
#pragma unroll
            for (int hh = 0; hh < HPW; ++hh) {
                const int h = h0 + hh;
                if (!(C32_ABL & 2)) { if (hh + 1 < HPW) { if (hh & 1) C2_ISSUE(0, hh + 1, c) else C2_ISSUE(1, hh + 1, c) } else C2_ISSUE(0, 0, c + 1) }
                float scc_[2]; f32x4 pac_[2];
                if (C32_CPIPE) {
                    scc_[0] = scn_[0]; pac_[0] = pan_[0];
                    if (C32_CPIPE == 2) { scc_[1] = scl[(par * BI2 + 16 + fm) * SCLD + h]; pac_[1] = *reinterpret_cast<const f32x4*>(sp + (buf * BI2 + 16 + fm) * SROW + sp_off(h, kq)); }
                    else { scc_[1] = scn_[1]; pac_[1] = pan_[1]; }
                    if (hh + 1 < HPW) C2_LDS(hh + 1)
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const float sc = C32_CPIPE ? scc_[rt] : scl[(par * BI2 + rt * 16 + fm) * SCLD + h];
                    const f32x4 pa = C32_CPIPE ? pac_[rt] : *reinterpret_cast<const f32x4*>(sp + (buf * BI2 + rt * 16 + fm) * SROW + sp_off(h, kq));
                    C32_CNT(2, 1) C32_CNT(3, __builtin_amdgcn_ballot_w64(sc != 1.f) == 0ull)
                    anych_ |= __builtin_amdgcn_ballot_w64(sc != 1.f);
                    // (all 16 rows of the tile kept their running maximum of this head: factors exactly 1, nothing to rescale)
                    if (!(C32_ABL & 1024) && !((C32_LAZY & 2) && FUSE && __builtin_amdgcn_ballot_w64(sc != 1.f) == 0ull)) { accV[hh][rt][0] *= sc; accV[hh][rt][1] *= sc; accT[hh][rt][0] *= sc; accT[hh][rt][1] *= sc; }
                    if (C32_ABL & 512) { accV[hh][rt][0] += pa; }
                    else if (C32_ABL & 16) { accV[hh][rt][0] += vf[hh & 1][0] * pa; accV[hh][rt][1] += vf[hh & 1][1] * pa; accT[hh][rt][0] += vf[hh & 1][2] * pa; accT[hh][rt][1] += vf[hh & 1][3] * pa; }
                    else
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        accV[hh][rt][0] = mfma4(vf[hh & 1][s][0], pa[s], accV[hh][rt][0]);
                        accV[hh][rt][1] = mfma4(vf[hh & 1][s][1], pa[s], accV[hh][rt][1]);
                        accT[hh][rt][0] = mfma4(vf[hh & 1][s][2], pa[s], accT[hh][rt][0]);
                        accT[hh][rt][1] = mfma4(vf[hh & 1][s][3], pa[s], accT[hh][rt][1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef C2_LDS
            C32_CNT(4, 1) C32_CNT(5, anych_ == 0ull)
            (void)anych_;
        };
        C2_ISSUE(0, 0, 0)
        fill_mask();
        __syncthreads();
        __syncthreads();                                                    // barrier #0
        C32_SYNC()                                                          // barrier #1: P(0) ready
        int buf = 0;
        for (int c = 1; c < nchunk; ++c) {
            consume(c - 1, buf);
            buf = (buf == 2) ? 0 : buf + 1;
            C32_SYNC()                                                      // barrier #(c + 1)
        }
        consume(nchunk - 1, buf);
        C32_REPORT(2)
        WT raw[RD];
        C32F_STAMP(0)
        __syncthreads();                                                    // F1
        float* pts = FUSE ? ptsf : sp;                                      // [BI2][H][24] aggregated global-frame points; the S/P tile is free now
        char* nbuf = stage + (wave - NPW2 - 2) * OT_STAGE;                  // FUSE: heads 0..5 are chunk 4 = position 0, heads 6..11 chunk 5 = position 1
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
            const int h = h0 + hh;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int il = rt * 16 + fm, i = i0 + il;
                const bool mi = (i < L) && mk[min(i, L - 1)] != 0;
                const float inv = mi ? 1.f / lsum[il * SCLD + h] : 0.f;
                if constexpr (FUSE) {
                    const f32x4 a0 = accV[hh][rt][0], a1 = accV[hh][rt][1];
                    if (!(C32F_ABL & 8)) {
                    stage_put4(nbuf, il, hh * D + kq * 4, mul_rn(a0[0], inv), mul_rn(a0[1], inv), mul_rn(a0[2], inv), mul_rn(a0[3], inv));
                    stage_put4(nbuf, il, hh * D + 16 + kq * 4, mul_rn(a1[0], inv), mul_rn(a1[1], inv), mul_rn(a1[2], inv), mul_rn(a1[3], inv));
                    }
                } else if (i < L) {
                    float* fo = feat + (rowbase + i) * FEAT + H * C + h * D + kq * 4;
                    *reinterpret_cast<f32x4*>(fo) = accV[hh][rt][0] * inv;
                    *reinterpret_cast<f32x4*>(fo + 16) = accV[hh][rt][1] * inv;
                }
                float* po = pts + il * (FUSE ? C32F_PTSLD : H * P * 3) + h * (P * 3) + kq * 3;
                if (!(FUSE && (C32F_ABL & 16))) {
#pragma unroll
                for (int r = 0; r < 3; ++r) { po[r] = accT[hh][rt][0][r] * inv; po[12 + r] = accT[hh][rt][1][r] * inv; }
                }
            }
        }
        if constexpr (FUSE) {
            consumer_prefetch(wave - NPW2, raw);                            // (only now: 192 accumulators + 96 fragment registers do not fit)
            consumer(wave - NPW2, raw);                                     // E0 .. E10 inside
        }
        else __syncthreads();                                               // F2
#undef C2_ISSUE
    }
    if constexpr (!FUSE) {
        // ---------------------------------------------------------------- all waves: local frame, norm, direction of the aggregated points (ga.py:136-139)
        persist_point_epilogue(sp, R, t, feat, rowbase, i0, L, tid, NTH2, BI2);
    } else {
        // ---------------------------------------------------------------- all waves: LayerNorm1, mlp_transition, LayerNorm2 of the 32 rows (tail_common.h)
        const int64_t row0 = row0f, row_end = row_endf;
        pre = tail_p2_prefetch<NTH2 / 64>(ta.x, ta.ubias, mask, ta.g1, ta.be1, ta.wmf, row0, row_end, wave, lane);
        float (*bias)[F] = reinterpret_cast<float (*)[F]>(smem_raw + C32F_BIAS_OFF);
        C32F_STAMP(14)
        __syncthreads();                                                    // U: u and the biases are in LDS
        const float (*us)[XLD] = reinterpret_cast<const float (*)[XLD]>(smem_raw + C32F_U_OFF);
        auto get_u = [&](int rl) { return *reinterpret_cast<const float2*>(&us[rl][2 * lane]); };
        tail_p2_run<NTH2 / 64, false>(pre, get_u, reinterpret_cast<float (*)[XLD]>(smem_raw + C32F_YS_OFF), bias, smem_raw + C32F_APA_OFF,
                                      smem_raw + C32F_PTS_OFF, ta.wmf, ta.g2, ta.be2, ta.out, nullptr, 0, row0, row_end, wave, lane, ta.xt);
        C32F_STAMP(15)
    }
    if (blockIdx.x == 0 && tid == 0) { g_clock_probe[0] = clock64() - probe_c0; g_clock_probe[1] = wall_clock64() - probe_w0; }
    if (prof_slot >= 0 && tid == 0) atomicMax(&g_prof_span[prof_slot][1], (unsigned long long)wall_clock64());
}

// Pair-bias cache: lp[l][n,i,j,h] = z[n,i,j,:] . Wb_l[h,:] for every layer l in ONE pass over z (ga.py:88-90).  z and the weights do
// not change during the 100 steps of FullDPM.sample, so the sampler builds this once per call and the per-step kernel reads 48
// bytes per (i,j) instead of spending 64 more MFMAs per (row, chunk) and an LDS transpose on it.
// Layout per layer: [N*L (query row)][nchunk][12 (head)][16 (key in chunk)] -- the float4 (head, 4 keys) an A wave adds to its
// S tile.  Same MFMA chain order as the in-kernel path => bit-identical logits.
struct WbList { const float* w[8]; };

__global__ __launch_bounds__(256) void pair_bias_cache_kernel(const float* __restrict__ z, WbList wl, int num_layers, float* __restrict__ cache,
                                                              int64_t rows, int L, int nchunk) {
    __shared__ __attribute__((aligned(16))) float zst[4][JC][ZSLD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fm = lane & 15, kq = lane >> 4;
    // a wave walks the chunks q, q + 4, q + 8, ... of one query row with the next chunk's z already requested (one chunk per wave and no
    // prefetch ran at 2.7 TB/s: load -> transpose -> 96 MFMAs -> store, every round trip exposed)
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;                  // (query row, chunk phase q)
    if (unit >= rows * 4) return;
    const int64_t row = unit >> 2;
    const int q0 = (int)(unit & 3);
    if (q0 >= nchunk) return;
    const float* zi = z + (row * (int64_t)L) * C;
    f32x4 wv[6][4];                                                        // this lane's weights of (up to) six layers: head fm, channels 16 kq + 4 q ..
#pragma unroll
    for (int l = 0; l < 6; ++l)
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[l][q] = (l < num_layers && fm < H) ? reinterpret_cast<const f32x4*>(wl.w[l] + fm * C + kq * 16)[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 zn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) zn[r] = *(reinterpret_cast<const f32x4*>(zi + (int64_t)min(q0 * JC + kq * 4 + r, L - 1) * C) + fm);
    for (int ch = q0; ch < nchunk; ch += 4) {
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(&zst[wave][kq * 4 + r][fm * 4]) = zn[r];
        if (ch + 4 < nchunk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) zn[r] = *(reinterpret_cast<const f32x4*>(zi + (int64_t)min((ch + 4) * JC + kq * 4 + r, L - 1) * C) + fm);
        }
        wave_lds_sync();                                                  // cross-lane transpose through LDS
        f32x4 za[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) za[q] = *reinterpret_cast<const f32x4*>(&zst[wave][fm][kq * 16 + q * 4]);
        const int64_t u = PBC_CHUNK_MAJOR ? (int64_t)ch * rows + row : row * nchunk + ch;
        auto layer = [&](int l, const f32x4 (&w4)[4]) {
            f32x4 acc4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc4[q] = mfma4(za[q][0], w4[q][0], (f32x4){0.f, 0.f, 0.f, 0.f});
                acc4[q] = mfma4(za[q][1], w4[q][1], acc4[q]); acc4[q] = mfma4(za[q][2], w4[q][2], acc4[q]); acc4[q] = mfma4(za[q][3], w4[q][3], acc4[q]);
            }
            const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);   // accumulator row 4 kq + r = key, column fm = head
            if (fm < H) *reinterpret_cast<f32x4*>(cache + ((int64_t)l * rows * nchunk + u) * (H * JC) + fm * JC + kq * 4) = acc;
        };
#pragma unroll
        for (int l = 0; l < 6; ++l)
            if (l < num_layers) layer(l, wv[l]);
        for (int l = 6; l < num_layers; ++l) {                            // (more than six blocks: weights from L1 / L2 per chunk)
            f32x4 w4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[q] = (fm < H) ? reinterpret_cast<const f32x4*>(wl.w[l] + fm * C + kq * 16)[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
            layer(l, w4);
        }
    }
}

size_t pair_bias_layer_floats(int N, int L) { return (size_t)N * L * ((L + JC - 1) / JC) * (H * JC); }

int launch_pair_bias_cache(const float* z, const float* const* wb, int num_layers, float* cache, int N, int L, hipStream_t st) {
    ABOPT_CHECK_ARG(num_layers >= 1 && num_layers <= 8, "pair_bias_cache: 1..8 layers supported (got %d)", num_layers);
    WbList wl;
    for (int l = 0; l < 8; ++l) wl.w[l] = l < num_layers ? wb[l] : nullptr;
    const int nchunk = (L + JC - 1) / JC;
    const int64_t units = (int64_t)N * L * nchunk;
    if (units == 0) return ABOPT_OK;
    (void)units;
    hipLaunchKernelGGL(pair_bias_cache_kernel, dim3((unsigned)((int64_t)N * L)), dim3(256), 0, st, z, wl, num_layers, cache, (int64_t)N * L, L, nchunk);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// Pair terms (round 6): pair_feat as the K-packed two-term fp16 operands of ipa_core32_kernel<*, true> -- built once per sample() / optimize() call, like the
// bias cache (z is constant over the steps, dpm_full.py:274-283).  One workgroup per query row (n, i): pass A folds max_j |z[n, i, j, c]| per CHANNEL (the
// row's 64 KB come from HBM once; pass B re-reads them from L2), S_ic = 2^k with max |z| S_ic in [2^13, 2^14) -- every value within 2^-17 of the largest of
// its (row, channel) column keeps a NORMAL low term, i.e. 22 significant bits, whatever the magnitudes of other channels or rows; pass B writes, per chunk of
// 16 keys and channel tile mt, lane (fm, kq)'s 16 bytes
//     {h(k0), h(k1) | h(k2), h(k3) | l(k0), l(k1) | l(k2), l(k3)},  k_e = key 16 ch + 4 kq + e, of channel c = 4 fm + mt;  h = fp16(S z), l = fp16(S z - h)
// at float offset ((ch ROWS + row) 4 + mt) 256 + 4 lane with ROWS = N L rows of the whole batch (chunk-major: ZT_CHUNK_MAJOR; keys past L: zeros), and the 64 factors 2^-14 / S_ic of every row (the consumer multiplies its probabilities by 2^14) behind the terms.  Same bytes as z (+ 1.6 %).
size_t pair_terms_floats(int Nz, int L) { return (size_t)Nz * L * ((L + JC - 1) / JC) * (JC * C); }
size_t pair_terms_blob_floats(int Nz, int L) { return pair_terms_floats(Nz, L) + (size_t)Nz * L * C; }

__global__ __launch_bounds__(256) void pair_terms_kernel(const float* __restrict__ z, float* __restrict__ terms, float* __restrict__ zsc, int L, int nchunk) {
    __shared__ float wmax[4][C];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fm = lane & 15, kq = lane >> 4;
    const int64_t row = blockIdx.x;
    const float* zi = z + (row * (int64_t)L) * C;
    auto load = [&](int ch, f32x4 (&zn)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = ch * JC + kq * 4 + r;
            zn[r] = key < L ? *(reinterpret_cast<const f32x4*>(zi + (int64_t)key * C) + fm) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 am = (f32x4){0.f, 0.f, 0.f, 0.f};                                  // channels 4 fm .. 4 fm + 3
    for (int ch = wave; ch < nchunk; ch += 4) {
        f32x4 zn[4];
        load(ch, zn);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) am[e] = fmaxf(am[e], fabsf(zn[r][e]));        // (a NaN is dropped here and travels through the split below)
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { am[e] = fmaxf(am[e], __shfl_xor(am[e], 16)); am[e] = fmaxf(am[e], __shfl_xor(am[e], 32)); }
    if (kq == 0) *reinterpret_cast<f32x4*>(&wmax[wave][fm * 4]) = am;
    __syncthreads();
    f32x4 S;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float m = fmaxf(fmaxf(wmax[0][fm * 4 + e], wmax[1][fm * 4 + e]), fmaxf(wmax[2][fm * 4 + e], wmax[3][fm * 4 + e]));
        // S = 2^(13 - floor(log2 m)); exponent field clamped so that S, 1 / S and the products with the softmax normalisation stay normal numbers whatever the column holds
        const int ex = min(max((int)((__float_as_uint(m) >> 23) & 0xffu), 64), 190);
        S[e] = __uint_as_float((unsigned)(267 - ex) << 23);
        if (wave == 0 && kq == 0) zsc[row * C + fm * 4 + e] = __uint_as_float((unsigned)(ex - 27) << 23);     // 2^-14 / S: the consumer's probabilities carry 2^14
    }
    const int64_t smp = row / L, ri = row % L;                               // (ZT_CHUNK_MAJOR: [sample][chunk][row of the sample], 256 vectors per (row, chunk))
    u32x4* out = reinterpret_cast<u32x4*>(terms) + (ZT_CHUNK_MAJOR == 2 ? row * 256 : (ZT_CHUNK_MAJOR ? smp * (int64_t)L * nchunk * 256 + ri * 256 : row * (int64_t)nchunk * 256)) + lane;
    for (int ch = wave; ch < nchunk; ch += 4) {
        f32x4 zn[4];
        load(ch, zn);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            unsigned h01, l01, h23, l23;
            split_pair2(zn[0][mt] * S[mt], zn[1][mt] * S[mt], h01, l01);
            split_pair2(zn[2][mt] * S[mt], zn[3][mt] * S[mt], h23, l23);
            out[(ZT_CHUNK_MAJOR == 2 ? (int64_t)ch * gridDim.x * 256 : (ZT_CHUNK_MAJOR ? (int64_t)ch * L * 256 : (int64_t)ch * 256)) + mt * 64] = (u32x4){h01, h23, l01, l23};
        }
    }
}

int launch_pair_terms(const float* z, float* blob, int Nz, int L, hipStream_t st) {
    if ((int64_t)Nz * L == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(pair_terms_floats(Nz, L) * sizeof(float) < ((size_t)1 << 32), "pair_terms: %d x %d x %d values exceed the 4 GB a 32-bit buffer offset reaches (the consuming kernels address the whole "
                    "batch's terms chunk-major through one descriptor)", Nz, L, L);
    const int nchunk = (L + JC - 1) / JC;
    hipLaunchKernelGGL(pair_terms_kernel, dim3((unsigned)((int64_t)Nz * L)), dim3(256), 0, st, z, blob, blob + pair_terms_floats(Nz, L), L, nchunk);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// Softmax merge of the key slices of a SPLIT launch + the point epilogue (ga.py:133-139): one workgroup per query row.
//   w_s[h] = 2^(m_s - M) / sum_s' l_s' 2^(m_s' - M),  M = max_s m_s   (0 for a masked query row, ga.py:24-25)
__global__ __launch_bounds__(256) void ipa_split_merge_kernel(const float* __restrict__ part, const float* __restrict__ pstats, const uint8_t* __restrict__ mask,
                                                              const float* __restrict__ R, const float* __restrict__ t, float* __restrict__ feat,
                                                              int64_t rows, int nsplit) {
    __shared__ float w[4][16];
    __shared__ float pts[H * P * 3];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < H) {
        float m[4], l[4], mx = -INFINITY;
        for (int s_ = 0; s_ < nsplit; ++s_) {
            const float2 ml = *reinterpret_cast<const float2*>(pstats + (((int64_t)s_ * rows + row) * H + tid) * 2);
            m[s_] = ml.x; l[s_] = ml.y; mx = fmaxf(mx, ml.x);
        }
        float den = 0.f;
        for (int s_ = 0; s_ < nsplit; ++s_) { m[s_] = __builtin_amdgcn_exp2f(m[s_] - mx); den += l[s_] * m[s_]; }
        const float inv = mask[row] ? 1.f / den : 0.f;
        for (int s_ = 0; s_ < nsplit; ++s_) w[s_][tid] = m[s_] * inv;
    }
    __syncthreads();
    float* fo = feat + row * FEAT;
    for (int idx = tid; idx < SPLIT_ROW; idx += 256) {
        const int h = idx < H * C ? idx / C : (idx < H * C + H * D ? (idx - H * C) / D : (idx - H * C - H * D) / (P * 3));
        float v = 0.f;
        for (int s_ = 0; s_ < nsplit; ++s_) v += part[((int64_t)s_ * rows + row) * SPLIT_ROW + idx] * w[s_][h];
        if (idx < H * C + H * D) fo[idx] = v; else pts[idx - H * C - H * D] = v;
    }
    __syncthreads();
    if (tid < H * P) {
        const float* Rr = R + row * 9;
        const float* tr = t + row * 3;
        const float dx = pts[3 * tid] - tr[0], dy = pts[3 * tid + 1] - tr[1], dz = pts[3 * tid + 2] - tr[2];
        float lx, ly, lz, d, inv;
        point_local(dx, dy, dz, Rr[0], Rr[1], Rr[2], Rr[3], Rr[4], Rr[5], Rr[6], Rr[7], Rr[8], lx, ly, lz, d, inv);
        float* fp = fo + H * C + H * D;
        fp[3 * tid] = lx; fp[3 * tid + 1] = ly; fp[3 * tid + 2] = lz;
        fp[H * P * 3 + tid] = d;
        fp[H * P * 3 + H * P + 3 * tid] = lx * inv; fp[H * P * 3 + H * P + 3 * tid + 1] = ly * inv; fp[H * P * 3 + H * P + 3 * tid + 2] = lz * inv;
    }
}

size_t ipa_split_ws_floats(int N, int L) {
    const int nib = (L + BI - 1) / BI;
    return ((int64_t)N * nib * 2 <= 256) ? (size_t)4 * N * L * (SPLIT_ROW + 2 * H) : 0;
}

template <bool DUMP, bool CACHED>
static int launch_core_variant(const float* qfrag, const float* kvfrag, const float* z, const uint8_t* mask, const float* R, const float* t,
                               const float* Wb, float* feat, float* dump, float* dump_stats, const float* pbc, int N, int L, hipStream_t st, int z_shared) {
    const int nib = (L + BI - 1) / BI, nchunk = (L + JC - 1) / JC;
    const size_t lds = core_lds_fixed_bytes<CACHED>() + (size_t)nchunk * JC;
    ABOPT_CHECK_ARG(lds <= 160 * 1024, "ipa_core: L=%d needs %zu bytes of LDS for the key mask (max 163840)", L, lds);
    static LdsConfig lds_cfg;                                               // per instantiation
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_core_kernel<DUMP, CACHED>), lds, lds_cfg)) return rc;
    prof::begin(st);
    hipLaunchKernelGGL((ipa_core_kernel<DUMP, CACHED>), dim3((unsigned)(N * nib)), dim3(NTH), lds, st, qfrag, kvfrag, z, mask, R, t, Wb, feat, dump, dump_stats, pbc,
                       N, L, nib, (N % 8 == 0) ? 1 : 0, z_shared);
    prof::end(st);
    ABOPT_LAUNCH_CHECK();
#ifdef CORE_TIMING
    {
        long long h[3][8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_core_timing), sizeof(h));
        static int calls = 0;
        if (++calls == 8)
            for (int r = 0; r < 3; ++r)
                fprintf(stderr, "[core timing, cycles of WG 17] %s: prologue %lld | work %lld | barrier wait %lld | own epilogue %lld | F1/F2 wait %lld | common epilogue %lld\n",
                        r == 0 ? "pair" : (r == 1 ? "A   " : "C   "), h[r][0], h[r][1], h[r][2], h[r][3], h[r][4], h[r][5]);
    }
#endif
    return ABOPT_OK;
}

// The 32-row kernel runs one block per workgroup, so it pays where its N * ceil(L / 32) workgroups fill the CUs in whole rounds
// (tools/r03_c32_sweep.sh, r03_c32_sweep2.sh; microseconds per launch against the 16-row kernels on the same box):
//   one round, more than half full, where the 16-row blocks no longer fit one round themselves (N L / 16 > CUs: the persistent kernel
//   then walks two blocks per CU):  N = 23 / 24 / 28 / 32 at L = 256: 145 / 145 / 154 / 167 against 156 / 155 / 168 / 175; N = 32,
//   L = 200: 123 against 132.  Up to N L / 16 = CUs the one-block 16-row kernel is the faster one (N = 16: 90 against 124).
//   several rounds at least 95 % full:  N = 62 / 64: 334 / 335 against 344 / 345.  A half-empty last round loses (N = 48: 288 against
//   259; N = 20, L = 400: 385 against 258).
//   short lengths gain nothing (L = 128: equal; L = 64: 134 against 128).
// ABOPT_CORE32=0 / 1 overrides (1: whenever 16 < L <= 2048).
// block -> (sample, query block) mapping of the 32-row kernels: 2 = by complex (groups of z_shared samples, a multiple of 8 complexes),
// 1 = all query blocks of a sample on one XCD (N % 8 == 0), 0 = plain
static int core32_remap(int N, int z_shared) {
    if (z_shared > 1 && z_shared < N && N % z_shared == 0 && (N / z_shared) % 8 == 0) return 2;
    return (N % 8 == 0) ? 1 : 0;
}
static bool use_core32(int N, int L, int cus, int z_shared) {
    const char* e = getenv("ABOPT_CORE32");
    if (L > 2048) return false;                                 // its buffer descriptors address a sample's z slab (L^2 * 256 bytes) with 32-bit offsets
    // ... and ONE descriptor addresses a block's whole chunk-major slab of the bias cache (distinct samples x L rows x chunks x 768 bytes): 1365 distinct samples at L = 256
    if ((int64_t)(z_shared > 1 ? N / z_shared : N) * L * ((L + JC - 1) / JC) * (H * JC * 4) >= (1ll << 32)) return false;
    if (e && e[0] == '0') return false;
    if (e && e[0] == '1') return L > BI;
    if (cus < 8) return false;
    const int64_t total = (int64_t)N * ((L + BI2 - 1) / BI2), rounds = (total + cus - 1) / cus;
    // short crops (pose sampling: N = 1000 x L = 48): since the epilogue runs on two fp16 terms (round 5) the fused 32-row kernel wins wherever it fills
    // the chip once -- 4.10 -> 3.80 ms per step at N = 1000 x L = 48, 3.60 -> 2.88 at 600 x 64, 0.95 -> 0.81 at 64 x 128; it loses below one workgroup
    // per CU (32 x 128: 0.64 -> 0.72).  Rounds 3-4 had excluded L < 192 (three key chunks did not amortise a 40 us epilogue).
    if (L < 192) return L > BI && total >= cus;
    if (rounds == 1) return total * 100 >= (int64_t)cus * 53 && (int64_t)N * ((L + BI - 1) / BI) > cus;
    return total * 100 >= rounds * cus * 95;
}

bool ipa_core32_applies(int N, int L, int z_shared) {
    int cus = 0;
    if (device_cu_count(&cus)) return false;
    return !CORE_ABL && use_core32(N, L, cus, z_shared);
}

// The whole block behind the projections in ONE launch (ipa_core32_kernel<true, ZT>: core + tail) where the 32-row kernel is the core of
// choice; *fused = 0 and nothing launched otherwise (the caller then runs core and tail separately -- same results bit for bit).
// ABOPT_FUSE_TAIL=0 keeps the two-launch form (A/B, tests).
int launch_ipa_block_fused(const float* qfrag, const float* kvfrag, const float* z, const uint8_t* mask, const float* R, const float* t,
                           const float* pair_bias_cache, int N, int L, hipStream_t st, int z_shared, const float* wot, const float* wmf, const float* x,
                           const float* ubias, const float* g1, const float* be1, const float* b0, const float* b1, const float* b2, const float* g2,
                           const float* be2, float* out, int* fused, const float* pair_terms, float* xt_out) {
    *fused = 0;
    const char* e = getenv("ABOPT_FUSE_TAIL");
    if (!pair_bias_cache || !wot || !wmf || (e && e[0] == '0') || CORE_ABL || C32_ABL) return ABOPT_OK;
    int cus = 0;
    if (int rc = device_cu_count(&cus)) return rc;
    if (!use_core32(N, L, cus, z_shared)) return ABOPT_OK;
    const int nib2 = (L + BI2 - 1) / BI2;
    static LdsConfig lds_cfg, lds_cfg_t;
    TailArgs ta{wot, wmf, x, ubias, g1, be1, b0, b1, b2, g2, be2, out, reinterpret_cast<unsigned*>(xt_out)};
    const float* zsc = pair_terms ? pair_terms + pair_terms_floats(z_shared ? N / z_shared : N, L) : nullptr;
    if (pair_terms) {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_core32_kernel<true, true>), C32F_LDS_BYTES, lds_cfg_t)) return rc;
        prof::begin(st);
        hipLaunchKernelGGL((ipa_core32_kernel<true, true>), dim3((unsigned)(N * nib2)), dim3(NTH2), C32F_LDS_BYTES, st, qfrag, kvfrag, z, mask, R, t, (float*)nullptr,
                           pair_bias_cache, L, nib2, core32_remap(N, z_shared), z_shared, ta, prof::next_span_slot(), pair_terms, zsc);
    } else {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_core32_kernel<true, false>), C32F_LDS_BYTES, lds_cfg)) return rc;
        prof::begin(st);
        hipLaunchKernelGGL((ipa_core32_kernel<true, false>), dim3((unsigned)(N * nib2)), dim3(NTH2), C32F_LDS_BYTES, st, qfrag, kvfrag, z, mask, R, t, (float*)nullptr,
                           pair_bias_cache, L, nib2, core32_remap(N, z_shared), z_shared, ta, prof::next_span_slot(), (const float*)nullptr, (const float*)nullptr);
    }
    prof::end(st);
    ABOPT_LAUNCH_CHECK();
#ifdef C32F_TIMING
    {
        long long h[8][16];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c32f_timing), sizeof(h));
        static int calls = 0;
        if (++calls == 8)
            for (int w = 0; w < 8; ++w) {
                fprintf(stderr, "[c32f timing WG 17 wave %d] cycles since kernel start: at F1 %lld | at E0 %lld, past %lld | arrival at E1..E10:", w, h[w][0], h[w][1], h[w][2]);
                for (int k = 3; k < 13; ++k) fprintf(stderr, " %lld", h[w][k]);
                fprintf(stderr, " | past E10 %lld | at U %lld | end %lld\n", h[w][13], h[w][14], h[w][15]);
            }
    }
#endif
    *fused = 1;
    return ABOPT_OK;
}

int prof_spans_reset(hipStream_t st) {
    hipLaunchKernelGGL(prof_span_reset_kernel, dim3(PROF_SLOTS / 256), dim3(256), 0, st);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}
int prof_spans_read(int nslots, int* launches, double* total_ms) {
    static unsigned long long h[PROF_SLOTS][2];
    ABOPT_HIP(hipDeviceSynchronize());
    ABOPT_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof_span), sizeof(h)));
    int n = 0;
    double tot = 0.0;
    for (int i = 0; i < nslots && i < PROF_SLOTS; ++i)
        if (h[i][1] > h[i][0]) { ++n; tot += (double)(h[i][1] - h[i][0]) * 1e-5; }      // 100 MHz ticks -> ms
    *launches = n; *total_ms = tot;
    return ABOPT_OK;
}
int read_clock_probe(long long* cycles, long long* wall_ticks) {
    long long h[2] = {0, 0};
    ABOPT_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clock_probe), sizeof(h)));
    *cycles = h[0]; *wall_ticks = h[1];
    return ABOPT_OK;
}

int launch_ipa_core_kernel(const float* qfrag, const float* kvfrag, const float* z, const uint8_t* mask, const float* R, const float* t,
                           const float* w_pair_bias, float* feat, float* dump, float* dump_stats, const float* pair_bias_cache, int N, int L, hipStream_t st,
                           int z_shared, float* split_ws, size_t split_ws_floats, const float* pair_terms) {
    ABOPT_CHECK_ARG(!dump == !dump_stats, "ipa_core: the logits dump and its row statistics come together");
    ABOPT_CHECK_ARG(!dump || (int64_t)H * L * L * 4 < (1ll << 31), "ipa_core: L=%d too long for the logits dump", L);
    int cus32 = 0;
    if (pair_bias_cache && !dump && !CORE_ABL) { if (int rc = device_cu_count(&cus32)) return rc; }
    if (pair_bias_cache && !dump && !CORE_ABL && use_core32(N, L, cus32, z_shared)) {
        const int nib2 = (L + BI2 - 1) / BI2, nchunk = (L + JC - 1) / JC;
        const size_t lds = sizeof(float) * (3 * BI2 * SROW + 2 * BI2 * SCLD + BI2 * SCLD + BI2 * 32) + (size_t)nchunk * JC;
        ABOPT_CHECK_ARG(lds <= 160 * 1024, "ipa_core: L=%d needs %zu bytes of LDS for the key mask (max 163840)", L, lds);
        static LdsConfig lds_cfg, lds_cfg_t;
        if (pair_terms) {
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_core32_kernel<false, true>), lds, lds_cfg_t)) return rc;
            prof::begin(st);
            hipLaunchKernelGGL((ipa_core32_kernel<false, true>), dim3((unsigned)(N * nib2)), dim3(NTH2), lds, st, qfrag, kvfrag, z, mask, R, t, feat, pair_bias_cache, L, nib2,
                               core32_remap(N, z_shared), z_shared, TailArgs{}, prof::next_span_slot(), pair_terms, pair_terms + pair_terms_floats(z_shared ? N / z_shared : N, L));
        } else {
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_core32_kernel<false, false>), lds, lds_cfg)) return rc;
            prof::begin(st);
            hipLaunchKernelGGL((ipa_core32_kernel<false, false>), dim3((unsigned)(N * nib2)), dim3(NTH2), lds, st, qfrag, kvfrag, z, mask, R, t, feat, pair_bias_cache, L, nib2,
                               core32_remap(N, z_shared), z_shared, TailArgs{}, prof::next_span_slot(), (const float*)nullptr, (const float*)nullptr);
        }
        prof::end(st);
        ABOPT_LAUNCH_CHECK();
#ifdef C32_COUNT
        {
            static int calls = 0;
            if (++calls % 60 == 0) {
                unsigned long long h[8];
                (void)hipDeviceSynchronize();
                (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c32_count), sizeof(h));
                fprintf(stderr, "[c32 count after %d launches] pair row-chunks %llu, unchanged %llu (%.1f %%) | C (head, tile, chunk) %llu, unchanged %llu (%.1f %%) | C chunks %llu, unchanged %llu (%.1f %%)\n",
                        calls, h[0], h[1], 100.0 * h[1] / (h[0] + 1), h[2], h[3], 100.0 * h[3] / (h[2] + 1), h[4], h[5], 100.0 * h[5] / (h[4] + 1));
            }
        }
#endif
#ifdef C32_TIMING
        {
            long long h[3][8];
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_core_timing), sizeof(h));
            static int calls = 0;
            if (++calls == 8)
                for (int r = 0; r < 3; ++r)
                    fprintf(stderr, "[core32 timing, WG 17] %s: %lld cycles to the end of its chunk loop, %lld of them at barriers; %lld wall ticks of 10 ns -> %.2f GHz\n",
                            r == 0 ? "pair" : (r == 1 ? "A   " : "C   "), h[r][0], h[r][1], h[r][2], h[r][0] / (10.0 * h[r][2]));
            if (calls == 8) fprintf(stderr, "[core32 timing] SIMD of waves 0..7 (summed over launches, byte per wave): %llx %llx\n", (unsigned long long)h[0][4], (unsigned long long)h[0][5]);
        }
#endif
        return ABOPT_OK;
    }
    if (pair_bias_cache && !dump) {
        const int nib = (L + BI - 1) / BI, nchunk = (L + JC - 1) / JC;
        int cus = 0;
        if (int rc = device_cu_count(&cus)) return rc;
        cus &= ~7;                                                          // a multiple of 8 keeps blockIdx & 7 = XCD for every block of a workgroup
        const int total = N * nib;
#ifndef CORE_NO_PERSIST      // developer A/B switch
        if (nchunk >= 2 && cus >= 8 && total > cus && !CORE_ABL) {
            const size_t lds = sizeof(float) * (3 * BI * SROW + H * 4 * 64 * 4 + 2 * BI * SCLD + BI * SCLD + BI * H * P * 3) + 2 * (size_t)nchunk * JC;
            ABOPT_CHECK_ARG(lds <= 160 * 1024, "ipa_core: L=%d needs %zu bytes of LDS for the key masks (max 163840)", L, lds);
            static LdsConfig lds_cfg;
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_core_persist_kernel), lds, lds_cfg)) return rc;
            prof::begin(st);
            hipLaunchKernelGGL(ipa_core_persist_kernel, dim3((unsigned)cus), dim3(NTH), lds, st, qfrag, kvfrag, z, mask, R, t, feat, pair_bias_cache, L, nib, total,
                               (N % 8 == 0) ? 1 : 0, z_shared);
            prof::end(st);
            ABOPT_LAUNCH_CHECK();
#ifdef PERSIST_TIMING
            {
                long long h[3][8];
                (void)hipDeviceSynchronize();
                (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_core_timing), sizeof(h));
                static int calls = 0;
                if (++calls == 8)
                    for (int r = 0; r < 3; ++r)
                        fprintf(stderr, "[persist timing, cycles of WG 17] %s: total %lld | barrier wait %lld | waiting for its fragment loads %lld | (C) LDS reads %lld, scale + MFMA issue %lld | barrier wait by interval class (first / middle / last chunk of a block) %lld / %lld / %lld\n", r == 0 ? "pair" : (r == 1 ? "A   " : "C   "), h[r][0], h[r][1], h[r][2], h[r][3], h[r][4], h[r][5], h[r][6], h[r][7]);
            }
#endif
            return ABOPT_OK;
        }
#endif
    }
    if (pair_bias_cache && !dump && split_ws && !CORE_ABL && !getenv("ABOPT_CORE_NO_SPLIT")) {
        // small batches: split the keys of every query block over 2 or 4 workgroups (see the SPLIT note at ipa_core_kernel)
        const int nib = (L + BI - 1) / BI, nchunk = (L + JC - 1) / JC, total = N * nib;
        int cus = 0;
        if (int rc = device_cu_count(&cus)) return rc;
        int nsplit = (total * 4 <= cus && nchunk >= 8) ? 4 : ((total * 2 <= cus && nchunk >= 4) ? 2 : 1);
        const int64_t rows = (int64_t)N * L;
        if (nsplit > 1 && (size_t)nsplit * rows * (SPLIT_ROW + 2 * H) <= split_ws_floats) {
            float* part = split_ws;
            float* pstats = split_ws + (size_t)nsplit * rows * SPLIT_ROW;
            const size_t lds = core_lds_fixed_bytes<true>() + (size_t)nchunk * JC;
            ABOPT_CHECK_ARG(lds <= 160 * 1024, "ipa_core: L=%d needs %zu bytes of LDS for the key mask (max 163840)", L, lds);
            static LdsConfig lds_cfg;
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_core_kernel<false, true, true>), lds, lds_cfg)) return rc;
            prof::begin(st);
            hipLaunchKernelGGL((ipa_core_kernel<false, true, true>), dim3((unsigned)(total * nsplit)), dim3(NTH), lds, st, qfrag, kvfrag, z, mask, R, t, w_pair_bias, feat,
                               nullptr, nullptr, pair_bias_cache, N, L, nib, (N % 8 == 0) ? 1 : 0, z_shared, part, pstats, nsplit);
            ABOPT_LAUNCH_CHECK();
            hipLaunchKernelGGL(ipa_split_merge_kernel, dim3((unsigned)rows), dim3(256), 0, st, part, pstats, mask, R, t, feat, rows, nsplit);
            prof::end(st);
            ABOPT_LAUNCH_CHECK();
            return ABOPT_OK;
        }
    }
    if (pair_bias_cache) {
        if (dump) return launch_core_variant<true, true>(qfrag, kvfrag, z, mask, R, t, w_pair_bias, feat, dump, dump_stats, pair_bias_cache, N, L, st, z_shared);
        return launch_core_variant<false, true>(qfrag, kvfrag, z, mask, R, t, w_pair_bias, feat, nullptr, nullptr, pair_bias_cache, N, L, st, z_shared);
    }
    if (dump) return launch_core_variant<true, false>(qfrag, kvfrag, z, mask, R, t, w_pair_bias, feat, dump, dump_stats, nullptr, N, L, st, z_shared);
    return launch_core_variant<false, false>(qfrag, kvfrag, z, mask, R, t, w_pair_bias, feat, nullptr, nullptr, nullptr, N, L, st, z_shared);
}

}  // namespace abopt

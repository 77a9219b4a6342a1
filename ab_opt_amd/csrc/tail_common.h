// Shared pieces of the GABlock tail (out_transform + LayerNorm + mlp_transition + LayerNorm, ga.py:174-177) used by the stand-alone
// tail kernel (mlp.hip: out_ln_mlp_kernel) and by the fused IPA-core + tail kernel (ipa_core.hip: ipa_core32_kernel<true>).  Both run
// the SAME arithmetic in the SAME order per output element -- k-steps of a (column block, K group) chain in the chunk order
// ot_chunk_at(), partial sums ((p0 + p1) + (p2 + p3)) + bias, phase 2 row-local -- so their results are bit-identical
// (tests/test_hip_parity.py::test_fused_block_is_bit_identical).
// Arithmetic of the FORWARD tail since round 5: two fp16 terms per operand, three products per k-step (ipa_common.h: split_pair2); all weights arrive
// pre-split and scaled by a power of two per matrix (pack_tail_weights_kernel), the inverse scales ride in w_mlp_frag behind the layer weights.  The
// backward chain (tail_backward_kernel) keeps the three-term bf16 products and its own helpers below (store_terms2, MlpW, mlp_partial).
#pragma once
#include "ipa_common.h"

namespace abopt {

constexpr int F = 128, XLD = F + 4;
constexpr int MR = 32;                     // rows per workgroup

namespace {
constexpr int OT_K = ABOPT_IPA_FEAT;          // 1824
constexpr int OT_KC = 192, OT_NCH = (OT_K + OT_KC - 1) / OT_KC;           // 10 chunks of 192 columns = 12 k-steps of 16 (the last one half full)
constexpr int OT_ST = OT_K / 16;              // 114 k-steps
constexpr int OT_SPC = OT_KC / 16, OT_SPW = OT_SPC / 4;                   // 12 k-steps per chunk, 3 per wave
constexpr int OT_TH = 1024, OT_NW = OT_TH / 64;
constexpr int OT_SROW = OT_KC * 2 + 16;       // bytes per row of one bf16 plane of a chunk: 400, rows 36 banks apart (conflict-free b128 reads)
constexpr int OT_NT = 2;                      // terms per value in the forward tail (fp16 h | l)
constexpr int OT_PLANE = MR * OT_SROW, OT_STAGE = OT_NT * OT_PLANE;       // 12800, 25600 bytes
constexpr int AP_ROW = F * 2 + 16, AP_PLANE = MR * AP_ROW;                // activation planes: 272 bytes per row (rows 4 banks apart)
constexpr int OT_WVEC = OT_NT * 64;           // 16-byte vectors per (column block, k-step) of w_out_frag: [term][lane]
constexpr int OT_SCALE_OFF = 3 * F * F;       // float offset of {S_out, S_0, S_1, S_2, 1/S_out, 1/S_0, 1/S_1, 1/S_2} in w_mlp_frag
constexpr int OT_MS = F / 16;                 // 8 k-steps per MLP layer
static_assert(OT_K % 16 == 0 && OT_KC % 64 == 0 && MR == 32 && F == 128, "out_transform tiling");

struct OtSmem {
    float ys[MR][XLD];                        // y = LayerNorm1(...) in fp32 (residual of the MLP)
    float bias[3][F];                         // b_mlp0..2
    char ap[OT_NT * AP_PLANE];                // input of the current layer as [term][row][128 fp16 + pad]
    union {
        char stage[2][OT_STAGE];              // phase 1: feat chunks as [term][row][192 fp16 + pad]
        float part[4][MR][XLD];               // partial sums of the four K groups
    };
};

__device__ __forceinline__ void acc_zero(f32x16& a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
}
// accumulator register 4 g + i = output column 32 cb + 8 g + 4 (lane >> 5) + i, lane & 31 = residue
__device__ __forceinline__ void store_partial1(float (*dst)[XLD], const f32x16& a0, int cb, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(&dst[lane & 31][cb * 32 + g * 8 + (lane >> 5) * 4]) = (f32x4){a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]};
}
__device__ __forceinline__ void store_partial(float (*dst)[XLD], const f32x16& a0, const f32x16& a1, int cb, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(&dst[lane & 31][cb * 32 + g * 8 + (lane >> 5) * 4]) =
            (f32x4){a0[4 * g] + a1[4 * g], a0[4 * g + 1] + a1[4 * g + 1], a0[4 * g + 2] + a1[4 * g + 2], a0[4 * g + 3] + a1[4 * g + 3]};
}
// two adjacent values -> one 4-byte entry in each of the two fp16 planes (forward)
__device__ __forceinline__ void store_terms2h(char* ap, int byte_off, float e0, float e1) {
    unsigned h, l;
    split_pair2(e0, e1, h, l);
    *reinterpret_cast<unsigned*>(ap + byte_off) = h;
    *reinterpret_cast<unsigned*>(ap + AP_PLANE + byte_off) = l;
}
// two adjacent values -> one 4-byte entry in each of the three bf16 planes (backward)
__device__ __forceinline__ void store_terms2(char* ap, int byte_off, float e0, float e1) {
    const unsigned h = pk_bf16(e0, e1);
    const float r0 = e0 - __uint_as_float(h << 16), r1 = e1 - __uint_as_float(h & 0xffff0000u);
    const unsigned m = pk_bf16(r0, r1);
    const unsigned l = pk_bf16(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
    *reinterpret_cast<unsigned*>(ap + byte_off) = h;
    *reinterpret_cast<unsigned*>(ap + AP_PLANE + byte_off) = m;
    *reinterpret_cast<unsigned*>(ap + 2 * AP_PLANE + byte_off) = l;
}
struct MlpW { u32x4 v[2][3]; };               // a wave's two k-steps of one layer: [step][term]
__device__ __forceinline__ MlpW load_mlp_w(const float* __restrict__ wm, int layer, int cb, int kg, int lane) {
    const u32x4* p = reinterpret_cast<const u32x4*>(wm) + ((int64_t)(layer * 4 + cb) * OT_MS + kg * 2) * 192 + lane;
    MlpW w;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) w.v[j][sp] = p[j * 192 + sp * 64];
    return w;
}
// partial [32 rows x 32 columns] of one layer for the wave's two k-steps
__device__ __forceinline__ void mlp_partial(const char* ap, const MlpW& w, float (*dst)[XLD], int cb, int kg, int lane) {
    f32x16 a0, a1;
    acc_zero(a0); acc_zero(a1);
    const char* xp = ap + (lane & 31) * AP_ROW + kg * 64 + (lane >> 5) * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const u32x4 xh = *reinterpret_cast<const u32x4*>(xp + j * 32), xm = *reinterpret_cast<const u32x4*>(xp + j * 32 + AP_PLANE),
                    xl = *reinterpret_cast<const u32x4*>(xp + j * 32 + 2 * AP_PLANE);
        a0 = mfma_bf32(w.v[j][0], xl, a0); a1 = mfma_bf32(w.v[j][2], xh, a1);
        a0 = mfma_bf32(w.v[j][1], xm, a0); a1 = mfma_bf32(w.v[j][0], xm, a1);
        a0 = mfma_bf32(w.v[j][1], xh, a0); a1 = mfma_bf32(w.v[j][0], xh, a1);
    }
    store_partial(dst, a0, a1, cb, lane);
}
}  // namespace

// ---- phase-2 layer weights for the 16x16x32 form: wmf [layer][ct 8][k-step 4][term 2][lane 64] x 8 fp16, lane (m = lane & 15, kq = lane >> 4)
// holds the terms of S_layer W[16 ct + m][32 s + 8 kq + i].  4 bytes per weight and no VALU work in the kernel (rounds 2-4 streamed fp32 and
// split in registers: 176 VALU operations per wave and layer; three bf16 terms would have been 6 bytes).  Eight waves compute (wave = column
// tile ct, both row tiles with the same weight registers); a layer's fragment is 32 registers, requested one layer ahead.
namespace {
struct MlpRaw { u32x4 v[4][2]; };             // [k-step][term]
__device__ __forceinline__ MlpRaw load_mlp_raw(const float* __restrict__ wm, int layer, int ct, int lane) {
    const u32x4* p = reinterpret_cast<const u32x4*>(wm) + (int64_t)((layer * 8 + ct) * 4) * OT_WVEC + lane;
    MlpRaw w;
#pragma unroll
    for (int s = 0; s < 4; ++s) { w.v[s][0] = p[s * OT_WVEC]; w.v[s][1] = p[s * OT_WVEC + 64]; }
    return w;
}
// [16 output columns of tile ct] x [32 rows] of one layer, K = 128, no K split: o[rt] register r of lane (n, kq) = S_layer x output column
// 16 ct + 4 kq + r of row 16 rt + n (the caller multiplies by 1 / S_layer).
__device__ __forceinline__ void mlp16_layer(const char* ap, const MlpRaw& w, int lane, f32x4 (&o)[2]) {
    f32x4 a[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { a[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; a[rt][1] = a[rt][0]; }
    const char* xp = ap + (lane & 15) * AP_ROW + (lane >> 4) * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 xh[2], xl[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const char* xr = xp + rt * 16 * AP_ROW + s * 64;
            xh[rt] = *reinterpret_cast<const u32x4*>(xr); xl[rt] = *reinterpret_cast<const u32x4*>(xr + AP_PLANE);
        }
        // small products first; four independent chains (two per row tile) so consecutive MFMAs never depend on each other
        a[0][0] = mfma_h(w.v[s][0], xl[0], a[0][0]); a[1][0] = mfma_h(w.v[s][0], xl[1], a[1][0]);
        a[0][1] = mfma_h(w.v[s][1], xh[0], a[0][1]); a[1][1] = mfma_h(w.v[s][1], xh[1], a[1][1]);
        a[0][0] = mfma_h(w.v[s][0], xh[0], a[0][0]); a[1][0] = mfma_h(w.v[s][0], xh[1], a[1][0]);
    }
    o[0] = a[0][0] + a[0][1]; o[1] = a[1][0] + a[1][1];
}
}  // namespace

namespace {
// Order in which the ten 192-column chunks of feat enter the out_transform sums.  The fused kernel gets the node features (chunks 4, 5:
// the C waves' accumulators, which must leave their registers before those waves can take part in the products) first, then the pair
// features (0..3, the pair waves' accumulators), then what the point epilogue derives (6..9; the last one half full).
__device__ __forceinline__ constexpr int ot_chunk_at(int p) { return p < 2 ? 4 + p : (p < 6 ? p - 2 : p); }
// Which feature column sits at staging column j (0..191) of chunk c.  The 768 pair-feature columns (h, channel) are dealt to chunks 0..3
// by (channel % 16) / 4, not by head: chunk c, j = 16 h + 4 (channel / 16) + channel % 4.  In the fused kernel a pair wave's lane
// (head, key group kq) holds the channels 16 kq + 4 r + 0..3 of its head: chunk r takes ONE 8-byte entry per row from EVERY lane (24
// stores per wave and chunk instead of 96 from 12 of its 64 lanes, which cost the consumers 2..3k cycles per interval).  W_out is packed
// in the same column order (pack_tail_weights_kernel), so the products are unchanged, only the order of the sum.  Chunks 4..9: identity.
__device__ __forceinline__ constexpr int ot_feat_col(int c, int j) {
    return c < 4 ? (j >> 4) * 64 + ((j >> 2) & 3) * 16 + 4 * c + (j & 3) : c * OT_KC + j;
}

// the three fp16-term products of one k-step (small ones first) into ONE accumulator chain per (column block, K group).  (Two chains per K
// group -- round 3 -- cost the fused kernel's consumer waves, which run all four K groups, 128 accumulator registers; with 64 they keep their
// W_out fragments in flight, which is what the epilogue needs while other workgroups still stream z: L2 round trips of ~2000 cycles.)
__device__ __forceinline__ void ot_kstep3(const u32x4& wH, const u32x4& wL, const u32x4& xh, const u32x4& xl, f32x16& acc) {
    acc = mfma_h32(wH, xl, acc); acc = mfma_h32(wL, xh, acc); acc = mfma_h32(wH, xh, acc);
}

// ---- phase 2 of the tail: LayerNorm1, three 128 x 128 layers, LayerNorm2 on the 32 rows of a workgroup of NW waves (16 or 8).
// What it needs from global memory is requested by tail_p2_prefetch BEFORE the caller publishes u (loads return in order, and the layer
// weights behind them are a 64 KB burst).
template <int NW>
struct TailP2Pre {
    float2 xv[MR / NW];
    bool keep[MR / NW];
    float2 bb, g1v, be1v;
    f32x4 inv;                                // 1 / S of W_out, W_mlp0..2
    MlpRaw mw;
};
template <int NW>
__device__ __forceinline__ TailP2Pre<NW> tail_p2_prefetch(const float* __restrict__ x, const float* __restrict__ ubias, const uint8_t* __restrict__ mask,
                                                         const float* __restrict__ g1, const float* __restrict__ be1, const float* __restrict__ wmf,
                                                         int64_t row0, int64_t row_end, int wave, int lane) {
    constexpr int RW = MR / NW;
    TailP2Pre<NW> p;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int64_t row = min(row0 + wave * RW + rr, row_end - 1);
        p.keep[rr] = mask ? (mask[row] != 0) : true;
        p.xv[rr] = reinterpret_cast<const float2*>(x + row * F)[lane];
    }
    p.bb = ubias ? reinterpret_cast<const float2*>(ubias)[lane] : make_float2(0.f, 0.f);
    p.g1v = reinterpret_cast<const float2*>(g1)[lane];
    p.be1v = reinterpret_cast<const float2*>(be1)[lane];
    p.inv = *reinterpret_cast<const f32x4*>(wmf + OT_SCALE_OFF + 4);
    if (wave < 8) p.mw = load_mlp_raw(wmf, 0, wave & 7, lane);
    return p;
}
// the three bias vectors -> LDS (read per layer by the compute waves); call before the barrier that publishes u
__device__ __forceinline__ void tail_p2_stage_bias(float (*bias)[F], const float* __restrict__ b0, const float* __restrict__ b1, const float* __restrict__ b2, int tid) {
    if (tid < 3 * F / 4) {
        const float* bsrc = tid < F / 4 ? b0 : (tid < F / 2 ? b1 : b2);
        *reinterpret_cast<f32x4*>(&bias[tid >> 5][(tid & 31) * 4]) = *reinterpret_cast<const f32x4*>(bsrc + (tid & 31) * 4);
    }
}
// GetU(rl) -> this lane's two columns (2 lane, 2 lane + 1) of S_out u, u = feat . W_out^T for local row rl, WITHOUT the bias.  The caller has
// passed the barrier that publishes u and the staged biases.  ys [MR][XLD] fp32, apA / apB two sets of OT_NT fp16 planes (AP_PLANE each);
// none of them may alias what GetU reads.  DUMP (training): five [rows, 128] slabs, see out_ln_mlp_kernel.
template <int NW, bool DUMP, class GetU>
__device__ __forceinline__ void tail_p2_run(TailP2Pre<NW>& pre, GetU&& get_u, float (*ys)[XLD], float (*bias)[F], char* apA, char* apB,
                                            const float* __restrict__ wmf, const float* __restrict__ g2, const float* __restrict__ be2,
                                            float* __restrict__ out, float* __restrict__ dump, int64_t slab, int64_t row0, int64_t row_end,
                                            int wave, int lane, unsigned* __restrict__ xt_out = nullptr) {
    constexpr int RW = MR / NW;
    const int fm = lane & 15, kq = lane >> 4;
    const bool mlpw = wave < 8;                                                                          // waves 0..7 own the eight 16-column tiles
    const int ct = wave & 7;
    const int ocol = ct * 16 + kq * 4;                                                                   // this lane's four output columns in every layer
    MlpRaw mw = pre.mw;
    {
        const float2 g = pre.g1v, bt = pre.be1v;
        float a_[RW], b_[RW], mean[RW], var[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int rl = wave * RW + rr;
            const float2 u = get_u(rl);
            float2 us = make_float2(fmaf(u.x, pre.inv[0], pre.bb.x), fmaf(u.y, pre.inv[0], pre.bb.y));    // the scaling is exact: one rounding, in the sum
            if (!pre.keep[rr]) us = make_float2(0.f, 0.f);
            a_[rr] = pre.xv[rr].x + us.x; b_[rr] = pre.xv[rr].y + us.y;
            if (DUMP && row0 + rl < row_end) reinterpret_cast<float2*>(dump + (row0 + rl) * F)[lane] = make_float2(a_[rr], b_[rr]);
        }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) mean[rr] = wave_sum(a_[rr] + b_[rr]) * (1.f / F);
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) { a_[rr] -= mean[rr]; b_[rr] -= mean[rr]; var[rr] = wave_sum(a_[rr] * a_[rr] + b_[rr] * b_[rr]) * (1.f / F); }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int rl = wave * RW + rr;
            const float sd = sqrtf(var[rr] + 1e-10f);
            const float y0 = a_[rr] / sd * g.x + bt.x, y1 = b_[rr] / sd * g.y + bt.y;
            *reinterpret_cast<float2*>(&ys[rl][2 * lane]) = make_float2(y0, y1);
            store_terms2h(apA, rl * AP_ROW + lane * 4, y0, y1);
            if (DUMP && row0 + rl < row_end) reinterpret_cast<float2*>(dump + slab + (row0 + rl) * F)[lane] = make_float2(y0, y1);
        }
    }
    __syncthreads();                                                                                     // y planes complete; every read of u is done
    // ---- layer 0: relu(W0 y + b0) -> planes B ; layer 1: relu(W1 h + b1) -> planes A (their last readers passed the barrier in between)
#pragma unroll
    for (int layer = 0; layer < 2; ++layer) {
        if (mlpw) {
            const char* src = layer == 0 ? apA : apB;
            char* dst = layer == 0 ? apB : apA;
            const MlpRaw nxt = load_mlp_raw(wmf, layer + 1, ct, lane);                                  // the next layer's fragment travels while this one computes
            f32x4 o[2];
            mlp16_layer(src, mw, lane, o);
            mw = nxt;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias[layer][ocol]);
            const float isc = pre.inv[1 + layer];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int orow = rt * 16 + fm;
                const f32x4 v = (f32x4){fmaf(o[rt][0], isc, bv[0]), fmaf(o[rt][1], isc, bv[1]), fmaf(o[rt][2], isc, bv[2]), fmaf(o[rt][3], isc, bv[3])};
                const f32x4 hv = (f32x4){relu_nan(v[0]), relu_nan(v[1]), relu_nan(v[2]), relu_nan(v[3])};
                store_terms2h(dst, orow * AP_ROW + ocol * 2, hv[0], hv[1]);
                store_terms2h(dst, orow * AP_ROW + ocol * 2 + 4, hv[2], hv[3]);
                if (DUMP && row0 + orow < row_end) *reinterpret_cast<f32x4*>(dump + (2 + layer) * slab + (row0 + orow) * F + ocol) = hv;
            }
        }
        __syncthreads();
    }
    // ---- layer 2 + residual (in place in ys: every element has exactly one owner), then LayerNorm2
    if (mlpw) {
        f32x4 o[2];
        mlp16_layer(apA, mw, lane, o);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias[2][ocol]);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int orow = rt * 16 + fm;
            f32x4 yv = *reinterpret_cast<const f32x4*>(&ys[orow][ocol]);
            const float isc = pre.inv[3];
            yv += (f32x4){fmaf(o[rt][0], isc, bv[0]), fmaf(o[rt][1], isc, bv[1]), fmaf(o[rt][2], isc, bv[2]), fmaf(o[rt][3], isc, bv[3])};
            *reinterpret_cast<f32x4*>(&ys[orow][ocol]) = yv;
            if (DUMP && row0 + orow < row_end) *reinterpret_cast<f32x4*>(dump + 4 * slab + (row0 + orow) * F + ocol) = yv;
        }
    }
    __syncthreads();
    {
        const float2 g = reinterpret_cast<const float2*>(g2)[lane], bt = reinterpret_cast<const float2*>(be2)[lane];
        float2 v[RW];
        float mean[RW], var[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) { v[rr] = *reinterpret_cast<const float2*>(&ys[wave * RW + rr][2 * lane]); mean[rr] = wave_sum(v[rr].x + v[rr].y) * (1.f / F); }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) { v[rr].x -= mean[rr]; v[rr].y -= mean[rr]; var[rr] = wave_sum(v[rr].x * v[rr].x + v[rr].y * v[rr].y) * (1.f / F); }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int64_t row = row0 + wave * RW + rr;
            const float sd = sqrtf(var[rr] + 1e-10f);
            if (row < row_end) {
                const float o0 = v[rr].x / sd * g.x + bt.x, o1 = v[rr].y / sd * g.y + bt.y;
                reinterpret_cast<float2*>(out + row * F)[lane] = make_float2(o0, o1);
                // round 6: the same row as two fp16 terms ([64 words of high terms | 64 words of low terms], split_pair2) for the NEXT block's node_frags, whose 24
                // (head, half) workgroups would otherwise each split these values again
                if (xt_out) { unsigned h_, l_; split_pair2(o0, o1, h_, l_); xt_out[row * F + lane] = h_; xt_out[row * F + 64 + lane] = l_; }
            }
        }
    }
}
}  // namespace

}  // namespace abopt

// Shared constants / helpers of the IPA-core kernels (gfx950).
#pragma once
#include "abopt_common.h"

namespace abopt {

constexpr int H = ABOPT_HEADS, D = ABOPT_QK_DIM, P = ABOPT_POINTS, C = 64;
// Row layout of the per-residue projection buffer: the 2016 outputs of the fused node projection GEMM
// (q|k|v|q_pts|k_pts|v_pts) followed by the squared norms of the 8 global-frame query / key points of every head
// (written by points_to_global), padded to 2048 floats = one 8 KB row.
constexpr int NP = 2048;
constexpr int OFF_Q = 0, OFF_K = H * D, OFF_V = 2 * H * D, OFF_QP = 3 * H * D, OFF_KP = OFF_QP + H * P * 3, OFF_VP = OFF_KP + H * P * 3;
constexpr int OFF_NQ = ABOPT_NODE_PROJ, OFF_NK = OFF_NQ + H;
constexpr int FEAT = ABOPT_IPA_FEAT;         // 1824

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BI = 16;            // query rows per workgroup
constexpr int JC = 16;            // key rows per chunk
constexpr int PLD = JC + 4;       // row stride of the S/P tile (floats)
constexpr int ZSLD = C + 4;       // row stride of the z staging tile (floats)
constexpr int NPT = H * P * 3;    // 288 point coordinates per residue

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// reductions over the four 16-lane rows of a wave (lanes with equal lane & 15) with the gfx950 row-swap instructions
__device__ __forceinline__ float rows_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// ---- fp32 x fp32 on the bf16 matrix pipe: an fp32 value is exactly h + m + l with three bf16 terms (see node_frags.hip header)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// (lo, hi) -> one register of two bf16, round to nearest even.  PK_BF16_MODE (developer switch): 0 (shipped) the instruction as an asm
// statement | 1 the same + two wait states inside the string | 2 the compiler's own float -> bf16 conversion (it lowers to the same instruction).
// Round 5 (DESIGN_LOG.md): mode 2 lets the SLP vectoriser pack the splits' subtractions into v_pk_add_f32 and made the fused block kernel
// NON-DETERMINISTIC (repeat tests fail at the first repeat; with -fno-slp-vectorize they pass) -- not understood, so the asm form, which every
// bit-identity and repeat test of rounds 2-5 has covered, stays.  What is known about it: hipcc pads no MFMA hazard for an asm statement's result
// register.  Measured on gfx950 (tools/micro/mfma_hazard.hip): a VALU overwrite of an in-flight v_mfma_f32_16x16x32_bf16's SrcC is safe at 0
// wait states, a VALU read of D needs 8, a VALU overwrite of D needs 4.  tools/r05/mfma_hazard_scan.py lists the sequences of a build: the
// shipped kernels have no D read within 8 states; they do have cvt results allocated to a register an MFMA wrote 2-3 instructions earlier
// (out_ln_mlp 12, fused block kernel 11) -- each behind a dependent MFMA, which cannot issue before that write has landed.
#ifndef PK_BF16_MODE
#define PK_BF16_MODE 0
#endif
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
#if PK_BF16_MODE == 2
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){lo, hi}, bf16x2_));
#else
    unsigned r;
#if PK_BF16_MODE == 1
    asm("v_cvt_pk_bf16_f32 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(lo), "v"(hi));
#else
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#endif
    return r;
#endif
}
// 8 consecutive fp32 values -> their three bf16 terms, two values per register (element 2p in the low half)
struct Split3 { u32x4 h, m, l; };
__device__ __forceinline__ Split3 split3(const f32x4& lo, const f32x4& hi) {
    Split3 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float e0 = p < 2 ? lo[2 * p] : hi[2 * p - 4], e1 = p < 2 ? lo[2 * p + 1] : hi[2 * p - 3];
        o.h[p] = pk_bf16(e0, e1);
        const float r0 = e0 - __uint_as_float(o.h[p] << 16), r1 = e1 - __uint_as_float(o.h[p] & 0xffff0000u);    // exact, |r| <= 2^-9 |e|
        o.m[p] = pk_bf16(r0, r1);
        const float q0 = r0 - __uint_as_float(o.m[p] << 16), q1 = r1 - __uint_as_float(o.m[p] & 0xffff0000u);    // exact, <= 8 significant bits left
        o.l[p] = pk_bf16(q0, q1);                                                  // exact
    }
    return o;
}
__device__ __forceinline__ f32x4 mfma_bf(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_bf32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// ---- fp32 x fp32 with TWO fp16 terms per operand (round 5; the forward tail of a GABlock, tail_common.h).  h = fp16(x), l = fp16(x - h), both round to
// nearest: |x - h - l| <= 2^-22 |x| (two 11-bit significands; x - h is exact in fp32), absolute floor 2^-25 where l is subnormal (the gfx950 matrix
// pipe and both conversions keep fp16 subnormals: tools/micro/f16_denorm.hip).  Weights are multiplied by a power of two S per tensor before the split
// (max |w| S in [2^14, 2^15): their low terms stay normal) and the result by 1 / S -- both exact.  THREE products per k-step
//     x w S = h_x l_w + l_x h_w + h_x h_w   [+ l_x l_w, <= 2^-22 relative, dropped]
// instead of the six of the three-term bf16 scheme, and 4 instead of 6 bytes per staged value.  Error against an exact product <= 3 x 2^-22 relative
// worst case, ~2^-23 typical: in a K = 128..1824 sum that is the size of the fp32 accumulation error both schemes share (measured on random
// operands, max / mean error of [512 x 1824] . [1824 x 128] against fp64: 3.9e-6 / 2.1e-7 for this scheme, 4.9e-6 / 2.7e-7 for the six bf16 products,
// 3.3e-6 / 3.2e-7 for an fp32 GEMM).  Limit: |activation| < 65504 (fp16 overflow -> inf -> NaN in the output, nothing silent).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifndef PK_F16_MODE
#define PK_F16_MODE 0     // developer switch: 0 (shipped) an asm statement like pk_bf16 (see there) | 2 the compiler's own float -> half conversion
#endif
__device__ __forceinline__ unsigned pk_f16(float lo, float hi) {
#if PK_F16_MODE == 2
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){lo, hi}, f16x2_));
#else
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#endif
}
__device__ __forceinline__ float f16lo_f32(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float f16hi_f32(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
// two adjacent fp32 values -> their two packed fp16 term words
__device__ __forceinline__ void split_pair2(float e0, float e1, unsigned& h, unsigned& l) {
    h = pk_f16(e0, e1);
    l = pk_f16(e0 - f16lo_f32(h), e1 - f16hi_f32(h));                     // the differences are exact
}
struct Split2 { u32x4 h, l; };
__device__ __forceinline__ Split2 split2(const f32x4& lo, const f32x4& hi) {
    Split2 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float e0 = p < 2 ? lo[2 * p] : hi[2 * p - 4], e1 = p < 2 ? lo[2 * p + 1] : hi[2 * p - 3];
        unsigned h, l;
        split_pair2(e0, e1, h, l);
        o.h[p] = h; o.l[p] = l;
    }
    return o;
}
__device__ __forceinline__ f32x4 mfma_h(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// K = 16 form (two registers per operand): the second product of a K-packed pair, P_l against the high terms only
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mfma_h16(const u32x2& a, const u32x2& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_h32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float f4get(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

namespace prof { void begin(hipStream_t st); void end(hipStream_t st); int next_span_slot(); }
int prof_spans_reset(hipStream_t st);
int prof_spans_read(int nslots, int* launches, double* total_ms);

int launch_ipa_core_kernel(const float* qfrag, const float* kvfrag, const float* z, const uint8_t* mask, const float* R, const float* t,
                           const float* w_pair_bias, float* feat, float* dump, float* dump_stats, const float* pair_bias_cache, int N, int L,
                           hipStream_t st, int z_shared, float* split_ws = nullptr, size_t split_ws_floats = 0, const float* pair_terms = nullptr);
bool ipa_core32_applies(int N, int L, int z_shared = 0);           // the launch geometry takes the 32-row kernels (with a bias cache)
size_t pair_terms_floats(int Nz, int L);
size_t pair_terms_blob_floats(int Nz, int L);
int launch_pair_terms(const float* z, float* blob, int Nz, int L, hipStream_t st);
size_t ipa_split_ws_floats(int N, int L);
int read_clock_probe(long long* cycles, long long* wall_ticks);
int launch_pair_bias_cache(const float* z, const float* const* wb, int num_layers, float* cache, int N, int L, hipStream_t st);

}  // namespace abopt

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
__device__ __forceinline__ float f4get(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

namespace prof { void begin(hipStream_t st); void end(hipStream_t st); }

int launch_ipa_core_kernel(const float* qfrag, const float* kvfrag, const float* z, const uint8_t* mask, const float* R, const float* t,
                           const float* w_pair_bias, float* feat, float* dbg_logits, const float* pair_bias_cache, int N, int L, hipStream_t st, int z_shared);
int launch_pair_bias_cache(const float* z, const float* const* wb, int num_layers, float* cache, int N, int L, hipStream_t st);

}  // namespace abopt

// Shared device helpers for the gfx950 kernels: SO(3)/frame algebra with the reference's exact
// epsilons, wave64 reductions, Philox4x32-10.  Written for CDNA4 only (wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/abopt.h"

#define ABOPT_WAVE 64

namespace abopt {

// ---------------------------------------------------------------- host-side error plumbing
void set_error(const char* fmt, ...);
#define ABOPT_CHECK_ARG(cond, ...) do { if (!(cond)) { ::abopt::set_error(__VA_ARGS__); return ABOPT_EINVAL; } } while (0)
#define ABOPT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    ::abopt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return ABOPT_EHIP; } } while (0)
#define ABOPT_LAUNCH_CHECK() ABOPT_HIP(hipGetLastError())

// Per-DEVICE launch state (api.hip).  One process may drive several GPUs (`with torch.cuda.device(...)`): the CU count and the
// "this kernel may use > 64 KB of dynamic LDS" attribute belong to the device that is current at the launch, not to the first one seen.
constexpr int kMaxDevices = 64;
struct LdsConfig { size_t bytes[kMaxDevices] = {}; };       // one static instance per kernel: bytes already granted, by device ordinal
int device_cu_count(int* cus);                               // CUs of the current device (cached per ordinal, thread-safe)
int ensure_dynamic_lds(const void* kernel, size_t bytes, LdsConfig& cfg);

// Lanes of ONE wave exchanging data through LDS (write, then read what other lanes wrote): the hardware executes a wave's
// LDS instructions in order, but the compiler may reorder a ds_read above a ds_write it cannot prove aliased.  This pins the
// order at wavefront scope (no s_barrier; at most an s_waitcnt).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
// All-reduce over the 64 lanes without touching the LDS crossbar (ds_bpermute costs ~60 cycles per step): two gfx950 row
// swaps fold the four 16-lane rows, four DPP row rotations (row_ror:8,4,2,1) finish inside a row.
#define ABOPT_DPP_ROR(x, n) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x120 + (n), 0xf, 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    v += ABOPT_DPP_ROR(v, 8); v += ABOPT_DPP_ROR(v, 4); v += ABOPT_DPP_ROR(v, 2); v += ABOPT_DPP_ROR(v, 1);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
    v = fmaxf(v, ABOPT_DPP_ROR(v, 8)); v = fmaxf(v, ABOPT_DPP_ROR(v, 4)); v = fmaxf(v, ABOPT_DPP_ROR(v, 2)); v = fmaxf(v, ABOPT_DPP_ROR(v, 1));
    return v;
}

// ---------------------------------------------------------------- 3x3 helpers (row-major float[9])
struct Mat3 { float m[9]; };
struct Vec3 { float x, y, z; };

__device__ __forceinline__ Mat3 matmul3(const Mat3& a, const Mat3& b) {
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c.m[i * 3 + j] = a.m[i * 3 + 0] * b.m[0 * 3 + j] + a.m[i * 3 + 1] * b.m[1 * 3 + j] + a.m[i * 3 + 2] * b.m[2 * 3 + j];
    return c;
}

// exp map, reference so3.py:33-57.  The 'skew' layout there is rows (0,z,-y),(-z,0,x),(y,-x,0).
__device__ __forceinline__ Mat3 so3_exp(float x, float y, float z) {
    Mat3 S = {{0.f, z, -y, -z, 0.f, x, y, -x, 0.f}};
    float th = sqrtf(x * x + y * y + z * z);
    float b = (sinf(th) + 1e-8f) / (th + 1e-8f);
    float c = (1.f - cosf(th) + 1e-8f) / (th * th + 2e-8f);
    Mat3 S2 = matmul3(S, S);
    Mat3 R;
#pragma unroll
    for (int i = 0; i < 9; ++i) R.m[i] = ((i == 0 || i == 4 || i == 8) ? 1.f : 0.f) + b * S.m[i] + c * S2.m[i];
    return R;
}

// log map, reference so3.py:10-30,60-63.  min_cos = -0.999 when the reference runs with autograd on.
__device__ __forceinline__ Vec3 so3_log(const Mat3& R, bool grad_mode) {
    float tr = R.m[0] + R.m[4] + R.m[8];
    float cmin = grad_mode ? -0.999f : -1.0f;
    float ct = fmaxf((tr - 1.f) / 2.f, cmin);
    float st = sqrtf(1.f - ct * ct);
    float th = acosf(ct);
    float coef = (th + 1e-8f) / (2.f * st + 2e-8f);
    Vec3 w;
    w.x = coef * (R.m[1 * 3 + 2] - R.m[2 * 3 + 1]);
    w.y = coef * (R.m[2 * 3 + 0] - R.m[0 * 3 + 2]);
    w.z = coef * (R.m[0 * 3 + 1] - R.m[1 * 3 + 0]);
    return w;
}

// (1 + b i + c j + d k) -> R, reference geometry.py:215-233.
__device__ __forceinline__ Mat3 quat1ijk_to_rot(float qb, float qc, float qd) {
    float s = sqrtf(1.f + qb * qb + qc * qc + qd * qd);
    float a = 1.f / s, b = qb / s, c = qc / s, d = qd / s;
    Mat3 o = {{a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c,
               2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b,
               2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d}};
    return o;
}

// ReLU that lets a NaN through, like torch.relu (v_max_f32 returns the other operand): the range guard of the two-term fp16 layers needs an overflow (inf -> NaN in the
// products) to REACH the heads' outputs, and the reference's own relu(NaN) is NaN.  One compare + select.
__device__ __forceinline__ float relu_nan(float x) { return x < 0.f ? 0.f : x; }

// Geometric epilogue of the three denoiser heads for ONE residue i (dpm_full.py:95-107): eps_pos = gen ? R eps_crd : 0;
// R_next = R * U(eps_rot); v_next = gen ? log(R_next) : v_t; c = softmax(seq logits).  crd / rot / seq point at the row's head outputs
// (global memory: heads_epilogue_kernel; LDS: the tail of heads_mlp_kernel).  seq == nullptr: training path, the softmax stays in autograd.
__device__ __forceinline__ void heads_epilogue_row(int64_t i, const float* __restrict__ R, const float* __restrict__ v_t, const float* crd, const float* rot,
                                                   const float* seq, const uint8_t* __restrict__ mask_generate, float* __restrict__ v_next,
                                                   float* __restrict__ R_next, float* __restrict__ eps_pos, float* __restrict__ c_den, int grad_mode,
                                                   unsigned* __restrict__ nonfinite = nullptr) {
    const bool gen = mask_generate[i] != 0;
    // Range guard (round 6): the dense layers multiply on two fp16 terms per operand, so an activation beyond 65504 becomes inf and reaches the heads'
    // outputs as inf / NaN through every path (LayerNorm, attention); a non-finite head output of ANY row raises the flag the host reads once per call
    // (abopt_nonfinite_flag) and answers with the fp32-range GEMM path.  One compare per row; the racing stores all write 1.
    if (nonfinite && !(fabsf(crd[0] + crd[1] + crd[2] + rot[0] + rot[1] + rot[2]) < INFINITY)) *nonfinite = 1u;
    Mat3 Rm;
#pragma unroll
    for (int k = 0; k < 9; ++k) Rm.m[k] = R[i * 9 + k];
    const float cx = crd[0], cy = crd[1], cz = crd[2];
    // apply_rotation_to_vector = R p + 0 (geometry.py:116-117)
    eps_pos[i * 3 + 0] = gen ? (Rm.m[0] * cx + Rm.m[1] * cy + Rm.m[2] * cz + 0.f) : 0.f;
    eps_pos[i * 3 + 1] = gen ? (Rm.m[3] * cx + Rm.m[4] * cy + Rm.m[5] * cz + 0.f) : 0.f;
    eps_pos[i * 3 + 2] = gen ? (Rm.m[6] * cx + Rm.m[7] * cy + Rm.m[8] * cz + 0.f) : 0.f;
    const Mat3 U = quat1ijk_to_rot(rot[0], rot[1], rot[2]);
    const Mat3 Rn = matmul3(Rm, U);
#pragma unroll
    for (int k = 0; k < 9; ++k) R_next[i * 9 + k] = Rn.m[k];
    if (v_next) {
        const Vec3 w = so3_log(Rn, grad_mode != 0);
        v_next[i * 3 + 0] = gen ? w.x : v_t[i * 3 + 0];
        v_next[i * 3 + 1] = gen ? w.y : v_t[i * 3 + 1];
        v_next[i * 3 + 2] = gen ? w.z : v_t[i * 3 + 2];
    }
    if (!seq) return;
    float lgt[ABOPT_AA], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < ABOPT_AA; ++k) { lgt[k] = seq[k]; mx = fmaxf(mx, lgt[k]); }
    if (nonfinite && !(fabsf(mx) < INFINITY)) *nonfinite = 1u;
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < ABOPT_AA; ++k) { lgt[k] = expf(lgt[k] - mx); sm += lgt[k]; }
#pragma unroll
    for (int k = 0; k < ABOPT_AA; ++k) c_den[i * ABOPT_AA + k] = lgt[k] / sm;
}

// general quaternion (real first) -> R with normalisation, reference geometry.py:148-175.
__device__ __forceinline__ Mat3 quat_to_rot(float r, float i, float j, float k) {
    float n = fmaxf(sqrtf(r * r + i * i + j * j + k * k), 1e-12f);
    r /= n; i /= n; j /= n; k /= n;
    float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    Mat3 o = {{1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
               two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
               two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)}};
    return o;
}

// ---------------------------------------------------------------- Philox4x32-10 (Salmon et al. 2011)
struct Philox {
    uint32_t key0, key1;
    __device__ __forceinline__ Philox(uint64_t seed) : key0((uint32_t)seed), key1((uint32_t)(seed >> 32)) {}
    __device__ __forceinline__ uint4 operator()(uint64_t ctr_lo, uint64_t ctr_hi) const {
        uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
        uint32_t k0 = key0, k1 = key1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }           // [0,1)
__device__ __forceinline__ float u01_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); } // (0,1)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    float r = sqrtf(-2.0f * logf(u01_open(a)));
    float ph = 6.283185307179586f * u01(b);
    n0 = r * cosf(ph);
    n1 = r * sinf(ph);
}

}  // namespace abopt

// Device code of tools/micro/dih_share.hip (first form: no diagnostic stores) for assembler-level experiments: hipcc -S --cuda-device-only, edit, assemble, load as a module
// (dih_driver.cpp).  MODE as in dih_share.hip: 0 literal | 16 raw v_sqrt_f32 + six IEEE divisions | 32 raw v_sqrt_f32 + two IEEE divisions.
#include <hip/hip_runtime.h>
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float norm3(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

template <int MODE>
__device__ __forceinline__ float dihedral(V3 p0, V3 p1, V3 p2, V3 p3) {
    const V3 v0 = p2 - p1, v1 = p0 - p1, v2 = p3 - p2;
    const V3 u1 = cross3(v0, v1), u2 = cross3(v0, v2);
    const float sd = dot3(cross3(v1, v2), v0);
    float c;
    if constexpr (MODE & 16) {
        const float l1 = __builtin_amdgcn_sqrtf(dot3(u1, u1)), l2 = __builtin_amdgcn_sqrtf(dot3(u2, u2));
        const V3 n1 = v3(u1.x / l1, u1.y / l1, u1.z / l1), n2 = v3(u2.x / l2, u2.y / l2, u2.z / l2);
        c = dot3(n1, n2);
    } else if constexpr (MODE & 32) {
        const float l1 = __builtin_amdgcn_sqrtf(dot3(u1, u1)), l2 = __builtin_amdgcn_sqrtf(dot3(u2, u2));
        c = dot3(u1 * (1.f / l1), u2 * (1.f / l2));
    } else {
        const float l1 = norm3(u1), l2 = norm3(u2);
        const V3 n1 = v3(u1.x / l1, u1.y / l1, u1.z / l1), n2 = v3(u2.x / l2, u2.y / l2, u2.z / l2);
        c = dot3(n1, n2);
    }
    c = fminf(fmaxf(c, -0.999999f), 0.999999f);
    const float ac = acosf(c);
    const float sgn = (sd > 0.f) ? 1.f : ((sd < 0.f) ? -1.f : 0.f);
    const float d = sgn * ac;
    return (d != d) ? 0.f : d;
}

constexpr int SETS = 16;
template <int MODE, int LOADS = 0>
__global__ __launch_bounds__(256) void dih_kernel(const float4* __restrict__ pts, float* __restrict__ out, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int w = t >> 6;
    const float4 a0 = pts[(w % n) * 4 + 0], a1 = pts[(w % n) * 4 + 1], a2 = pts[(w % n) * 4 + 2];
    float acc0 = 0.f, acc1 = 0.f;
    for (int s = 0; s < SETS; ++s) {
        const int j = (t * SETS + s) % n;
        float4 b0, b1, b2;
        if constexpr (LOADS == 0) { b0 = pts[j * 4 + 0]; b1 = pts[j * 4 + 1]; b2 = pts[j * 4 + 2]; }             // only x, y, z are used below: hipcc emits global_load_dwordx3
        else if constexpr (LOADS == 1) {                                                                           // all four components used: global_load_dwordx4
            b0 = pts[j * 4 + 0]; b1 = pts[j * 4 + 1]; b2 = pts[j * 4 + 2];
            acc1 += (b0.w + b1.w + b2.w) * 1e-30f;
        } else {                                                                                                   // dwordx2 + dword
            const float* q = reinterpret_cast<const float*>(pts + j * 4);
            const float2 c0 = *reinterpret_cast<const float2*>(q), c1 = *reinterpret_cast<const float2*>(q + 4), c2 = *reinterpret_cast<const float2*>(q + 8);
            b0 = make_float4(c0.x, c0.y, q[2], 0.f); b1 = make_float4(c1.x, c1.y, q[6], 0.f); b2 = make_float4(c2.x, c2.y, q[10], 0.f);
        }
        const float x0 = dihedral<MODE>(v3(a2.x, a2.y, a2.z), v3(b0.x, b0.y, b0.z), v3(b1.x, b1.y, b1.z), v3(b2.x, b2.y, b2.z));
        const float x1 = dihedral<MODE>(v3(a0.x, a0.y, a0.z), v3(a1.x, a1.y, a1.z), v3(a2.x, a2.y, a2.z), v3(b0.x, b0.y, b0.z));
        out[(size_t)(t * SETS + s) * 2 + 0] = x0;
        out[(size_t)(t * SETS + s) * 2 + 1] = x1;
        acc0 += x0; acc1 += x1;
    }
    if (acc0 == 1.2345e-30f && acc1 == 5.4321e-30f) out[0] = 0.f;
}
template __global__ void dih_kernel<0>(const float4*, float*, int);
template __global__ void dih_kernel<16>(const float4*, float*, int);
template __global__ void dih_kernel<32>(const float4*, float*, int);
template __global__ void dih_kernel<32, 1>(const float4*, float*, int);
template __global__ void dih_kernel<32, 2>(const float4*, float*, int);
template __global__ void dih_kernel<0, 1>(const float4*, float*, int);
template __global__ void dih_kernel<0, 2>(const float4*, float*, int);

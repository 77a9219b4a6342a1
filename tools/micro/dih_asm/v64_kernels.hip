// Victims of other 64-bit-wide VALU classes for partner_classes.cpp (same launch shape and output layout as dih_kernels.hip): f64 arithmetic, 64-bit integer arithmetic, and -- as the
// positive control -- explicit packed-FP32 arithmetic.   hipcc --genco --offload-arch=gfx950 -O3 -o v64.hsaco v64_kernels.hip
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int SETS = 16;
extern "C" __global__ __launch_bounds__(256) void victim_f64(const float4* __restrict__ pts, float* __restrict__ out, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int s = 0; s < SETS; ++s) {
        const float4 b = pts[((t * SETS + s) % n) * 4];
        double x = b.x, y = b.y, z = b.z;
        for (int i = 0; i < 12; ++i) { x = __builtin_fma(x, 0.99990001, y * 1.0000003); y = __builtin_fma(y, z, -x) * 0.015625; z = z + x * 1e-3 - y; }
        out[(size_t)(t * SETS + s) * 2 + 0] = (float)x;
        out[(size_t)(t * SETS + s) * 2 + 1] = (float)(y + z);
    }
}
extern "C" __global__ __launch_bounds__(256) void victim_u64(const float4* __restrict__ pts, float* __restrict__ out, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int s = 0; s < SETS; ++s) {
        const float4 b = pts[((t * SETS + s) % n) * 4];
        unsigned long long x = __float_as_uint(b.x), y = __float_as_uint(b.y) | 1ull;
        for (int i = 0; i < 12; ++i) { x = x * 0x9E3779B97F4A7C15ull + (y << 3); y = (y << 5) + (x >> 7) + (unsigned)(x * (unsigned)y); }
        out[(size_t)(t * SETS + s) * 2 + 0] = __uint_as_float((unsigned)(x >> 9) & 0x3fffffffu);
        out[(size_t)(t * SETS + s) * 2 + 1] = __uint_as_float((unsigned)(y >> 11) & 0x3fffffffu);
    }
}
extern "C" __global__ __launch_bounds__(256) void victim_pk(const float4* __restrict__ pts, float* __restrict__ out, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int s = 0; s < SETS; ++s) {
        const float4 b = pts[((t * SETS + s) % n) * 4];
        f32x2 x = {b.x, b.y}, y = {b.z, b.x};
        for (int i = 0; i < 12; ++i) { x = x * 0.9999f + y * 1.0000003f; y = (y * x - x) * 0.015625f; }
        out[(size_t)(t * SETS + s) * 2 + 0] = x[0] + y[1];
        out[(size_t)(t * SETS + s) * 2 + 1] = x[1] + y[0];
    }
}
extern "C" __global__ __launch_bounds__(256) void victim_f32(const float4* __restrict__ pts, float* __restrict__ out, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int s = 0; s < SETS; ++s) {
        const float4 b = pts[((t * SETS + s) % n) * 4];
        float x = b.x, y = b.y, z = b.z;
        for (int i = 0; i < 24; ++i) { x = __builtin_fmaf(x, 0.9999f, y * 1.0000003f); y = __builtin_fmaf(y, z, -x) * 0.015625f; z = z + x * 1e-3f - y; }
        out[(size_t)(t * SETS + s) * 2 + 0] = x;
        out[(size_t)(t * SETS + s) * 2 + 1] = y + z;
    }
}

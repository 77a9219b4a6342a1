// Does the gfx950 matrix pipe keep fp16 SUBNORMAL inputs (v_mfma_f32_16x16x32_f16 / 32x32x16_f16), and do v_cvt_pk_f16_f32 / v_cvt_f32_f16 produce / read them
// under the default MODE of a HIP kernel?  (The two-term fp16 scheme of the dense layers puts the low terms of small activations there.)
//   hipcc --offload-arch=gfx950 -O3 -o f16_denorm f16_denorm.hip && ./f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out, unsigned* bits) {
    unsigned pa, pb;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pa) : "v"(a_val));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pb) : "v"(b_val));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 A = {pa, pa, pa, pa}, B = {pb, pb, pb, pb};
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), c, 0, 0, 0);
    f32x16 d;
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), d, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0]; out[1] = d[0];
        bits[0] = pa;
        out[2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(pa & 0xffff));
        out[3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(pa >> 16));
    }
}
int main() {
    float* out; unsigned* bits;
    hipMalloc(&out, 64); hipMalloc(&bits, 64);
    const float avals[] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */, 3.0517578125e-05f /* 2^-15 */, 1.0f, 1.00048828125f /* 1 + 2^-11: tie */, 70000.f};
    for (float a : avals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1024.f, out, bits);
        float h[4]; unsigned b;
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); hipMemcpy(&b, bits, 4, hipMemcpyDeviceToHost);
        printf("a = %.10g: cvt bits 0x%08x, back lo %.10g hi %.10g | 16x16x32 sum %.10g (expect %.10g) | 32x32x16 sum %.10g (expect %.10g)\n", a, b, h[2], h[3], h[0], 32.0 * h[2] * 1024.0, h[1], 16.0 * h[2] * 1024.0);
    }
    return 0;
}

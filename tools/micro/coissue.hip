// Do VALU instructions of one wave issue under the MFMAs of another wave of the same SIMD on gfx950?  Workgroups of 4 or 8 waves (1 or 2 per
// SIMD); each wave runs either a chain-free MFMA stream (v_mfma_f32_16x16x32_bf16, 4 passes, or v_mfma_f32_32x32x16_bf16, 8 passes; four
// independent accumulators) or a VALU stream (eight independent v_fma chains).  Prints cycles per instruction for each mix.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// role per wave: 0 idle, 1 MFMA 16x16x32, 2 VALU, 3 MFMA 32x32x16, 4 one wave alternating 1 MFMA 16x16x32 : 3 VALU
__global__ __launch_bounds__(512) void mix(int role_lo, int role_hi, int iters, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wave < 4 ? role_lo : role_hi;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0.f; d1[i] = 0.f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + lane * 0.01f + i;
    const float m = 1.0001f, q = 0.0003f;
    __syncthreads();
    const long long t0 = clock64();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], m, q);
        }
    } else if (role == 4) {
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); v[0] = __builtin_fmaf(v[0], m, q); v[1] = __builtin_fmaf(v[1], m, q); v[2] = __builtin_fmaf(v[2], m, q);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0); v[3] = __builtin_fmaf(v[3], m, q); v[4] = __builtin_fmaf(v[4], m, q); v[5] = __builtin_fmaf(v[5], m, q);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); v[6] = __builtin_fmaf(v[6], m, q); v[7] = __builtin_fmaf(v[7], m, q); v[0] = __builtin_fmaf(v[0], m, q);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0); v[1] = __builtin_fmaf(v[1], m, q); v[2] = __builtin_fmaf(v[2], m, q); v[3] = __builtin_fmaf(v[3], m, q);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 16; ++i) s += d0[i] + d1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 20000;
    struct { int lo, hi, waves; const char* what; } cases[] = {
        {1, 0, 4, "MFMA 16x16x32 alone (1 wave per SIMD)"}, {3, 0, 4, "MFMA 32x32x16 alone"}, {2, 0, 4, "VALU alone"},
        {1, 2, 8, "MFMA 16x16x32 + VALU wave on the same SIMD"}, {3, 2, 8, "MFMA 32x32x16 + VALU wave on the same SIMD"},
        {1, 1, 8, "two MFMA 16x16x32 waves per SIMD"}, {2, 2, 8, "two VALU waves per SIMD"}, {4, 0, 4, "one wave: 1 MFMA 16x16x32 : 3 VALU interleaved"}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(mix, dim3(256), dim3(c.waves * 64), 0, 0, c.lo, c.hi, iters, out, cyc);
        hipDeviceSynchronize();
        long long h[256 * 8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double lo = 0, hi = 0;
        for (int b = 0; b < 256; ++b) { lo += h[b * 8]; hi += h[b * 8 + 4]; }
        printf("%-55s waves 0-3: %7.2f cycles per loop trip", c.what, lo / 256 / iters);
        if (c.waves == 8) printf("   waves 4-7: %7.2f", hi / 256 / iters);
        printf("\n");
    }
    printf("(a trip = 4 MFMAs, or 16 v_fma, or 4 MFMAs + 12 v_fma)\n");
    return 0;
}

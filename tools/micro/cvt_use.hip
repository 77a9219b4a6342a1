// Does the VALU instruction right behind v_cvt_pk_bf16_f32 see its result on gfx950?  (round 5: kernels whose bf16 splits went through the
// compiler's own float -> bf16 conversion were non-deterministic; hipcc 7.2 puts the dependent v_lshlrev_b32 directly behind the conversion,
// while behind an asm statement it pads one state.)  Each lane converts pairs, then immediately uses the packed result; the same with one
// s_nop in between; mismatches are counted.  Several waves per SIMD and an MFMA stream next to it vary the issue timing.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cvt_use.hip -o /tmp/cvt_use && /tmp/cvt_use
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024) void k(const float* __restrict__ x, unsigned* __restrict__ bad, int iters, int with_mfma) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, wave = threadIdx.x >> 6;
    float a = x[tid * 2], b = x[tid * 2 + 1];
    unsigned nbad = 0;
    if (with_mfma && (wave & 1)) {                       // every other wave keeps the matrix pipe busy
        bf16x8 p, q;
        for (int i = 0; i < 8; ++i) { p[i] = (__bf16)a; q[i] = (__bf16)b; }
        f32x4 c = {0, 0, 0, 0};
        for (int it = 0; it < iters * 4; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, c, 0, 0, 0);
        if (c[0] == 123.456f) bad[0] = 1;
        return;
    }
    for (int it = 0; it < iters; ++it) {
        unsigned h0, lo0, h1, lo1;
        // back to back: the conversion, then a shift of its result in the very next instruction
        asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_lshlrev_b32 %1, 16, %0" : "=&v"(h0), "=v"(lo0) : "v"(a), "v"(b));
        // one wait state in between
        asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\ts_nop 0\n\tv_lshlrev_b32 %1, 16, %0" : "=&v"(h1), "=v"(lo1) : "v"(a), "v"(b));
        nbad += (lo0 != lo1) | (h0 != h1) | (lo0 != (h0 << 16));
        a = a * 1.0001f + 0.5f; b = b * 0.9999f - 0.25f;
    }
    if (nbad) atomicAdd(&bad[1], nbad);
}
int main() {
    const int blocks = 512, n = blocks * 1024;
    float* x; unsigned* bad;
    hipMalloc(&x, n * 2 * 4); hipMalloc(&bad, 8);
    float* hx = new float[n * 2];
    for (int i = 0; i < n * 2; ++i) hx[i] = (float)((i * 2654435761u) % 100003) * 0.01f - 400.f;
    hipMemcpy(x, hx, n * 2 * 4, hipMemcpyHostToDevice);
    for (int mf = 0; mf < 2; ++mf) {
        hipMemset(bad, 0, 8);
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, x, bad, 2000, mf);
        hipDeviceSynchronize();
        unsigned hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
        printf("v_cvt_pk_bf16_f32 followed directly by a VALU read of its result, %s: %u mismatching lane-iterations of %.1f G\n",
               mf ? "MFMA waves on the same SIMDs" : "VALU only", hb[1], 5.0 * n * 2000 * (mf ? 0.5 : 1.0) / 1e9);
    }
    return 0;
}

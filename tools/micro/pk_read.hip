// Does v_pk_add_f32 see a register written by the VALU instruction right in front of it on gfx950?  hipcc 7.2 emits (SLP-vectorised bf16 splits)
//     v_lshlrev_b32 v150, 16, v148 ; v_and_b32 v151, 0xffff0000, v148 ; v_pk_add_f32 v[146:147], v[146:147], v[150:151] neg_lo:[0,1] neg_hi:[0,1]
// with no wait state in between; kernels built that way were non-deterministic in round 5, the same kernels with -fno-slp-vectorize were not.
// Per lane and iteration: the pair v[100:101] is produced by two VALU instructions from a packed bf16 pair, the packed subtraction follows after
// K wait states (K = 0, 1, 2, 4); its result is compared with the scalar subtractions.  Second experiment: the conversion itself in front.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_read.hip -o /tmp/pk_read && /tmp/pk_read
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106"
#define BODY(NOPS)                                                                                               \
    asm volatile("v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\ts_nop 4\n\t"                                       \
                 "v_cvt_pk_bf16_f32 v106, v102, v103\n\t"                                                        \
                 "v_lshlrev_b32 v100, 16, v106\n\tv_and_b32 v101, 0xffff0000, v106\n\t" NOPS                     \
                 "v_pk_add_f32 v[104:105], v[102:103], v[100:101] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 4\n\t"      \
                 "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105" : "=v"(r0), "=v"(r1) : "v"(a), "v"(b) : CLOB);
__device__ unsigned g_bad[4];
__global__ __launch_bounds__(1024) void k(const float* __restrict__ x, int iters, int busy) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, wave = threadIdx.x >> 6;
    float a = x[tid * 2], b = x[tid * 2 + 1];
    if (busy && (wave & 1)) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 p, q; for (int i = 0; i < 8; ++i) { p[i] = (__bf16)a; q[i] = (__bf16)b; }
        f32x4 c = {0, 0, 0, 0};
        for (int it = 0; it < iters * 8; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, c, 0, 0, 0);
        if (c[0] == 123.456f) g_bad[0] = 0xffffffffu;
        return;
    }
    unsigned bad[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        float r0, r1;
        const unsigned h = __builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
        const float e0 = a - __uint_as_float(h << 16), e1 = b - __uint_as_float(h & 0xffff0000u);
        BODY("") bad[0] += (r0 != e0) | (r1 != e1);
        BODY("s_nop 0\n\t") bad[1] += (r0 != e0) | (r1 != e1);
        BODY("s_nop 1\n\t") bad[2] += (r0 != e0) | (r1 != e1);
        BODY("s_nop 3\n\t") bad[3] += (r0 != e0) | (r1 != e1);
        a = a * 1.0001f + 0.5f; b = b * 0.9999f - 0.25f;
    }
    for (int i = 0; i < 4; ++i) if (bad[i]) atomicAdd(&g_bad[i], bad[i]);
}
int main() {
    const int blocks = 512, n = blocks * 1024;
    float* x; hipMalloc(&x, n * 2 * 4);
    float* hx = new float[n * 2];
    for (int i = 0; i < n * 2; ++i) hx[i] = (float)((i * 2654435761u) % 100003) * 0.01f - 400.f;
    hipMemcpy(x, hx, n * 2 * 4, hipMemcpyHostToDevice);
    for (int busy = 0; busy < 2; ++busy) {
        unsigned z[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_bad), z, 16);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, x, 1000, busy);
        hipDeviceSynchronize();
        unsigned h[4]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bad), 16);
        printf("v_lshlrev / v_and -> v_pk_add_f32 reading both (%s): wrong lane-iterations with 0 / 1 / 2 / 4 wait states: %u %u %u %u of %.2f G\n",
               busy ? "MFMA waves on the same SIMDs" : "VALU only", h[0], h[1], h[2], h[3], 3.0 * n * 1000 * (busy ? 0.5 : 1.0) / 1e9);
    }
    return 0;
}

// Write-after-read on a lane mask: a VALU instruction that reads an SGPR pair / VCC as its lane mask (v_cndmask_b32, v_div_fmas_f32), followed K wait states later by an
// instruction that OVERWRITES that mask (s_mov_b64, or a VALU compare) -- the pattern hipcc emits all over its division / sqrtf / select code, with K = 0.
// gfx11 documents this as a hazard in wave64 (LLVM: VALUMaskWriteHazard); for gfx9 / gfx940 no rule exists.  Does the last quarter-wave of the reader see the new mask on
// gfx950 -- alone, and beside other waves?
//   hipcc --offload-arch=gfx950 -O3 -o sgpr_war tools/micro/sgpr_war.hip ;  ./sgpr_war <seconds> [own_partner]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

// mask m (random per wave, from a compare), reader r = m ? a : b, then the mask register is overwritten with ~m K states later; expected r is formed from a copy of m
#define CASE_S(NOPS, SLOT)                                                                                            \
    {                                                                                                                 \
        float r_; unsigned long long m_, k_;                                                                          \
        asm volatile("v_cmp_gt_f32_e64 %1, %3, %4\n\ts_nop 7\n\ts_mov_b64 %2, %1\n\ts_nop 3\n\t"                        \
                     "v_cndmask_b32_e64 %0, %5, %6, %1\n\t" NOPS "s_not_b64 %1, %1\n\t"                                 \
                     : "=&v"(r_), "=&s"(m_), "=&s"(k_) : "v"(x), "v"(thr), "v"(a), "v"(b) : "scc");                     \
        asm volatile("s_nop 15" ::: "memory");                                                                         \
        const float e_ = ((k_ >> lane) & 1ull) ? b : a;                                                               \
        if (__float_as_uint(e_) != __float_as_uint(r_)) atomicAdd(&cnt[(SLOT) * 4 + (lane >> 4)], 1u);                \
    }
// the same with a VALU writer of the mask behind the reader
#define CASE_V(NOPS, SLOT)                                                                                            \
    {                                                                                                                 \
        float r_; unsigned long long m_, k_;                                                                          \
        asm volatile("v_cmp_gt_f32_e64 %1, %3, %4\n\ts_nop 7\n\ts_mov_b64 %2, %1\n\ts_nop 3\n\t"                        \
                     "v_cndmask_b32_e64 %0, %5, %6, %1\n\t" NOPS "v_cmp_le_f32_e64 %1, %3, %4\n\t"                      \
                     : "=&v"(r_), "=&s"(m_), "=&s"(k_) : "v"(x), "v"(thr), "v"(a), "v"(b));                             \
        asm volatile("s_nop 15" ::: "memory");                                                                         \
        const float e_ = ((k_ >> lane) & 1ull) ? b : a;                                                               \
        if (__float_as_uint(e_) != __float_as_uint(r_)) atomicAdd(&cnt[(SLOT) * 4 + (lane >> 4)], 1u);                \
    }
// VCC: v_cndmask_b32_e32 (implicit VCC) then s_mov_b64 vcc
#define CASE_C(NOPS, SLOT)                                                                                            \
    {                                                                                                                 \
        float r_; unsigned long long k_;                                                                              \
        asm volatile("v_cmp_gt_f32_e32 vcc, %2, %3\n\ts_nop 7\n\ts_mov_b64 %1, vcc\n\ts_nop 3\n\t"                      \
                     "v_cndmask_b32_e32 %0, %4, %5, vcc\n\t" NOPS "s_not_b64 vcc, vcc\n\t"                              \
                     : "=&v"(r_), "=&s"(k_) : "v"(x), "v"(thr), "v"(a), "v"(b) : "vcc", "scc");                         \
        asm volatile("s_nop 15" ::: "memory");                                                                         \
        const float e_ = ((k_ >> lane) & 1ull) ? b : a;                                                               \
        if (__float_as_uint(e_) != __float_as_uint(r_)) atomicAdd(&cnt[(SLOT) * 4 + (lane >> 4)], 1u);                \
    }
#define KERNEL(NAME, C)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(const float* __restrict__ in, unsigned* __restrict__ cnt, int iters) { \
        const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;                                        \
        for (int it = 0; it < iters; ++it) {                                                                          \
            const float x = in[(t * 61 + it * 977) & 65535], thr = in[(t * 17 + it * 331 + 7) & 65535], a = x + 1.f, b = thr + 2.f; \
            C("", 0) C("s_nop 0\n\t", 1) C("s_nop 1\n\t", 2) C("s_nop 3\n\t", 3) C("s_nop 7\n\t", 4)                    \
        }                                                                                                             \
    }
KERNEL(k_salu, CASE_S)
KERNEL(k_valu, CASE_V)
KERNEL(k_vcc, CASE_C)

__global__ __launch_bounds__(256) void k_partner(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;
    for (int i = 0; i < iters; ++i) {
        a = __builtin_amdgcn_exp2f(a) * 0.25f; b = __builtin_amdgcn_exp2f(b) * 0.25f; c = __builtin_amdgcn_rsqf(c + 1.f); d = __builtin_amdgcn_rcpf(d + 1.f);
    }
    if (a + b + c + d == 1.2345e-30f) out[0] = a;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int own = argc > 2 ? atoi(argv[2]) : 0;
    float* in; unsigned* cnt; float* pout;
    (void)hipMalloc(&in, 65536 * 4); (void)hipMalloc(&cnt, 64 * 4); (void)hipMalloc(&pout, 4);
    float* h = (float*)malloc(65536 * 4);
    unsigned st = 777u;
    for (int i = 0; i < 65536; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 65536.0f; }
    (void)hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2;
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const char* names[] = {"v_cndmask(s[n:n+1]) ; s_not_b64 s[n:n+1]", "v_cndmask(s[n:n+1]) ; v_cmp_e64 s[n:n+1]", "v_cndmask_e32(vcc) ; s_not_b64 vcc"};
    void (*kern[])(const float*, unsigned*, int) = {k_salu, k_valu, k_vcc};
    const int states[] = {0, 1, 2, 4, 8};
    for (int k = 0; k < 3; ++k) {
        (void)hipMemset(cnt, 0, 64 * 4);
        long launches = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            if (own) hipLaunchKernelGGL(k_partner, dim3(1024), dim3(256), 0, s2, pout, 20000);
            for (int j = 0; j < 20; ++j) hipLaunchKernelGGL(kern[k], dim3(64), dim3(256), 0, s1, in, cnt, 256);
            (void)hipStreamSynchronize(s1);
            launches += 20;
        }
        (void)hipDeviceSynchronize();
        unsigned hc[64];
        (void)hipMemcpy(hc, cnt, 64 * 4, hipMemcpyDeviceToHost);
        printf("%-44s %ld launches x 16384 threads x 256; wrong selects per quarter-wave [lanes 0-15, 16-31, 32-47, 48-63]:\n", names[k], launches);
        for (int c = 0; c < 5; ++c) printf("    %2d wait states: %u %u %u %u\n", states[c], hc[c * 4], hc[c * 4 + 1], hc[c * 4 + 2], hc[c * 4 + 3]);
        fflush(stdout);
    }
    return 0;
}

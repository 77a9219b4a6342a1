// How many wait states does gfx950 need between v_mfma_f32_16x16x32_bf16 and a VALU instruction that touches its registers?  hipcc 7.2 keeps
// >= 3 states before a VALU overwrite of SrcC and >= 8 before a VALU read / overwrite of D (tools/r05/mfma_hazard_scan.py on its own code);
// round 5's bf16-term pair embedding was wrong with exactly those distances.  Each test is ONE asm block on fixed registers:
//   warc_dep   MFMA_a writes C;  MFMA_b = A.B + C (must wait for MFMA_a);  K states;  v_mov C[0] <- poison.   Expect D_b = 2 A.B + c0.
//   warc       MFMA = A.B + C;  K states;  v_mov C[0] <- poison.                                            Expect D = A.B + c0.
//   raw        MFMA -> D;  K states;  v_mov out <- D[0].                                                      Expect out = A.B + c0.
//   waw        MFMA -> D;  K states;  v_mov D[0] <- 777;  later read D[0].                                    Expect 777.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_hazard.hip -o /tmp/mfma_hazard && /tmp/mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"
#define SETUP "v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n" \
              "v_mov_b32 v104, %2\n v_mov_b32 v105, %2\n v_mov_b32 v106, %2\n v_mov_b32 v107, %2\n" \
              "v_mov_b32 v112, %2\n v_mov_b32 v113, %2\n v_mov_b32 v114, %2\n v_mov_b32 v115, %2\n" \
              "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n s_nop 7\n s_nop 7\n"
#define TAIL "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"
#define NOPS0 ""
#define NOPS1 "s_nop 0\n"
#define NOPS2 "s_nop 1\n"
#define NOPS3 "s_nop 2\n"
#define NOPS4 "s_nop 3\n"
#define NOPS6 "s_nop 5\n"
#define NOPS8 "s_nop 7\n"
#define NOPS12 "s_nop 7\n s_nop 3\n"
#define NOPS16 "s_nop 7\n s_nop 7\n"
template <int K> struct T;
#define DEF(K)                                                                                                                         \
template <> struct T<K> {                                                                                                              \
    static __device__ float warc(float c0, unsigned ab, float poison) { float o;                                                       \
        asm volatile(SETUP "v_mfma_f32_16x16x32_bf16 v[108:111], v[104:107], v[112:115], v[100:103]\n" NOPS##K "v_mov_b32 v100, %3\n" TAIL "v_mov_b32 %0, v108\n" \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(poison) : CLOB); return o; }                                                     \
    static __device__ float warc_dep(float c0, unsigned ab, float poison) { float o;                                                   \
        asm volatile(SETUP "v_mfma_f32_16x16x32_bf16 v[100:103], v[104:107], v[112:115], v[100:103]\n"                                 \
                           "v_mfma_f32_16x16x32_bf16 v[108:111], v[104:107], v[112:115], v[100:103]\n" NOPS##K "v_mov_b32 v100, %3\n" TAIL "v_mov_b32 %0, v108\n" \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(poison) : CLOB); return o; }                                                     \
    static __device__ float raw(float c0, unsigned ab, float poison) { float o;                                                        \
        asm volatile(SETUP "v_mfma_f32_16x16x32_bf16 v[108:111], v[104:107], v[112:115], v[100:103]\n" NOPS##K "v_mov_b32 %0, v108\n" TAIL \
                     : "=&v"(o) : "v"(c0), "v"(ab), "v"(poison) : CLOB); return o; }                                                    \
    static __device__ float raw_pk(float c0, unsigned ab, float poison) { float o;                                                     \
        asm volatile(SETUP "v_mfma_f32_16x16x32_bf16 v[108:111], v[104:107], v[112:115], v[100:103]\n" NOPS##K "v_pk_add_f32 v[116:117], v[108:109], v[108:109]\n" TAIL "v_mov_b32 %0, v116\n" \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(poison) : CLOB); return o; }                                                     \
    static __device__ float waw_pk(float c0, unsigned ab, float poison) { float o;                                                     \
        asm volatile(SETUP "v_mov_b32 v118, %3\n v_mov_b32 v119, %3\n s_nop 4\n"                                                       \
                           "v_mfma_f32_16x16x32_bf16 v[108:111], v[104:107], v[112:115], v[100:103]\n" NOPS##K "v_pk_add_f32 v[108:109], v[118:119], v[118:119]\n" TAIL "v_mov_b32 %0, v108\n" \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(poison) : CLOB); return o; }                                                     \
    static __device__ float waw(float c0, unsigned ab, float poison) { float o;                                                        \
        asm volatile(SETUP "v_mfma_f32_16x16x32_bf16 v[108:111], v[104:107], v[112:115], v[100:103]\n" NOPS##K "v_mov_b32 v108, %3\n" TAIL "v_mov_b32 %0, v108\n" \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(poison) : CLOB); return o; }                                                     \
};
DEF(0) DEF(1) DEF(2) DEF(3) DEF(4) DEF(6) DEF(8) DEF(12) DEF(16)
__device__ unsigned g_bad[6][9];
template <int K, int SLOT> __device__ void run(float c0, unsigned ab, float ab_dot, int busy) {
    const float poison = 777.f;
    for (int it = 0; it < 200; ++it) {
        if (T<K>::warc(c0, ab, poison) != ab_dot + c0) atomicAdd(&g_bad[0][SLOT], 1u);
        if (T<K>::warc_dep(c0, ab, poison) != 2.f * ab_dot + c0) atomicAdd(&g_bad[1][SLOT], 1u);
        if (T<K>::raw(c0, ab, poison) != ab_dot + c0) atomicAdd(&g_bad[2][SLOT], 1u);
        if (T<K>::waw(c0, ab, poison) != poison) atomicAdd(&g_bad[3][SLOT], 1u);
        if (T<K>::raw_pk(c0, ab, poison) != 2.f * (ab_dot + c0)) atomicAdd(&g_bad[4][SLOT], 1u);
        if (T<K>::waw_pk(c0, ab, poison) != 2.f * poison) atomicAdd(&g_bad[5][SLOT], 1u);
    }
}
__global__ __launch_bounds__(512) void k(int busy) {
    const int wave = threadIdx.x >> 6;
    if (busy && (wave & 1)) {                 // the other wave of each SIMD keeps the matrix pipe and the VALU busy
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 p, q; for (int i = 0; i < 8; ++i) { p[i] = (__bf16)1.f; q[i] = (__bf16)0.5f; }
        f32x4 c = {0, 0, 0, 0}; float v = threadIdx.x;
        for (int it = 0; it < 60000; ++it) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, c, 0, 0, 0); v = v * 1.0001f + 0.5f; }
        if (c[0] + v == 123.456f) g_bad[0][0] = 0xffffffffu;
        return;
    }
    // A = B = all 1.0 (bf16 0x3f80 twice per register): A.B over K = 32 -> 32
    const unsigned ab = 0x3f803f80u; const float ab_dot = 32.f, c0 = 5.f;
    run<0, 0>(c0, ab, ab_dot, busy); run<1, 1>(c0, ab, ab_dot, busy); run<2, 2>(c0, ab, ab_dot, busy); run<3, 3>(c0, ab, ab_dot, busy); run<4, 4>(c0, ab, ab_dot, busy);
    run<6, 5>(c0, ab, ab_dot, busy); run<8, 6>(c0, ab, ab_dot, busy); run<12, 7>(c0, ab, ab_dot, busy); run<16, 8>(c0, ab, ab_dot, busy);
}
int main() {
    const int ks[9] = {0, 1, 2, 3, 4, 6, 8, 12, 16};
    const char* names[6] = {"VALU overwrites SrcC (WAR)           ", "... of an MFMA waiting for its SrcC    ", "VALU reads D (RAW)                     ", "VALU overwrites D (WAW)                ",
                            "v_pk_add_f32 reads D (RAW)             ", "v_pk_add_f32 overwrites D (WAW)        "};
    for (int busy = 0; busy < 2; ++busy) {
        unsigned z[6][9] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(g_bad), z, sizeof(z));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, busy);
        hipDeviceSynchronize();
        unsigned h[6][9]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bad), sizeof(h));
        printf("v_mfma_f32_16x16x32_bf16, wrong results per wait states between the MFMA and the VALU instruction (%s):\n   states:", busy ? "a second wave per SIMD issuing MFMAs and VALU" : "one wave per SIMD");
        for (int i = 0; i < 9; ++i) printf(" %8d", ks[i]);
        printf("\n");
        for (int t = 0; t < 6; ++t) { printf("   %s", names[t]); for (int i = 0; i < 9; ++i) printf(" %8u", h[t][i]); printf("\n"); }
    }
    return 0;
}

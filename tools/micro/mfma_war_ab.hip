// Does a VALU write to the SrcA / SrcB register of a just-issued v_mfma_f32_16x16x32_f16 corrupt the product -- alone, and while ANOTHER wave of the
// SIMD keeps the matrix pipe busy with MFMAs of the same or of a longer latency (16x16x4 f32: 8 passes, 32x32x2 f32: 16 passes)?  The ISA manuals
// list no wait states for this pair (only for SrcC and D), hipcc pads nothing, and the pair embedding's two-term path does exactly this: the
// split of the next 16-pair tile lands in the registers that were the B operand of the previous tile's last products (DESIGN_LOG round 5).
//   wara / warb   MFMA D = A.B + C;  K states;  v_mov A[0] / B[0] <- other value;  expect D = A.B + c0 with the OLD operand
//   raw           MFMA -> D;  K states;  v_mov out <- D[0]                          expect A.B + c0
//   waw           MFMA -> D;  K states;  v_mov D[0] <- 777; later read D[0]         expect 777
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_war_ab.hip -o /tmp/mfma_war_ab && /tmp/mfma_war_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115"
// C = v[100:103] = c0, A = v[104:107] = ab, B = v[112:115] = ab, D = v[108:111]
#define SETUP "v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n" \
              "v_mov_b32 v104, %2\n v_mov_b32 v105, %2\n v_mov_b32 v106, %2\n v_mov_b32 v107, %2\n" \
              "v_mov_b32 v112, %2\n v_mov_b32 v113, %2\n v_mov_b32 v114, %2\n v_mov_b32 v115, %2\n" \
              "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n s_nop 7\n s_nop 7\n"
#define TAIL "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"
#define MF "v_mfma_f32_16x16x32_f16 v[108:111], v[104:107], v[112:115], v[100:103]\n"
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
    static __device__ float wara(float c0, unsigned ab, unsigned other) { float o;                                                     \
        asm volatile(SETUP MF NOPS##K "v_mov_b32 v104, %3\n v_mov_b32 v105, %3\n v_mov_b32 v106, %3\n v_mov_b32 v107, %3\n" TAIL "v_mov_b32 %0, v108\n" \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(other) : CLOB); return o; }                                                      \
    static __device__ float warb(float c0, unsigned ab, unsigned other) { float o;                                                     \
        asm volatile(SETUP MF NOPS##K "v_mov_b32 v112, %3\n v_mov_b32 v113, %3\n v_mov_b32 v114, %3\n v_mov_b32 v115, %3\n" TAIL "v_mov_b32 %0, v108\n" \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(other) : CLOB); return o; }                                                      \
    static __device__ float raw(float c0, unsigned ab, unsigned other) { float o;                                                      \
        asm volatile(SETUP MF NOPS##K "v_mov_b32 %0, v108\n" TAIL : "=&v"(o) : "v"(c0), "v"(ab), "v"(other) : CLOB); return o; }         \
    static __device__ float chain3(float c0, unsigned ab, unsigned other) { float o;     /* c = A.B + 0; c = A.B + c; c = A.B + c (in place, K states between) */ \
        asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[108:111], v[104:107], v[112:115], 0\n" NOPS##K                                    \
                           "v_mfma_f32_16x16x32_f16 v[108:111], v[104:107], v[112:115], v[108:111]\n" NOPS##K                          \
                           "v_mfma_f32_16x16x32_f16 v[108:111], v[104:107], v[112:115], v[108:111]\n" TAIL "v_mov_b32 %0, v108\n"    \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(other) : CLOB); return o; }                                                      \
    static __device__ float chain2x(float c0, unsigned ab, unsigned other) { float o;    /* d = A.B + c0 (into another register); e = A.B + d */ \
        asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[116:119], v[104:107], v[112:115], v[100:103]\n" NOPS##K                         \
                           "v_mfma_f32_16x16x32_f16 v[108:111], v[104:107], v[112:115], v[116:119]\n" TAIL "v_mov_b32 %0, v108\n"    \
                     : "=v"(o) : "v"(c0), "v"(ab), "v"(other) : CLOB, "v116", "v117", "v118", "v119"); return o; }                      \
    static __device__ float waw(float c0, unsigned ab, unsigned other) { float o;                                                      \
        asm volatile(SETUP MF NOPS##K "v_mov_b32 v108, 0x44424000\n" TAIL "v_mov_b32 %0, v108\n" : "=v"(o) : "v"(c0), "v"(ab), "v"(other) : CLOB); return o; } \
};
DEF(0) DEF(1) DEF(2) DEF(3) DEF(4) DEF(6) DEF(8) DEF(12) DEF(16)
__device__ unsigned g_bad[6][9];
template <int K, int SLOT> __device__ void run(float c0, unsigned ab, float ab_dot, int iters) {
    const unsigned other = 0x40004000u;       // 2.0, 2.0 in fp16: a product that read it gives 64 (one operand) per 32 of K instead of 32
    for (int it = 0; it < iters; ++it) {
        if (T<K>::wara(c0, ab, other) != ab_dot + c0) atomicAdd(&g_bad[0][SLOT], 1u);
        if (T<K>::warb(c0, ab, other) != ab_dot + c0) atomicAdd(&g_bad[1][SLOT], 1u);
        if (T<K>::raw(c0, ab, other) != ab_dot + c0) atomicAdd(&g_bad[2][SLOT], 1u);
        if (T<K>::waw(c0, ab, other) != 777.f) atomicAdd(&g_bad[3][SLOT], 1u);
        if (T<K>::chain3(c0, ab, other) != 3.f * ab_dot) atomicAdd(&g_bad[4][SLOT], 1u);
        if (T<K>::chain2x(c0, ab, other) != 2.f * ab_dot + c0) atomicAdd(&g_bad[5][SLOT], 1u);
    }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// partner: 0 none | 1 f16 16x16x32 (4 passes) | 2 f32 16x16x4 (8 passes) | 3 f32 32x32x2 (16 passes) | 4 VALU only
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(int partner, int testers_per_simd) {
    const int wave = threadIdx.x >> 6;
    // waves are dealt to the four SIMDs round-robin: wave w runs on SIMD w % 4; slot w / 4 of that SIMD.  Slots < testers_per_simd test, the others are partners
    const bool tester = (wave >> 2) < testers_per_simd;
    if (!tester) {
        if (partner == 0) return;
        float v = threadIdx.x;
        if (partner == 1) {
            f16x8 p, q; for (int i = 0; i < 8; ++i) { p[i] = (_Float16)1.f; q[i] = (_Float16)0.5f; }
            f32x4 c = {0, 0, 0, 0}, d = {1, 1, 1, 1};
            for (int it = 0; it < 40000; ++it) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(p, q, c, 0, 0, 0); d = __builtin_amdgcn_mfma_f32_16x16x32_f16(p, q, d, 0, 0, 0); v = v * 1.0001f + 0.5f; }
            if (c[0] + d[0] + v == 123.456f) g_bad[0][0] = 0xffffffffu;
        } else if (partner == 2) {
            f32x4 c = {0, 0, 0, 0}, d = {1, 1, 1, 1};
            for (int it = 0; it < 40000; ++it) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(v, 0.5f, c, 0, 0, 0); d = __builtin_amdgcn_mfma_f32_16x16x4f32(v, 0.25f, d, 0, 0, 0); v = v * 1.0001f + 0.5f; }
            if (c[0] + d[0] + v == 123.456f) g_bad[0][0] = 0xffffffffu;
        } else if (partner == 3) {
            f32x16 c, d; for (int i = 0; i < 16; ++i) { c[i] = 0.f; d[i] = 1.f; }
            for (int it = 0; it < 20000; ++it) { c = __builtin_amdgcn_mfma_f32_32x32x2f32(v, 0.5f, c, 0, 0, 0); d = __builtin_amdgcn_mfma_f32_32x32x2f32(v, 0.25f, d, 0, 0, 0); v = v * 1.0001f + 0.5f; }
            if (c[0] + d[0] + v == 123.456f) g_bad[0][0] = 0xffffffffu;
        } else {
            float w = v + 1.f;
            for (int it = 0; it < 200000; ++it) { v = v * 1.0001f + 0.5f; w = w * 0.9999f + v; }
            if (w + v == 123.456f) g_bad[0][0] = 0xffffffffu;
        }
        return;
    }
    const unsigned ab = 0x3c003c00u;          // 1.0, 1.0 in fp16: A.B over K = 32 -> 32
    const float ab_dot = 32.f, c0 = 5.f;
    const int iters = 100;
    run<0, 0>(c0, ab, ab_dot, iters); run<1, 1>(c0, ab, ab_dot, iters); run<2, 2>(c0, ab, ab_dot, iters); run<3, 3>(c0, ab, ab_dot, iters); run<4, 4>(c0, ab, ab_dot, iters);
    run<6, 5>(c0, ab, ab_dot, iters); run<8, 6>(c0, ab, ab_dot, iters); run<12, 7>(c0, ab, ab_dot, iters); run<16, 8>(c0, ab, ab_dot, iters);
}
int main() {
    const int ks[9] = {0, 1, 2, 3, 4, 6, 8, 12, 16};
    const char* names[6] = {"VALU overwrites SrcA (WAR)   ", "VALU overwrites SrcB (WAR)   ", "VALU reads D (RAW)           ", "VALU overwrites D (WAW)      ", "3 dependent MFMAs in place   ", "MFMA reads another MFMA's D  "};
    const char* pn[5] = {"no partner wave", "partner: f16 16x16x32 MFMAs (4 passes)", "partner: f32 16x16x4 MFMAs (8 passes)", "partner: f32 32x32x2 MFMAs (16 passes)", "partner: VALU only"};
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int waves = cfg == 0 ? 8 : 16, testers = cfg == 2 ? 2 : 1;          // 2 waves per SIMD (1 tester + 1 partner) | 4 per SIMD (1 + 3) | 4 per SIMD (2 + 2)
        for (int partner = 0; partner < 5; ++partner) {
            unsigned z[6][9] = {};
            hipMemcpyToSymbol(HIP_SYMBOL(g_bad), z, sizeof(z));
            if (waves == 8) hipLaunchKernelGGL(k<8>, dim3(512), dim3(512), 0, 0, partner, testers);
            else hipLaunchKernelGGL(k<16>, dim3(512), dim3(1024), 0, 0, partner, testers);
            hipDeviceSynchronize();
            unsigned h[6][9]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bad), sizeof(h));
            printf("v_mfma_f32_16x16x32_f16, %d waves per SIMD (%d testing), %s: wrong results by wait states between the MFMA and the VALU instruction\n   states:                    ", waves / 4, testers, pn[partner]);
            for (int i = 0; i < 9; ++i) printf(" %8d", ks[i]);
            printf("\n");
            for (int t = 0; t < 6; ++t) { printf("   %s", names[t]); for (int i = 0; i < 9; ++i) printf(" %8u", h[t][i]); printf("\n"); }
        }
    }
    return 0;
}

// Is a VALU write of a register safe when an MFMA that WROTE it has been issued a few instructions earlier and a DEPENDENT MFMA (reading it as SrcC) sits in between?
// (The sequence hipcc emits in pair_embed_kernel's term path: the register allocator hands the just-consumed accumulator to the next conversion's result.)
//   mfma  D1 = A B + C0          ; D1 = v[20:23]
//   NIND independent mfmas
//   mfma  D4 = A B + D1          ; reads D1 as SrcC (DEP = 1) or not (DEP = 0)
//   NOPS wait states
//   v_mov v20, VAL               ; VALU overwrite of D1's first register
//   ... later: v20 must be VAL on every lane.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_waw_chain mfma_waw_chain.hip && ./mfma_waw_chain
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NIND, int DEP, int NOPS>
__global__ __launch_bounds__(1024) void k(unsigned* bad, float seed, int iters) {
    unsigned nbad = 0;
    float a = seed + threadIdx.x * 0.001f, val = 12345.f;
    for (int it = 0; it < iters; ++it) {
        float got;
        asm volatile(
            "v_cvt_pk_f16_f32 v0, %1, %1\n v_mov_b32 v1, v0\n v_mov_b32 v2, v0\n v_mov_b32 v3, v0\n"
            "v_mov_b32 v4, v0\n v_mov_b32 v5, v0\n v_mov_b32 v6, v0\n v_mov_b32 v7, v0\n"
            "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n"
            "s_nop 7\n s_nop 7\n"
            "v_mfma_f32_16x16x32_f16 v[20:23], v[0:3], v[4:7], v[8:11]\n"
            ".rept %c3\n v_mfma_f32_16x16x32_f16 v[24:27], v[0:3], v[4:7], v[8:11]\n .endr\n"
            ".if %c4\n v_mfma_f32_16x16x32_f16 v[28:31], v[0:3], v[4:7], v[20:23]\n .endif\n"
            ".rept %c5\n s_nop 0\n .endr\n"
            "v_mov_b32 v20, %2\n"
            "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"
            "v_mov_b32 %0, v20\n"
            : "=v"(got) : "v"(a), "v"(val), "n"(NIND), "n"(DEP), "n"(NOPS)
            : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
        nbad += (got != val);
        a += 0.25f;
    }
    if (nbad) atomicAdd(bad, nbad);
}
template <int NIND, int DEP, int NOPS>
void run(unsigned* bad) {
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((k<NIND, DEP, NOPS>), dim3(1024), dim3(1024), 0, 0, bad, 0.5f, 2000);
    unsigned h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("independent mfmas %d, dependent mfma %d, wait states %d: %u wrong lane-iterations of %.0f\n", NIND, DEP, NOPS, h, 1024.0 * 1024 * 2000);
}
int main() {
    unsigned* bad; hipMalloc(&bad, 4);
    run<0, 0, 0>(bad); run<0, 0, 2>(bad); run<0, 0, 4>(bad); run<0, 0, 6>(bad);
    run<0, 1, 0>(bad); run<0, 1, 2>(bad); run<0, 1, 4>(bad);
    run<2, 0, 0>(bad); run<2, 1, 0>(bad); run<2, 1, 1>(bad); run<2, 1, 2>(bad); run<2, 1, 4>(bad);
    run<3, 1, 0>(bad); run<3, 0, 0>(bad); run<1, 1, 0>(bad); run<1, 0, 0>(bad);
    return 0;
}

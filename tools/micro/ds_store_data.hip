// Does an LDS write read its data registers late on gfx950?  hipcc 7.2 emits (SLP-vectorised build of the fused block kernel, non-deterministic in round 5)
//     ds_write2_b64 v159, v[140:141], v[146:147] offset0:64 offset1:114 ; v_mov_b32 v140, v67
// i.e. a VALU overwrite of a 128-bit LDS write's data register in the very next instruction.  Each lane writes four dwords with ONE instruction
// (ds_write_b64, ds_write2_b64, ds_write_b128), overwrites the first / last data register after K wait states, then reads the LDS back.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ds_store_data.hip -o /tmp/ds_store_data && /tmp/ds_store_data
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory"
#define LOADV "v_mov_b32 v100, %1\n\tv_add_u32 v101, 1, %1\n\tv_add_u32 v102, 2, %1\n\tv_add_u32 v103, 3, %1\n\ts_nop 4\n\t"
#define RB "s_waitcnt lgkmcnt(0)\n\ts_nop 4\n\tds_read_b128 v[104:107], %2\n\ts_waitcnt lgkmcnt(0)\n\tv_sub_u32 v104, v104, %1\n\tv_sub_u32 v105, v105, %1\n\tv_sub_u32 v106, v106, %1\n\tv_sub_u32 v107, v107, %1\n\t" \
           "v_or_b32 v104, v104, v105\n\tv_sub_u32 v106, v106, 2\n\tv_sub_u32 v107, v107, 3\n\tv_sub_u32 v105, v105, 1\n\tv_mov_b32 %0, 0\n\t" \
           "v_cmp_ne_u32 vcc, 0, v106\n\tv_cndmask_b32 %0, %0, 1, vcc\n\tv_cmp_ne_u32 vcc, 0, v107\n\tv_cndmask_b32 %0, %0, 1, vcc\n\tv_cmp_ne_u32 vcc, 0, v105\n\tv_cndmask_b32 %0, %0, 1, vcc\n\t"
#define T(NAME, WRITE, NOPS, POISON)                                                                        \
__device__ __forceinline__ unsigned NAME(unsigned base, unsigned addr) { unsigned bad;                                  \
    asm volatile(LOADV WRITE "\n\t" NOPS POISON "\n\t" RB : "=&v"(bad) : "v"(base), "v"(addr) : CLOB, "vcc"); return bad; }
#define W128 "ds_write_b128 %2, v[100:103]"
#define W2X64 "ds_write2_b64 %2, v[100:101], v[102:103] offset1:1"
#define P_FIRST "v_mov_b32 v100, 0x7fffffff"
#define P_LAST "v_mov_b32 v103, 0x7fffffff"
T(w128_first_0, W128, "", P_FIRST) T(w128_first_1, W128, "s_nop 0\n\t", P_FIRST) T(w128_first_2, W128, "s_nop 1\n\t", P_FIRST)
T(w128_last_0, W128, "", P_LAST) T(w128_last_1, W128, "s_nop 0\n\t", P_LAST) T(w128_last_2, W128, "s_nop 1\n\t", P_LAST)
T(w2_first_0, W2X64, "", P_FIRST) T(w2_first_1, W2X64, "s_nop 0\n\t", P_FIRST) T(w2_first_2, W2X64, "s_nop 1\n\t", P_FIRST)
T(w2_last_0, W2X64, "", P_LAST) T(w2_last_1, W2X64, "s_nop 0\n\t", P_LAST) T(w2_last_2, W2X64, "s_nop 1\n\t", P_LAST)
__device__ unsigned g_bad[12];
__global__ __launch_bounds__(1024) void k(int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024 * 4];
    const unsigned addr = (unsigned)(size_t)(&lds[threadIdx.x * 4]) & 0xffffu;     // LDS byte address of this lane's four dwords
    unsigned bad[12] = {};
    for (int it = 0; it < iters; ++it) {
        const unsigned base = (blockIdx.x * 1024 + threadIdx.x) * 16u + it * 64u;
        bad[0] += w128_first_0(base, addr); bad[1] += w128_first_1(base, addr); bad[2] += w128_first_2(base, addr);
        bad[3] += w128_last_0(base, addr); bad[4] += w128_last_1(base, addr); bad[5] += w128_last_2(base, addr);
        bad[6] += w2_first_0(base, addr); bad[7] += w2_first_1(base, addr); bad[8] += w2_first_2(base, addr);
        bad[9] += w2_last_0(base, addr); bad[10] += w2_last_1(base, addr); bad[11] += w2_last_2(base, addr);
    }
    for (int i = 0; i < 12; ++i) if (bad[i]) atomicAdd(&g_bad[i], bad[i]);
}
int main() {
    unsigned z[12] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_bad), z, sizeof(z));
    hipLaunchKernelGGL(k, dim3(1024), dim3(1024), 0, 0, 500);
    hipDeviceSynchronize();
    unsigned h[12]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bad), sizeof(h));
    const char* n[4] = {"ds_write_b128, first data register overwritten", "ds_write_b128, last data register overwritten ", "ds_write2_b64, first data register overwritten", "ds_write2_b64, last data register overwritten "};
    for (int t = 0; t < 4; ++t) printf("%s after 0 / 1 / 2 wait states: %u %u %u corrupted writes of %.0f M\n", n[t], h[3 * t], h[3 * t + 1], h[3 * t + 2], 1024.0 * 1024 * 500 / 1e6);
    return 0;
}

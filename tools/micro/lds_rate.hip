// LDS read rate on gfx950 by access width: one workgroup per CU, W waves, every lane reads consecutive 4 / 8 / 16-byte words (the access
// pattern of the MFMA operand reads in node_frags / the tail / the fused consumer).  Prints bytes per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_rate.hip -o /tmp/lds_rate && /tmp/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int WIDTH>
__global__ __launch_bounds__(1024) void rd(unsigned* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int base = ((wave * 16 + u) * 1024 + it * 64) & 65535 & ~1023;
            if (WIDTH == 16) { const u32x4 v = *reinterpret_cast<const u32x4*>(sm + base + lane * 16); acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
            else if (WIDTH == 8) {
                const u32x2 a = *reinterpret_cast<const u32x2*>(sm + base + lane * 8), b = *reinterpret_cast<const u32x2*>(sm + base + 512 + lane * 8);
                acc += a[0] ^ a[1] ^ b[0] ^ b[1];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += *reinterpret_cast<const unsigned*>(sm + base + q * 256 + lane * 4);
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    for (int waves : {4, 8, 16}) for (int width : {16, 8, 4}) {
        auto k = width == 16 ? rd<16> : (width == 8 ? rd<8> : rd<4>);
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, out, cyc, iters);
        hipDeviceSynchronize();
        long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        const double bytes = (double)iters * 16 * 1024 * waves;
        printf("%2d waves, %2d-byte reads: %.1f B/clk per CU (%.0f cycles for %.0f KB per CU)\n", waves, width, bytes / avg, avg, bytes / 1024);
    }
    return 0;
}

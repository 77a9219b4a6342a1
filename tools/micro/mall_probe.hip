// Does the memory-side cache (256 MB Infinity Cache) of MI355X serve a stream that is re-read?  One kernel reads `bytes` of a buffer with 16-byte loads (all
// CUs, coalesced 1 KB per wave request, temporal or non-temporal), in forward or REVERSED block order.  Cases: a working set that fits (128 MB) against one
// that does not (640 MB), read forward every pass (cyclic: an LRU cache hits nothing) or alternating forward / backward (the last-read ~256 MB are the first
// read of the next pass).  Prints GB/s per case.   hipcc --offload-arch=gfx950 -O3 tools/micro/mall_probe.hip -o tools/micro/mall_probe && tools/micro/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, size_t nvec, int reverse, float* out) {
    // block b walks a contiguous span; spans are visited in forward or reversed order of TIME by giving early blocks the far end
    const size_t per = nvec / gridDim.x;
    const size_t blk = reverse ? (gridDim.x - 1 - blockIdx.x) : blockIdx.x;
    const f32x4* q = p + blk * per;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per; i += 256 * 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t j = reverse ? (per - 1 - (i + u * 256)) : (i + u * 256);
            if (i + u * 256 < per) v[u] = NT ? __builtin_nontemporal_load(q + j) : q[j]; else v[u] = (f32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) out[0] = 1.f;
}
int main() {
    const size_t cap = (size_t)1 << 30;
    f32x4* buf; float* out;
    hipMalloc(&buf, cap); hipMalloc(&out, 4);
    hipMemset(buf, 0, cap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
    for (size_t mb : {64, 128, 192, 256, 384, 640, 1024})
    for (int alt = 0; alt < 2; ++alt) {
        const size_t nvec = mb * 1024 * 1024 / 16;
        const int grid = 256 * 8, passes = 12;
        auto launch = [&](int rev) { if (nt) hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(256), 0, 0, buf, nvec, rev, out); else hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(256), 0, 0, buf, nvec, rev, out); };
        launch(0); launch(alt ? 1 : 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int p = 0; p < passes; ++p) launch(alt ? (p & 1) : 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s loads, %4zu MB, %s: %.2f TB/s\n", nt ? "nt" : "temporal", mb, alt ? "alternating forward / backward" : "forward every pass       ", passes * (double)mb * 1.048576e-6 / (ms * 1e-3));
    }
    return 0;
}

// ./dih_driver <module.hsaco> <seconds> <kernel name> [<kernel name> ...]: launches each kernel of the module (64 x 256 threads) in a loop, compares every launch with the first on the device
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void diff_kernel(const unsigned* a, const unsigned* b, int n, unsigned* count, unsigned* where) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && a[i] != b[i]) { if (atomicAdd(count, 1u) == 0) *where = (unsigned)i; }
}
int main(int argc, char** argv) {
    if (argc < 4) return 1;
    const double seconds = atof(argv[2]);
    hipModule_t mod;
    if (hipModuleLoad(&mod, argv[1]) != hipSuccess) { printf("cannot load %s\n", argv[1]); return 1; }
    const int npts = 4096, SETS = 16;
    std::vector<float4> h(npts * 4);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f; };
    for (int i = 0; i < npts; ++i) {
        const float cx = rnd() * 40.f, cy = rnd() * 40.f, cz = rnd() * 40.f;
        for (int a = 0; a < 4; ++a) h[i * 4 + a] = make_float4(cx + rnd() * 3.f, cy + rnd() * 3.f, cz + rnd() * 3.f, 1.f);
    }
    float4* pts; float *ref, *out; unsigned* cnt;
    const int nout = 64 * 256 * SETS * 2;
    (void)hipMalloc(&pts, h.size() * sizeof(float4)); (void)hipMalloc(&ref, (size_t)nout * 4); (void)hipMalloc(&out, (size_t)nout * 4); (void)hipMalloc(&cnt, 8);
    (void)hipMemcpy(pts, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    for (int k = 3; k < argc; ++k) {
        hipFunction_t fn;
        if (hipModuleGetFunction(&fn, mod, argv[k]) != hipSuccess) { printf("no kernel %s\n", argv[k]); continue; }
        int n = npts;
        float* dst = ref;
        void* args[] = {&pts, &dst, &n};
        (void)hipModuleLaunchKernel(fn, 64, 1, 1, 256, 1, 1, 0, 0, args, nullptr);
        (void)hipDeviceSynchronize();
        dst = out;
        long launches = 0, bad = 0; unsigned first = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int r = 0; r < 50; ++r) {
                (void)hipMemsetAsync(cnt, 0, 8, 0);
                (void)hipModuleLaunchKernel(fn, 64, 1, 1, 256, 1, 1, 0, 0, args, nullptr);
                hipLaunchKernelGGL(diff_kernel, dim3((nout + 255) / 256), dim3(256), 0, 0, (const unsigned*)ref, (const unsigned*)out, nout, cnt, cnt + 1);
                unsigned hc[2];
                (void)hipMemcpy(hc, cnt, 8, hipMemcpyDeviceToHost);
                ++launches;
                if (hc[0]) { if (!bad) first = hc[1]; ++bad; }
            }
        }
        printf("%-60s %ld of %ld launches differ from the first", argv[k], bad, launches);
        if (bad) printf("; first: lane %u, set %u, angle %u", (first / (SETS * 2)) & 63, (first / 2) % SETS, first & 1);
        printf("\n"); fflush(stdout);
    }
    return 0;
}

// Optimizer step of the training loop (the last third of AbDesign/train.py:108-118 / AbDock/train.py:104-114):
//     orig_grad_norm = clip_grad_norm_(model.parameters(), max_grad_norm);  optimizer.step()        with torch.optim.Adam
// (AbDesign/diffab/utils/train.py:28-36: lr, betas, weight_decay; eps 1e-8, no amsgrad).  torch runs this as ~1000 small launches for
// the model's 207 parameter tensors (1.8 ms of a 16 ms step on MI355X); here the whole list is walked by a handful of launches:
//   grad_sqsum_kernel   per-block partial sums of g^2                      (clipping only)
//   adam_prepare_kernel total norm, clip coefficient, step += 1             (one workgroup)
//   adam_kernel         g' = coef g (+ wd p); m, v, p updates, one pass over p, g, m, v
// Tensors are handed over as host arrays of device pointers; AD_K of them travel in the kernel arguments of one launch (no pointer
// table in device memory, nothing to upload, capturable into a hipGraph as is).  Everything is deterministic: block partials are
// summed in a fixed order.
#include <cmath>
#include "abopt_common.h"

namespace abopt {

constexpr int AD_K = 32;            // tensors per launch
constexpr int AD_THREADS = 256;
constexpr int AD_BLK = 4096;        // elements per workgroup (16 per thread, coalesced)

struct AdamChunk {
    float* p[AD_K];
    const float* g[AD_K];
    float* m[AD_K];
    float* v[AD_K];
    int64_t n[AD_K];
    int blk0[AD_K + 1];             // first workgroup of every tensor of this launch
    int count;
};

__device__ __forceinline__ int chunk_tensor_of(const AdamChunk& c, int b) {
    int ti = 0;
    while (ti + 1 < c.count && b >= c.blk0[ti + 1]) ++ti;
    return ti;
}

__global__ __launch_bounds__(AD_THREADS) void grad_sqsum_kernel(AdamChunk c, float* __restrict__ partials) {
    const int b = blockIdx.x, ti = chunk_tensor_of(c, b);
    const int64_t off = (int64_t)(b - c.blk0[ti]) * AD_BLK, n = c.n[ti];
    const float* g = c.g[ti];
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < AD_BLK / AD_THREADS; ++k) {
        const int64_t i = off + k * AD_THREADS + threadIdx.x;
        if (i < n) { const float x = g[i]; s = fmaf(x, x, s); }
    }
    s = wave_sum(s);
    __shared__ float red[AD_THREADS / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ws[0] = total gradient norm, ws[1] = clip coefficient (torch.nn.utils.clip_grad_norm_: max_norm / (norm + 1e-6), clamped to 1);
// the step counter moves on by one.  nparts == 0: no clipping.
// hyper (optional, DEVICE): {lr, beta1, beta2, eps, weight_decay, max_grad_norm} as doubles, read at execution time instead of the by-value
// arguments -- a captured hipGraph of the training step then follows a learning-rate scheduler (the host refreshes the buffer before a replay).
__global__ __launch_bounds__(AD_THREADS) void adam_prepare_kernel(const float* __restrict__ partials, int nparts, float max_norm, float* __restrict__ ws,
                                                                  int64_t* __restrict__ step, float* __restrict__ norm_out, const double* __restrict__ hyper) {
    if (hyper) max_norm = (float)hyper[5];
    __shared__ double red[AD_THREADS];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += AD_THREADS) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = AD_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        float coef = 1.f;
        if (nparts > 0) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
        ws[0] = norm; ws[1] = coef;
        if (norm_out) norm_out[0] = norm;
        step[0] += 1;
    }
}

// torch/optim/adam.py _single_tensor_adam, in its order of operations:
//   grad (+= weight_decay * param);  exp_avg.lerp_(grad, 1 - beta1);  exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps;  param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^t))
__global__ __launch_bounds__(AD_THREADS) void adam_kernel(AdamChunk c, const float* __restrict__ ws, const int64_t* __restrict__ step, double lr, double beta1d,
                                                          double beta2d, float eps, float weight_decay, const double* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; beta1d = hyper[1]; beta2d = hyper[2]; eps = (float)hyper[3]; weight_decay = (float)hyper[4]; }
    const int b = blockIdx.x, ti = chunk_tensor_of(c, b);
    const int64_t off = (int64_t)(b - c.blk0[ti]) * AD_BLK, n = c.n[ti];
    float* p = c.p[ti];
    const float* g = c.g[ti];
    float* m = c.m[ti];
    float* v = c.v[ti];
    const float coef = ws[1];
    const double t = (double)step[0];
    // the hyper-parameters arrive as the doubles Python holds, and are rounded where torch rounds them (1 - beta in double first)
    const float step_size = (float)(lr / (1.0 - pow(beta1d, t)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2d, t));
    const float w1 = (float)(1.0 - beta1d), w2 = (float)(1.0 - beta2d), beta2 = (float)beta2d;
#pragma unroll 4
    for (int k = 0; k < AD_BLK / AD_THREADS; ++k) {
        const int64_t i = off + k * AD_THREADS + threadIdx.x;
        if (i >= n) continue;
        float gi = g[i] * coef;
        const float pi = p[i];
        if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
        const float mi = fmaf(w1, gi - m[i], m[i]);
        const float vi = fmaf(w2, gi * gi, v[i] * beta2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        m[i] = mi; v[i] = vi;
        p[i] = fmaf(-step_size, mi / denom, pi);
    }
}

static int64_t adam_blocks(int count, const int64_t* numel) {
    int64_t b = 0;
    for (int i = 0; i < count; ++i) b += (numel[i] + AD_BLK - 1) / AD_BLK;
    return b;
}

}  // namespace abopt

using namespace abopt;

extern "C" size_t abopt_adam_ws_floats(int count, const int64_t* numel) { return (size_t)adam_blocks(count, numel) + 8; }

extern "C" int abopt_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                               const int64_t* numel, double lr, double beta1, double beta2, double eps, double weight_decay, double max_grad_norm,
                               int64_t* step, float* ws, size_t ws_floats, float* grad_norm_out, const double* hyper_dev, abopt_stream stream) {
    hipStream_t st = (hipStream_t)stream;
    ABOPT_CHECK_ARG(count >= 0 && step && ws, "adam_step: step counter and workspace are required");
    ABOPT_CHECK_ARG(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && lr >= 0. && weight_decay >= 0., "adam_step: lr %g, betas (%g, %g), eps %g", lr, beta1, beta2, eps);
    for (int i = 0; i < count; ++i) ABOPT_CHECK_ARG(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && numel[i] >= 0, "adam_step: tensor %d has a null pointer", i);
    const int64_t total_blocks = adam_blocks(count, numel);
    ABOPT_CHECK_ARG((size_t)total_blocks + 8 <= ws_floats, "adam_step: workspace of %zu floats, %lld needed", ws_floats, (long long)total_blocks + 8);
    ABOPT_CHECK_ARG(total_blocks < (1ll << 31), "adam_step: too many elements");
    const bool clip = max_grad_norm > 0.;
    float* partials = ws + 8;
    auto for_chunks = [&](auto&& launch) -> int {
        int64_t pbase = 0;
        for (int i0 = 0; i0 < count; i0 += AD_K) {
            AdamChunk c;
            c.count = 0;
            int blocks = 0;
            for (int i = i0; i < count && i < i0 + AD_K; ++i) {
                if (numel[i] == 0) continue;
                const int k = c.count++;
                c.p[k] = params[i]; c.g[k] = grads[i]; c.m[k] = exp_avg[i]; c.v[k] = exp_avg_sq[i]; c.n[k] = numel[i];
                c.blk0[k] = blocks;
                blocks += (int)((numel[i] + AD_BLK - 1) / AD_BLK);
            }
            c.blk0[c.count] = blocks;
            if (blocks == 0) continue;
            if (int rc = launch(c, blocks, pbase)) return rc;
            pbase += blocks;
        }
        return ABOPT_OK;
    };
    if (clip) {
        if (int rc = for_chunks([&](const AdamChunk& c, int blocks, int64_t pbase) -> int {
                hipLaunchKernelGGL(grad_sqsum_kernel, dim3((unsigned)blocks), dim3(AD_THREADS), 0, st, c, partials + pbase);
                ABOPT_LAUNCH_CHECK();
                return ABOPT_OK;
            })) return rc;
    }
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(AD_THREADS), 0, st, partials, clip ? (int)total_blocks : 0, (float)max_grad_norm, ws, step, grad_norm_out, hyper_dev);
    ABOPT_LAUNCH_CHECK();
    return for_chunks([&](const AdamChunk& c, int blocks, int64_t) -> int {
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(AD_THREADS), 0, st, c, ws, step, lr, beta1, beta2, (float)eps, (float)weight_decay, hyper_dev);
        ABOPT_LAUNCH_CHECK();
        return ABOPT_OK;
    });
}

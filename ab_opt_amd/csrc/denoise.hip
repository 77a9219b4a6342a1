// Per-step diffusion transitions and sampler initialisation for CDNA4.
//
// Replaces (reference AbDock/src/, identical maths in AbDesign/diffab/):
//   PositionTransition.pred_noise_from_start / denoise      modules/diffusion/transition.py:42-50, 80-101
//   RotationTransition.denoise                              modules/diffusion/transition.py:146-160
//   ApproxAngularDistribution.sample, random_normal_so3     modules/common/so3.py:111-146
//   AminoacidCategoricalTransition.posterior/denoise/_sample modules/diffusion/transition.py:166-245
//   pRMSDCa.compute_prmsd, calc_perplexity                  modules/common/prmsd.py:31-47, modules/diffusion/dpm_full.py:380-399
//   FullDPM.sample initial state                            modules/diffusion/dpm_full.py:255-269
//   calc_per_rmsd / rank_commoness score                    tools/runner/design_for_testset.py:556-589
// The reference gathers an (N*L, 8192) histogram copy per step and calls multinomial on it (240 ms at
// N=4, L=256 on CPU); here one 8191-entry CDF row is binary-searched per residue.
#include "denoise_row.h"

namespace abopt {

__device__ __forceinline__ float torch_linspace(float a, float b, int steps, int i) {
    // at::linspace: symmetric evaluation around the midpoint
    const float step = (b - a) / (float)(steps - 1);
    return (i < steps / 2) ? a + step * (float)i : b - step * (float)(steps - 1 - i);
}

// one workgroup per sample n; threads stride over residues; block-reduce the two per-sample scalars.
__global__ __launch_bounds__(256) void denoise_step_kernel(abopt_step_params sp, abopt_step_noise nz, uint64_t seed, uint64_t offset,
                                                           const uint64_t* __restrict__ seed_dev,
                                                           const float* __restrict__ v_t, const float* __restrict__ p_t,
                                                           const int64_t* __restrict__ s_t, const float* __restrict__ v_net,
                                                           const float* __restrict__ p_net, const float* __restrict__ c_net,
                                                           const float* __restrict__ prmsd_logits, const uint8_t* __restrict__ mask_generate,
                                                           const float* __restrict__ igX, const float* __restrict__ igCdf, int bins, int num_bins,
                                                           float* __restrict__ v_next, float* __restrict__ p_next, int64_t* __restrict__ s_next,
                                                           float* __restrict__ prmsd, float* __restrict__ ppl, float* __restrict__ post_out,
                                                           float* __restrict__ p_next_norm, int L, int ppl_masked) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const bool injected = nz.axis != nullptr;
    if (seed_dev) { seed = seed_dev[0]; offset = seed_dev[1]; }     // graph replays: the stream position comes from device memory
    const Philox rng(seed);
    float ppl_num = 0.f, ppl_den = 0.f;
    // The histogram bin is the only long dependent chain of a residue (13 probes of the 8191-entry CDF row): the row is staged in LDS by
    // coalesced loads first (same comparisons on the same values), and not searched at all where its result cannot reach the output
    // (Gaussian branch of so3.py:129-138, t <= 1: e = 0, injected noise).
    __shared__ float cdf_s[8192];
    const bool need_bin = !injected && !sp.igso3_gaussian && sp.t > 1;
    const bool cdf_lds = need_bin && bins - 1 <= 8192;
    if (cdf_lds) {
        for (int k = tid; k < bins - 1; k += 256) cdf_s[k] = igCdf[k];
        __syncthreads();
    }

    const DenoiseRowIO io{v_t, p_t, s_t, v_net, p_net, c_net, mask_generate, igX, igCdf, bins, v_next, p_next, s_next, post_out, p_next_norm};
    for (int l = tid; l < L; l += 256) {
        const int64_t i = (int64_t)n * L + l;
        float nx_, ny_, nz_;
        denoise_row(i, sp, nz, injected, rng, offset, io, cdf_lds ? cdf_s : nullptr, need_bin, ppl_masked, ppl_num, ppl_den, nx_, ny_, nz_);
    }

    // ---- per-sample scalars
    __shared__ float red[2][4];
    ppl_num = wave_sum(ppl_num); ppl_den = wave_sum(ppl_den);
    if ((tid & 63) == 0) { red[0][tid >> 6] = ppl_num; red[1][tid >> 6] = ppl_den; }
    __syncthreads();
    if (tid == 0) {
        if (ppl) ppl[n] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        if (prmsd && prmsd_logits) {
            float mx = -INFINITY;
            for (int b = 0; b < num_bins; ++b) mx = fmaxf(mx, prmsd_logits[(int64_t)n * num_bins + b]);
            float sm = 0.f;
            for (int b = 0; b < num_bins; ++b) sm += expf(prmsd_logits[(int64_t)n * num_bins + b] - mx);
            float acc = 0.f;
            for (int b = 0; b < num_bins; ++b)
                acc += (expf(prmsd_logits[(int64_t)n * num_bins + b] - mx) / sm) * torch_linspace(sp.dist_min, sp.dist_max, num_bins, b);
            prmsd[n] = acc;
        }
    }
}

// FullDPM.sample init (dpm_full.py:255-269)
__global__ __launch_bounds__(256) void sample_init_kernel(const float* __restrict__ v, const float* __restrict__ p, const int64_t* __restrict__ s,
                                                          const uint8_t* __restrict__ mask_generate, const float* __restrict__ q4,
                                                          const float* __restrict__ pn, const int64_t* __restrict__ sr, uint64_t seed, uint64_t offset,
                                                          float scale, float m0, float m1, float m2, int sample_structure, int sample_sequence,
                                                          float* __restrict__ v_init, float* __restrict__ p_init, int64_t* __restrict__ s_init, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const bool gen = mask_generate[i] != 0;
    const float mean[3] = {m0, m1, m2};
    float q[4], g[3];
    int64_t srand_;
    if (q4) {
        q[0] = q4[i * 4]; q[1] = q4[i * 4 + 1]; q[2] = q4[i * 4 + 2]; q[3] = q4[i * 4 + 3];
        g[0] = pn[i * 3]; g[1] = pn[i * 3 + 1]; g[2] = pn[i * 3 + 2];
        srand_ = sr ? sr[i] : 0;
    } else {
        const Philox rng(seed);
        const uint4 r0 = rng(offset + (uint64_t)i, 0xFFFF00ull), r1 = rng(offset + (uint64_t)i, 0xFFFF01ull);
        float d0;
        box_muller(r0.x, r0.y, q[0], q[1]);
        box_muller(r0.z, r0.w, q[2], q[3]);
        box_muller(r1.x, r1.y, g[0], g[1]);
        box_muller(r1.z, r1.w, g[2], d0);
        const Philox rng2(seed ^ 0x5bd1e995ull);
        srand_ = (int64_t)(rng2(offset + (uint64_t)i, 0xFFFF02ull).x % 19u);     // randint_like(low=0, high=19): TYR never drawn
        if (sr) srand_ = sr[i];      // sample_structure = False draws the sequence alone (dpm_full.py:262-267): it can be injected alone
    }
    // random_uniform_so3: F.normalize then quaternion_to_rotation_matrix (which normalises again), so3.py:66-68
    const float nq = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
    const Vec3 w = so3_log(quat_to_rot(q[0] / nq, q[1] / nq, q[2] / nq, q[3] / nq), false);
    const float wr[3] = {w.x, w.y, w.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pnorm = (p[i * 3 + k] - mean[k]) / scale;
        const bool rnd = gen && sample_structure;
        v_init[i * 3 + k] = rnd ? wr[k] : v[i * 3 + k];
        p_init[i * 3 + k] = (rnd ? g[k] : pnorm) * scale + mean[k];
    }
    s_init[i] = (gen && sample_sequence) ? srand_ : s[i];
}

// Forward noising of all three modalities (transition.py:62-78,120-144,179-200); one thread per residue.
__global__ __launch_bounds__(256) void add_noise_kernel(const int64_t* __restrict__ t, const float* __restrict__ alpha_bars,
                                                        const float* __restrict__ fstd, const uint8_t* __restrict__ fapprox,
                                                        const float* __restrict__ fX, const float* __restrict__ fcdf, int bins,
                                                        abopt_addnoise_noise nz, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ seed_dev,
                                                        const float* __restrict__ v_0, const float* __restrict__ p_0, const int64_t* __restrict__ s_0,
                                                        const uint8_t* __restrict__ mask_generate, float scale, float m0, float m1, float m2,
                                                        int noise_structure, int noise_sequence, int grad_mode,
                                                        float* __restrict__ v_noisy, float* __restrict__ p_noisy, int64_t* __restrict__ s_noisy,
                                                        float* __restrict__ eps_p, float* __restrict__ c_noisy, int L, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const int64_t tt = t[i / L];
    const bool gen = mask_generate[i] != 0;
    const float abar = alpha_bars[tt];
    const float c0 = sqrtf(abar), c1 = sqrtf(1.f - abar);
    const float mean[3] = {m0, m1, m2};
    float ax = 0.f, ay = 0.f, az = 0.f, ubin = 0.f, gss = 0.f, ex = 0.f, ey = 0.f, ez = 0.f, useq = 0.f;
    int64_t bin = 0;
    // injected draws: all of them, or -- when the structure is not noised (dpm_full.py:169-173 draws nothing for it) -- s_noisy alone
    const bool injected = nz.axis != nullptr || (!noise_structure && nz.s_noisy != nullptr);
    if (nz.axis) {
        ax = nz.axis[i * 3]; ay = nz.axis[i * 3 + 1]; az = nz.axis[i * 3 + 2];
        bin = nz.bin[i]; ubin = nz.ubin[i]; gss = nz.gauss[i];
        ex = nz.pos[i * 3]; ey = nz.pos[i * 3 + 1]; ez = nz.pos[i * 3 + 2];
    } else if (!injected) {
        if (seed_dev) { seed = seed_dev[0]; offset = seed_dev[1]; }
        const Philox rng(seed);
        const uint4 r0 = rng(offset + (uint64_t)i, 0xA00000ull), r1 = rng(offset + (uint64_t)i, 0xA00001ull), r2 = rng(offset + (uint64_t)i, 0xA00002ull);
        float d0;
        box_muller(r0.x, r0.y, ax, ay);
        box_muller(r0.z, r0.w, az, gss);
        box_muller(r1.x, r1.y, ex, ey);
        box_muller(r1.z, r1.w, ez, d0);
        ubin = u01(r2.x); useq = u01(r2.y);
        const float ub = u01(r2.z);
        const float* cdf = fcdf + tt * (int64_t)(bins - 1);
        int lo = 0, hi = bins - 2;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > ub) hi = mid; else lo = mid + 1; }
        bin = lo;
    }
    // rotation
    const float vx = v_0[i * 3], vy = v_0[i * 3 + 1], vz = v_0[i * 3 + 2];
    float nvx = vx, nvy = vy, nvz = vz;
    if (noise_structure) {
        const float* X = fX + tt * (int64_t)bins;
        const float sd = fstd[tt];
        const float nrm = fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-12f);
        const float hist = X[bin] + ubin * (X[bin + 1] - X[bin]);
        const float gau = fmodf(fabsf(sd * 2.f + gss * sd), PI_F);
        const float th = fapprox[tt] ? gau : hist;
        const Mat3 E = so3_exp(ax / nrm * th, ay / nrm * th, az / nrm * th);
        const Mat3 Rn = matmul3(E, so3_exp(c0 * vx, c0 * vy, c0 * vz));
        const Vec3 w = so3_log(Rn, grad_mode != 0);
        if (gen) { nvx = w.x; nvy = w.y; nvz = w.z; }
    }
    v_noisy[i * 3] = nvx; v_noisy[i * 3 + 1] = nvy; v_noisy[i * 3 + 2] = nvz;
    // position
    const float en[3] = {ex, ey, ez};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p0n = (p_0[i * 3 + k] - mean[k]) / scale;
        const float pn = (gen && noise_structure) ? c0 * p0n + c1 * en[k] : p0n;
        p_noisy[i * 3 + k] = pn * scale + mean[k];
        if (eps_p) eps_p[i * 3 + k] = noise_structure ? en[k] : 0.f;
    }
    // sequence
    const int64_t s0 = s_0[i];
    int64_t sn = s0;
    if (noise_sequence) {
        const bool ok = s0 >= 0 && s0 < KAA;
        float tot = 0.f, c[KAA];
#pragma unroll
        for (int k = 0; k < KAA; ++k) {
            const float oh = (ok && s0 == k) ? 1.f : 0.f;
            const float ck = gen ? (abar * oh) + ((1.f - abar) / (float)KAA) : oh;       // c_t, transition.py:196-198
            if (c_noisy) c_noisy[i * KAA + k] = ck;
            c[k] = ck + 1e-8f;                                                             // _sample adds 1e-8 (transition.py:179)
            tot += c[k];
        }
        if (injected) sn = nz.s_noisy[i];
        else {
            const float target = useq * tot;
            float cum = 0.f;
            sn = KAA - 1;
            for (int k = 0; k < KAA; ++k) { cum += c[k]; if (cum > target) { sn = k; break; } }
        }
    } else if (c_noisy) {
#pragma unroll
        for (int k = 0; k < KAA; ++k) c_noisy[i * KAA + k] = (s0 == k) ? 1.f : 0.f;      // c_0 = clampped_one_hot(s_0), transition.py:189
    }
    s_noisy[i] = sn;
}

// score[b] = sum_b' sqrt(mean_n |x_b - x_b'|^2) / (B - 1)   (design_for_testset.py:556-563,586-588)
__global__ __launch_bounds__(256) void commonness_kernel(const float* __restrict__ x, float* __restrict__ score, int B, int n) {
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ float part[4];
    float tot = 0.f;
    for (int o = wave; o < B; o += 4) {
        float s = 0.f;
        for (int k = lane; k < n * 3; k += 64) { const float d = x[((int64_t)b * n) * 3 + k] - x[((int64_t)o * n) * 3 + k]; s = fmaf(d, d, s); }
        s = wave_sum(s);
        tot += sqrtf(s / (float)n);
    }
    if (lane == 0) part[wave] = tot;
    __syncthreads();
    if (tid == 0) score[b] = (part[0] + part[1] + part[2] + part[3]) / (float)(B - 1);
}

}  // namespace abopt

using namespace abopt;

extern "C" int abopt_denoise_step(const abopt_step_params* sp, const abopt_step_noise* noise, uint64_t seed, uint64_t offset,
                                  const float* v_t, const float* p_t, const int64_t* s_t,
                                  const float* v_net, const float* p_net, const float* c_net, const float* prmsd_logits,
                                  const uint8_t* mask_generate, const float* igso3_X, const float* igso3_cdf, int igso3_bins, int num_bins,
                                  float* v_next, float* p_next, int64_t* s_next, float* prmsd, float* perplexity,
                                  float* post_out, float* p_next_norm, const uint64_t* seed_offset_dev, int N, int L, abopt_stream stream) {
    ABOPT_CHECK_ARG(sp && v_t && p_t && s_t && v_net && p_net && c_net && mask_generate && v_next && p_next && s_next, "denoise_step: NULL argument");
    ABOPT_CHECK_ARG(igso3_X && igso3_bins >= 2, "denoise_step: IGSO(3) histogram row missing");
    abopt_step_noise nz = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (noise && noise->axis) {
        ABOPT_CHECK_ARG(noise->bin && noise->ubin && noise->gauss && noise->z && noise->s_next, "denoise_step: injected noise must provide all six draws");
        nz = *noise;
    } else {
        ABOPT_CHECK_ARG(igso3_cdf, "denoise_step: device RNG path needs the CDF row");
    }
    if (N == 0 || L == 0) return ABOPT_OK;
    hipLaunchKernelGGL(denoise_step_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, *sp, nz, seed, offset, seed_offset_dev, v_t, p_t, s_t, v_net, p_net, c_net,
                       prmsd_logits, mask_generate, igso3_X, igso3_cdf, igso3_bins, num_bins, v_next, p_next, s_next, prmsd, perplexity, post_out,
                       p_next_norm, L, sp->ppl_masked);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

extern "C" int abopt_sample_init(const float* v, const float* p, const int64_t* s, const uint8_t* mask_generate,
                                 const float* q4, const float* pn, const int64_t* sr, uint64_t seed, uint64_t offset,
                                 float position_scale, const float* position_mean, int sample_structure, int sample_sequence,
                                 float* v_init, float* p_init, int64_t* s_init, int N, int L, abopt_stream stream) {
    ABOPT_CHECK_ARG(v && p && s && mask_generate && v_init && p_init && s_init && position_mean, "sample_init: NULL argument");
    ABOPT_CHECK_ARG((q4 == nullptr) == (pn == nullptr), "sample_init: q4 and pn must be given together");
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(sample_init_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, p, s, mask_generate, q4, pn, sr,
                       seed, offset, position_scale, position_mean[0], position_mean[1], position_mean[2], sample_structure, sample_sequence,
                       v_init, p_init, s_init, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

extern "C" int abopt_add_noise(const int64_t* t, const float* alpha_bars, const float* fwd_stddevs, const uint8_t* fwd_approx,
                               const float* fwd_X, const float* fwd_cdf, int bins, int num_sched,
                               const abopt_addnoise_noise* noise, uint64_t seed, uint64_t offset,
                               const float* v_0, const float* p_0, const int64_t* s_0, const uint8_t* mask_generate,
                               float position_scale, const float* position_mean, int noise_structure, int noise_sequence, int grad_mode,
                               float* v_noisy, float* p_noisy, int64_t* s_noisy, float* eps_p, float* c_noisy, const uint64_t* seed_offset_dev,
                               int N, int L, abopt_stream stream) {
    ABOPT_CHECK_ARG(t && alpha_bars && fwd_stddevs && fwd_approx && fwd_X && v_0 && p_0 && s_0 && mask_generate && position_mean &&
                    v_noisy && p_noisy && s_noisy && bins >= 2 && num_sched >= 1, "add_noise: bad arguments");
    abopt_addnoise_noise nz = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (noise && noise->axis) {
        ABOPT_CHECK_ARG(noise->bin && noise->ubin && noise->gauss && noise->pos && (noise->s_noisy || !noise_sequence), "add_noise: injected noise must provide every draw");
        nz = *noise;
    } else if (noise && noise->s_noisy && !noise_structure) {
        nz.s_noisy = noise->s_noisy;          // sequence-only noising (train_structure = False): the one draw this mode makes
    } else {
        ABOPT_CHECK_ARG(fwd_cdf, "add_noise: device RNG path needs the CDF table");
    }
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, alpha_bars, fwd_stddevs, fwd_approx,
                       fwd_X, fwd_cdf, bins, nz, seed, offset, seed_offset_dev, v_0, p_0, s_0, mask_generate, position_scale, position_mean[0], position_mean[1],
                       position_mean[2], noise_structure, noise_sequence, grad_mode, v_noisy, p_noisy, s_noisy, eps_p, c_noisy, L, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

extern "C" int abopt_commonness_score(const float* structs, float* score, int B, int n, abopt_stream stream) {
    ABOPT_CHECK_ARG(structs && score && B >= 2 && n >= 1, "commonness_score: bad arguments (B=%d n=%d)", B, n);
    hipLaunchKernelGGL(commonness_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, structs, score, B, n);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

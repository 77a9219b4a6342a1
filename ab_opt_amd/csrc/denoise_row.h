// One residue of a denoising step (the loop body of FullDPM.sample after the network, dpm_full.py:284-297): the three transitions
//   RotationTransition.denoise (transition.py:146-160, so3.py:111-146), PositionTransition.denoise / pred_noise_from_start (transition.py:42-50,80-101),
//   AminoacidCategoricalTransition.denoise (transition.py:202-245), and the residue's perplexity term (dpm_full.py:392-396)
// as ONE device function shared by denoise_step_kernel (denoise.hip: one workgroup per sample) and by the mixer kernel of the NEXT network evaluation
// (heads.hip, round 5: the step's transitions run on an otherwise idle wave of that launch) -- the same arithmetic in the same order, bit for bit.
#pragma once
#include "abopt_common.h"
#include "kernels.h"

namespace abopt {

constexpr int KAA = ABOPT_AA;
constexpr float PI_F = 3.14159265358979323846f;

struct DenoiseRowIO {
    const float* v_t; const float* p_t; const int64_t* s_t; const float* v_net; const float* p_net; const float* c_net; const uint8_t* mask_generate;
    const float* igX; const float* igCdf; int bins;
    float* v_next; float* p_next; int64_t* s_next; float* post_out; float* p_next_norm;
};

// cdf_s: the CDF row staged in LDS (or nullptr: searched in global memory).  Returns the next orientation vector through (nvx, nvy, nvz) as well.
__device__ __forceinline__ void denoise_row(int64_t i, const abopt_step_params& sp, const abopt_step_noise& nz, bool injected, const Philox& rng, uint64_t offset,
                                            const DenoiseRowIO& io, const float* cdf_s, bool need_bin, int ppl_masked, float& ppl_num, float& ppl_den,
                                            float& nvx_out, float& nvy_out, float& nvz_out) {
    const uint8_t* mask_generate = io.mask_generate;
    const float *v_t = io.v_t, *p_t = io.p_t, *v_net = io.v_net, *p_net = io.p_net, *c_net = io.c_net, *igX = io.igX, *igCdf = io.igCdf;
    const int64_t* s_t = io.s_t;
    float *v_next = io.v_next, *p_next = io.p_next, *post_out = io.post_out, *p_next_norm = io.p_next_norm;
    int64_t* s_next = io.s_next;
    const int bins = io.bins;
    const bool cdf_lds = cdf_s != nullptr;
    const bool gen = mask_generate[i] != 0;
    // ---- draws
    float ax, ay, az, ubin, gss, zx, zy, zz, useq;
    int64_t bin = 0;
    if (injected) {
        ax = nz.axis[i * 3]; ay = nz.axis[i * 3 + 1]; az = nz.axis[i * 3 + 2];
        bin = nz.bin[i]; ubin = nz.ubin[i]; gss = nz.gauss[i];
        zx = nz.z[i * 3]; zy = nz.z[i * 3 + 1]; zz = nz.z[i * 3 + 2];
        useq = 0.f;
    } else {
        const uint64_t ctr = offset + (uint64_t)i;
        const uint4 r0 = rng(ctr, ((uint64_t)sp.t << 8) | 0u), r1 = rng(ctr, ((uint64_t)sp.t << 8) | 1u), r2 = rng(ctr, ((uint64_t)sp.t << 8) | 2u);
        float d0;
        box_muller(r0.x, r0.y, ax, ay);
        box_muller(r0.z, r0.w, az, gss);
        box_muller(r1.x, r1.y, zx, zy);
        box_muller(r1.z, r1.w, zz, d0);
        ubin = u01(r2.x); useq = u01(r2.y);
        const float ub = u01(r2.z);
        // inverse CDF over bins-1 histogram cells == multinomial(Y[t, :-1]) (so3.py:122)
        int lo = 0, hi = bins - 2;
        if (cdf_lds) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf_s[mid] > ub) hi = mid; else lo = mid + 1; } }
        else if (need_bin) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (igCdf[mid] > ub) hi = mid; else lo = mid + 1; } }
        bin = lo;
    }
    // ---- rotation (transition.py:146-160)
    const float vx = v_t[i * 3], vy = v_t[i * 3 + 1], vz = v_t[i * 3 + 2];
    float nvx = vx, nvy = vy, nvz = vz;
    {
        const float nrm = fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-12f);
        const float hist = igX[bin] + ubin * (igX[bin + 1] - igX[bin]);
        const float gau = fmodf(fabsf(sp.igso3_std * 2.f + gss * sp.igso3_std), PI_F);
        const float th = sp.igso3_gaussian ? gau : hist;
        float ex = ax / nrm * th, ey = ay / nrm * th, ez = az / nrm * th;
        if (!(sp.t > 1)) { ex = 0.f; ey = 0.f; ez = 0.f; }
        const Mat3 E = so3_exp(ex, ey, ez);
        const Mat3 Rn = matmul3(E, so3_exp(v_net[i * 3], v_net[i * 3 + 1], v_net[i * 3 + 2]));
        const Vec3 w = so3_log(Rn, false);
        if (gen) { nvx = w.x; nvy = w.y; nvz = w.z; }
    }
    // ---- position (transition.py:42-50, 80-101); state is kept in Angstrom like the reference traj
    const float pa[3] = {p_t[i * 3], p_t[i * 3 + 1], p_t[i * 3 + 2]};
    const float zn[3] = {zx, zy, zz};
    float pn[3], pt[3];
    {
        const float c0 = 1.0f / sqrtf(sp.alpha_clamped + 1e-8f);
        const float c1 = (1.f - sp.alpha_clamped) / sqrtf(1.f - sp.alpha_bar + 1e-8f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pt[k] = (pa[k] - sp.position_mean[k]) / sp.position_scale;
            const float pnet = p_net[i * 3 + k];
            float eps = pnet;
            if (sp.pred_x0) eps = gen ? (sp.sqrt_recip_abar * pt[k] - pnet) / sp.sqrt_recipm1_abar : pt[k];
            const float zk = (sp.t > 1) ? zn[k] : 0.f;
            const float nx = c0 * (pt[k] - c1 * eps) + sp.sigma * zk;
            pn[k] = gen ? nx : pt[k];
        }
    }
    if (!sp.sample_structure) { nvx = vx; nvy = vy; nvz = vz; pn[0] = pt[0]; pn[1] = pt[1]; pn[2] = pt[2]; }
    v_next[i * 3] = nvx; v_next[i * 3 + 1] = nvy; v_next[i * 3 + 2] = nvz;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pa_next = pn[k] * sp.position_scale + sp.position_mean[k];
        p_next[i * 3 + k] = pa_next;
        // what the next step feeds the network (dpm_full.py:276 normalises the STORED Angstrom value again): saves the host two launches per step
        if (p_next_norm) p_next_norm[i * 3 + k] = (pa_next - sp.position_mean[k]) / sp.position_scale;
    }

    // ---- sequence (transition.py:202-245): NOTE alpha_bar_t multiplies both factors (reference quirk)
    const int64_t st = s_t[i];
    const bool st_ok = st >= 0 && st < KAA;
    float post[KAA], tot = 0.f;
    const float ab = sp.alpha_bar, unif = (1.f - ab) / (float)KAA;
#pragma unroll
    for (int k = 0; k < KAA; ++k) {
        const float ct = (st_ok && st == k) ? 1.f : 0.f;
        post[k] = ((ab * ct) + unif) * ((ab * c_net[i * KAA + k]) + unif);
        tot += post[k];
    }
    float pmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < KAA; ++k) {
        const float ct = (st_ok && st == k) ? 1.f : 0.f;
        post[k] = gen ? post[k] / (tot + 1e-8f) : ct;
        pmax = fmaxf(pmax, post[k]);
        if (post_out) post_out[i * KAA + k] = post[k];
    }
    int64_t sn;
    if (injected) sn = nz.s_next[i];
    else {
        float cum = 0.f, total = 0.f;
#pragma unroll
        for (int k = 0; k < KAA; ++k) total += post[k] + 1e-8f;
        const float target = useq * total;
        sn = KAA - 1;
        for (int k = 0; k < KAA; ++k) { cum += post[k] + 1e-8f; if (cum > target) { sn = k; break; } }
    }
    s_next[i] = sp.sample_sequence ? sn : st;
    // perplexity term: max softmax(post) (dpm_full.py:392-396)
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < KAA; ++k) se += expf(post[k] - pmax);
    const float w = (!ppl_masked || gen) ? 1.f : 0.f;
    ppl_num += (1.f / se) * w;
    ppl_den += w;
    nvx_out = nvx; nvy_out = nvy; nvz_out = nvz;
}

}  // namespace abopt

// Operand preparation and launch plumbing of the IPA core (the kernel itself is ipa_core.hip).
//
// Replaces, together with ipa_core.hip (reference AbDock/src/modules/encoders/ga.py):
//   _node_logits :81-86, _pair_logits :88-90, _spatial_logits :92-112, _alpha_from_logits :11-26,
//   _pair_aggregation :114-118, _node_aggregation :120-125, _spatial_aggregation :127-147.
#include "ipa_common.h"
#include "kernels.h"
#include <vector>
#include <utility>

namespace abopt {

// ------------------------------------------------------------------ node projections -> MFMA fragment order
// geometry.py:72-91 applied to proj_{query,key,value}_point outputs (ga.py:96-105,129-132): p <- R p + t, then the operands of
// the IPA core in fragment order, one workgroup per (sample, 16-residue block).  The block's residues are at once one KEY chunk
// (kvfrag) and one QUERY block (qfrag).  float4 units, lane = (fm = lane & 15, kq = lane >> 4):
//
//   kvfrag[((n * nchunk + ch) * H + h) * 8 + slot][lane]
//     slot 0,1: k[j = 16 ch + fm][h][4 kq + 0..3], [16 + 4 kq + 0..3]      slot 2: k_pts point kq (x, y, z), norm-step value
//     slot 3:   k_pts point 4 + kq (x, y, z), 0
//     slot 4 + s: (v[j = 16 ch + 4 kq + s][h][fm], v[..][16 + fm], v_pts coordinate (fm & 3) of point (fm >> 2), of point 4 + (fm >> 2); 0 for fm & 3 == 3)
//   qfrag[((n * nib + ib) * H + h) * 4 + slot][lane]   -- the same K order as slots 0..3 above, rows = queries i = 16 ib + fm, PRE-SCALED:
//     q / sqrt(D)   |   -2 c_h q_pts   |   norm-step value            c_h = -softplus(spatial_coef_h) sqrt(2/(9 P)) / 2   (ga.py:108-111)
//   norm step (one MFMA K-slice, the .w of slot 2):  kq = 0: (k: 1, q: c_h |q_pts|^2)   kq = 1: (k: |k_pts|^2, q: c_h)   kq = 2,3: 0
// (This is exactly the accumulator layout of the fused projection kernel node_frags.hip, whose weight rows are permuted so that a
//  16-column MFMA tile holds 4 points as (x, y, z, pad) quadruples; this kernel produces the same layout from a plain proj buffer.)
// so that  sum_K k'_j q'_i = q_i.k_j / sqrt(D) + c_h |q_pts_i - k_pts_j|^2  is ONE 15-step MFMA chain per head in the core
// (ga.py:84-85,108-111; the cancellation error of the expanded square is <= 2e-6 on the logit in the global frame, |p| <~ 10).
__global__ __launch_bounds__(256) void ipa_frags_kernel(const float* __restrict__ proj, const float* __restrict__ R, const float* __restrict__ t,
                                                        const float* __restrict__ spatial_coef, float* __restrict__ qfrag,
                                                        float* __restrict__ kvfrag, int L, int nchunk, int ldp) {
    __shared__ float pts[JC][3 * NPT + 2 * H + 4];               // q_pts | k_pts | v_pts (global frame) | |q_pts|^2 | |k_pts|^2 of the block's 16 rows
    __shared__ float coef[H];
    const int n = blockIdx.x / nchunk, ch = blockIdx.x % nchunk, tid = threadIdx.x;
    const int64_t rowbase = (int64_t)n * L;
    if (tid < H) {
        const float sc = spatial_coef[tid];
        const float gamma = (sc > 20.f) ? sc : log1pf(expf(sc));                     // softplus, ga.py:108
        coef[tid] = (-1.f * gamma * 0.16666666666666666f) / 2.f;                      // -gamma sqrt(2/(9*8)) / 2, ga.py:109-110
    }
    for (int item = tid; item < JC * 3 * H; item += 256) {
        const int r = item / (3 * H), sh = item % (3 * H), set = sh / H, h = sh % H;
        const int64_t row = rowbase + min(ch * JC + r, L - 1);                        // rows past the end are clamped copies (finite, never stored by the core)
        const float* p = proj + row * ldp + OFF_QP + set * NPT + h * (P * 3);
        const float* Rr = R + row * 9;
        const float* tr = t + row * 3;
        const float r0 = Rr[0], r1 = Rr[1], r2 = Rr[2], r3 = Rr[3], r4 = Rr[4], r5 = Rr[5], r6 = Rr[6], r7 = Rr[7], r8 = Rr[8];
        const float t0 = tr[0], t1 = tr[1], t2 = tr[2];
        float4 v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = reinterpret_cast<const float4*>(p)[q];
        const float* f = reinterpret_cast<const float*>(v);
        float nrm = 0.f;
        float* dst = &pts[r][set * NPT + h * (P * 3)];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float x = f[k * 3], y = f[k * 3 + 1], z = f[k * 3 + 2];
            const float gx = r0 * x + r1 * y + r2 * z + t0, gy = r3 * x + r4 * y + r5 * z + t1, gz = r6 * x + r7 * y + r8 * z + t2;
            dst[k * 3] = gx; dst[k * 3 + 1] = gy; dst[k * 3 + 2] = gz;
            nrm = fmaf(gx, gx, nrm); nrm = fmaf(gy, gy, nrm); nrm = fmaf(gz, gz, nrm);
        }
        if (set < 2) pts[r][3 * NPT + set * H + h] = nrm;
    }
    __syncthreads();
    float4* outk = reinterpret_cast<float4*>(kvfrag) + (int64_t)blockIdx.x * H * 8 * 64;
    for (int e = tid; e < H * 8 * 64; e += 256) {
        const int lane = e & 63, slot = (e >> 6) & 7, h = e >> 9, fm = lane & 15, kq = lane >> 4;
        float4 o;
        if (slot < 2) {
            const int64_t row = rowbase + min(ch * JC + fm, L - 1);
            o = *reinterpret_cast<const float4*>(proj + row * ldp + OFF_K + h * D + slot * 16 + kq * 4);
        } else if (slot == 2) {
            const float* kp = &pts[fm][NPT + h * (P * 3) + kq * 3];
            o = make_float4(kp[0], kp[1], kp[2], kq == 0 ? 1.f : (kq == 1 ? pts[fm][3 * NPT + H + h] : 0.f));
        } else if (slot == 3) {
            const float* kp = &pts[fm][NPT + h * (P * 3) + (4 + kq) * 3];
            o = make_float4(kp[0], kp[1], kp[2], 0.f);
        } else {
            const int r = kq * 4 + (slot - 4);
            const int64_t row = rowbase + min(ch * JC + r, L - 1);
            const float* vrow = proj + row * ldp + OFF_V + h * D;
            const float* vp = &pts[r][2 * NPT + h * (P * 3)];
            const int pt = fm >> 2, c = fm & 3;
            o = make_float4(vrow[fm], vrow[16 + fm], c < 3 ? vp[pt * 3 + c] : 0.f, c < 3 ? vp[(4 + pt) * 3 + c] : 0.f);
        }
        outk[e] = o;
    }
    float4* outq = reinterpret_cast<float4*>(qfrag) + (int64_t)blockIdx.x * H * 4 * 64;       // nib == nchunk (BI == JC)
    for (int e = tid; e < H * 4 * 64; e += 256) {
        const int lane = e & 63, slot = (e >> 6) & 3, h = e >> 8, fm = lane & 15, kq = lane >> 4;
        const float c = coef[h], m2c = -2.f * c;
        float4 o;
        if (slot < 2) {
            const int64_t row = rowbase + min(ch * JC + fm, L - 1);
            const float4 qv = *reinterpret_cast<const float4*>(proj + row * ldp + OFF_Q + h * D + slot * 16 + kq * 4);
            const float s = 0.17677669529663687f;                                      // 1 / sqrt(D), ga.py:84
            o = make_float4(qv.x * s, qv.y * s, qv.z * s, qv.w * s);
        } else if (slot == 2) {
            const float* qp = &pts[fm][h * (P * 3) + kq * 3];
            o = make_float4(m2c * qp[0], m2c * qp[1], m2c * qp[2], kq == 0 ? c * pts[fm][3 * NPT + h] : (kq == 1 ? c : 0.f));
        } else {
            const float* qp = &pts[fm][h * (P * 3) + (4 + kq) * 3];
            o = make_float4(m2c * qp[0], m2c * qp[1], m2c * qp[2], 0.f);
        }
        outq[e] = o;
    }
}

static_assert(BI == JC, "a 16-residue block is both a key chunk and a query block");
size_t ipa_kvfrag_floats(int N, int L) { return (size_t)N * ((L + JC - 1) / JC) * H * 8 * 64 * 4; }
size_t ipa_qfrag_floats(int N, int L) { return (size_t)N * ((L + BI - 1) / BI) * H * 4 * 64 * 4; }

int launch_ipa_frags(const float* proj, const float* R, const float* t, const float* spatial_coef, float* qfrag, float* kvfrag, int N, int L, hipStream_t st, int ldp) {
    if ((int64_t)N * L == 0) return ABOPT_OK;
    const int nchunk = (L + JC - 1) / JC;
    hipLaunchKernelGGL(ipa_frags_kernel, dim3((unsigned)(N * nchunk)), dim3(256), 0, st, proj, R, t, spatial_coef, qfrag, kvfrag, L, nchunk, ldp);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// debug only: the reference-layout intermediates [N,L,L,12] from the core's head-major dump (x = logit * sqrt(1/3) * log2 e, unmasked,
// and the final (max, sum) of every row): logits = x / log2 e, alpha = mask ? exp2(x - m) / l : 0  (ga.py:11-26,165-166)
__global__ __launch_bounds__(256) void debug_from_dump_kernel(const float* __restrict__ dump, const float* __restrict__ stats, const uint8_t* __restrict__ mask,
                                                              float* __restrict__ logits, float* __restrict__ alpha, int L, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // ((n * L + i) * L + j) * H + h
    if (e >= total) return;
    const int h = (int)(e % H);
    const int64_t rj = e / H, row = rj / L;
    const int j = (int)(rj % L);
    const int64_t n = row / L;
    const int i = (int)(row % L);
    const float x = dump[((n * H + h) * L + i) * L + j];
    if (logits) logits[e] = x * 0.6931471805599453f;
    if (alpha) {
        const bool live = mask[row] != 0 && mask[n * L + j] != 0;
        alpha[e] = live ? __builtin_amdgcn_exp2f(x - stats[(row * H + h) * 2]) / stats[(row * H + h) * 2 + 1] : 0.f;
    }
}

// ------------------------------------------------------------------ measurement hook (see abopt_prof_enable)
namespace prof {
static bool g_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
static size_t g_used = 0;
static bool g_span_on = false;               // launch spans (abopt_prof_enable(3)): slot numbers handed to the 32-row launches
static int g_span_next = 0;
int next_span_slot() { return (g_span_on && g_span_next < 2048) ? g_span_next++ : -1; }
void begin(hipStream_t st) {
    if (!g_on) return;
    if (g_used == g_pool.size()) {
        hipEvent_t a, b;
        // no system-scope fence at the record: the default flavour writes back / invalidates L2 around every bracket, which evicts the
        // key/value fragments node_frags has just produced and makes the bracketed kernel ~9 % slower than it runs unobserved
        // (166 vs 153 us against rocprofv3 at the bench shape); timestamps are unaffected
        if (hipEventCreateWithFlags(&a, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableSystemFence) != hipSuccess) { g_on = false; return; }
        g_pool.emplace_back(a, b);
    }
    (void)hipEventRecord(g_pool[g_used].first, st);
}
void end(hipStream_t st) {
    if (!g_on) return;
    (void)hipEventRecord(g_pool[g_used].second, st);
    ++g_used;
}
}  // namespace prof

int launch_ipa_core(const float* qfrag, const float* kvfrag, const float* z, const uint8_t* mask, const float* R, const float* t,
                    const float* w_pair_bias, float* feat, float* dbg_logits, float* dbg_alpha, const float* pair_bias_cache,
                    int N, int L, hipStream_t st, int z_shared, float* split_ws, size_t split_ws_floats, const float* pair_terms) {
    if (N == 0 || L == 0) return ABOPT_OK;
    const int nib = (L + BI - 1) / BI;
    ABOPT_CHECK_ARG((int64_t)N * nib < (1ll << 31), "ipa_core: grid too large");
    ABOPT_CHECK_ARG(!z_shared || pair_bias_cache, "ipa_core: a shared pair_feat comes with its shared pair-bias cache");
    if (!dbg_logits && !dbg_alpha)
        return launch_ipa_core_kernel(qfrag, kvfrag, z, mask, R, t, w_pair_bias, feat, nullptr, nullptr, pair_bias_cache, N, L, st, z_shared, split_ws, split_ws_floats, pair_terms);
    // parity-test path only: a stream-ordered temporary for the head-major dump (the product path never allocates)
    const size_t nd = (size_t)N * H * L * L, ns = (size_t)N * L * H * 2;
    float* tmp = nullptr;
    ABOPT_HIP(hipMallocAsync(reinterpret_cast<void**>(&tmp), (nd + ns) * sizeof(float), st));
    int rc = launch_ipa_core_kernel(qfrag, kvfrag, z, mask, R, t, w_pair_bias, feat, tmp, tmp + nd, pair_bias_cache, N, L, st, z_shared);
    if (!rc) {
        const int64_t total = (int64_t)N * L * L * H;
        hipLaunchKernelGGL(debug_from_dump_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tmp, tmp + nd, mask, dbg_logits, dbg_alpha, L, total);
        if (hipGetLastError() != hipSuccess) rc = ABOPT_EHIP;
    }
    (void)hipFreeAsync(tmp, st);
    if (rc) return rc;
    return ABOPT_OK;
}

}  // namespace abopt

extern "C" int abopt_prof_enable(int on) {
    abopt::prof::g_on = on == 1;
    if (on != 2 && on != 4) abopt::prof::g_used = 0;        // 2: stop bracketing new launches but keep the recorded pairs (see abopt_prof_peek)
    if (on == 3) { abopt::prof::g_span_on = true; abopt::prof::g_span_next = 0; }      // 3: launch spans on, slots from 0 (use while capturing a graph)
    else if (on != 4) abopt::prof::g_span_on = false;                                  // 4: stop handing out slots, keep the count
    else abopt::prof::g_span_on = false;
    return ABOPT_OK;
}
extern "C" int abopt_prof_spans_reset(abopt_stream stream) { return abopt::prof_spans_reset((hipStream_t)stream); }
extern "C" int abopt_prof_spans(int* launches, double* total_ms) {
    ABOPT_CHECK_ARG(launches && total_ms, "prof_spans: NULL argument");
    return abopt::prof_spans_read(abopt::prof::g_span_next, launches, total_ms);
}

static int prof_sum(int* launches, double* total_ms, bool reset) {
    double tot = 0.0;
    for (size_t i = 0; i < abopt::prof::g_used; ++i) {
        float ms = 0.f;
        ABOPT_HIP(hipEventSynchronize(abopt::prof::g_pool[i].second));
        ABOPT_HIP(hipEventElapsedTime(&ms, abopt::prof::g_pool[i].first, abopt::prof::g_pool[i].second));
        tot += ms;
    }
    if (launches) *launches = (int)abopt::prof::g_used;
    if (total_ms) *total_ms = tot;
    if (reset) abopt::prof::g_used = 0;
    return ABOPT_OK;
}
extern "C" int abopt_prof_collect(int* launches, double* total_ms) { return prof_sum(launches, total_ms, true); }
extern "C" int abopt_prof_clock(long long* cycles, long long* wall_ticks_100mhz) {
    ABOPT_CHECK_ARG(cycles && wall_ticks_100mhz, "prof_clock: NULL argument");
    ABOPT_HIP(hipDeviceSynchronize());
    return abopt::read_clock_probe(cycles, wall_ticks_100mhz);
}
// the same without forgetting the pairs: event records captured into a hipGraph are re-recorded by every replay
extern "C" int abopt_prof_peek(int* launches, double* total_ms) { return prof_sum(launches, total_ms, false); }

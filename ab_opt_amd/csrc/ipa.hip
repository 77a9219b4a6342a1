// Invariant-point-attention core for CDNA4 (gfx950).
//
// Replaces, fused in one pass over the pair features z (reference AbDock/src/modules/encoders/ga.py):
//   _node_logits :81-86, _pair_logits :88-90, _spatial_logits :92-112, _alpha_from_logits :11-26,
//   _pair_aggregation :114-118, _node_aggregation :120-125, _spatial_aggregation :127-147.
// The reference materialises (N,L,L,12,{24,32,64}) temporaries; here z[n,i,:,:] is read from HBM
// exactly once per (n,i) and everything else stays on chip.
#include "ipa_common.h"
#include "kernels.h"
#include <vector>
#include <utility>

namespace abopt {

// ------------------------------------------------------------------ local -> global of the point sets
// the wave-specialised kernel stages the key mask in LDS (WS_MAX_L = 2048 keys); longer complexes take the single-role kernel
bool ipa_uses_kvfrag(int L) { return L <= 2048; }

// geometry.py:72-91 applied to proj_{query,key,value}_point outputs (ga.py:96-105,129-132): p <- R p + t, in place, one
// thread per (residue, point set, head).  Also emits |p|^2 summed over the head's 8 points for the query and key sets:
// the wave-specialised IPA kernel evaluates the squared point distances as |q|^2 + |k|^2 - 2 q.k on the matrix cores.
__global__ __launch_bounds__(256) void points_to_global_kernel(float* __restrict__ proj, const float* __restrict__ R,
                                                               const float* __restrict__ t, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * 3 * H) return;
    const int64_t row = idx / (3 * H);
    const int sh = (int)(idx % (3 * H)), set = sh / H, h = sh % H;
    float* p = proj + row * NP + OFF_QP + set * (H * P * 3) + h * (P * 3);
    const float* Rr = R + row * 9;
    const float* tr = t + row * 3;
    const float r0 = Rr[0], r1 = Rr[1], r2 = Rr[2], r3 = Rr[3], r4 = Rr[4], r5 = Rr[5], r6 = Rr[6], r7 = Rr[7], r8 = Rr[8];
    const float t0 = tr[0], t1 = tr[1], t2 = tr[2];
    float4 v[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = reinterpret_cast<const float4*>(p)[q];
    float* f = reinterpret_cast<float*>(v);
    float nrm = 0.f;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const float x = f[k * 3], y = f[k * 3 + 1], z = f[k * 3 + 2];
        const float gx = r0 * x + r1 * y + r2 * z + t0, gy = r3 * x + r4 * y + r5 * z + t1, gz = r6 * x + r7 * y + r8 * z + t2;
        f[k * 3] = gx; f[k * 3 + 1] = gy; f[k * 3 + 2] = gz;
        nrm = fmaf(gx, gx, nrm); nrm = fmaf(gy, gy, nrm); nrm = fmaf(gz, gz, nrm);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) reinterpret_cast<float4*>(p)[q] = v[q];
    if (set == 0) proj[row * NP + OFF_NQ + h] = nrm;
    if (set == 1) proj[row * NP + OFF_NK + h] = nrm;
}

// Same transform, plus the key / value operands of the wave-specialised IPA kernel re-laid out in MFMA fragment order
// ("kvfrag"): one workgroup per (sample, 16-key chunk).  Every i-block of a sample consumes the same 16 x (k | k_pts | v |
// v_pts) tile per chunk; gathering it row by row from proj costs the node waves ~40 poorly coalesced loads per chunk, the
// fragment copy turns that into 24 fully coalesced 1 KB loads.  Layout, float4 units:
//   kvfrag[((n * nchunk + ch) * H + h) * 8 + slot][lane],  lane = (fm = lane & 15, kq = lane >> 4)
//   slot 0,1: k[j = 16 ch + fm][h][8 kq + 0..3], [.. + 4..7]          slot 2: k_pts coords (2 kq + {0,1}, 8 + 2 kq + {0,1}) of head h
//   slot 3:   k_pts coords 16 + 2 kq + {0,1}, |k_pts|^2, 0             slot 4 + s: (v[j = 16 ch + 4 kq + s][h][2 fm + {0,1}], v_pts coords 2 fm + {0,1} or 0 for fm >= 12)
__global__ __launch_bounds__(256) void points_to_global_frags_kernel(float* __restrict__ proj, const float* __restrict__ R,
                                                                     const float* __restrict__ t, float* __restrict__ kvfrag, int L, int nchunk) {
    __shared__ float pts[JC][2 * NPT + H + 4];                   // k_pts | v_pts (global frame) | |k_pts|^2 of the chunk's 16 rows
    const int n = blockIdx.x / nchunk, ch = blockIdx.x % nchunk, tid = threadIdx.x;
    const int64_t rowbase = (int64_t)n * L;
    for (int item = tid; item < JC * 3 * H; item += 256) {
        const int r = item / (3 * H), sh = item % (3 * H), set = sh / H, h = sh % H;
        const int j = ch * JC + r;
        const int64_t row = rowbase + min(j, L - 1);
        float* p = proj + row * NP + OFF_QP + set * NPT + h * (P * 3);
        const float* Rr = R + row * 9;
        const float* tr = t + row * 3;
        const float r0 = Rr[0], r1 = Rr[1], r2 = Rr[2], r3 = Rr[3], r4 = Rr[4], r5 = Rr[5], r6 = Rr[6], r7 = Rr[7], r8 = Rr[8];
        const float t0 = tr[0], t1 = tr[1], t2 = tr[2];
        float4 v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = reinterpret_cast<const float4*>(p)[q];
        float* f = reinterpret_cast<float*>(v);
        float nrm = 0.f;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float x = f[k * 3], y = f[k * 3 + 1], z = f[k * 3 + 2];
            const float gx = r0 * x + r1 * y + r2 * z + t0, gy = r3 * x + r4 * y + r5 * z + t1, gz = r6 * x + r7 * y + r8 * z + t2;
            f[k * 3] = gx; f[k * 3 + 1] = gy; f[k * 3 + 2] = gz;
            nrm = fmaf(gx, gx, nrm); nrm = fmaf(gy, gy, nrm); nrm = fmaf(gz, gz, nrm);
        }
        if (set >= 1) {
#pragma unroll
            for (int k = 0; k < P * 3; ++k) pts[r][(set - 1) * NPT + h * (P * 3) + k] = f[k];
            if (set == 1) pts[r][2 * NPT + h] = nrm;
        }
        if (j < L && set == 0) {                                   // only the query points are read back from proj (the wave-specialised kernel
#pragma unroll                                                     // takes key/value points from kvfrag); rows past the end are clamped copies
            for (int q = 0; q < 6; ++q) reinterpret_cast<float4*>(p)[q] = v[q];
            proj[row * NP + OFF_NQ + h] = nrm;
        }
    }
    __syncthreads();
    float4* out = reinterpret_cast<float4*>(kvfrag) + (int64_t)blockIdx.x * H * 8 * 64;
    for (int e = tid; e < H * 8 * 64; e += 256) {
        const int lane = e & 63, slot = (e >> 6) & 7, h = e >> 9, fm = lane & 15, kq = lane >> 4;
        float4 o;
        if (slot < 2) {
            const int64_t row = rowbase + min(ch * JC + fm, L - 1);
            o = *reinterpret_cast<const float4*>(proj + row * NP + OFF_K + h * D + kq * 8 + slot * 4);
        } else if (slot == 2) {
            const float* kp = &pts[fm][h * (P * 3)];
            o = make_float4(kp[2 * kq], kp[2 * kq + 1], kp[8 + 2 * kq], kp[8 + 2 * kq + 1]);
        } else if (slot == 3) {
            const float* kp = &pts[fm][h * (P * 3)];
            o = make_float4(kp[16 + 2 * kq], kp[16 + 2 * kq + 1], pts[fm][2 * NPT + h], 0.f);
        } else {
            const int r = kq * 4 + (slot - 4);
            const int64_t row = rowbase + min(ch * JC + r, L - 1);
            const float2 vv = *reinterpret_cast<const float2*>(proj + row * NP + OFF_V + h * D + 2 * fm);
            const float* vp = &pts[r][NPT + h * (P * 3)];
            o = make_float4(vv.x, vv.y, fm < 12 ? vp[2 * fm] : 0.f, fm < 12 ? vp[2 * fm + 1] : 0.f);
        }
        out[e] = o;
    }
}

size_t ipa_kvfrag_floats(int N, int L) { return (size_t)N * ((L + JC - 1) / JC) * H * 8 * 64 * 4; }

int launch_points_to_global(float* proj, const float* R, const float* t, int64_t rows, hipStream_t st, float* kvfrag, int N, int L) {
    const int64_t total = rows * 3 * H;
    if (total == 0) return ABOPT_OK;
    if (kvfrag && ipa_uses_kvfrag(L)) {
        const int nchunk = (L + JC - 1) / JC;
        hipLaunchKernelGGL(points_to_global_frags_kernel, dim3((unsigned)(N * nchunk)), dim3(256), 0, st, proj, R, t, kvfrag, L, nchunk);
    } else {
        hipLaunchKernelGGL(points_to_global_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, proj, R, t, rows);
    }
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// ------------------------------------------------------------------ IPA core, version 1 (MFMA, flash-style over j)
// One 256-thread workgroup = 16 query residues (i-block) of one sample.  Keys are consumed JC=16 at a time with an
// online softmax, so any L works and z[n,i,:,:] is read from HBM exactly once.  All five contractions run on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32: an exact fp32 FMA chain, so the 1e-5 parity budget holds):
//
//   "i-batched" phases (waves split the 12 heads, M = the 16 query rows):
//     A: S_node[i,j] = q_i.k_j  (K=32)      + the squared point distances on the VALU            -> LDS  S[i][h][j]
//     C: fn[i,d] += P[i,j] v[j,d] (K=j), pts[i,e] += P[i,j] vp[j,e]                                <- LDS  P[i][h][j]
//   "per-i" phase (each wave owns 4 query rows, N = heads):
//     B: pair bias  lp[j,h] = z[j,:].Wb[h,:] (M=j, K=64);  S = (S_node + lp) sqrt(1/3), mask, running max/sum;
//        P = exp(S - m);  fp[c,h] += z[j,c] P[j,h] (M=c, K=j).  The softmax statistics live in lanes (h = lane&15), which is
//        exactly the B-operand layout of the aggregation MFMA and its accumulator column, so P never moves between lanes.
//   z chunk of a row: one fully coalesced global load (4 rows x 256 B per wave instruction) that is already the A operand
//   of the aggregation; a wave-private LDS tile transposes it into the A operand of the pair-bias MFMA.
struct IpaSmem {
    float sp[BI][16 * PLD + 4];   // S (phase A -> B), then P (phase B -> C), [i][h*PLD + j]; +4: odd slot stride across i; reused for the points at the end
    float zst[4][JC][ZSLD];       // per-wave z staging
    float qg[BI][NPT + 4];        // global-frame query points of the 16 rows (+4: rows 4 kq + r land on distinct banks)
    float scl[BI][16];            // per-(i,h) rescale factor of the current chunk
    float lsum[BI][16];           // softmax denominators
    float wbs[16][C + 4];         // pair-bias weights, rows 12..15 zero
    float coef[16];               // -softplus(spatial_coef) sqrt(2/(9 P)) / 2 per head
};

template <bool DBG>
__global__ __launch_bounds__(256, 2) void ipa_core_v1_kernel(const float* __restrict__ proj, const float* __restrict__ z,
                                                             const uint8_t* __restrict__ mask, const float* __restrict__ R,
                                                             const float* __restrict__ t, const float* __restrict__ Wb,
                                                             const float* __restrict__ spatial_coef, float* __restrict__ feat,
                                                             float* __restrict__ dbg_logits, int N, int L, int nib, int xcd_remap) {
    __shared__ __attribute__((aligned(16))) IpaSmem sm;
    // ---- block -> (sample, i-block).  With N % 8 == 0 all i-blocks of a sample run on one XCD (blocks are dealt
    // round-robin to the 8 XCDs), so the sample's k/v/point tiles stay in that XCD's L2.  Speed only, never correctness.
    int n, ib;
    {
        const int b = blockIdx.x;
        if (xcd_remap) { const int xcd = b & 7, k = b >> 3; n = xcd + 8 * (k / nib); ib = k % nib; }
        else { n = b / nib; ib = b % nib; }
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fm = lane & 15, kq = lane >> 4;
    const int i0 = ib * BI;
    const int64_t rowbase = (int64_t)n * L;
    const float* projn = proj + rowbase * NP;

    // ---- prologue
    for (int e = tid; e < BI * (NPT / 4); e += 256) {          // query points of the block -> LDS
        const int il = e / (NPT / 4), c4 = e % (NPT / 4);
        const int i = min(i0 + il, L - 1);
        reinterpret_cast<float4*>(&sm.qg[il][0])[c4] = reinterpret_cast<const float4*>(projn + (int64_t)i * NP + OFF_QP)[c4];
    }
    if (tid < H) {
        const float sc = spatial_coef[tid];
        const float gamma = (sc > 20.f) ? sc : log1pf(expf(sc));                       // softplus, ga.py:108
        sm.coef[tid] = (-1.f * gamma * 0.16666666666666666f) / 2.f;                     // -gamma sqrt(2/(9*8)) / 2, ga.py:109-110
    }
    const float* qrow = projn + (int64_t)min(i0 + fm, L - 1) * NP + OFF_Q + kq * 8;   // q fragments are re-read per chunk (L2 hits)
    for (int e = tid; e < 16 * (C / 4); e += 256) {             // pair-bias weights -> LDS (B operand: n = head, step s <-> c = 16 kq + s)
        const int h = e / (C / 4), c4 = e % (C / 4);
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h < H) w4 = reinterpret_cast<const float4*>(Wb + h * C)[c4];
        *reinterpret_cast<float4*>(&sm.wbs[h][c4 * 4]) = w4;
    }
    bool mi_b[4];                                                // masks of this wave's 4 query rows (phase B)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) { const int i = i0 + wave * 4 + ii; mi_b[ii] = (i < L) && mask[rowbase + i] != 0; }
    bool mi_a[4];                                                // masks of rows 4 kq + r (phases A / C)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int i = i0 + kq * 4 + r; mi_a[r] = (i < L) && mask[rowbase + i] != 0; }

    float m_run[4], l_run[4];
    f32x4 accP[4][4];                                            // pair aggregation: [row ii][c-tile]
    f32x4 accV[3][2], accT[3][2];                                // node / point aggregation: [head][n-tile]
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        m_run[ii] = -INFINITY; l_run[ii] = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) accP[ii][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int hh = 0; hh < 3; ++hh)
#pragma unroll
        for (int k = 0; k < 2; ++k) { accV[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; accT[hh][k] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();

    f32x4 znext[4];                                             // z prefetch: row (4 wave), chunk 0
    {
        const float* zi = z + ((rowbase + min(i0 + wave * 4, L - 1)) * (int64_t)L) * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) znext[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(zi + (int64_t)min(kq * 4 + r, L - 1) * C) + fm);
    }
    const int nchunk = (L + JC - 1) / JC;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int jc0 = ch * JC;
        // ================================================================= phase A: node + spatial logits -> sp
        {
            const int j = min(jc0 + fm, L - 1);
            const float* pj = projn + (int64_t)j * NP;
#pragma unroll 1
            for (int hh = 0; hh < 3; ++hh) {
                const int h = wave * 3 + hh;
                const float coefh = sm.coef[h];
                const float4 k0 = reinterpret_cast<const float4*>(pj + OFF_K + h * D + kq * 8)[0];
                const float4 k1 = reinterpret_cast<const float4*>(pj + OFF_K + h * D + kq * 8)[1];
                float4 kg[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) kg[q] = reinterpret_cast<const float4*>(pj + OFF_KP + h * (P * 3))[q];
                // A operand: row = query fm, K-permuted: step s <-> channel 8 kq + s (same permutation on the key side)
                const float4 q0 = reinterpret_cast<const float4*>(qrow + h * D)[0], q1 = reinterpret_cast<const float4*>(qrow + h * D)[1];
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = mfma4(s < 4 ? f4get(q0, s) : f4get(q1, s - 4), s < 4 ? f4get(k0, s) : f4get(k1, s - 4), acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {                    // accumulator row = query 4 kq + r, column = key fm
                    const float4* qgp = reinterpret_cast<const float4*>(&sm.qg[kq * 4 + r][h * (P * 3)]);
                    float d2 = 0.f;
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const float4 a = qgp[q];
                        const float dx = a.x - kg[q].x, dy = a.y - kg[q].y, dz = a.z - kg[q].z, dw = a.w - kg[q].w;
                        d2 = fmaf(dx, dx, d2); d2 = fmaf(dy, dy, d2); d2 = fmaf(dz, dz, d2); d2 = fmaf(dw, dw, d2);
                    }
                    sm.sp[kq * 4 + r][h * PLD + fm] = acc[r] * 0.17677669529663687f + d2 * coefh;
                }
            }
        }
        __syncthreads();
        // ================================================================= phase B: pair bias, softmax, pair aggregation
        {
            bool mj[4], jv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int j = jc0 + kq * 4 + r; jv[r] = j < L; mj[r] = jv[r] && mask[rowbase + j] != 0; }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int il = wave * 4 + ii;
                f32x4 zr[4];                                     // this row's chunk (prefetched): aggregation A operand
#pragma unroll
                for (int r = 0; r < 4; ++r) zr[r] = znext[r];
                {                                                // prefetch the next row's chunk (next chunk's first row after ii = 3)
                    const int ni = (ii < 3) ? ii + 1 : 0;
                    const int njc0 = (ii < 3) ? jc0 : jc0 + JC;
                    const int i = min(i0 + wave * 4 + ni, L - 1);
                    const float* zi = z + ((rowbase + i) * (int64_t)L) * C;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        znext[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(zi + (int64_t)min(njc0 + kq * 4 + r, L - 1) * C) + fm);
                }
                wave_lds_sync();                                            // previous row's transposed reads are done
#pragma unroll
                for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(&sm.zst[wave][kq * 4 + r][fm * 4]) = zr[r];
                wave_lds_sync();                                            // cross-lane transpose through LDS
                f32x4 acc4[4];                                   // four independent chains (the 16x16x4 MFMA has a 40-cycle dependent latency)
#pragma unroll
                for (int q = 0; q < 4; ++q) {                    // pair-bias: A row fm, channels 16 kq + 4 q ..; B = Wb rows (heads)
                    const float4 za = *reinterpret_cast<const float4*>(&sm.zst[wave][fm][kq * 16 + q * 4]);
                    const float4 wv = *reinterpret_cast<const float4*>(&sm.wbs[fm][kq * 16 + q * 4]);
                    acc4[q] = mfma4(za.x, wv.x, (f32x4){0.f, 0.f, 0.f, 0.f});
                    acc4[q] = mfma4(za.y, wv.y, acc4[q]); acc4[q] = mfma4(za.z, wv.z, acc4[q]); acc4[q] = mfma4(za.w, wv.w, acc4[q]);
                }
                const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
                const float4 tns = *reinterpret_cast<const float4*>(&sm.sp[il][fm * PLD + kq * 4]);
                float sv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {                    // accumulator row = key 4 kq + r, column = head fm
                    float lt = (f4get(tns, r) + acc[r]) * 0.5773502691896258f;
                    if (DBG && jv[r] && fm < H && (i0 + il) < L) dbg_logits[((rowbase + i0 + il) * L + jc0 + kq * 4 + r) * H + fm] = lt;
                    if (!(mi_b[ii] && mj[r])) lt -= 1e5f;        // ga.py:20-23
                    sv[r] = (fm < H) ? lt : 0.f;
                    sv[r] = jv[r] ? sv[r] : -INFINITY;
                }
                float mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
                mx = rows_max(mx);
                const float m_new = fmaxf(m_run[ii], mx);
                const float sc = __expf(m_run[ii] - m_new);
                float pv[4], ps = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pv[r] = __expf(sv[r] - m_new); ps += pv[r]; }
                ps = rows_sum(ps);
                l_run[ii] = l_run[ii] * sc + ps;
                m_run[ii] = m_new;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) accP[ii][mt] *= sc;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) accP[ii][mt] = mfma4(zr[r][mt], pv[r], accP[ii][mt]);
                *reinterpret_cast<float4*>(&sm.sp[il][fm * PLD + kq * 4]) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                if (kq == 0) sm.scl[il][fm] = sc;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ================================================================= phase C: node / point aggregation
        {
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                const int h = wave * 3 + hh;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = sm.scl[kq * 4 + r][h];
                    accV[hh][0][r] *= sc; accV[hh][1][r] *= sc; accT[hh][0][r] *= sc; accT[hh][1][r] *= sc;
                }
                const float4 pa = *reinterpret_cast<const float4*>(&sm.sp[fm][h * PLD + kq * 4]);     // A: row = query fm, step s <-> key 4 kq + s
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float* pj = projn + (int64_t)min(jc0 + kq * 4 + s, L - 1) * NP;
                    const float2 vb = reinterpret_cast<const float2*>(pj + OFF_V + h * D)[fm];    // channels 2 fm (+1): n-tile 0 / 1
                    float2 tb = make_float2(0.f, 0.f);
                    if (fm < 12) tb = reinterpret_cast<const float2*>(pj + OFF_VP + h * (P * 3))[fm];
                    const float a = f4get(pa, s);
                    accV[hh][0] = mfma4(a, vb.x, accV[hh][0]);
                    accV[hh][1] = mfma4(a, vb.y, accV[hh][1]);
                    accT[hh][0] = mfma4(a, tb.x, accT[hh][0]);
                    accT[hh][1] = mfma4(a, tb.y, accT[hh][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // no barrier needed here: phase A of the next chunk writes only this wave's own head slices of sp, which only this
        // wave read in phase C; phase B of the next chunk starts behind the barrier after phase A.
    }

    // ---- finalisation: alpha = P / l, zero for masked queries (ga.py:24-25)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int il = wave * 4 + ii, i = i0 + il;
        if (kq == 0) sm.lsum[il][fm] = l_run[ii];
        if (i < L && fm < H) {
            const float inv = mi_b[ii] ? 1.f / l_run[ii] : 0.f;
            float* fo = feat + (rowbase + i) * FEAT + fm * C + kq * 16;        // accumulator row c_local = 4 kq + r', tile mt: c = 16 kq + 4 r' + mt
#pragma unroll
            for (int r = 0; r < 4; ++r)
                reinterpret_cast<float4*>(fo)[r] = make_float4(accP[ii][0][r] * inv, accP[ii][1][r] * inv, accP[ii][2][r] * inv, accP[ii][3][r] * inv);
        }
    }
    __syncthreads();                                             // lsum visible; every wave is done reading sp
    float* pts = &sm.sp[0][0];                                // [BI][H][24] aggregated global-frame points
#pragma unroll
    for (int hh = 0; hh < 3; ++hh) {
        const int h = wave * 3 + hh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int il = kq * 4 + r, i = i0 + il;
            const float inv = mi_a[r] ? 1.f / sm.lsum[il][h] : 0.f;
            if (i < L)
                reinterpret_cast<float2*>(feat + (rowbase + i) * FEAT + H * C + h * D)[fm] = make_float2(accV[hh][0][r] * inv, accV[hh][1][r] * inv);
            if (fm < 12) *reinterpret_cast<float2*>(&pts[(il * H + h) * (P * 3) + 2 * fm]) = make_float2(accT[hh][0][r] * inv, accT[hh][1][r] * inv);
        }
    }
    __syncthreads();
    for (int e = tid; e < BI * H * P; e += 256) {                // local frame, norm, direction (ga.py:136-139)
        const int il = e / (H * P), hp = e % (H * P), i = i0 + il;
        if (i >= L) continue;
        const float* Rr = R + (rowbase + i) * 9;
        const float* tr = t + (rowbase + i) * 3;
        const float* a = pts + (il * H * P + hp) * 3;
        const float dx = a[0] - tr[0], dy = a[1] - tr[1], dz = a[2] - tr[2];
        const float lx = Rr[0] * dx + Rr[3] * dy + Rr[6] * dz;
        const float ly = Rr[1] * dx + Rr[4] * dy + Rr[7] * dz;
        const float lz = Rr[2] * dx + Rr[5] * dy + Rr[8] * dz;
        const float dist = sqrtf(lx * lx + ly * ly + lz * lz);
        const float inv = 1.f / (dist + 1e-4f);
        float* fpnt = feat + (rowbase + i) * FEAT + H * C + H * D;
        fpnt[hp * 3 + 0] = lx; fpnt[hp * 3 + 1] = ly; fpnt[hp * 3 + 2] = lz;
        fpnt[H * P * 3 + hp] = dist;
        float* fdir = fpnt + H * P * 3 + H * P;
        fdir[hp * 3 + 0] = lx * inv; fdir[hp * 3 + 1] = ly * inv; fdir[hp * 3 + 2] = lz * inv;
    }
}

// debug only: alpha from the unmasked logits the fused kernel dumped (ga.py:11-26), one wave per (n, i, h)
__global__ __launch_bounds__(64) void alpha_from_logits_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                               float* __restrict__ alpha, int L) {
    const int64_t row = blockIdx.x;                              // n * L + i
    const int h = blockIdx.y, lane = threadIdx.x;
    const int64_t nbase = (row / L) * L;
    const bool mi = mask[row] != 0;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
        float v = logits[(row * L + j) * H + h];
        if (!(mi && mask[nbase + j] != 0)) v -= 1e5f;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sm = 0.f;
    for (int j = lane; j < L; j += 64) {
        float v = logits[(row * L + j) * H + h];
        if (!(mi && mask[nbase + j] != 0)) v -= 1e5f;
        sm += expf(v - mx);
    }
    sm = wave_sum(sm);
    for (int j = lane; j < L; j += 64) {
        float v = logits[(row * L + j) * H + h];
        if (!(mi && mask[nbase + j] != 0)) v -= 1e5f;
        alpha[(row * L + j) * H + h] = mi ? expf(v - mx) / sm : 0.f;
    }
}

// ------------------------------------------------------------------ measurement hook (see abopt_prof_enable)
namespace prof {
static bool g_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
static size_t g_used = 0;
void begin(hipStream_t st) {
    if (!g_on) return;
    if (g_used == g_pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { g_on = false; return; }
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

int launch_ipa_core(const float* proj, const float* z, const uint8_t* mask, const float* R, const float* t,
                    const float* w_pair_bias, const float* spatial_coef, float* feat,
                    float* dbg_logits, float* dbg_alpha, const float* pair_bias_cache, const float* kvfrag, int N, int L, hipStream_t st, int z_shared) {
    if (N == 0 || L == 0) return ABOPT_OK;
    if (z_shared && !ipa_uses_kvfrag(L)) { set_error("ipa_core: a shared pair_feat needs the wave-specialised kernel (L <= 2048)"); return ABOPT_EUNSUPPORTED; }
    const int nib = (L + BI - 1) / BI;
    ABOPT_CHECK_ARG((int64_t)N * nib < (1ll << 31), "ipa_core: grid too large");
    if (ipa_uses_kvfrag(L)) {
        ABOPT_CHECK_ARG(kvfrag != nullptr, "ipa_core: the wave-specialised kernel needs the key/value fragment buffer");
        int rc = launch_ipa_core_ws(proj, z, mask, R, t, w_pair_bias, spatial_coef, feat, dbg_logits, pair_bias_cache, kvfrag, N, L, st, z_shared);
        if (rc) return rc;
        if (dbg_alpha) {
            ABOPT_CHECK_ARG(dbg_logits != nullptr, "ipa_core: alpha dump needs the logits dump");
            hipLaunchKernelGGL(alpha_from_logits_kernel, dim3((unsigned)(N * L), H), dim3(64), 0, st, dbg_logits, mask, dbg_alpha, L);
            ABOPT_LAUNCH_CHECK();
        }
        return ABOPT_OK;
    }
    prof::begin(st);
    if (dbg_logits)
        hipLaunchKernelGGL(ipa_core_v1_kernel<true>, dim3((unsigned)(N * nib)), dim3(256), 0, st, proj, z, mask, R, t, w_pair_bias, spatial_coef,
                           feat, dbg_logits, N, L, nib, (N % 8 == 0) ? 1 : 0);
    else
        hipLaunchKernelGGL(ipa_core_v1_kernel<false>, dim3((unsigned)(N * nib)), dim3(256), 0, st, proj, z, mask, R, t, w_pair_bias, spatial_coef,
                           feat, dbg_logits, N, L, nib, (N % 8 == 0) ? 1 : 0);
    prof::end(st);
    ABOPT_LAUNCH_CHECK();
    if (dbg_alpha) {
        ABOPT_CHECK_ARG(dbg_logits != nullptr, "ipa_core: alpha dump needs the logits dump");
        hipLaunchKernelGGL(alpha_from_logits_kernel, dim3((unsigned)(N * L), H), dim3(64), 0, st, dbg_logits, mask, dbg_alpha, L);
        ABOPT_LAUNCH_CHECK();
    }
    return ABOPT_OK;
}

}  // namespace abopt

extern "C" int abopt_prof_enable(int on) {
    abopt::prof::g_on = on != 0;
    abopt::prof::g_used = 0;
    return ABOPT_OK;
}

extern "C" int abopt_prof_collect(int* launches, double* total_ms) {
    double tot = 0.0;
    for (size_t i = 0; i < abopt::prof::g_used; ++i) {
        float ms = 0.f;
        ABOPT_HIP(hipEventSynchronize(abopt::prof::g_pool[i].second));
        ABOPT_HIP(hipEventElapsedTime(&ms, abopt::prof::g_pool[i].first, abopt::prof::g_pool[i].second));
        tot += ms;
    }
    if (launches) *launches = (int)abopt::prof::g_used;
    if (total_ms) *total_ms = tot;
    abopt::prof::g_used = 0;
    return ABOPT_OK;
}

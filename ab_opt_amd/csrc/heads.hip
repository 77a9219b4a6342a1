// The three denoiser heads of EpsilonNet after the encoder (AbDock/src/modules/diffusion/dpm_full.py:92-101: eps_crd_net, eps_rot_net,
// eps_seq_net, each Linear(F+3,F) ReLU Linear(F,F) ReLU Linear(F,out)) in ONE launch; round 1 ran them as nine small GEMMs plus a
// feature-building pass (81 + 5 us of a 1.5 ms step, each GEMM a 10 us launch with M = 8192, N <= 384, K = 128).
//
// One 1024-thread workgroup owns 32 residues.  Everything is K = 128 on the fp16 matrix pipe with two-term splits (ipa_common.h: split_pair2
// explains the arithmetic; one power-of-two scale S for the whole weight buffer, 1 / S behind its last block): the three time features [beta, sin beta, cos beta] of in_feat are constant per sample, so their part of the
// first layer is an affine term added in the epilogue (3 FMAs per output) instead of a ragged K = 131 product.  Weights arrive pre-split in
// MFMA operand order from L2 (27 blocks of [32 outputs x 128]: 12 for the fused first layers, 4 per head for the second, one zero-padded
// block per head for the third), activations live in LDS as fp16 planes.  Wave w < 12 owns one 32-column block of layers 1 and 2; waves
// 0..2 the three output blocks.  Output: out3 [rows, 32] = eps_crd (0..2) | eps_rot (4..6) | sequence logits (8..27), consumed by the
// geometric epilogue (rows.hip: heads_epilogue_kernel).
#include "ipa_common.h"
#include "kernels.h"

namespace abopt {
namespace {
constexpr int HF = 128, HR = 32, HTH = 1024;
constexpr int HP_ROW = HF * 2 + 16, HP_PLANE = HR * HP_ROW;      // one fp16 plane of [32 rows x 128]: rows 4 banks apart (conflict-free b128 reads)
constexpr int HNT = 2;                                           // fp16 terms per value
constexpr int HBLK = 8 * HNT * 64;                               // 16-byte vectors per weight block: [k-step][term][lane]

struct HeadsSmem {
    char xp[HNT * HP_PLANE];                  // input x as two planes (h | l)
    char hp[3][HNT * HP_PLANE];               // per head: hidden activations (layer 1 output, then layer 2 output in place)
};
static_assert(HNT * HP_PLANE >= HR * 32 * 4, "the x planes later hold the rows' 32 head outputs");

// two adjacent values -> one 4-byte entry in each of the two planes
__device__ __forceinline__ void put_terms2(char* planes, int byte_off, float e0, float e1) {
    unsigned h, l;
    split_pair2(e0, e1, h, l);
    *reinterpret_cast<unsigned*>(planes + byte_off) = h;
    *reinterpret_cast<unsigned*>(planes + HP_PLANE + byte_off) = l;
}

// a0 + a1 [32 rows x 32 outputs] = S planes[32 x 128] . block^T ; the whole block's operands (8 k-steps x 2 terms = 64 registers) are requested up front
__device__ __forceinline__ void block_gemm(const char* planes, const u32x4* __restrict__ wb, int lane, f32x16& a0, f32x16& a1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    const u32x4* wl = wb + lane;
    u32x4 w[8][HNT];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int sp = 0; sp < HNT; ++sp) w[s][sp] = wl[(s * HNT + sp) * 64];
    const char* xp = planes + (lane & 31) * HP_ROW + (lane >> 5) * 16;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const u32x4 xh = *reinterpret_cast<const u32x4*>(xp + s * 32), xl = *reinterpret_cast<const u32x4*>(xp + s * 32 + HP_PLANE);
        a0 = mfma_h32(w[s][0], xl, a0); a1 = mfma_h32(w[s][1], xh, a1);
        a0 = mfma_h32(w[s][0], xh, a0);
    }
}
}  // namespace

__global__ __launch_bounds__(HTH) void heads_mlp_kernel(const float* __restrict__ xe, const float* __restrict__ beta, const float* __restrict__ wfrag,
                                                        const float* __restrict__ w1 /* [384, ld1]: columns 128..130 = the time features */, int ld1,
                                                        const float* __restrict__ b1, const float* __restrict__ b2c, const float* __restrict__ b2r,
                                                        const float* __restrict__ b2s, const float* __restrict__ b3c, const float* __restrict__ b3r,
                                                        const float* __restrict__ b3s, float* __restrict__ out3, int64_t rows, int L, HeadsEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) char hd_raw[];
    HeadsSmem& sm = *reinterpret_cast<HeadsSmem*>(hd_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * HR;
    const u32x4* wf = reinterpret_cast<const u32x4*>(wfrag);
    const float winv = wfrag[(int64_t)27 * HBLK * 4 + 1];                                            // 1 / S behind the last block
    {   // x rows -> planes: thread -> (row tid >> 5, 4 columns)
        const int r = tid >> 5, c = (tid & 31) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xe + min(row0 + r, rows - 1) * HF + c);
        put_terms2(sm.xp, r * HP_ROW + c * 2, v[0], v[1]);
        put_terms2(sm.xp, r * HP_ROW + c * 2 + 4, v[2], v[3]);
    }
    __syncthreads();
    const int mrow = lane & 31, csub = (lane >> 5) * 4;           // accumulator register 4 g + i = output column 8 g + csub + i of the block, row mrow
    f32x16 a0, a1;
    if (wave < 12) {
        // ---- layer 1 (three heads side by side): block `wave` = outputs 32 wave .. 32 wave + 31 of the 384
        block_gemm(sm.xp, wf + (int64_t)wave * HBLK, lane, a0, a1);
        const float bt = beta[min(row0 + mrow, rows - 1) / L];
        const float sb = sinf(bt), cb = cosf(bt);
        char* dst = sm.hp[wave >> 2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = wave * 32 + g * 8 + csub + i;
                const float* wt = w1 + (int64_t)o * ld1 + HF;
                v[i] = relu_nan(fmaf(a0[4 * g + i] + a1[4 * g + i], winv, b1[o] + (wt[0] * bt + (wt[1] * sb + wt[2] * cb))));
            }
            const int col = (wave & 3) * 32 + g * 8 + csub;
            put_terms2(dst, mrow * HP_ROW + col * 2, v[0], v[1]);
            put_terms2(dst, mrow * HP_ROW + col * 2 + 4, v[2], v[3]);
        }
    }
    __syncthreads();
    if (wave < 12) block_gemm(sm.hp[wave >> 2], wf + (int64_t)(12 + wave) * HBLK, lane, a0, a1);      // ---- layer 2: head wave / 4, block wave % 4
    __syncthreads();                                                                                  // every read of the layer-1 planes is done: overwrite in place
    if (wave < 12) {
        const float* b2 = (wave >> 2) == 0 ? b2c : ((wave >> 2) == 1 ? b2r : b2s);
        char* dst = sm.hp[wave >> 2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = (wave & 3) * 32 + g * 8 + csub;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu_nan(fmaf(a0[4 * g + i] + a1[4 * g + i], winv, b2[col + i]));
            put_terms2(dst, mrow * HP_ROW + col * 2, v[0], v[1]);
            put_terms2(dst, mrow * HP_ROW + col * 2 + 4, v[2], v[3]);
        }
    }
    __syncthreads();
    if (wave < 3) {
        // ---- layer 3: head `wave`, outputs zero-padded to 32
        block_gemm(sm.hp[wave], wf + (int64_t)(24 + wave) * HBLK, lane, a0, a1);
        const int nout = wave == 2 ? ABOPT_AA : 3, base = wave == 0 ? 0 : (wave == 1 ? 4 : 8);
        const float* b3 = wave == 0 ? b3c : (wave == 1 ? b3r : b3s);
        float* os = reinterpret_cast<float*>(sm.xp) + mrow * 32 + base;       // the x planes are dead since layer 1: the row's 32 head outputs for the epilogue
        if (row0 + mrow < rows) {
            float* o = out3 + (row0 + mrow) * 32 + base;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = g * 8 + csub + i;
                    if (c < nout) { const float v = fmaf(a0[4 * g + i] + a1[4 * g + i], winv, b3[c]); o[c] = v; os[c] = v; }
                }
        }
    }
    if (!ep.R_next) return;
    // ---- geometric epilogue of the same rows (dpm_full.py:95-107; rows.hip: heads_epilogue_kernel is the stand-alone form of the same function):
    // one launch and one dependent kernel boundary less per denoising step
    __syncthreads();
    if (wave == 3 && lane < HR && row0 + lane < rows) {
        const float* os = reinterpret_cast<const float*>(sm.xp) + lane * 32;
        heads_epilogue_row(row0 + lane, ep.R, ep.v_t, os, os + 4, os + 8, ep.mask_generate, ep.v_next, ep.R_next, ep.eps_pos, ep.c_den, ep.grad_mode, ep.nonfinite);
    }
}

// res_feat_mixer of EpsilonNet (dpm_full.py:56-59,89: Linear(2F,F) ReLU Linear(F,F) on [res_feat | Embedding(s_t)]) in one launch: the
// embedding half of the first layer is a 25-row table T[s] = W0[:, F:] embed[s] + b0 (built once at pack time), so both layers are K = 128
// products on the same path as the heads.  Waves 0..3 own the four 32-column blocks of each layer.
__global__ __launch_bounds__(HTH) void mixer_kernel(const float* __restrict__ res_feat, const int64_t* __restrict__ s_t, const float* __restrict__ wfrag,
                                                    const float* __restrict__ table, const float* __restrict__ b1, float* __restrict__ x_out,
                                                    int64_t rows, const float* __restrict__ v_t, float* __restrict__ R_out, unsigned* __restrict__ xt_out) {
    extern __shared__ __attribute__((aligned(16))) char hd_raw[];
    HeadsSmem& sm = *reinterpret_cast<HeadsSmem*>(hd_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * HR;
    const u32x4* wf = reinterpret_cast<const u32x4*>(wfrag);
    const float winv = wfrag[(int64_t)8 * HBLK * 4 + 1];
    if (R_out && wave == 4 && lane < HR && row0 + lane < rows) {
        // dpm_full.py:86  R = exp(v_t) of this workgroup's rows, on a wave that has no layer to compute (saves the so3_exp launch of a step)
        const int64_t i = row0 + lane;
        const Mat3 m = so3_exp(v_t[i * 3], v_t[i * 3 + 1], v_t[i * 3 + 2]);
#pragma unroll
        for (int k = 0; k < 9; ++k) R_out[i * 9 + k] = m.m[k];
    }
    {
        const int r = tid >> 5, c = (tid & 31) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(res_feat + min(row0 + r, rows - 1) * HF + c);
        put_terms2(sm.xp, r * HP_ROW + c * 2, v[0], v[1]);
        put_terms2(sm.xp, r * HP_ROW + c * 2 + 4, v[2], v[3]);
    }
    __syncthreads();
    const int mrow = lane & 31, csub = (lane >> 5) * 4;
    f32x16 a0, a1;
    if (wave < 4) {
        block_gemm(sm.xp, wf + (int64_t)wave * HBLK, lane, a0, a1);
        const int64_t s = s_t[min(row0 + mrow, rows - 1)];
        const bool ok = s >= 0 && s < 25;                                        // nn.Embedding(25) raises on anything else: poison instead of reading out of bounds
        const float* tr = table + (ok ? s : 0) * HF;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = wave * 32 + g * 8 + csub;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? relu_nan(fmaf(a0[4 * g + i] + a1[4 * g + i], winv, tr[col + i])) : __builtin_nanf("");
            put_terms2(sm.hp[0], mrow * HP_ROW + col * 2, v[0], v[1]);
            put_terms2(sm.hp[0], mrow * HP_ROW + col * 2 + 4, v[2], v[3]);
        }
    }
    __syncthreads();
    if (wave < 4) {
        block_gemm(sm.hp[0], wf + (int64_t)(4 + wave) * HBLK, lane, a0, a1);
        if (row0 + mrow < rows) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wave * 32 + g * 8 + csub;
                const f32x4 o = (f32x4){fmaf(a0[4 * g] + a1[4 * g], winv, b1[col]), fmaf(a0[4 * g + 1] + a1[4 * g + 1], winv, b1[col + 1]),
                                        fmaf(a0[4 * g + 2] + a1[4 * g + 2], winv, b1[col + 2]), fmaf(a0[4 * g + 3] + a1[4 * g + 3], winv, b1[col + 3])};
                *reinterpret_cast<f32x4*>(x_out + (row0 + mrow) * HF + col) = o;
                if (xt_out) {                                                  // the first block's node_frags reads x as terms (tail_common.h: tail_p2_run writes the same layout)
                    unsigned h0_, l0_, h1_, l1_;
                    split_pair2(o[0], o[1], h0_, l0_); split_pair2(o[2], o[3], h1_, l1_);
                    unsigned* d = xt_out + (row0 + mrow) * HF + col / 2;
                    *reinterpret_cast<uint2*>(d) = make_uint2(h0_, h1_); *reinterpret_cast<uint2*>(d + 64) = make_uint2(l0_, l1_);
                }
            }
        }
    }
}

int launch_mixer(const float* res_feat, const int64_t* s_t, const float* wfrag, const float* table, const float* b1, float* x_out, int64_t rows,
                 hipStream_t st, const float* v_t, float* R_out, float* xt_out) {
    if (rows == 0) return ABOPT_OK;
    static LdsConfig lds_cfg;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(mixer_kernel), sizeof(HeadsSmem), lds_cfg)) return rc;
    hipLaunchKernelGGL(mixer_kernel, dim3((unsigned)((rows + HR - 1) / HR)), dim3(HTH), sizeof(HeadsSmem), st, res_feat, s_t, wfrag, table, b1, x_out, rows, v_t, R_out, reinterpret_cast<unsigned*>(xt_out));
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

size_t heads_wfrag_floats() { return (size_t)27 * HBLK * 4 + 4; }     // + {S, 1 / S, 0, 0}
size_t mixer_wfrag_floats() { return (size_t)8 * HBLK * 4 + 4; }

int launch_heads_mlp(const float* xe, const float* beta, const float* wfrag, const float* w1, int ld1, const float* b1, const float* b2c,
                     const float* b2r, const float* b2s, const float* b3c, const float* b3r, const float* b3s, float* out3, int64_t rows, int L,
                     hipStream_t st, const HeadsEpilogue* ep) {
    if (rows == 0) return ABOPT_OK;
    static LdsConfig lds_cfg;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(heads_mlp_kernel), sizeof(HeadsSmem), lds_cfg)) return rc;
    hipLaunchKernelGGL(heads_mlp_kernel, dim3((unsigned)((rows + HR - 1) / HR)), dim3(HTH), sizeof(HeadsSmem), st, xe, beta, wfrag, w1, ld1, b1,
                       b2c, b2r, b2s, b3c, b3r, b3s, out3, rows, L, ep ? *ep : HeadsEpilogue{});
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

// encode(): residue and pair embeddings for CDNA4 (reference AbDock/src/models/diffab.py:39-83).
//
//   ResidueEmbedding.forward   AbDock/src/modules/encoders/residue.py:26-92   (AbDesign adds a hotspot embedding, A/.../residue.py:19-21)
//   PairEmbedding.forward      AbDock/src/modules/encoders/pair.py:37-101
//   construct_3d_basis         AbDock/src/modules/common/geometry.py:47-69
//   get_backbone_dihedral_angles / pairwise_dihedrals   geometry.py:301-401 ; AngularEncoding layers.py:59-80
//
// The pair embedding is 5.3 GFLOP per 256-residue sample of small dense layers over L^2 pairs.  One wave owns a strip of
// 64 pairs (i, j0..j0+63) and keeps it in registers through all five layers: with the weight matrix as the MFMA A operand
// the accumulator of layer l (lane = pair, 4 consecutive output features per register quad) IS the B operand of layer l+1
// under the K permutation k = 16 blk + 4 (lane >> 4) + r, so there is no LDS round trip and no cross-lane traffic between
// layers.  Weights are re-laid out once per call into that fragment order ("swizzled": 1 KB contiguous per 16x16 block)
// and streamed from L2 with fully coalesced loads.  The 225 Gaussian atom-pair features are produced directly in operand
// layout (lane = (pair, 4 atoms of residue j), loop over the atoms of residue i).
#include "ipa_common.h"
#include "kernels.h"

namespace abopt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4e(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int EF = 128;       // res_feat_dim
constexpr int EC = 64;        // pair_feat_dim
constexpr int AAT = 22;       // max_aa_types
constexpr int NREL = 65;      // 2 * max_relpos + 1

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float norm3(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 xyz(f32x4 a) { return v3(a[0], a[1], a[2]); }

// The 225 Gaussian atom-pair features of a residue pair (pair.py:62-73: d = |x_a - x_b| / 10, g = exp(-softplus(coef) d^2)) are the
// VALU hot spot of the pair embedding (PMC, round 4: 9 VALU instructions per MFMA, the two pipes together account for the run time).  The
// squared scaled distance is formed without the square root and the IEEE division of the literal formula ((sqrt(s) / 10)^2 = s / 100 up
// to 2 ulp), the exponential is one v_exp_f32 on the base-2 argument instead of libm's range-reduced expf (~1 ulp against ~0.5): the
// forward and the backward's recomputation share these two functions, so T = dg / d softplus(coef) stays bit-identical between them.
__device__ __forceinline__ float gauss_d2(float dx, float dy, float dz) { return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) * 0.01f; }
__device__ __forceinline__ float gauss_exp(float c, float dd) { return __builtin_amdgcn_exp2f(-1.4426950408889634f * c * dd); }

// geometry.py:336-362: signed dihedral of p0-p1-p2-p3, NaN -> 0 -- the reference's arithmetic op for op (IEEE divisions by the norms, acos, sign).
// Round 6: hipcc turned parts of this into v_pk_*_f32 instructions, and those return wrong lanes 48-63 while another wave of the SIMD runs 16x16x32 f16 / bf16 MFMAs (the
// pair embedding beside this library's denoiser in a second process: 1-3 % of its small launches wrong) -- the library is built without packed-FP32 instructions since
// (csrc/Makefile: NOPK; tests: test_pair_embedding_repeats_beside_a_second_process, test_no_packed_fp32_instructions_in_the_library).
__device__ __forceinline__ float dihedral_from_four_points(V3 p0, V3 p1, V3 p2, V3 p3) {
    const V3 v0 = p2 - p1, v1 = p0 - p1, v2 = p3 - p2;
    const V3 u1 = cross3(v0, v1), u2 = cross3(v0, v2);
    const float l1 = norm3(u1), l2 = norm3(u2);
    const V3 n1 = v3(u1.x / l1, u1.y / l1, u1.z / l1), n2 = v3(u2.x / l2, u2.y / l2, u2.z / l2);
    const float sd = dot3(cross3(v1, v2), v0);
    const float sgn = (sd > 0.f) ? 1.f : ((sd < 0.f) ? -1.f : 0.f);
    float c = dot3(n1, n2);
    c = fminf(fmaxf(c, -0.999999f), 0.999999f);
    const float d = sgn * acosf(c);
    return (d != d) ? 0.f : d;
}

// ------------------------------------------------------------------------------------------------ per-residue pack
// atoms4[row][16] = (x, y, z, mask) of the first A atoms (others 0); meta; frames R [rows,3,3] and CA positions p [rows,3].
__global__ void residue_pack_kernel(const int64_t* __restrict__ aa, const int64_t* __restrict__ res_nb, const int64_t* __restrict__ chain_nb,
                                    const float* __restrict__ pos, const uint8_t* __restrict__ matom,
                                    const uint8_t* __restrict__ structure_mask, const uint8_t* __restrict__ sequence_mask,
                                    int atoms_in, int A, int64_t rows,
                                    f32x4* __restrict__ atoms4, int* __restrict__ aa_eff, int* __restrict__ resnb32, int* __restrict__ chain32,
                                    uint8_t* __restrict__ flags, float* __restrict__ R, float* __restrict__ p) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 4);
    const int a = threadIdx.x & 15;
    if (row >= rows) return;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a < A) {
        const float* q = pos + (row * atoms_in + a) * 3;
        v = (f32x4){q[0], q[1], q[2], matom[row * atoms_in + a] ? 1.f : 0.f};
    }
    atoms4[row * 16 + a] = v;
    if (a == 0) {
        int e = (int)aa[row];
        if (sequence_mask && !sequence_mask[row]) e = 20;                             // AA.UNK, residue.py:37-40
        aa_eff[row] = e;
        resnb32[row] = (int)res_nb[row];
        chain32[row] = (int)chain_nb[row];
        flags[row] = (uint8_t)((matom[row * atoms_in + 1] ? 1 : 0) | ((!structure_mask || structure_mask[row]) ? 2 : 0));
        if (R) {
            const float* q = pos + row * atoms_in * 3;
            const V3 n = v3(q[0], q[1], q[2]), ca = v3(q[3], q[4], q[5]), c = v3(q[6], q[7], q[8]);
            const V3 v1 = c - ca, v2 = n - ca;                                         // construct_3d_basis(CA, C, N)
            const float l1 = norm3(v1) + 1e-6f;                                          // normalize_vector eps, geometry.py:32-33
            const V3 u1 = v3(v1.x / l1, v1.y / l1, v1.z / l1);
            const float pr = dot3(u1, v2);
            const V3 w2 = v2 - u1 * pr;
            const float l2 = norm3(w2) + 1e-6f;
            const V3 u2 = v3(w2.x / l2, w2.y / l2, w2.z / l2);
            const V3 u3 = cross3(u1, u2);
            float* Rr = R + row * 9;                                                    // columns e1, e2, e3
            Rr[0] = u1.x; Rr[1] = u2.x; Rr[2] = u3.x;
            Rr[3] = u1.y; Rr[4] = u2.y; Rr[5] = u3.y;
            Rr[6] = u1.z; Rr[7] = u2.z; Rr[8] = u3.z;
            p[row * 3 + 0] = ca.x; p[row * 3 + 1] = ca.y; p[row * 3 + 2] = ca.z;
        }
    }
}

// ------------------------------------------------------------------------------------------------ residue features
// feat[row] = [aatype_embed(aa) | per-aa-slotted local coords (22*A*3) | dihedral encoding (39) | type_embed | hotspot_embed], zero padded to ld.
__global__ __launch_bounds__(256) void residue_feat_kernel(const f32x4* __restrict__ atoms4, const int* __restrict__ aa_eff, const int* __restrict__ resnb,
                                                           const int* __restrict__ chain, const uint8_t* __restrict__ flags, const float* __restrict__ R,
                                                           const int64_t* __restrict__ fragment_type, const int64_t* __restrict__ hotspot,
                                                           const float* __restrict__ aatype_embed, const float* __restrict__ type_embed,
                                                           const float* __restrict__ hotspot_embed, const float* __restrict__ freq,
                                                           int A, int L, int has_struct_mask, float* __restrict__ feat, int ld) {
    __shared__ float crd[16][3];
    __shared__ float dih[3][13];
    const int64_t row = blockIdx.x;
    const int l = (int)(row % L), tid = threadIdx.x;
    float* fr = feat + row * ld;
    const int aa = aa_eff[row];
    const int ncrd = AAT * A * 3;
    const uint8_t fl = flags[row];
    if (tid < A) {                                                                     // R^T (x - CA), masked atoms -> 0 (residue.py:47-53)
        const f32x4 at = atoms4[row * 16 + tid], ca = atoms4[row * 16 + 1];
        const float* Rr = R + row * 9;
        const float dx = at[0] - ca[0], dy = at[1] - ca[1], dz = at[2] - ca[2];
        const bool ok = at[3] != 0.f;
        crd[tid][0] = ok ? (Rr[0] * dx + Rr[3] * dy + Rr[6] * dz) : 0.f;
        crd[tid][1] = ok ? (Rr[1] * dx + Rr[4] * dy + Rr[7] * dz) : 0.f;
        crd[tid][2] = ok ? (Rr[2] * dx + Rr[5] * dy + Rr[8] * dz) : 0.f;
    }
    if (tid >= 64 && tid < 67) {                                                       // omega, phi, psi (geometry.py:364-401)
        const int which = tid - 64;
        const bool has_prev = l > 0, has_next = l < L - 1;
        auto consec = [&](int64_t r0) {                                                // residues r0, r0+1 consecutive on one chain, r0 present
            return (abs(resnb[r0 + 1] - resnb[r0]) == 1) && (chain[r0 + 1] == chain[r0]) && (flags[r0] & 1);
        };
        float ang = 0.f;
        bool ok;
        if (which < 2) {
            ok = has_prev && consec(row - 1);                                          // ~N-terminus
            if (has_prev) {
                const V3 cam = xyz(atoms4[(row - 1) * 16 + 1]), cm = xyz(atoms4[(row - 1) * 16 + 2]);
                const V3 n = xyz(atoms4[row * 16 + 0]), ca = xyz(atoms4[row * 16 + 1]), c = xyz(atoms4[row * 16 + 2]);
                ang = (which == 0) ? dihedral_from_four_points(cam, cm, n, ca) : dihedral_from_four_points(cm, n, ca, c);
            }
        } else {
            ok = has_next && consec(row);                                              // ~C-terminus
            if (has_next) {
                const V3 n = xyz(atoms4[row * 16 + 0]), ca = xyz(atoms4[row * 16 + 1]), c = xyz(atoms4[row * 16 + 2]);
                ang = dihedral_from_four_points(n, ca, c, xyz(atoms4[(row + 1) * 16 + 0]));
            }
        }
        float keep = ok ? 1.f : 0.f;
        ang *= keep;                                                                   // dihedral * mask, then encoding * mask (residue.py:60-62)
        if (has_struct_mask) {                                                         // residue.py:63-68: own, previous and next residue (rolled, wraps)
            const int64_t base = row - l;
            const bool m0 = flags[row] & 2, mp = flags[base + (l + L - 1) % L] & 2, mn = flags[base + (l + 1) % L] & 2;
            if (!(m0 && mp && mn)) keep = 0.f;
        }
        dih[which][0] = ang * keep;
#pragma unroll
        for (int k = 0; k < 6; ++k) { dih[which][1 + k] = sinf(ang * freq[k]) * keep; dih[which][7 + k] = cosf(ang * freq[k]) * keep; }
    }
    __syncthreads();
    const bool smask = (fl & 2) != 0;
    for (int e = tid; e < ld; e += blockDim.x) {
        float v = 0.f;
        if (e < EF) v = aatype_embed[aa * EF + e];
        else if (e < EF + ncrd) {
            const int k = e - EF, slot = k / (A * 3), rem = k % (A * 3);
            v = (slot == aa && smask) ? crd[rem / 3][rem % 3] : 0.f;
        } else if (e < EF + ncrd + 39) { const int k = e - EF - ncrd; v = dih[k / 13][k % 13]; }
        else if (e < EF + ncrd + 39 + EF) v = type_embed[fragment_type[row] * EF + (e - EF - ncrd - 39)];
        else if (hotspot_embed && e < EF + ncrd + 39 + 2 * EF) v = hotspot_embed[(hotspot ? hotspot[row] : 0) * EF + (e - EF - ncrd - 39 - EF)];
        fr[e] = v;
    }
}

__global__ void pad_rows_kernel(const float* __restrict__ src, int ncols, float* __restrict__ dst, int ld, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * ld) return;
    const int64_t r = idx / ld;
    const int c = (int)(idx % ld);
    dst[idx] = (c < ncols) ? src[r * ncols + c] : 0.f;
}

__global__ void mask_rows_kernel(float* __restrict__ x, const uint8_t* __restrict__ flags, int64_t rows) {   // residue.py:91
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * EF) return;
    if (!(flags[idx / EF] & 1)) x[idx] = 0.f;
}

// ------------------------------------------------------------------------------------------------ pair embedding: weight prep
// T_aap[e][n] = sum_k aa_pair_embed[e][k] wo0[n][k];  T_rel[e][n] = sum_k relpos_embed[e][k] wo0[n][C + k]  (first out_mlp layer is
// linear in the two looked-up embeddings, pair.py:52-60,98);  SP[e][a][16] = softplus(aapair_to_distcoef[e][a*A + b]) (pair.py:66)
__global__ void pair_tables_kernel(const float* __restrict__ e_aap, const float* __restrict__ e_rel, const float* __restrict__ coef,
                                   const float* __restrict__ wo0, int ldwo0, int A,
                                   float* __restrict__ t_aap, float* __restrict__ t_rel, float* __restrict__ sp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_aap = AAT * AAT * EC, n_rel = NREL * EC, n_sp = AAT * AAT * A * 16;
    if (idx < n_aap) {
        const int e = idx / EC, n = idx % EC;
        float s = 0.f;
        for (int k = 0; k < EC; ++k) s += e_aap[e * EC + k] * wo0[n * ldwo0 + k];
        t_aap[idx] = s;
    } else if (idx < n_aap + n_rel) {
        const int q = idx - n_aap, e = q / EC, n = q % EC;
        float s = 0.f;
        for (int k = 0; k < EC; ++k) s += e_rel[e * EC + k] * wo0[n * ldwo0 + EC + k];
        t_rel[q] = s;
    } else if (idx < n_aap + n_rel + n_sp) {
        const int q = idx - n_aap - n_rel, b = q & 15, a = (q >> 4) % A, e = (q >> 4) / A;
        float v = 0.f;
        if (b < A) { const float x = coef[e * A * A + a * A + b]; v = (x > 20.f) ? x : log1pf(expf(x)); }
        sp[q] = v;
    }
}

// out[(blk*4 + nt)*64 + lane] = (W[nt*16 + fm][col(blk, 4 kq + q)])_{q<4};  col(blk, c) = col0 + blk*stride + c, valid while
// c < width and blk*stride + c < kreal (else 0): the MFMA A-operand fragment order of a [64, K] weight matrix.
// A-operand fragments of W^T restricted to columns [col0, col0 + 64) of W [64, ldw]: rows of the operand = input features
// (what the backward multiplies by), K = the 64 output features.  out[(blk * 4 + nt) * 64 + lane][q] = W[16 blk + 4 kq + q][col0 + 16 nt + fm]
__global__ void swizzle_transposed_kernel(const float* __restrict__ W, int ldw, int col0, f32x4* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * 4 * 64) return;
    const int lane = idx & 63, nt = (idx >> 6) & 3, blk = idx >> 8, fm = lane & 15, kq = lane >> 4;
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = W[(blk * 16 + kq * 4 + q) * ldw + col0 + nt * 16 + fm];
    out[idx] = v;
}
// same for distance_embed.0 [64, A*A]: operand rows = (a, b padded to 16): out[(blk * A + a) * 64 + lane][q] = W[16 blk + 4 kq + q][a * A + fm] (0 for fm >= A)
__global__ void swizzle_wd0_transposed_kernel(const float* __restrict__ W, int A, f32x4* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * A * 64) return;
    const int lane = idx & 63, a = (idx >> 6) % A, blk = (idx >> 6) / A, fm = lane & 15, kq = lane >> 4;
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (fm < A) ? W[(blk * 16 + kq * 4 + q) * A * A + a * A + fm] : 0.f;
    out[idx] = v;
}

__global__ void swizzle_weights_kernel(const float* __restrict__ W, int ldw, int col0, int stride, int width, int kreal, int nblk, f32x4* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nblk * 4 * 64) return;
    const int lane = idx & 63, nt = (idx >> 6) & 3, blk = idx >> 8, fm = lane & 15, kq = lane >> 4;
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = kq * 4 + q;
        v[q] = (c < width && blk * stride + c < kreal) ? W[(nt * 16 + fm) * ldw + col0 + blk * stride + c] : 0.f;
    }
    out[idx] = v;
}

// ------------------------------------------------------------------------------------------------ pair embedding: main kernel
struct PairArgs {
    const f32x4* atoms4; const int* aa_eff; const int* res_nb; const int* chain_nb; const uint8_t* flags;
    const float* t_aap; const float* t_rel; const float* sp; const float* freq;
    const f32x4* wd0; const float* bd0; const f32x4* wd1; const float* bd1;
    const f32x4* wo0; const float* bo0; const f32x4* wo1; const float* bo1; const f32x4* wo2; const float* bo2;
    float* out; int N, L, A, has_struct;
    float* gsave; float* tsave;   // training: Gaussian features and d/d softplus(coef), [pair][A][16] (b padded to 16), NULL for inference
    float* acts;          // training: per pair [relu(D0) 64 | f_dist 64 | f_dih 32 | relu(O0) 64 | relu(O1) 64] (PAIR_ACT floats), NULL for inference
};
constexpr int PAIR_ACT = 288;

constexpr int PMT = 4;        // 16-pair tiles per wave

// one 64 -> 64 layer for the wave's PMT tiles: DST[mt][nt] += W . SRC[mt]; block blk of K is the source's accumulator quad nt = blk
#define PAIR_DENSE(DST, SRC, WPTR)                                                                                        \
    _Pragma("unroll") for (int blk = 0; blk < 4; ++blk) {                                                                 \
        f32x4 w_[4];                                                                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) w_[nt] = (WPTR)[(blk * 4 + nt) * 64 + lane];                     \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                     \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                              \
                _Pragma("unroll") for (int mt = 0; mt < PMT; ++mt)                                                        \
                    DST[mt][nt] = mfma4e(w_[nt][q], SRC[mt][blk][q], DST[mt][nt]);                                        \
    }
// training: dump a 64-wide activation tile (lane = pair fm of tile mt, features 16 nt + 4 kq ..) at float offset OFF of the pair's record
#define PAIR_SAVE(TILE, OFF)                                                                                              \
    if (a.acts) {                                                                                                         \
        _Pragma("unroll") for (int mt = 0; mt < PMT; ++mt) {                                                              \
            const int j_ = j0 + mt * 16 + fm;                                                                             \
            if (j_ < L) {                                                                                                 \
                float* d_ = a.acts + prow * PAIR_ACT + (unsigned)(j_ * PAIR_ACT + (OFF) + kq * 4);                        \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(d_ + nt * 16) = TILE[mt][nt];  \
            }                                                                                                             \
        }                                                                                                                 \
    }
#define SEL4(ARR, I) ((I) == 0 ? ARR[0] : (I) == 1 ? ARR[1] : (I) == 2 ? ARR[2] : ARR[3])

#ifndef PE_LB
#define PE_LB 2
#endif
__global__ __launch_bounds__(256, PE_LB) void pair_embed_kernel(PairArgs a) {
    const int lane = threadIdx.x & 63, fm = lane & 15, kq = lane >> 4;
    const int L = a.L, A = a.A;
    const int jblocks = (L + 16 * PMT - 1) / (16 * PMT);
    const int64_t unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (unit >= (int64_t)a.N * L * jblocks) return;
    const int64_t row_i = unit / jblocks;
    const int j0 = (int)(unit % jblocks) * 16 * PMT;
    const int64_t base = (row_i / L) * L;
    const int aa_i = a.aa_eff[row_i], res_i = a.res_nb[row_i], chain_i = a.chain_nb[row_i];
    const uint8_t fl_i = a.flags[row_i];

    // Addresses = a wave-uniform 64-bit base (the sample's first row / this query row: SGPRs) + an unsigned 32-bit lane offset.  (Per-lane 64-bit
    // row numbers and sign-extended offsets kept live across the five layers were what the register allocator spilled: 104 bytes of scratch per lane
    // until round 5.)
    const int* __restrict__ aa_b = a.aa_eff + base;
    const int* __restrict__ res_b = a.res_nb + base;
    const int* __restrict__ chain_b = a.chain_nb + base;
    const uint8_t* __restrict__ flags_b = a.flags + base;
    const f32x4* __restrict__ atoms_b = a.atoms4 + base * 16;
    const int64_t prow = row_i * L;                                    // first pair record of this query row
    unsigned jl[PMT];
    int aap[PMT];
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt) {
        jl[mt] = (unsigned)min(j0 + mt * 16 + fm, L - 1);
        aap[mt] = aa_i * AAT + aa_b[jl[mt]];
    }

    // ---- distance_embed.0: 225 Gaussian atom-pair features -> 64, K block = atom a of residue i, lane group kq = atoms 4kq..4kq+3 of j
    f32x4 h0[PMT][4], h1[PMT][4];
    {
        f32x4 pj[PMT][4];
#pragma unroll
        for (int mt = 0; mt < PMT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) pj[mt][q] = atoms_b[jl[mt] * 16u + (unsigned)(kq * 4 + q)];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bd0 + nt * 16 + kq * 4);
#pragma unroll
            for (int mt = 0; mt < PMT; ++mt) h0[mt][nt] = bv;
        }
#pragma unroll 1
        for (int at = 0; at < A; ++at) {
            const f32x4 pi = a.atoms4[row_i * 16 + at];
            // atom `at` of residue i absent (mask_heavyatom, pair.py:69-73): its 16 features are exact zeros for every pair of the strip and add
            // nothing to the layer -- the whole K block is skipped (wave-uniform: a wave owns one i).  Backbone-only inputs skip 10 of 15 blocks.
            if (__builtin_amdgcn_readfirstlane(__float_as_uint(pi[3])) == 0u) {
                if (a.gsave) {
#pragma unroll
                    for (int mt = 0; mt < PMT; ++mt) {
                        const int j_ = j0 + mt * 16 + fm;
                        if (j_ < L) {
                            const unsigned o_ = (unsigned)((j_ * A + at) * 16 + kq * 4);
                            *reinterpret_cast<f32x4*>(a.gsave + prow * A * 16 + o_) = (f32x4){0.f, 0.f, 0.f, 0.f};
                            if (a.tsave) *reinterpret_cast<f32x4*>(a.tsave + prow * A * 16 + o_) = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                    }
                }
                continue;
            }
            f32x4 w_[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) w_[nt] = a.wd0[(at * 4 + nt) * 64 + lane];
            f32x4 g[PMT];
#pragma unroll
            for (int mt = 0; mt < PMT; ++mt) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(a.sp + (unsigned)((aap[mt] * A + at) * 16 + kq * 4));
                f32x4 tq;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dx = pi[0] - pj[mt][q][0], dy = pi[1] - pj[mt][q][1], dz = pi[2] - pj[mt][q][2];
                    const float dd = gauss_d2(dx, dy, dz);                                      // (|x_i - x_j| / 10)^2, pair.py:64
                    const float gv = gauss_exp(c4[q], dd);                                      // exp(-c d^2), pair.py:67
                    g[mt][q] = (pi[3] != 0.f && pj[mt][q][3] != 0.f) ? gv : 0.f;               // pair.py:69-73
                    tq[q] = -dd * g[mt][q];
                }
                if (a.gsave) {
                    const int j_ = j0 + mt * 16 + fm;
                    if (j_ < L) {
                        const unsigned o_ = (unsigned)((j_ * A + at) * 16 + kq * 4);
                        *reinterpret_cast<f32x4*>(a.gsave + prow * A * 16 + o_) = g[mt];
                        if (a.tsave) *reinterpret_cast<f32x4*>(a.tsave + prow * A * 16 + o_) = tq;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < PMT; ++mt) h0[mt][nt] = mfma4e(w_[nt][q], g[mt][q], h0[mt][nt]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bd1 + nt * 16 + kq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) h0[mt][nt][r] = relu_nan(h0[mt][nt][r]);
            h1[mt][nt] = bv;
        }
    PAIR_SAVE(h0, 0)
    // ---- distance_embed.2 + ReLU, structure mask (pair.py:74-76)
    PAIR_DENSE(h1, h0, a.wd1)
    float ps[PMT], same[PMT], mp[PMT];
    int rel[PMT];
    // the key rows are re-derived from the lane index here and in the dihedral block (two VALU operations) instead of living in registers across
    // distance_embed: the empty asm statement keeps the compiler from merging them with the copies above
    int fm_late = fm;
    asm volatile("" : "+v"(fm_late));
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt) {
        const unsigned jl_ = (unsigned)min(j0 + mt * 16 + fm_late, L - 1);
        const uint8_t fl_j = flags_b[jl_];
        ps[mt] = (!a.has_struct || ((fl_i & 2) && (fl_j & 2))) ? 1.f : 0.f;
        mp[mt] = ((fl_i & 1) && (fl_j & 1)) ? 1.f : 0.f;
        same[mt] = (chain_b[jl_] == chain_i) ? 1.f : 0.f;
        rel[mt] = min(max(res_i - res_b[jl_], -32), 32) + 32;                           // pair.py:55-60
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[mt][nt][r] = relu_nan(h1[mt][nt][r]) * ps[mt];
    }
    PAIR_SAVE(h1, 64)
    // ---- out_mlp.0: folded embedding tables + f_dist block + dihedral block
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nt * 16 + kq * 4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bo0 + col);
            const f32x4 ta = *reinterpret_cast<const f32x4*>(a.t_aap + (unsigned)(aap[mt] * EC + col));
            const f32x4 tr = *reinterpret_cast<const f32x4*>(a.t_rel + (unsigned)(rel[mt] * EC + col));
            h0[mt][nt] = (bv + ta) + tr * same[mt];
        }
    PAIR_DENSE(h0, h1, a.wo0)
    {   // inter-residue dihedrals (pair.py:80-92): phi-like (C_i, N_j, CA_j, C_j), psi-like (N_i, CA_i, C_i, N_j); AngularEncoding -> 26 (+6 pad)
        const V3 ni = xyz(a.atoms4[row_i * 16 + 0]), cai = xyz(a.atoms4[row_i * 16 + 1]), ci = xyz(a.atoms4[row_i * 16 + 2]);
        f32x4 dh[PMT][2];
#pragma unroll 1
        for (int mt = 0; mt < PMT; ++mt) {
            const unsigned jr = (unsigned)min(j0 + mt * 16 + fm_late, L - 1);
            const float psm = SEL4(ps, mt);
            const V3 nj = xyz(atoms_b[jr * 16u + 0u]), caj = xyz(atoms_b[jr * 16u + 1u]), cj = xyz(atoms_b[jr * 16u + 2u]);
#if defined(PE_ABL) && (PE_ABL & 1)      // developer build: no dihedral geometry
            const float x0 = nj.x, x1 = caj.y;
#else
            const float x0 = dihedral_from_four_points(ci, nj, caj, cj);
            const float x1 = dihedral_from_four_points(ni, cai, ci, nj);
#endif
            f32x4 d0, d1;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = blk * 16 + kq * 4 + q;
                    const int ang = k >= 13, m = k - 13 * ang;
                    const float x = ang ? x1 : x0;
                    float v = 0.f;
#if defined(PE_ABL) && (PE_ABL & 2)      // developer build: no sin / cos
                    if (k < 26) v = (m == 0) ? x : x * a.freq[m <= 6 ? m - 1 : m - 7];
#else
                    if (k < 26) v = (m == 0) ? x : ((m <= 6) ? sinf(x * a.freq[m - 1]) : cosf(x * a.freq[m - 7]));
#endif
                    if (blk == 0) d0[q] = v * psm; else d1[q] = v * psm;
                }
            // register-array write with a loop-variant index would go to scratch: select per tile
            if (mt == 0) { dh[0][0] = d0; dh[0][1] = d1; } else if (mt == 1) { dh[1][0] = d0; dh[1][1] = d1; }
            else if (mt == 2) { dh[2][0] = d0; dh[2][1] = d1; } else { dh[3][0] = d0; dh[3][1] = d1; }
        }
        if (a.acts) {
#pragma unroll
            for (int mt = 0; mt < PMT; ++mt) {
                const int j_ = j0 + mt * 16 + fm;
                if (j_ < L) {
                    float* d_ = a.acts + prow * PAIR_ACT + (unsigned)(j_ * PAIR_ACT + 128 + kq * 4);
                    *reinterpret_cast<f32x4*>(d_) = dh[mt][0]; *reinterpret_cast<f32x4*>(d_ + 16) = dh[mt][1];
                }
            }
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x4 w_[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) w_[nt] = a.wo0[((4 + blk) * 4 + nt) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < PMT; ++mt) h0[mt][nt] = mfma4e(w_[nt][q], dh[mt][blk][q], h0[mt][nt]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bo1 + nt * 16 + kq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) h0[mt][nt][r] = relu_nan(h0[mt][nt][r]);
            h1[mt][nt] = bv;
        }
    PAIR_SAVE(h0, 160)
    PAIR_DENSE(h1, h0, a.wo1)
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bo2 + nt * 16 + kq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[mt][nt][r] = relu_nan(h1[mt][nt][r]);
            h0[mt][nt] = bv;
        }
    PAIR_SAVE(h1, 224)
    PAIR_DENSE(h0, h1, a.wo2)
    // ---- pair mask, store (pair.py:100): a lane owns features 16 nt + 4 kq .. +3 of pair (i, j0 + 16 mt + fm)
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt) {
        const int j = j0 + mt * 16 + fm;
        if (j >= L) continue;
        float* o = a.out + prow * EC + (unsigned)(j * EC + kq * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(o + nt * 16) = h0[mt][nt] * mp[mt];
    }
}

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct PackBufs { f32x4* atoms4; int* aa_eff; int* resnb; int* chain; uint8_t* flags; char* end; };
static PackBufs carve_pack(char* p, int64_t rows) {
    PackBufs b;
    b.atoms4 = (f32x4*)p; p += al256((size_t)rows * 16 * sizeof(f32x4));
    b.aa_eff = (int*)p; p += al256((size_t)rows * 4);
    b.resnb = (int*)p; p += al256((size_t)rows * 4);
    b.chain = (int*)p; p += al256((size_t)rows * 4);
    b.flags = (uint8_t*)p; p += al256((size_t)rows);
    b.end = p;
    return b;
}
static size_t pack_bytes(int64_t rows) { return al256(rows * 16 * sizeof(f32x4)) + 3 * al256(rows * 4) + al256(rows); }

// ------------------------------------------------------------------------------------------------ pair embedding: backward chain
// d(out) -> gradients w.r.t. the pre-activation of every layer, for the 64-pair strip of a wave, in registers exactly like the
// forward (the accumulator of W^T . dY is the B operand of the next W^T): per pair it writes
//   dys[0:64] = d out x pair mask (= dY of out_mlp.4)   dys[64:128] = dY of out_mlp.2   dys[128:192] = dY of out_mlp.0
//   dys[192:256] = dY of distance_embed.2                dys[256:320] = dY of distance_embed.0
//   ds [A][16]  = d loss / d softplus(coef) per atom pair (= (Wd0^T dY_d0) x T)
// The weight gradients are then tall GEMMs of these against the saved activations (host side).
struct PairBwdArgs {
    const float* dout; const float* acts; const float* tsave; const uint8_t* flags;
    const f32x4* wo2t; const f32x4* wo1t; const f32x4* wo0dt; const f32x4* wd1t; const f32x4* wd0t;
    float* dys; float* ds; int N, L, A, has_struct;
    float* dsum;                                                  // [units][PAIR_DY] per-wave column sums of dys (the five bias gradients), or NULL
    const f32x4* atoms4; const int* aa_eff; const float* sp;      // tsave == NULL: T = -d^2 g is recomputed from the atoms, as the forward does
};
constexpr int PAIR_DY = 320;

#define PAIR_LOAD_ACT(DST, OFF)                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < PMT; ++mt) {                                                                  \
        const float* s_ = b.acts + ((row_i * L) + jc[mt]) * PAIR_ACT + (OFF) + kq * 4;                                    \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) DST[mt][nt] = *reinterpret_cast<const f32x4*>(s_ + nt * 16);     \
    }
#define PAIR_STORE_DY(SRC, OFF)                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < PMT; ++mt) {                                                                  \
        if (j0 + mt * 16 + fm < L) {                                                                                      \
            float* d_ = b.dys + ((row_i * L) + jc[mt]) * PAIR_DY + (OFF) + kq * 4;                                        \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(d_ + nt * 16) = SRC[mt][nt];       \
        }                                                                                                                 \
    }                                                                                                                     \
    if (b.dsum) {                                                       /* the strip's column sums: tiles, then the 16 pairs of a tile (one DPP row) */ \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                                \
            f32x4 t_ = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                       \
            _Pragma("unroll") for (int mt = 0; mt < PMT; ++mt) if (j0 + mt * 16 + fm < L) t_ += SRC[mt][nt];              \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                               \
                float v_ = t_[r];                                                                                         \
                v_ += ABOPT_DPP_ROR(v_, 8); v_ += ABOPT_DPP_ROR(v_, 4); v_ += ABOPT_DPP_ROR(v_, 2); v_ += ABOPT_DPP_ROR(v_, 1); \
                t_[r] = v_;                                                                                               \
            }                                                                                                             \
            if (fm == 0) *reinterpret_cast<f32x4*>(b.dsum + unit * PAIR_DY + (OFF) + nt * 16 + kq * 4) = t_;              \
        }                                                                                                                 \
    }
// DST = (W^T . SRC) masked by ACT > 0
#define PAIR_BACK(DST, SRC, WT, ACT)                                                                                      \
    {                                                                                                                    \
        _Pragma("unroll") for (int mt = 0; mt < PMT; ++mt)                                                                \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) DST[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};                   \
        _Pragma("unroll") for (int blk = 0; blk < 4; ++blk) {                                                             \
            f32x4 w_[4];                                                                                                 \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) w_[nt] = (WT)[(blk * 4 + nt) * 64 + lane];                   \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                 \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                          \
                    _Pragma("unroll") for (int mt = 0; mt < PMT; ++mt)                                                    \
                        DST[mt][nt] = mfma4e(w_[nt][q], SRC[mt][blk][q], DST[mt][nt]);                                    \
        }                                                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < PMT; ++mt)                                                                \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                              \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) DST[mt][nt][r] = (ACT[mt][nt][r] > 0.f) ? DST[mt][nt][r] : 0.f; \
    }

__global__ __launch_bounds__(256, 2) void pair_embed_backward_kernel(PairBwdArgs b) {
    const int lane = threadIdx.x & 63, fm = lane & 15, kq = lane >> 4;
    const int L = b.L, A = b.A;
    const int jblocks = (L + 16 * PMT - 1) / (16 * PMT);
    const int64_t unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (unit >= (int64_t)b.N * L * jblocks) return;
    const int64_t row_i = unit / jblocks;
    const int j0 = (int)(unit % jblocks) * 16 * PMT;
    const int64_t base = (row_i / L) * L;
    const uint8_t fl_i = b.flags[row_i];
    int jc[PMT];
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt) jc[mt] = min(j0 + mt * 16 + fm, L - 1);
    f32x4 da[PMT][4], db[PMT][4], act[PMT][4];
    // d out x pair mask (pair.py:100)
#pragma unroll
    for (int mt = 0; mt < PMT; ++mt) {
        const float mp = ((fl_i & 1) && (b.flags[base + jc[mt]] & 1)) ? 1.f : 0.f;
        const float* s_ = b.dout + ((row_i * L) + jc[mt]) * EC + kq * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) da[mt][nt] = *reinterpret_cast<const f32x4*>(s_ + nt * 16) * mp;
    }
    PAIR_STORE_DY(da, 0)
    PAIR_LOAD_ACT(act, 224)                      // relu(out_mlp.2 pre-activation)
    PAIR_BACK(db, da, b.wo2t, act)
    PAIR_STORE_DY(db, 64)
    PAIR_LOAD_ACT(act, 160)                      // relu(out_mlp.0)
    PAIR_BACK(da, db, b.wo1t, act)
    PAIR_STORE_DY(da, 128)
    PAIR_LOAD_ACT(act, 64)                       // f_dist = relu(distance_embed.2) x structure mask: zero where masked
    PAIR_BACK(db, da, b.wo0dt, act)
    PAIR_STORE_DY(db, 192)
    PAIR_LOAD_ACT(act, 0)                        // relu(distance_embed.0)
    PAIR_BACK(da, db, b.wd1t, act)
    PAIR_STORE_DY(da, 256)
    // d G = Wd0^T dY_d0 per atom block a (operand rows = the 16 padded b), times T = dG / d softplus(coef)
    f32x4 pj[PMT][4];
    int aap[PMT];
    if (!b.tsave) {
        const int aa_i = b.aa_eff[row_i];
#pragma unroll
        for (int mt = 0; mt < PMT; ++mt) {
            const int64_t jr = base + jc[mt];
            aap[mt] = aa_i * AAT + b.aa_eff[jr];
#pragma unroll
            for (int q = 0; q < 4; ++q) pj[mt][q] = b.atoms4[jr * 16 + kq * 4 + q];
        }
    }
#pragma unroll 1
    for (int at = 0; at < A; ++at) {
        if (b.atoms4 && __builtin_amdgcn_readfirstlane(__float_as_uint(b.atoms4[row_i * 16 + at][3])) == 0u) {
            // atom `at` of residue i absent: T = 0 for the whole strip, so ds = 0 whatever Wd0^T dY is (the forward skips the same K block)
#pragma unroll
            for (int mt = 0; mt < PMT; ++mt)
                if (j0 + mt * 16 + fm < L) *reinterpret_cast<f32x4*>(b.ds + (((row_i * L) + jc[mt]) * A + at) * 16 + kq * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            continue;
        }
        f32x4 acc[PMT];
#pragma unroll
        for (int mt = 0; mt < PMT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const f32x4 w_ = b.wd0t[(blk * A + at) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mt = 0; mt < PMT; ++mt) acc[mt] = mfma4e(w_[q], da[mt][blk][q], acc[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < PMT; ++mt) {
            if (j0 + mt * 16 + fm < L) {
                const int64_t o_ = (((row_i * L) + jc[mt]) * A + at) * 16 + kq * 4;
                f32x4 tq;
                if (b.tsave) tq = *reinterpret_cast<const f32x4*>(b.tsave + o_);
                else {                                                                              // the forward's arithmetic (pair_embed_kernel), bit for bit
                    const f32x4 pi = b.atoms4[row_i * 16 + at];
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(b.sp + ((int64_t)aap[mt] * A + at) * 16 + kq * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dx = pi[0] - pj[mt][q][0], dy = pi[1] - pj[mt][q][1], dz = pi[2] - pj[mt][q][2];
                        const float dd = gauss_d2(dx, dy, dz);
                        const float gv = gauss_exp(c4[q], dd);
                        tq[q] = -dd * ((pi[3] != 0.f && pj[mt][q][3] != 0.f) ? gv : 0.f);
                    }
                }
                *reinterpret_cast<f32x4*>(b.ds + o_) = acc[mt] * tq;
            }
        }
    }
}

static size_t pair_tables_floats(int A) { return (size_t)AAT * AAT * EC + (size_t)NREL * EC + (size_t)AAT * AAT * A * 16; }
static int64_t pair_units(int N, int L) { return (int64_t)N * L * ((L + 16 * PMT - 1) / (16 * PMT)); }
size_t pair_embed_backward_ws_bytes(int N, int L, int A) {
    return pack_bytes((int64_t)N * L) + al256((size_t)(4 * 4 + 4 * A) * 1024 * 4 + 4096) + al256(pair_tables_floats(A) * 4) +
           al256((size_t)pair_units(N, L) * PAIR_DY * 4) + al256((size_t)1024 * PAIR_DY * 4);
}

int launch_colsum(const float* x, int ld, int64_t rows, int cols, float* out, float* ws, size_t ws_floats, hipStream_t st);    // gemm.hip

int launch_pair_embed_backward(const abopt_encode_inputs* in, const abopt_pair_embed_weights* w, const float* dout, const float* acts, const float* tsave,
                               float* dys, float* ds, float* dy_colsum, void* ws, size_t ws_bytes, hipStream_t st) {
    const int N = in->N, L = in->L, A = in->atoms;
    ABOPT_CHECK_ARG(A >= 3 && A <= 15 && A <= in->atoms_in, "pair_embed_backward: atoms=%d must be in [3, min(15, atoms_in=%d)]", A, in->atoms_in);
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    if (ws_bytes < pair_embed_backward_ws_bytes(N, L, A)) { set_error("pair_embed_backward: workspace too small"); return ABOPT_EWORKSPACE; }
    PackBufs pb = carve_pack((char*)ws, rows);
    f32x4* wt = (f32x4*)pb.end;
    f32x4 *wo2t = wt, *wo1t = wt + 1024, *wo0dt = wt + 2048, *wd1t = wt + 3072, *wd0t = wt + 4096;
    hipLaunchKernelGGL(residue_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(64), 0, st, in->aa, in->res_nb, in->chain_nb, in->pos_atoms, in->mask_atoms,
                       in->structure_mask, in->sequence_mask, in->atoms_in, A, rows, pb.atoms4, pb.aa_eff, pb.resnb, pb.chain, pb.flags, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(swizzle_transposed_kernel, dim3(4), dim3(256), 0, st, w->wo2, EC, 0, wo2t);
    hipLaunchKernelGGL(swizzle_transposed_kernel, dim3(4), dim3(256), 0, st, w->wo1, EC, 0, wo1t);
    hipLaunchKernelGGL(swizzle_transposed_kernel, dim3(4), dim3(256), 0, st, w->wo0, 3 * EC + 26, 2 * EC, wo0dt);
    hipLaunchKernelGGL(swizzle_transposed_kernel, dim3(4), dim3(256), 0, st, w->wd1, EC, 0, wd1t);
    hipLaunchKernelGGL(swizzle_wd0_transposed_kernel, dim3(A), dim3(256), 0, st, w->wd0, A, wd0t);
    ABOPT_LAUNCH_CHECK();
    PairBwdArgs b;
    b.atoms4 = pb.atoms4; b.aa_eff = pb.aa_eff; b.sp = nullptr;
    if (!tsave) {                                                   // softplus(coef) table, as the forward builds it
        float* tab = (float*)((char*)pb.end + al256((size_t)(4 * 4 + 4 * A) * 1024 * 4 + 4096));
        float* sp = tab + AAT * AAT * EC + NREL * EC;
        const int n = (int)pair_tables_floats(A);
        hipLaunchKernelGGL(pair_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w->aa_pair_embed, w->relpos_embed, w->aapair_to_distcoef, w->wo0, 3 * EC + 26, A,
                           tab, tab + AAT * AAT * EC, sp);
        ABOPT_LAUNCH_CHECK();
        b.sp = sp;
    }
    float* partials = (float*)((char*)pb.end + al256((size_t)(4 * 4 + 4 * A) * 1024 * 4 + 4096) + al256(pair_tables_floats(A) * 4));
    float* cs_ws = (float*)((char*)partials + al256((size_t)pair_units(N, L) * PAIR_DY * 4));
    b.dsum = dy_colsum ? partials : nullptr;
    b.dout = dout; b.acts = acts; b.tsave = tsave; b.flags = pb.flags;
    b.wo2t = wo2t; b.wo1t = wo1t; b.wo0dt = wo0dt; b.wd1t = wd1t; b.wd0t = wd0t;
    b.dys = dys; b.ds = ds; b.N = N; b.L = L; b.A = A; b.has_struct = in->structure_mask ? 1 : 0;
    const int jblocks = (L + 16 * PMT - 1) / (16 * PMT);
    hipLaunchKernelGGL(pair_embed_backward_kernel, dim3((unsigned)((rows * jblocks + 3) / 4)), dim3(256), 0, st, b);
    ABOPT_LAUNCH_CHECK();
    if (dy_colsum) return launch_colsum(partials, PAIR_DY, pair_units(N, L), PAIR_DY, dy_colsum, cs_ws, (size_t)1024 * PAIR_DY, st);
    return ABOPT_OK;
}

// ------------------------------------------------------------------------------------------------ backbone reconstruction
// reconstruct_backbone_partially (AbDock/src/modules/common/geometry.py:404-480): the step right after the sampler
// (design_for_pdb.py:166-223).  One thread per residue: N, CA, C = R l + t with l the ideal local coordinates of the residue
// type; O through the psi frame (psi of the RECONSTRUCTED backbone and the next residue's N, zero at chain ends / breaks);
// merged into the context atoms where mask_recons.
__global__ void reconstruct_backbone_kernel(const float* __restrict__ pos_ctx, const float* __restrict__ Rn, const float* __restrict__ tn,
                                            const int64_t* __restrict__ aa, const int64_t* __restrict__ chain_nb, const int64_t* __restrict__ res_nb,
                                            const uint8_t* __restrict__ mask_atoms, const uint8_t* __restrict__ mask_recons,
                                            const float* __restrict__ bb_table, const float* __restrict__ o_table,
                                            float* __restrict__ pos_new, uint8_t* __restrict__ mask_new, int64_t rows, int L, int A) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int l = (int)(row % L);
    const bool rec = mask_recons[row] != 0;
    float* po = pos_new + row * A * 3;
    uint8_t* mo = mask_new + row * A;
    if (!rec) {
        for (int k = 0; k < A * 3; ++k) po[k] = pos_ctx[row * A * 3 + k];
        for (int a = 0; a < A; ++a) mo[a] = mask_atoms[row * A + a];
        return;
    }
    auto to_global = [&](int64_t r, const float* loc) {
        const float* R = Rn + r * 9;
        const float* t = tn + r * 3;
        return v3(R[0] * loc[0] + R[1] * loc[1] + R[2] * loc[2] + t[0], R[3] * loc[0] + R[4] * loc[1] + R[5] * loc[2] + t[1],
                  R[6] * loc[0] + R[7] * loc[1] + R[8] * loc[2] + t[2]);
    };
    const int a0 = (int)min(max(aa[row], (int64_t)0), (int64_t)20);
    const float* bb = bb_table + a0 * 9;
    const V3 n = to_global(row, bb), ca = to_global(row, bb + 3), c = to_global(row, bb + 6);
    float psi = 0.f;
    if (l < L - 1 && llabs((long long)(res_nb[row + 1] - res_nb[row])) == 1 && chain_nb[row + 1] == chain_nb[row] && mask_atoms[row * A + 1]) {
        const int a1 = (int)min(max(aa[row + 1], (int64_t)0), (int64_t)20);
        psi = dihedral_from_four_points(n, ca, c, to_global(row + 1, bb_table + a1 * 9));
    }
    const float sp = sinf(psi), cp = cosf(psi);
    // O = R R_psi o + t, R_psi = rot_x(psi)
    const float* o = o_table + a0 * 3;
    const float lo[3] = {o[0], cp * o[1] - sp * o[2], sp * o[1] + cp * o[2]};
    const V3 ox = to_global(row, lo);
    const V3 at[4] = {n, ca, c, ox};
    for (int a = 0; a < A; ++a) {
        po[a * 3 + 0] = a < 4 ? at[a].x : 0.f; po[a * 3 + 1] = a < 4 ? at[a].y : 0.f; po[a * 3 + 2] = a < 4 ? at[a].z : 0.f;
        mo[a] = a < 4 ? 1 : 0;
    }
}

int launch_reconstruct_backbone(const float* pos_ctx, const float* R_new, const float* t_new, const int64_t* aa, const int64_t* chain_nb,
                                const int64_t* res_nb, const uint8_t* mask_atoms, const uint8_t* mask_recons, const float* bb_table,
                                const float* o_table, float* pos_new, uint8_t* mask_new, int N, int L, int A, hipStream_t st) {
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    hipLaunchKernelGGL(reconstruct_backbone_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, st, pos_ctx, R_new, t_new, aa, chain_nb, res_nb,
                       mask_atoms, mask_recons, bb_table, o_table, pos_new, mask_new, rows, L, A);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// ------------------------------------------------------------------------------------------------ host
static int residue_in_dim(int A, bool hotspot) { return EF + AAT * A * 3 + 39 + EF + (hotspot ? EF : 0); }

size_t residue_embed_ws_bytes(int N, int L, int A, int hotspot) {
    const int64_t rows = (int64_t)N * L;
    const int ld = (residue_in_dim(A, hotspot) + 3) & ~3;
    return pack_bytes(rows) + al256((size_t)rows * ld * 4) + al256((size_t)2 * EF * ld * 4) + al256((size_t)rows * 2 * EF * 4) + 2 * al256((size_t)rows * EF * 4) + 1024;
}

int launch_residue_embed(const abopt_encode_inputs* in, const abopt_residue_embed_weights* w, float* res_feat, float* R, float* p,
                         void* ws, size_t ws_bytes, hipStream_t st) {
    const int N = in->N, L = in->L, A = in->atoms;
    ABOPT_CHECK_ARG(A >= 3 && A <= 15 && A <= in->atoms_in, "residue_embed: atoms=%d must be in [3, min(15, atoms_in=%d)]", A, in->atoms_in);
    ABOPT_CHECK_ARG(R && p && res_feat, "residue_embed: outputs must not be NULL");
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    const bool hs = w->hotspot_embed != nullptr;
    if (ws_bytes < residue_embed_ws_bytes(N, L, A, hs)) { set_error("residue_embed: workspace too small"); return ABOPT_EWORKSPACE; }
    const int in_dim = residue_in_dim(A, hs), ld = (in_dim + 3) & ~3;
    PackBufs pb = carve_pack((char*)ws, rows);
    char* q = pb.end;
    float* feat = (float*)q; q += al256((size_t)rows * ld * 4);
    float* w0p = (float*)q; q += al256((size_t)2 * EF * ld * 4);
    float* h0 = (float*)q; q += al256((size_t)rows * 2 * EF * 4);
    float* h1 = (float*)q; q += al256((size_t)rows * EF * 4);
    float* h2 = (float*)q;
    hipLaunchKernelGGL(residue_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(64), 0, st, in->aa, in->res_nb, in->chain_nb, in->pos_atoms, in->mask_atoms,
                       in->structure_mask, in->sequence_mask, in->atoms_in, A, rows, pb.atoms4, pb.aa_eff, pb.resnb, pb.chain, pb.flags, R, p);
    ABOPT_LAUNCH_CHECK();
    hipLaunchKernelGGL(residue_feat_kernel, dim3((unsigned)rows), dim3(256), 0, st, pb.atoms4, pb.aa_eff, pb.resnb, pb.chain, pb.flags, R, in->fragment_type,
                       in->hotspot, w->aatype_embed, w->type_embed, w->hotspot_embed, w->freq_bands, A, L, in->structure_mask ? 1 : 0, feat, ld);
    ABOPT_LAUNCH_CHECK();
    {
        const int64_t n = (int64_t)2 * EF * ld;
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w->w0, in_dim, w0p, ld, (int64_t)2 * EF);
        ABOPT_LAUNCH_CHECK();
    }
    int rc;
    if ((rc = launch_linear(feat, ld, w0p, ld, w->b0, h0, 2 * EF, (int)rows, 2 * EF, ld, true, st))) return rc;
    if ((rc = launch_linear(h0, 2 * EF, w->w1, 2 * EF, w->b1, h1, EF, (int)rows, EF, 2 * EF, true, st))) return rc;
    if ((rc = launch_linear(h1, EF, w->w2, EF, w->b2, h2, EF, (int)rows, EF, EF, true, st))) return rc;
    if ((rc = launch_linear(h2, EF, w->w3, EF, w->b3, res_feat, EF, (int)rows, EF, EF, false, st))) return rc;
    hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)((rows * EF + 255) / 256)), dim3(256), 0, st, res_feat, pb.flags, rows);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

// The input features of ResidueEmbedding's MLP on their own (residue.py:33-88): [rows, ld] with ld = in_dim rounded up to 4, for the
// training path (the MLP then runs under autograd on abopt_gemm).  Frames R and CA positions p come along as in the fused forward.
size_t residue_features_ws_bytes(int N, int L) { return pack_bytes((int64_t)N * L) + 1024; }
int launch_residue_features(const abopt_encode_inputs* in, const abopt_residue_embed_weights* w, float* feat, float* R, float* p, void* ws, size_t ws_bytes,
                            hipStream_t st) {
    const int N = in->N, L = in->L, A = in->atoms;
    ABOPT_CHECK_ARG(A >= 3 && A <= 15 && A <= in->atoms_in, "residue_features: atoms=%d must be in [3, min(15, atoms_in=%d)]", A, in->atoms_in);
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    if (ws_bytes < residue_features_ws_bytes(N, L)) { set_error("residue_features: workspace too small"); return ABOPT_EWORKSPACE; }
    const bool hs = w->hotspot_embed != nullptr;
    const int ld = (residue_in_dim(A, hs) + 3) & ~3;
    PackBufs pb = carve_pack((char*)ws, rows);
    hipLaunchKernelGGL(residue_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(64), 0, st, in->aa, in->res_nb, in->chain_nb, in->pos_atoms, in->mask_atoms,
                       in->structure_mask, in->sequence_mask, in->atoms_in, A, rows, pb.atoms4, pb.aa_eff, pb.resnb, pb.chain, pb.flags, R, p);
    ABOPT_LAUNCH_CHECK();
    hipLaunchKernelGGL(residue_feat_kernel, dim3((unsigned)rows), dim3(256), 0, st, pb.atoms4, pb.aa_eff, pb.resnb, pb.chain, pb.flags, R, in->fragment_type,
                       in->hotspot, w->aatype_embed, w->type_embed, w->hotspot_embed, w->freq_bands, A, L, in->structure_mask ? 1 : 0, feat, ld);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

static size_t pair_weight_floats(int A) {
    return (size_t)AAT * AAT * EC + NREL * EC + (size_t)AAT * AAT * A * 16 + (size_t)(A + 4 + 6 + 4 + 4) * 4 * 64 * 4;
}

size_t pair_embed_ws_bytes(int N, int L, int A) {
    return pack_bytes((int64_t)N * L) + al256(pair_weight_floats(A) * 4) + 4096;
}

int launch_pair_embed(const abopt_encode_inputs* in, const abopt_pair_embed_weights* w, float* pair_feat, float* acts, float* gsave, float* tsave,
                      void* ws, size_t ws_bytes, hipStream_t st) {
    const int N = in->N, L = in->L, A = in->atoms;
    ABOPT_CHECK_ARG(A >= 3 && A <= 15 && A <= in->atoms_in, "pair_embed: atoms=%d must be in [3, min(15, atoms_in=%d)]", A, in->atoms_in);
    const int64_t rows = (int64_t)N * L;
    if (rows == 0) return ABOPT_OK;
    if (ws_bytes < pair_embed_ws_bytes(N, L, A)) { set_error("pair_embed: workspace too small"); return ABOPT_EWORKSPACE; }
    PackBufs pb = carve_pack((char*)ws, rows);
    float* q = (float*)pb.end;
    float* t_aap = q; q += AAT * AAT * EC;
    float* t_rel = q; q += NREL * EC;
    float* sp = q; q += (size_t)AAT * AAT * A * 16;
    f32x4* wd0 = (f32x4*)q; q += (size_t)A * 1024;
    f32x4* wd1 = (f32x4*)q; q += 4 * 1024;
    f32x4* wo0 = (f32x4*)q; q += 6 * 1024;
    f32x4* wo1 = (f32x4*)q; q += 4 * 1024;
    f32x4* wo2 = (f32x4*)q; q += 4 * 1024;
    hipLaunchKernelGGL(residue_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(64), 0, st, in->aa, in->res_nb, in->chain_nb, in->pos_atoms, in->mask_atoms,
                       in->structure_mask, in->sequence_mask, in->atoms_in, A, rows, pb.atoms4, pb.aa_eff, pb.resnb, pb.chain, pb.flags, (float*)nullptr, (float*)nullptr);
    ABOPT_LAUNCH_CHECK();
    const int ldo0 = 3 * EC + 26;
    {
        const int n = AAT * AAT * EC + NREL * EC + AAT * AAT * A * 16;
        hipLaunchKernelGGL(pair_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w->aa_pair_embed, w->relpos_embed, w->aapair_to_distcoef, w->wo0, ldo0, A, t_aap, t_rel, sp);
        ABOPT_LAUNCH_CHECK();
    }
    auto swz = [&](const float* W, int ldw, int col0, int stride, int width, int kreal, int nblk, f32x4* out) {
        hipLaunchKernelGGL(swizzle_weights_kernel, dim3(nblk), dim3(256), 0, st, W, ldw, col0, stride, width, kreal, nblk, out);
    };
    swz(w->wd0, A * A, 0, A, A, A * A, A, wd0);
    swz(w->wd1, EC, 0, 16, 16, EC, 4, wd1);
    swz(w->wo0, ldo0, 2 * EC, 16, 16, EC, 4, wo0);                 // f_dist columns
    swz(w->wo0, ldo0, 3 * EC, 16, 16, 26, 2, wo0 + 4 * 256);       // dihedral columns
    swz(w->wo1, EC, 0, 16, 16, EC, 4, wo1);
    swz(w->wo2, EC, 0, 16, 16, EC, 4, wo2);
    ABOPT_LAUNCH_CHECK();
    PairArgs a;
    a.atoms4 = pb.atoms4; a.aa_eff = pb.aa_eff; a.res_nb = pb.resnb; a.chain_nb = pb.chain; a.flags = pb.flags;
    a.t_aap = t_aap; a.t_rel = t_rel; a.sp = sp; a.freq = w->freq_bands;
    a.wd0 = wd0; a.bd0 = w->bd0; a.wd1 = wd1; a.bd1 = w->bd1; a.wo0 = wo0; a.bo0 = w->bo0; a.wo1 = wo1; a.bo1 = w->bo1; a.wo2 = wo2; a.bo2 = w->bo2;
    a.out = pair_feat; a.N = N; a.L = L; a.A = A; a.has_struct = in->structure_mask ? 1 : 0; a.acts = acts; a.gsave = gsave; a.tsave = tsave;
    const int jblocks = (L + 16 * PMT - 1) / (16 * PMT);
    const int64_t units = rows * jblocks;
    hipLaunchKernelGGL(pair_embed_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, a);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

}  // namespace abopt

// DockQ scoring of docked candidates on the device (the step after the sampler in the reference's runner):
//   AbDock/src/tools/runner/design_for_pdb.py:316-321  calc_DockQ(model.pdb, native.pdb, use_CA_only=True) for EVERY candidate, which
//   AbDock/DockQ/DockQ.py:98-385 implements by shelling out to the `fnat` C program twice (DockQ/src/fnat.c:100-252, contacts at 5 A,
//   interface at 10 A, all heavy atoms) and by two Biopython superpositions (interface CA atoms -> iRMS; receptor CA atoms -> LRMS).
// Here the candidates never leave HBM: structures are tensors (pos [L,A,3], mask [L,A], chain group [L]) with the residue indexing of
// the batch; one launch scores all S candidates of a complex.  Latency-bound (a few hundred KB): one workgroup per candidate.
#include "abopt_common.h"
#include "kernels.h"

namespace abopt {

constexpr int DQ_CA = 1;

// minimum squared heavy-atom distance of residues a, b (molecule.c:581-612 crd()); +inf if either has no atom
__device__ __forceinline__ float res_min_d2(const float* __restrict__ pos, const uint8_t* __restrict__ mask, int a, int b, int A) {
    float best = INFINITY;
    for (int p = 0; p < A; ++p) {
        if (!mask[a * A + p]) continue;
        const float ax = pos[(a * A + p) * 3], ay = pos[(a * A + p) * 3 + 1], az = pos[(a * A + p) * 3 + 2];
        for (int q = 0; q < A; ++q) {
            if (!mask[b * A + q]) continue;
            const float dx = ax - pos[(b * A + q) * 3], dy = ay - pos[(b * A + q) * 3 + 1], dz = az - pos[(b * A + q) * 3 + 2];
            best = fminf(best, dx * dx + dy * dy + dz * dz);
        }
    }
    return best;
}

// native contacts: nat5[a*L+b] = 1 for a in chain 1, b in chain 2 with min distance <= 5 A (fnat.c:129-140); interface[r] = 1 for
// residues in a native contact at 10 A (DockQ.py:110,121-123 + parse_fnat :18-49); nat_total = number of 5 A contacts.
__global__ __launch_bounds__(256) void dockq_native_kernel(const float* __restrict__ pos, const uint8_t* __restrict__ mask, const int32_t* __restrict__ group,
                                                           uint8_t* __restrict__ nat5, uint8_t* __restrict__ interface, int* __restrict__ nat_total, int L, int A) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L * L) return;
    const int a = idx / L, b = idx % L;
    uint8_t c5 = 0;
    if (group[a] == 1 && group[b] == 2) {
        const float d2 = res_min_d2(pos, mask, a, b, A);
        c5 = d2 <= 25.f;
        if (d2 <= 100.f) { interface[a] = 1; interface[b] = 1; }          // benign race: every writer stores 1
        if (c5) atomicAdd(nat_total, 1);
    }
    nat5[idx] = c5;
}

__device__ __forceinline__ double block_sum(double v, double* red) {          // 256 threads; red[4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Largest-eigenvalue eigenvector of the symmetric 4x4 matrix N (cyclic Jacobi, double): Horn's unit quaternion of the best rotation.
__device__ void jacobi4_max_eigvec(double N[4][4], double q[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0;
        for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += N[i][j] * N[i][j];
        if (off < 1e-30) break;
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                if (fabs(N[p][r]) < 1e-300) continue;
                const double theta = (N[r][r] - N[p][p]) / (2.0 * N[p][r]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 4; ++k) { const double a = N[k][p], b = N[k][r]; N[k][p] = c * a - s * b; N[k][r] = s * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = N[p][k], b = N[r][k]; N[p][k] = c * a - s * b; N[r][k] = s * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = V[k][p], b = V[k][r]; V[k][p] = c * a - s * b; V[k][r] = s * a + c * b; }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i) if (N[i][i] > N[best][best]) best = i;
    for (int k = 0; k < 4; ++k) q[k] = V[k][best];
}

// Best rotation R (row-major) and centroids moving the model CA atoms selected by `fit` onto the native ones (SVDSuperimposer), then
// the RMSD of the atoms selected by `eval` under that transform (without refitting).  Block-wide; returns the RMSD to every thread.
template <typename FitSel, typename EvalSel>
__device__ double fit_and_rmsd(const float* __restrict__ mp, const float* __restrict__ np_, int L, int A, FitSel fit, EvalSel eval, double* red, double* shR) {
    const int tid = threadIdx.x;
    double sx[3] = {0, 0, 0}, sy[3] = {0, 0, 0}, cnt = 0;
    for (int r = tid; r < L; r += 256)
        if (fit(r)) {
            for (int k = 0; k < 3; ++k) { sx[k] += np_[(r * A + DQ_CA) * 3 + k]; sy[k] += mp[(r * A + DQ_CA) * 3 + k]; }
            cnt += 1;
        }
    const double n = block_sum(cnt, red);
    if (n < 0.5) return -1.0;                                                   // nothing to superimpose (block-uniform): undefined, see abopt.h
    double cx[3], cy[3];
    for (int k = 0; k < 3; ++k) { cx[k] = block_sum(sx[k], red) / n; cy[k] = block_sum(sy[k], red) / n; }
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                                 // S[a][b] = sum (y_a - cy_a)(x_b - cx_b)
    for (int r = tid; r < L; r += 256)
        if (fit(r))
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    S[a * 3 + b] += ((double)mp[(r * A + DQ_CA) * 3 + a] - cy[a]) * ((double)np_[(r * A + DQ_CA) * 3 + b] - cx[b]);
    for (int k = 0; k < 9; ++k) S[k] = block_sum(S[k], red);
    if (tid == 0) {
        const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
        double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                          {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                          {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                          {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
        double q[4];
        jacobi4_max_eigvec(N, q);
        const double a = q[0], b = q[1], c = q[2], d = q[3];
        shR[0] = a * a + b * b - c * c - d * d; shR[1] = 2 * (b * c - a * d); shR[2] = 2 * (b * d + a * c);
        shR[3] = 2 * (b * c + a * d); shR[4] = a * a - b * b + c * c - d * d; shR[5] = 2 * (c * d - a * b);
        shR[6] = 2 * (b * d - a * c); shR[7] = 2 * (c * d + a * b); shR[8] = a * a - b * b - c * c + d * d;
    }
    __syncthreads();
    double ss = 0, m = 0;
    for (int r = tid; r < L; r += 256)
        if (eval(r)) {
            double y[3], x[3];
            for (int k = 0; k < 3; ++k) { y[k] = (double)mp[(r * A + DQ_CA) * 3 + k] - cy[k]; x[k] = (double)np_[(r * A + DQ_CA) * 3 + k] - cx[k]; }
            for (int k = 0; k < 3; ++k) { const double e = shR[k * 3] * y[0] + shR[k * 3 + 1] * y[1] + shR[k * 3 + 2] * y[2] - x[k]; ss += e * e; }
            m += 1;
        }
    const double tot = block_sum(ss, red), mm = block_sum(m, red);
    return mm < 0.5 ? -1.0 : sqrt(tot / mm);
}

// one workgroup per candidate: Fnat (fnat.c:225-243), iRMS (DockQ.py:296-301), LRMS (DockQ.py:303-366), DockQ (:378)
__global__ __launch_bounds__(256) void dockq_model_kernel(const float* __restrict__ model_pos, const uint8_t* __restrict__ model_mask, int64_t mask_stride,
                                                          const float* __restrict__ native_pos, const uint8_t* __restrict__ native_mask,
                                                          const int32_t* __restrict__ group, const uint8_t* __restrict__ nat5,
                                                          const uint8_t* __restrict__ interface, const int* __restrict__ nat_total,
                                                          float* __restrict__ out, int L, int A) {
    __shared__ double red[4], shR[9];
    __shared__ int cnt_sh[2];
    const int s = blockIdx.x, tid = threadIdx.x;
    const float* mp = model_pos + (int64_t)s * L * A * 3;
    const uint8_t* mm = model_mask + (int64_t)s * mask_stride;
    // ---- Fnat: native 5 A contacts that are also contacts in the model
    double correct = 0;
    for (int idx = tid; idx < L * L; idx += 256)
        if (nat5[idx] && res_min_d2(mp, mm, idx / L, idx % L, A) <= 25.f) correct += 1;
    correct = block_sum(correct, red);
    const int ntot = *nat_total;
    const double fnat = ntot ? correct / (double)ntot : 0.0;
    // ---- common CA atoms (atoms_def_in_both, DockQ.py:150-188)
    auto both = [&](int r) { return group[r] > 0 && mm[r * A + DQ_CA] && native_mask[r * A + DQ_CA]; };
    double n1 = 0, n2 = 0;
    for (int r = tid; r < L; r += 256) if (both(r)) { n1 += group[r] == 1; n2 += group[r] == 2; }
    n1 = block_sum(n1, red); n2 = block_sum(n2, red);
    if (tid == 0) { cnt_sh[0] = (int)n1; cnt_sh[1] = (int)n2; }
    __syncthreads();
    const int rec = cnt_sh[0] > cnt_sh[1] ? 1 : 2, lig = 3 - rec;               // receptor = the chain with more common atoms (DockQ.py:314-318)
    auto isel = [&](int r) { return both(r) && interface[r]; };
    const double irms = fit_and_rmsd(mp, native_pos, L, A, isel, isel, red, shR);
    auto rsel = [&](int r) { return both(r) && group[r] == rec; };
    auto lsel = [&](int r) { return both(r) && group[r] == lig; };
    const double lrms = fit_and_rmsd(mp, native_pos, L, A, rsel, lsel, red, shR);
    if (tid == 0) {
        out[s * 4 + 0] = (float)fnat;
        out[s * 4 + 1] = (float)irms;
        out[s * 4 + 2] = (float)lrms;
        // an empty interface / receptor / ligand selection has no RMSD (calc_DockQ asserts there, DockQ.py:150-188,296-366): -1 marks it
        out[s * 4 + 3] = (irms < 0 || lrms < 0) ? -1.f
                                                : (float)((fnat + 1.0 / (1.0 + (irms / 1.5) * (irms / 1.5)) + 1.0 / (1.0 + (lrms / 8.5) * (lrms / 8.5))) / 3.0);
    }
}

}  // namespace abopt

using namespace abopt;

extern "C" size_t abopt_dockq_workspace_bytes(int L) { return (size_t)L * L + (size_t)L + 256; }

extern "C" int abopt_dockq_lite(const float* model_pos, const uint8_t* model_mask, int model_mask_shared, const float* native_pos,
                                const uint8_t* native_mask, const int32_t* group, int S, int L, int A, float* out,
                                void* ws, size_t ws_bytes, abopt_stream stream) {
    ABOPT_CHECK_ARG(S >= 0 && L >= 1 && A >= 2, "dockq_lite: bad dims S=%d L=%d A=%d", S, L, A);
    if (S == 0) return ABOPT_OK;
    ABOPT_CHECK_ARG(model_pos && model_mask && native_pos && native_mask && group && out && ws, "dockq_lite: NULL argument");
    if (ws_bytes < abopt_dockq_workspace_bytes(L)) { set_error("dockq_lite: workspace too small (%zu bytes given)", ws_bytes); return ABOPT_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    uint8_t* nat5 = (uint8_t*)ws;
    uint8_t* interface = nat5 + (size_t)L * L;
    int* nat_total = (int*)(((uintptr_t)(interface + L) + 63) & ~(uintptr_t)63);
    ABOPT_HIP(hipMemsetAsync(interface, 0, (size_t)L + 128, st));
    hipLaunchKernelGGL(dockq_native_kernel, dim3((unsigned)((L * L + 255) / 256)), dim3(256), 0, st, native_pos, native_mask, group, nat5, interface, nat_total, L, A);
    ABOPT_LAUNCH_CHECK();
    hipLaunchKernelGGL(dockq_model_kernel, dim3((unsigned)S), dim3(256), 0, st, model_pos, model_mask, model_mask_shared ? (int64_t)0 : (int64_t)L * A,
                       native_pos, native_mask, group, nat5, interface, nat_total, out, L, A);
    ABOPT_LAUNCH_CHECK();
    return ABOPT_OK;
}

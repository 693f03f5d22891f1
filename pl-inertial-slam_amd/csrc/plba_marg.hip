// plba_marg.hip — K9: marginalization of the oldest keyframe on the device.
//
// Replaces MarginalizationInfo::preMarginalize / marginalizeWithoutThread (IMU/marginalization.cpp:128-147,
// 291-384) and the factor selection of the call site (src/mapHandler.cpp:6075-6188).
//
//   host   : factor selection + parameter ordering (pure index work on the uploaded graph)
//   device : (1) every selected factor re-evaluated at the final estimate, raw Jacobians, no information
//                matrix and no robust weight (marginalization.cpp:67), written as columns of one stacked
//                Jacobian  J (R x pos, column-major);  each factor's OWN (r, J): SURVEY B-Q3 decision
//            (2) A = J^T J, b = J^T r                                   (ThreadsConstructA, cpp:8-36)
//            (3) Schur elimination of the dropped block with the eigen pseudo-inverse, eigenvalues <= eps -> 0
//                (cpp:351-362).  The dropped block is [oldest keyframe (<= 15) | landmark blocks (3 / 6)] and
//                the landmark blocks do not couple, so the pseudo-inverse is taken block by block (landmarks,
//                then the keyframe block of the reduced system) — identical to the dense pseudo-inverse
//                whenever the discarded eigen-directions are block-local null spaces (rank-deficient landmark
//                blocks), which is the case the threshold exists for.
//            (4) A' = V S V^T by one-sided Jacobi, J0 = sqrt(S) V^T, r0 = sqrt(S^-1) V^T b'   (cpp:364-372).
//                n <= 100 (the reference's 12-keyframe window keeps <= 7 x 9 + 6 + the old prior's vertices): ONE
//                workgroup holds G = A'V and V in LDS (2 n^2 doubles <= 160 KB) and runs every round of every sweep
//                without leaving the CU; n <= 140 with V in global memory; larger n: two launches per round (k_j2_cols / k_j2_rows).
//   One stream synchronisation per call: index lists go up through the pinned staging area before the first launch,
//   results come back through it in two asynchronous copies.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "plba_problem.h"

namespace plba {

#define MDEV __device__ __forceinline__

namespace {

constexpr int MAXB = 15;   // largest block eliminated at once (PVR 9 + bias 6)

// ---- (1) factor evaluation -------------------------------------------------------------------------------------
struct MargObs { int edge; int row; int col_lm; int col_kf; };

MDEV void marg_obs(const DevBuf& d, int state, const MargObs* f, int nf, double* J, double* r, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const MargObs o = f[i];
    const int e = o.edge;
    const double* s = d.kf[state] + (size_t)d.ob_kf[e] * KF_STRIDE;
    double kc[KFCAM_STRIDE];
    kfcam_make(d.cam, s, kc);
    const double* L = d.lm[state] + (size_t)d.ob_slot[e] * 6;
    double e2[2], Jp[12], Jl[6];
    bool dp;
    const bool is_pt = e < d.Ep;
    if (is_pt) {
        point_edge(d.cam, kc, v3(L[0], L[1], L[2]), d.po_uv[2 * (size_t)e], d.po_uv[2 * (size_t)e + 1], e2, Jp, Jl, dp, true);
    } else {
        const double* l = d.lo_l + (size_t)(e - d.Ep) * 3;
        line_edge(d.cam, kc, v3(L[0], L[1], L[2]), v3(L[3], L[4], L[5]), l[0], l[1], l[2], d.fix_q1 != 0, e2, Jp, Jl, dp, true);
    }
    r[o.row] = e2[0]; r[o.row + 1] = e2[1];
    for (int a = 0; a < 2; ++a) {
        for (int c = 0; c < 3; ++c) {
            J[(size_t)(o.col_kf + c) * R + o.row + a] = Jp[a * 6 + c];          // dp
            J[(size_t)(o.col_kf + 6 + c) * R + o.row + a] = Jp[a * 6 + 3 + c];  // dphi (velocity columns stay zero)
        }
        if (is_pt) for (int c = 0; c < 3; ++c) J[(size_t)(o.col_lm + c) * R + o.row + a] = Jl[a * 3 + c];
        else for (int c = 0; c < 3; ++c) J[(size_t)(o.col_lm + 3 * a + c) * R + o.row + a] = Jl[a * 3 + c];
    }
}

MDEV void marg_imu(const DevBuf& d, int state, int m, int row, int c_pi, int c_pj, int c_bi, int c_bj, double* J, double* r, int R) {
    if (threadIdx.x != 0) return;
    const double* si = d.kf[state] + (size_t)d.imu_i[m] * KF_STRIDE;
    const double* sj = d.kf[state] + (size_t)d.imu_j[m] * KF_STRIDE;
    const double* pre = d.imu_pre + (size_t)m * PRE_STRIDE;
    double e9[9], e6[6], J0[81], J1[81], J2[54];
    pvr_error(si, sj, pre, d.gw, e9);
    for (int t = 0; t < 81; ++t) { J0[t] = 0.0; J1[t] = 0.0; }
    for (int t = 0; t < 54; ++t) J2[t] = 0.0;
    pvr_jacobians(si, sj, pre, d.gw, e9, J0, J1, J2);
    bias_error(si, sj, e6);
    for (int a = 0; a < 9; ++a) {
        r[row + a] = e9[a];
        for (int c = 0; c < 9; ++c) { J[(size_t)(c_pi + c) * R + row + a] = J0[a * 9 + c]; J[(size_t)(c_pj + c) * R + row + a] = J1[a * 9 + c]; }
        for (int c = 0; c < 6; ++c) J[(size_t)(c_bi + c) * R + row + a] = J2[a * 6 + c];
    }
    for (int a = 0; a < 6; ++a) {
        r[row + 9 + a] = e6[a];
        J[(size_t)(c_bi + a) * R + row + 9 + a] = -1.0;
        J[(size_t)(c_bj + a) * R + row + 9 + a] = 1.0;
    }
}

// old prior as a factor: residual = EdgeMarginalization error at the final estimate, Jacobian = J0 columns
MDEV void marg_prior(const DevBuf& d, int blk, int nblk, int row, const int* vcol, double* J, double* r, int R) {
    const int n = d.pr_n, stride = nblk * blockDim.x, first = blk * blockDim.x + threadIdx.x;
    for (int t = first; t < n; t += stride) r[row + t] = d.pr_err[t];
    for (int v = 0; v < d.pr_nv; ++v) {
        const int sz = d.pr_size[v], ix = d.pr_idx[v], col = vcol[v];
        for (int t = first; t < sz * n; t += stride) {
            const int c = t / n, rr = t % n;
            J[(size_t)(col + c) * R + row + rr] = d.pr_J0[(size_t)(ix + c) * n + rr];
        }
    }
}
// one launch for every selected factor: blocks [0, ob) observation edges (a lane each), [ob, ob + nimu) one IMU edge pair each,
// the rest copy the old prior's rows
__global__ __launch_bounds__(64) void k_marg_factors(DevBuf d, int state, const MargObs* f, int nf, int ob, const int* imu, int nimu, int prior_row,
                                                    const int* vcol, double* J, double* r, int R) {
    const int b = blockIdx.x;
    if (b < ob) { marg_obs(d, state, f, nf, J, r, R); return; }
    if (b < ob + nimu) { const int* q = imu + 6 * (b - ob); marg_imu(d, state, q[0], q[1], q[2], q[3], q[4], q[5], J, r, R); return; }
    marg_prior(d, b - ob - nimu, gridDim.x - ob - nimu, prior_row, vcol, J, r, R);
}

__global__ __launch_bounds__(256) void k_jt_r(const double* J, const double* r, int R, int pos, double* b) {      // one wave per column
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= pos) return;
    const double* col = J + (size_t)c * R;
    double s = 0.0;
    for (int t = lane; t < R; t += 64) s += col[t] * r[t];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) b[c] = s;
}

// ---- (3) block pseudo-inverse + Schur update --------------------------------------------------------------------
// in-thread cyclic Jacobi eigen-decomposition of an s x s symmetric block (s <= 15), pinv with threshold eps
__global__ void k_block_pinv(const double* A, int pos, const int* boff, const int* bsize, int nblk, double eps, double* Pinv) {
    const int bi = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi >= nblk) return;
    const int s = bsize[bi], o = boff[bi];
    double M[MAXB * MAXB], V[MAXB * MAXB];
    for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j) {
            M[i * MAXB + j] = 0.5 * (A[(size_t)(o + i) * pos + o + j] + A[(size_t)(o + j) * pos + o + i]);   // cpp:351
            V[i * MAXB + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < s; ++i) { dg += M[i * MAXB + i] * M[i * MAXB + i]; for (int j = i + 1; j < s; ++j) off += M[i * MAXB + j] * M[i * MAXB + j]; }
        if (off <= 1e-32 * (dg + off) || off == 0.0) break;
        for (int p = 0; p < s - 1; ++p)
            for (int q = p + 1; q < s; ++q) {
                const double apq = M[p * MAXB + q];
                if (apq == 0.0) continue;
                const double th = (M[q * MAXB + q] - M[p * MAXB + p]) / (2.0 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < s; ++k) { const double a = M[k * MAXB + p], b = M[k * MAXB + q]; M[k * MAXB + p] = c * a - sn * b; M[k * MAXB + q] = sn * a + c * b; }
                for (int k = 0; k < s; ++k) { const double a = M[p * MAXB + k], b = M[q * MAXB + k]; M[p * MAXB + k] = c * a - sn * b; M[q * MAXB + k] = sn * a + c * b; }
                for (int k = 0; k < s; ++k) { const double a = V[k * MAXB + p], b = V[k * MAXB + q]; V[k * MAXB + p] = c * a - sn * b; V[k * MAXB + q] = sn * a + c * b; }
            }
    }
    double* out = Pinv + (size_t)bi * MAXB * MAXB;
    for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j) {
            double acc = 0.0;
            for (int k = 0; k < s; ++k) { const double w = M[k * MAXB + k]; if (w > eps) acc += V[i * MAXB + k] * V[j * MAXB + k] / w; }
            out[i * MAXB + j] = acc;
        }
}
// Z[row][o + c] = sum_t A[row][o + t] Pinv_b[t][c]     (only the eliminated columns of Z are written)
__global__ void k_block_Z(const double* A, int pos, const int* boff, const int* bsize, int nblk, const double* Pinv, double* Z) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.y;
    if (row >= pos || bi >= nblk) return;
    const int s = bsize[bi], o = boff[bi];
    const double* P = Pinv + (size_t)bi * MAXB * MAXB;
    for (int c = 0; c < s; ++c) {
        double acc = 0.0;
        for (int t = 0; t < s; ++t) acc += A[(size_t)row * pos + o + t] * P[t * MAXB + c];
        Z[(size_t)row * pos + o + c] = acc;
    }
}
__global__ void k_block_Z1(const double* A, int pos, int o, int s, const double* P, double* Z) {      // one block at offset o; a thread per (row, column)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pos * s) return;
    const int row = i / s, c = i - row * s;
    double acc = 0.0;
    for (int t = 0; t < s; ++t) acc += A[(size_t)row * pos + o + t] * P[t * MAXB + c];
    Z[(size_t)row * pos + o + c] = acc;
}
// A[r][c] -= sum_{k in elim} Z[r][k] A[k][c],  b[r] -= sum Z[r][k] b[k]   for r, c in `rest`
__global__ void k_schur_apply(double* A, double* b, int pos, const double* Z, const int* elim, int nelim, const uint8_t* is_rest) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y * blockDim.y + threadIdx.y;
    if (r >= pos || c > pos || !is_rest[r]) return;
    if (c < pos && !is_rest[c]) return;
    double acc = 0.0;
    if (c == pos) { for (int t = 0; t < nelim; ++t) { const int k = elim[t]; acc += Z[(size_t)r * pos + k] * b[k]; } b[r] -= acc; }
    else { for (int t = 0; t < nelim; ++t) { const int k = elim[t]; acc += Z[(size_t)r * pos + k] * A[(size_t)k * pos + c]; } A[(size_t)r * pos + c] -= acc; }
}


// landmark blocks in registers: S = 3 (point) / 6 (line), everything unrolled so that M and V never touch scratch memory.
// With `cert` the block also leaves what the certificate of marg_exact = 1 needs (see cert_* below): the shifted inverse
// sum_{w_k > hi} u_k u_k^T / (w_k - hi), the directions the block-wise path discards (w_k <= hi), the smallest eigenvalue it keeps.
struct CertBuf {
    double* pinv_hi;                  // nblk x 36
    double* udrop;                    // nblk x 36: discarded eigenvectors, one per row
    int* ndrop;                       // nblk
    unsigned long long* lam_min_bits; // min over blocks of the smallest kept eigenvalue (bits of a positive double order like integers)
    unsigned long long* w2max_bits;   // max over discarded directions of |A[:, block] u|^2
    int* band;                        // an eigenvalue in (eps, hi]: kept by the block-wise path, counted as discarded by the certificate
};
template <int S>
MDEV void pinv_small(const double* A, int pos, int o, double eps, double* out, bool cert, double hi, const CertBuf cb, int bi) {
    double M[S][S], V[S][S];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            M[i][j] = 0.5 * (A[(size_t)(o + i) * pos + o + j] + A[(size_t)(o + j) * pos + o + i]);   // cpp:351
            V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            dg += M[i][i] * M[i][i];
#pragma unroll
            for (int j = i + 1; j < S; ++j) off += M[i][j] * M[i][j];
        }
        if (off <= 1e-28 * (dg + off) || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < S - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < S; ++q) {
                const double apq = M[p][q];
                if (apq != 0.0) {
                    const double th = (M[q][q] - M[p][p]) / (2.0 * apq);
                    const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                    for (int k = 0; k < S; ++k) { const double a = M[k][p], b = M[k][q]; M[k][p] = c * a - sn * b; M[k][q] = sn * a + c * b; }
#pragma unroll
                    for (int k = 0; k < S; ++k) { const double a = M[p][k], b = M[q][k]; M[p][k] = c * a - sn * b; M[q][k] = sn * a + c * b; }
#pragma unroll
                    for (int k = 0; k < S; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - sn * b; V[k][q] = sn * a + c * b; }
                }
            }
    }
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < S; ++k) { const double w = M[k][k]; if (w > eps) acc += V[i][k] * V[j][k] / w; }
            out[i * MAXB + j] = acc;
        }
    if (!cert) return;
    double* ph = cb.pinv_hi + (size_t)bi * 36; double* ud = cb.udrop + (size_t)bi * 36;
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < S; ++k) { const double w = M[k][k]; if (w > hi) acc += V[i][k] * V[j][k] / (w - hi); }
            ph[i * 6 + j] = acc;
        }
    int nd = 0, band = 0;
    double lmin = 1e300;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const double w = M[k][k];
        if (w > hi) { lmin = fmin(lmin, w); continue; }
        if (w > eps) band = 1;
#pragma unroll
        for (int i = 0; i < S; ++i) ud[nd * 6 + i] = V[i][k];
        ++nd;
    }
    cb.ndrop[bi] = nd;
    if (band) atomicOr(cb.band, 1);
    atomicMin(cb.lam_min_bits, (unsigned long long)__double_as_longlong(lmin));
}
__global__ __launch_bounds__(64) void k_block_pinv_small(const double* A, int pos, const int* boff, const int* bsize, int nblk, double eps, double* Pinv, bool cert, double hi, CertBuf cb) {
    const int bi = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi >= nblk) return;
    double* out = Pinv + (size_t)bi * MAXB * MAXB;
    if (bsize[bi] == 3) pinv_small<3>(A, pos, boff[bi], eps, out, cert, hi, cb, bi);
    else pinv_small<6>(A, pos, boff[bi], eps, out, cert, hi, cb, bi);
}

// ---- one-sided Jacobi (Hestenes) of a symmetric positive semi-definite n x n matrix held in LDS by ONE workgroup ------------
// G (column-major, leading dimension n, starts as the matrix) and V (starts as I) stay in LDS through every round of
// every sweep; a round rotates the npad / 2 disjoint column pairs of the round-robin schedule, one wave per pair
// (each lane owns rows lane, lane + 64: n <= 128), and ends in one workgroup barrier.  On return the columns of G are
// orthogonal: g_j = lambda_j v_j.
MDEV double wave_sum(double x) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
template <int CTRL, int ROW_MASK>
MDEV double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over each 32-lane half of the wave, returned to every lane of that half (row_shr 1, 2, 4, 8, row_bcast:15, readlane)
MDEV double half_sum(double v, bool upper) {
    v += dpp_get<0x111, 0xf>(v);
    v += dpp_get<0x112, 0xf>(v);
    v += dpp_get<0x114, 0xf>(v);
    v += dpp_get<0x118, 0xf>(v);
    v += dpp_get<0x142, 0xa>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double s0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double s1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return upper ? s1 : s0;
}
// sum over the wave, to every lane: the same DPP steps, both halves' totals by v_readlane (the __shfl_xor form of wave_sum is six dependent
// ds_bpermute round trips per 32-bit half: ~800 cycles)
MDEV double wave_sum_dpp(double v) {
    v += dpp_get<0x111, 0xf>(v);
    v += dpp_get<0x112, 0xf>(v);
    v += dpp_get<0x114, 0xf>(v);
    v += dpp_get<0x118, 0xf>(v);
    v += dpp_get<0x142, 0xa>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31)) + __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
}
// A round rotates the npad / 2 disjoint column pairs of the round-robin schedule, one HALF-wave per pair (lane l of the half
// owns rows l, l + 32, l + 64, l + 96: n <= 128), and ends in one workgroup barrier.  A pair is left alone when
//   |g_p . g_q| <= tol |g_p| |g_q|                      (orthogonal to the rounding of the dot product), or
//   |g_p . g_q| <= delta (|g_p| + |g_q|),  delta = 4 n macheps |A|_F    (noise2 = 2 delta^2 against (|g_p|^2 + |g_q|^2)):
// every entry of G carries ~ delta of absolute rounding from the rotations, so a column that has sunk to that level (a
// numerically null direction: the gauge freedom of the first prior) can never pass the RELATIVE test and would keep every
// sweep rotating — round 1's criterion never terminated before its sweep limit.
MDEV void jacobi_lds(double* G, double* V, int n, double tol, double noise2, int max_sweeps, int* s_rot, double* dbg = nullptr) {
    __shared__ double s_norm[128];
    __shared__ int s_perm[128], s_live;
    const int lane = threadIdx.x & 63, hl = lane & 31, nh = (blockDim.x >> 6) * 2, half = (threadIdx.x >> 6) * 2 + (lane >> 5);
    const bool upper = lane >= 32;
    if (n < 2) return;
    const double tol2 = tol * tol;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        // columns at the noise floor (|g_j|^2 <= noise2: never rotated, see above — e.g. the exactly zero velocity columns of
        // keyframes no selected factor constrains) leave the schedule; the others are visited in order of decreasing norm
        for (int j = threadIdx.x; j < n; j += blockDim.x) { double a = 0.0; const double* g = G + (size_t)j * n; for (int t = 0; t < n; ++t) a += g[t] * g[t]; s_norm[j] = a; }
        if (threadIdx.x == 0) { *s_rot = 0; s_live = 0; }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const double a = s_norm[j];
            if (a > noise2) {
                int rank = 0;
                for (int k = 0; k < n; ++k) { const double b = s_norm[k]; rank += (b > noise2) && (b > a || (b == a && k < j)); }
                s_perm[rank] = j;
                atomicAdd(&s_live, 1);
            }
        }
        __syncthreads();
        const int nl = s_live;
        if (nl < 2) break;
        const int npad = (nl & 1) ? nl + 1 : nl, mm = npad - 1, npairs = npad / 2;
        for (int round = 0; round < mm; ++round) {
            for (int i0 = 0; i0 < npairs; i0 += nh) {      // both halves of a wave run the loop the same number of times (DPP / readlane need the whole wave)
                const int i = i0 + half;
                int p = 0, q = 0;
                bool live = i < npairs;
                if (live) {
                    if (i == 0) { p = mm; q = round % mm; }
                    else { p = (round + i) % mm; q = (round - i + mm) % mm; }
                    if (p > q) { const int t = p; p = q; q = t; }
                    live = q < nl;                               // padding column: bye
                    if (live) { p = s_perm[p]; q = s_perm[q]; }
                }
                double* gp = G + (size_t)p * n; double* gq = G + (size_t)q * n;
                double x[4], y[4];
                double pa = 0.0, pb = 0.0, pg = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = hl + 32 * k;
                    const bool in = live && t < n;
                    x[k] = in ? gp[t] : 0.0; y[k] = in ? gq[t] : 0.0;
                    pa += x[k] * x[k]; pb += y[k] * y[k]; pg += x[k] * y[k];
                }
                const double a = half_sum(pa, upper), b = half_sum(pb, upper), g = half_sum(pg, upper);
                if (!live || !(g * g > tol2 * a * b) || !(g * g > noise2 * (a + b)) || g == 0.0) continue;
                if (hl == 0) atomicAdd(s_rot, 1);
                // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with zeta = (b - a) / (2 g), written with one square root and one division
                const double dd = b - a;
                const double t = ((dd * g >= 0.0) ? 2.0 : -2.0) * fabs(g) / (fabs(dd) + sqrt(dd * dd + 4.0 * g * g));
                const double c = rsqrt(1.0 + t * t), s = c * t;
                double* vp = V + (size_t)p * n; double* vq = V + (size_t)q * n;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = hl + 32 * k;
                    if (r < n) {
                        gp[r] = c * x[k] - s * y[k]; gq[r] = s * x[k] + c * y[k];
                        const double u = vp[r], w = vq[r];
                        vp[r] = c * u - s * w; vq[r] = s * u + c * w;
                    }
                }
            }
            __syncthreads();
        }
        if (dbg && threadIdx.x == 0 && sweep < 40) dbg[sweep] = (double)*s_rot + 1e-3 * nl;
        if (*s_rot == 0) break;
        __syncthreads();
    }
}
// ---- two-sided (classical) Jacobi, all rotations of a round at once -----------------------------------------------------------
// A (n x n symmetric, leading dimension lda — odd, so that the row phase walks distinct LDS banks) and V (n x n, starts as I)
// live in LDS.  A round of the round-robin schedule holds n/2 disjoint pairs, one HALF-wave each: (0) c, s from a_pp, a_qq, a_pq —
// three scalars, no dot products, which is what the one-sided method spends its time on; (1) its column pair of A and V;
// barrier; (2) its row pair of A; barrier.  A pair is left alone
// when |a_pq| <= tol sqrt(|a_pp a_qq|) (the scaled criterion: small eigenvalues of a positive semi-definite matrix come out with
// RELATIVE accuracy, which the 1e-8 threshold needs while |A'| ~ 1e7), or when the rotation would be the identity in fp64,
// |a_pq| <= macheps |a_qq - a_pp| (sin(theta) below half an ulp: what terminates the pairs of a large and a numerically null
// diagonal entry, whose a_pq is rounding noise that never passes the scaled test), or when |a_pq| <= delta_s, an ABSOLUTE floor for
// the entries between numerically null directions (true zeros that come out of cancellations among ~1e6 terms: noise of a few
// macheps |A|, which no criterion relative to that noise can ever call converged).  Round 2 fixed that floor at 8 macheps |A|_F,
// ~ 9e-9 for these matrices — the size of the 1e-8 threshold itself: eigenvalues within a factor of two of it were classified by
// their rounding (measured on the far-landmark windows of tests/golden/marg_exact.npz).  Now the floor starts at
// delta = macheps |A|_F / 16 and doubles with every sweep from the 15th on, so a matrix whose noise is lower converges to that
// accuracy and the others still terminate (9e-9 is reached in sweep 22).
// On exit the eigenvalues are the diagonal of A, the eigenvectors the columns of V.
// 1 / sqrt(x) for a normal positive x: v_rsq_f64 (~ single precision) + two Newton steps; the library routine's range handling
// is not needed here and costs a third of a round
MDEV double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}
constexpr double JACOBI2_ANG = 1.2e-16;      // |sin(theta)| ~ |a_pq| / |a_qq - a_pp| below this: the rotation is the identity in fp64
template <int MAXIT>      // pairs a half-wave owns per round: ceil((n / 2) / half-waves) — 1024 threads: 2 up to n = 128, 3 up to 192; 256 threads, n <= 16: 1
MDEV void jacobi2_lds(double* A, int lda, double* V, int n, double tol, double delta0, int max_sweeps, int* s_rot, double* dbg = nullptr) {
    if (n < 2) return;
    const int lane = threadIdx.x & 63, hl = lane & 31, nh = (blockDim.x >> 6) * 2, half = (threadIdx.x >> 6) * 2 + (lane >> 5);
    const double tol2 = tol * tol;
    __shared__ int s_perm[160], s_live;
    __shared__ unsigned char s_lf[160];
    double delta = delta0;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (sweep >= 15) delta *= 2.0;
        // Columns that are already decoupled — every off-diagonal entry at the floor: the exactly zero velocity columns of keyframes
        // no selected factor constrains, and, sweep after sweep, whatever has converged — leave the schedule; a round costs two
        // workgroup barriers however few of its pairs rotate, so the sweeps shrink with the live set (60 -> 39 -> ... at configs[3])
        if (threadIdx.x == 0) s_live = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const double ajj = A[(size_t)j * lda + j];
            bool live = false;
            for (int k = 0; k < n; ++k) {
                const double v = A[(size_t)k * lda + j], akk = A[(size_t)k * lda + k];
                if (k != j && v != 0.0 && fabs(v) > delta + JACOBI2_ANG * fabs(akk - ajj) && v * v > tol2 * fabs(ajj * akk)) { live = true; break; }
            }
            s_lf[j] = live ? 1 : 0;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) {      // live columns in index order (deterministic)
            if (!s_lf[j]) continue;
            int rank = 0;
            for (int k = 0; k < j; ++k) rank += s_lf[k];
            s_perm[rank] = j;
            atomicAdd(&s_live, 1);
        }
        __syncthreads();
        const int nl = s_live;
        if (dbg && threadIdx.x == 0 && sweep < 40) dbg[sweep] = (double)nl;
        if (threadIdx.x == 0) *s_rot = nl;
        if (nl < 2) break;
        const int npad = (nl & 1) ? nl + 1 : nl, mm = npad - 1, npairs = npad / 2;
        // this half-wave's pairs follow a recurrence over the rounds: p and q of pair i advance by one modulo mm (pair 0 keeps index mm)
        int pr[MAXIT], qr[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) { const int i = it * nh + half; pr[it] = (i == 0) ? mm : i % mm; qr[it] = (i == 0) ? 0 : (mm - i % mm) % mm; }
        for (int round = 0; round < mm; ++round) {
            int pp[MAXIT], qq[MAXIT];
            double cc[MAXIT], ss[MAXIT];
            // (0) + (1): a half-wave per pair takes c, s from a_pp, a_qq, a_pq (every lane the same three reads) and rotates its two
            // columns of A and of V at once: no other pair's rotation touches these columns or those three entries in this phase
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int i = it * nh + half;
                int p = pr[it], q = qr[it];
                if (p > q) { const int t = p; p = q; q = t; }
                if (i >= npairs || q >= nl) q = -1;      // no such pair / padding index: bye
                else { p = s_perm[p]; q = s_perm[q]; }
                double c = 1.0, sn = 0.0;
                if (q >= 0) {
                    const double app = A[(size_t)p * lda + p], aqq = A[(size_t)q * lda + q], apq = A[(size_t)q * lda + p];
                    if (apq != 0.0 && apq * apq > tol2 * fabs(app * aqq) && fabs(apq) > delta + JACOBI2_ANG * fabs(aqq - app)) {
                        // tan(theta) = sign(a b) |b| / (|a| + h), a = a_qq - a_pp, b = 2 a_pq, h = hypot(a, b)
                        //   =>  c = (|a| + h) r,  s = sign(a b) |b| r,  r = 1 / sqrt(2 h (h + |a|)):  two reciprocal square roots, no division
                        const double a = aqq - app, bb = 2.0 * apq, fa = fabs(a);
                        const double h2 = fma(a, a, bb * bb);
                        const double h = h2 * rsqrt_nr(h2);
                        const double r = rsqrt_nr(2.0 * h * (h + fa));
                        c = (fa + h) * r;
                        sn = ((a * bb >= 0.0) ? fabs(bb) : -fabs(bb)) * r;
                    }
                }
                pp[it] = p; qq[it] = q; cc[it] = c; ss[it] = sn;
                if (q >= 0 && sn != 0.0) {
                    double* ap = A + (size_t)p * lda; double* aq = A + (size_t)q * lda;
                    double* vp = V + (size_t)p * n; double* vq = V + (size_t)q * n;
                    for (int r = hl; r < n; r += 32) {
                        const double x = ap[r], y = aq[r], u = vp[r], w = vq[r];
                        ap[r] = c * x - sn * y; aq[r] = sn * x + c * y;
                        vp[r] = c * u - sn * w; vq[r] = sn * u + c * w;
                    }
                }
                // next round's pair: both members advance by one modulo mm (pair 0: only its second member)
                if (i == 0) { qr[it] = (qr[it] + 1 == mm) ? 0 : qr[it] + 1; }
                else { pr[it] = (pr[it] + 1 == mm) ? 0 : pr[it] + 1; qr[it] = (qr[it] + 1 == mm) ? 0 : qr[it] + 1; }
            }
            __syncthreads();
            // (2) the same half-waves rotate their two ROWS of A (every column was touched by some pair in (1): hence the barrier)
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int p = pp[it], q = qq[it];
                const double c = cc[it], sn = ss[it];
                if (q >= 0 && sn != 0.0) {
                    for (int k = hl; k < n; k += 32) {
                        double* col = A + (size_t)k * lda;
                        const double x = col[p], y = col[q];
                        col[p] = c * x - sn * y; col[q] = sn * x + c * y;
                    }
                }
            }
            __syncthreads();
        }
    }
}
// ---- the same method with ONE barrier per round (the kept block of k_marg_finish, A' and V in LDS, n <= 128) ------------------------
// jacobi2_lds spends a round as  c, s -> column pairs | barrier | row pairs | barrier.  Here the row phase of round t rides in the column
// phase of round t + 1: a column is owned by exactly one pair per round, so the owner of (p', q') in round t + 1 first applies round t's
// row rotations to ITS two columns (every row r has one partner row and one (c, s): per-row tables prt / ca / cb, double-buffered by round
// parity), now holds the finished columns in registers, takes a_p'p', a_q'q', a_p'q' from them (lane broadcasts, no LDS round trip),
// rotates the two columns and writes them back in place.  The arithmetic per entry is jacobi2_lds' (c x - s y, s x + c y, same order of
// the two phases per entry); what changes is one barrier and one dependent LDS round trip less per round: 3.7 k -> 3.3 k cycles
// at the configs[3] shape (stamps, PLBA_JSTAMPS: tables + columns 0.85 k, c and s 0.65 k, write-back 0.95 k, barrier 0.85 k — the round is bound by
// the LDS instruction rate of 11 busy waves on one CU, not by the barriers: k_marg_finish 696 -> 610 us).  Columns that left the schedule are not owned by anybody, so their row rotations are not
// applied; their entries against live indices are restored from the (rotated) rows of the live columns by symmetry before the
// next sweep's liveness test reads them.  A sweep ends with the pending row rotations applied to every live column.
struct Jac2sScratch { double2 cs[2][128]; int prt[2][128]; int perm[128]; int live; unsigned char lf[128]; int pad_; };      // (in the kernel's dynamic LDS, behind A' and V)
template <int MAXIT, int NIT>      // pairs per half-wave and round | rows per lane: ceil(n / 32)
MDEV void jacobi2s_lds(double* A, int lda, double* V, int n, double tol, double delta0, int max_sweeps, int* s_rot, Jac2sScratch& sc, double* dbg = nullptr) {
    if (n < 2) return;
    const int lane = threadIdx.x & 63, hl = lane & 31, hb = lane & 32, nh = (blockDim.x >> 6) * 2, half = (threadIdx.x >> 6) * 2 + (lane >> 5);
    const double tol2 = tol * tol;
    int* const s_perm = sc.perm; int& s_live = sc.live; int (*const s_prt)[128] = sc.prt;
    unsigned char* const s_lf = sc.lf;
    double2 (*const s_cs)[128] = sc.cs;      // new[r] = cs.x col[r] + cs.y col[prt[r]]
    double delta = delta0;
    // finished (row-rotated) entries of a column: rows hl + 32 u
    // (the row tables are read once per round and lane — both columns of a pair use them: the loop is bound by the LDS instruction rate,
    // 16 waves on one CU's LDS; stamps: 1.1 k of a round's 3.7 k cycles were this load with the tables read per column)
    int t_pr[NIT]; double2 t_cs[NIT];
    auto load_tab = [&](const int tb) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) { const int r = hl + 32 * u; t_pr[u] = r < n ? s_prt[tb][r] : 0; t_cs[u] = r < n ? s_cs[tb][r] : make_double2(1.0, 0.0); }
    };
    auto load_col = [&](const double* col, const bool pending, double* out) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int r = hl + 32 * u;
            double v = 0.0;
            if (r < n) {
                v = col[r];
                if (pending) v = t_cs[u].x * v + t_cs[u].y * col[t_pr[u]];
            }
            out[u] = v;
        }
    };
    auto bcast = [&](const double* a, const int idx) {      // entry `idx` of a column held as above, to every lane of the half-wave
        double v = a[0];
#pragma unroll
        for (int u = 1; u < NIT; ++u) v = ((idx >> 5) == u) ? a[u] : v;
        return __shfl(v, (idx & 31) | hb);
    };
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (sweep >= 15) delta *= 2.0;
        if (threadIdx.x == 0) s_live = 0;
        if (sweep > 0) {      // columns out of the previous sweep's schedule: entries against the other indices from the rows (symmetry)
            for (int t = threadIdx.x; t < n * n; t += blockDim.x) { const int j = t / n, k = t % n; if (!s_lf[j] && s_lf[k]) A[(size_t)j * lda + k] = A[(size_t)k * lda + j]; }
        }
        __syncthreads();
        unsigned char mylive[(128 + 1023) / 1024 + 1];
        {
            int w = 0;
            for (int j = threadIdx.x; j < n; j += blockDim.x, ++w) {
                const double ajj = A[(size_t)j * lda + j];
                bool live = false;
                for (int k = 0; k < n; ++k) {
                    const double v = A[(size_t)k * lda + j], akk = A[(size_t)k * lda + k];
                    if (k != j && v != 0.0 && fabs(v) > delta + JACOBI2_ANG * fabs(akk - ajj) && v * v > tol2 * fabs(ajj * akk)) { live = true; break; }
                }
                mylive[w] = live ? 1 : 0;
            }
        }
        __syncthreads();      // (the symmetrisation above read the previous flags)
        {
            int w = 0;
            for (int j = threadIdx.x; j < n; j += blockDim.x, ++w) s_lf[j] = mylive[w];
        }
        for (int r = threadIdx.x; r < n; r += blockDim.x) { s_prt[0][r] = r; s_prt[1][r] = r; s_cs[0][r] = make_double2(1.0, 0.0); s_cs[1][r] = make_double2(1.0, 0.0); }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) {      // live columns in index order (deterministic; ordering them by their diagonal entries was tried: more sweeps)
            if (!s_lf[j]) continue;
            int rank = 0;
            for (int k = 0; k < j; ++k) rank += s_lf[k];
            s_perm[rank] = j;
            atomicAdd(&s_live, 1);
        }
        __syncthreads();
        const int nl = s_live;
        if (dbg && threadIdx.x == 0 && sweep < 40) dbg[sweep] = (double)nl;
        if (threadIdx.x == 0) *s_rot = nl;
        if (nl < 2) break;
        const int npad = (nl & 1) ? nl + 1 : nl, mm = npad - 1, npairs = npad / 2;
        int pr[MAXIT], qr[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) { const int i = it * nh + half; pr[it] = (i == 0) ? mm : i % mm; qr[it] = (i == 0) ? 0 : (mm - i % mm) % mm; }
        for (int round = 0; round < mm; ++round) {
            const bool pending = round > 0;
            const int tb_prev = (round + 1) & 1, tb = round & 1;
#ifdef PLBA_JSTAMPS
            const bool stampme = dbg && threadIdx.x == 64 && sweep == 1 && round == 5;
            unsigned long long js[6] = {0, 0, 0, 0, 0, 0};
            if (stampme) js[0] = __builtin_readcyclecounter();
#define JST(k) do { if (stampme) js[k] = __builtin_readcyclecounter(); } while (0)
#else
#define JST(k) do {} while (0)
#endif
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int i = it * nh + half;
                int p = pr[it], q = qr[it];
                if (p > q) { const int t = p; p = q; q = t; }
                const bool have = i < npairs;
                const bool bye = have && q >= nl;      // the padding index: this round's odd column out (its row rotations are still due)
                if (have) { p = s_perm[p]; q = bye ? -1 : s_perm[q]; }
                if (have) {
                    double* ap_ = A + (size_t)p * lda;
                    double* aq_ = bye ? ap_ : A + (size_t)q * lda;
                    double xp[NIT], xq[NIT];
                    if (pending && it == 0) load_tab(tb_prev);
                    load_col(ap_, pending, xp);
                    if (!bye) load_col(aq_, pending, xq);
                    JST(1);
                    double c = 1.0, sn = 0.0;
                    if (!bye) {
                        const double app = bcast(xp, p), aqq = bcast(xq, q), apq = bcast(xq, p);      // (a_pq: column q, row p — as jacobi2_lds reads it)
                        if (apq != 0.0 && apq * apq > tol2 * fabs(app * aqq) && fabs(apq) > delta + JACOBI2_ANG * fabs(aqq - app)) {
                            const double a = aqq - app, bb = 2.0 * apq, fa = fabs(a);
                            const double h2 = fma(a, a, bb * bb);
                            const double h = h2 * rsqrt_nr(h2);
                            const double r = rsqrt_nr(2.0 * h * (h + fa));
                            c = (fa + h) * r;
                            sn = ((a * bb >= 0.0) ? fabs(bb) : -fabs(bb)) * r;
                        }
                    }
                    JST(2);
                    if (hl == 0) {      // this round's row rotation of rows p, q, for the columns' next owners
                        const bool rot = sn != 0.0;
                        s_prt[tb][p] = rot ? q : p; s_cs[tb][p] = make_double2(c, -sn);
                        if (!bye) { s_prt[tb][q] = rot ? p : q; s_cs[tb][q] = make_double2(c, sn); }
                    }
                    if (sn != 0.0) {
                        double* vp = V + (size_t)p * n; double* vq = V + (size_t)q * n;      // (asking for V's columns before the c, s arithmetic was tried: no gain)
#pragma unroll
                        for (int u = 0; u < NIT; ++u) {
                            const int r = hl + 32 * u;
                            if (r < n) {
                                const double x = xp[u], y = xq[u], uu = vp[r], w = vq[r];
                                ap_[r] = c * x - sn * y; aq_[r] = sn * x + c * y;
                                vp[r] = c * uu - sn * w; vq[r] = sn * uu + c * w;
                            }
                        }
                    } else if (pending) {
#pragma unroll
                        for (int u = 0; u < NIT; ++u) { const int r = hl + 32 * u; if (r < n) { ap_[r] = xp[u]; if (!bye) aq_[r] = xq[u]; } }
                    }
                }
                if (i == 0) { qr[it] = (qr[it] + 1 == mm) ? 0 : qr[it] + 1; }
                else { pr[it] = (pr[it] + 1 == mm) ? 0 : pr[it] + 1; qr[it] = (qr[it] + 1 == mm) ? 0 : qr[it] + 1; }
            }
            JST(3);
            __syncthreads();
            JST(4);
#ifdef PLBA_JSTAMPS
            if (stampme) for (int k = 0; k < 5; ++k) dbg[48 + k] = (double)(js[k] - js[0]);
#endif
        }
        // the last round's row rotations, on every live column
        {
            const int tb_last = (mm - 1) & 1;
            for (int idx = half; idx < nl; idx += nh) {
                double* col = A + (size_t)s_perm[idx] * lda;
                double x[NIT];
                if (idx == half) load_tab(tb_last);
                load_col(col, true, x);
#pragma unroll
                for (int u = 0; u < NIT; ++u) { const int r = hl + 32 * u; if (r < n) col[r] = x[u]; }
            }
        }
        __syncthreads();
    }
}
// ---- pre-rotation of the kept block: Householder tridiagonalisation + implicit QL (round 5) -------------------------------------------
// The Jacobi above needs ~13 sweeps from a cold start — 7 of them over every live column before anything converges — and a round costs what
// the CU's LDS can issue: 0.6 ms at 45 live dims, 2.0 ms at the 75 of the reference's own 12-keyframe window, more than the bundle adjustment
// it follows.  The reference's solver class (Eigen::SelfAdjointEigenSolver, IMU/marginalization.cpp:352,364: tridiagonalisation + implicit
// QR) gets within n macheps |A| of the answer in O(n^3) flops with a short dependent chain per step; that is not the RELATIVE accuracy the
// 1e-8 threshold needs while |A'| ~ 1e7, so it is used as what it is good at: V0 with V0^T A' V0 diagonal to ~1e-9 absolute.
// The Jacobi then starts from (V0^T A' V0, V0) instead of (A', I): every column whose off-diagonal entries already pass its scaled test
// is out of the schedule at once, what remains are the numerically null / tiny directions (20 - 35 columns, 2 - 4 sweeps), and the
// result is the Jacobi's own — same criteria, same accuracy class, V = V0 V_jacobi orthogonal as a product of reflections and rotations.
//   stage 1  A' = Q T Q^T, T tridiagonal (d, e), Q = H_0 H_1 ... accumulated in V: n - 2 steps, every one a matrix-vector product and a
//            rank-2 update of the trailing block plus the same on V, the whole workgroup on each (three barriers per step)
//   stage 2  T = Z D Z^T by QL with implicit shifts (EISPACK tql2 / Numerical Recipes tqli, restated): one lane runs the chase — ~17
//            dependent fp64 operations per rotation — and leaves (c, s) of the pass in LDS; all rows of V take them afterwards, a thread per row
//   stage 3  G <- V^T A' V with A' streamed from global memory once more (the tridiagonalisation consumed the LDS copy)
struct TriScratch { double d[128], e[128], c[128], s[128], u[128]; int m[2], go[2]; };      // (m, go: by pass parity — one barrier per pass)
static_assert(sizeof(TriScratch) <= sizeof(Jac2sScratch), "the pre-rotation's scratch overlays the one-barrier Jacobi's row tables");
MDEV void tridiag_householder(double* G, const int lda, double* V, const int n, TriScratch& S) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, T = blockDim.x, grp = tid >> 3, part = tid & 7, ngrp = T >> 3;
    for (int k = 0; k + 2 < n; ++k) {
        const int mm = n - k - 1;
        const double* x = G + (size_t)k * lda + k + 1;      // column k below the diagonal (contiguous)
        // (1) every wave forms the reflector's scalars redundantly (no barrier for them); wave 0 publishes v
        double st = 0.0;
        for (int j = lane; j < mm; j += 64) if (j > 0) st += x[j] * x[j];
        st = wave_sum_dpp(st);
        const double x0 = x[0];
        const bool reflect = st > 0.0;      // (nothing below the sub-diagonal: H = I)
        const double alpha = reflect ? -copysign(sqrt(fma(x0, x0, st)), x0) : x0;
        const double v0 = x0 - alpha;
        const double beta = reflect ? 2.0 / fma(v0, v0, st) : 0.0;
        if (wv == 0) for (int j = lane; j < mm; j += 64) S.u[j] = j == 0 ? v0 : x[j];
        if (tid == 0) { S.d[k] = G[(size_t)k * lda + k]; S.e[k] = alpha; }
        __syncthreads();
        if (reflect) {
            // (2) p = beta G22 u -> S.c, t_r = sum_j V[r][k + 1 + j] u_j -> S.s: eight lanes per entry, fixed order
            for (int i = grp; i < mm + n; i += ngrp) {
                double acc = 0.0;
                if (i < mm) { const double* col = G + (size_t)(k + 1 + i) * lda + k + 1; for (int j = part; j < mm; j += 8) acc = fma(col[j], S.u[j], acc); }      // (row i of the symmetric block = its column i)
                else { const int r = i - mm; for (int j = part; j < mm; j += 8) acc = fma(V[(size_t)(k + 1 + j) * n + r], S.u[j], acc); }
                acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
                if (part == 0) { if (i < mm) S.c[i] = beta * acc; else S.s[i - mm] = beta * acc; }
            }
            __syncthreads();
            // (3) K = beta / 2 u^T p (every wave for itself), w = p - K u;  (4) G22 -= u w^T + w u^T,  V[:, k + 1 ..] -= (beta t) u^T
            double kk = 0.0;
            for (int j = lane; j < mm; j += 64) kk = fma(S.u[j], S.c[j], kk);
            kk = 0.5 * beta * wave_sum_dpp(kk);
            // (a 32 x 32 thread tile walks both updates: an index pair from one counter costs two integer divisions by a run-time
            // divisor per entry — they were half of a step's 8.4 k cycles at 75 dims)
            const int tx = tid & 31, ty = tid >> 5;
            for (int j = ty; j < mm; j += 32) {
                const double uj = S.u[j], wj = S.c[j] - kk * uj;
                double* col = G + (size_t)(k + 1 + j) * lda + k + 1;
                for (int i = tx; i < mm; i += 32) { const double ui = S.u[i]; col[i] -= ui * wj + (S.c[i] - kk * ui) * uj; }      // entry (row i, column j) of the trailing block
                double* vcol = V + (size_t)(k + 1 + j) * n;
                for (int r = tx; r < n; r += 32) vcol[r] -= S.s[r] * uj;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (n >= 2) { S.d[n - 2] = G[(size_t)(n - 2) * lda + n - 2]; S.e[n - 2] = G[(size_t)(n - 2) * lda + n - 1]; }
        S.d[n - 1] = G[(size_t)(n - 1) * lda + n - 1]; S.e[n - 1] = 0.0;
    }
    __syncthreads();
}
// e[i] couples i and i + 1.  Returns (uniform) false when an eigenvalue did not converge in 40 passes: the caller then starts the Jacobi
// from wherever V stands — it is orthogonal all the same.
// The chase runs on wave 0 with d and e held ACROSS ITS LANES (lane i: entries i and 64 + i) and moved by v_readlane / v_writelane: the first
// form kept them in LDS and one lane walked them — every rotation then paid several dependent LDS round trips (a store to e[i + 1] has to land
// before e[i - 1] may be read: the compiler cannot know better) and the stage cost 0.5 ms at 45 dims, more than the sweeps it saves.
MDEV double lane_get(const double v, const int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); }
MDEV void lane_put(double& v, const int l, const double x) { v = ((int)(threadIdx.x & 63) == l) ? x : v; }      // (x uniform: a compare and two selects)
template <bool HI>
struct LaneVec {      // up to 64 (HI: 128) doubles over the 64 lanes of a wave (lane i: entries i and 64 + i); index uniform.  No branches: a chase
    double lo, hi;    // step built from `i < 64 ? ... : ...` branches was 120 executed instructions and 25 jumps, 890 cycles per rotation
    MDEV double get(const int i) const { const double a = lane_get(lo, i & 63); if (!HI) return a; const double b = lane_get(hi, i & 63); return i < 64 ? a : b; }
    MDEV void put(const int i, const double x) { const int l = threadIdx.x & 63; lo = (l == i) ? x : lo; if (HI) hi = (l + 64 == i) ? x : hi; }
    // one half only, for the stretches of the chase that stay inside it (HALF = 1: index i is entry i of the upper half, i.e. 64 + i)
    template <int HALF> MDEV double get_h(const int i) const { return lane_get(HALF ? hi : lo, i); }
    template <int HALF> MDEV void put_h(const int i, const double x) { const int l = threadIdx.x & 63; if (HALF) hi = (l == i) ? x : hi; else lo = (l == i) ? x : lo; }
};
template <bool HI>
MDEV bool tridiag_ql(double* V, const int n, TriScratch& S, double* dbg = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    bool ok = true;
    int n_pass = 0, n_rot = 0;
    LaneVec<HI> d, e;
    d.lo = lane < n ? S.d[lane] : 0.0; d.hi = (HI && 64 + lane < n) ? S.d[64 + lane] : 0.0;
    e.lo = lane < n ? S.e[lane] : 0.0; e.hi = (HI && 64 + lane < n) ? S.e[64 + lane] : 0.0;
    __syncthreads();      // (S.d / S.e are free from here on: they become the second (c, s) buffer)
    // The chase of pass k + 1 (wave 0) runs while the rows of V take the rotations of pass k (threads 64 .. 64 + n): two (c, s) buffers,
    // one barrier per pass.
    int buf = 0, pend_m = -1, pend_lo = 0, pend_buf = 0, par = 0;
    const unsigned long long q0 = __builtin_readcyclecounter();
    for (int l = 0; l <= n; ++l) {      // (l == n: nothing to chase, the last pending pass is applied)
        for (int iter = 0;; ++iter) {
            double* const cb = buf ? S.d : S.c; double* const sb = buf ? S.e : S.s;
            if (tid < 64) {      // wave 0, every lane the same scalars
                int m = l, go = 0;
                if (l < n) {
                    // the first negligible sub-diagonal entry at or after l: every lane tests its own two, one ballot each
                    const double dn_lo = __shfl_down(d.lo, 1), dn_hi = __shfl_down(d.hi, 1);
                    const double d64 = HI ? lane_get(d.hi, 0) : 0.0;
                    const double nx_lo = lane == 63 ? d64 : dn_lo;      // d[i + 1] for i = lane
                    const bool sm_lo = fabs(e.lo) <= 1.1102230246251565e-16 * (fabs(d.lo) + fabs(nx_lo)), sm_hi = HI && fabs(e.hi) <= 1.1102230246251565e-16 * (fabs(d.hi) + fabs(dn_hi));
                    const unsigned long long b_lo = __ballot(sm_lo && lane >= l && lane < n - 1), b_hi = HI ? __ballot(sm_hi && 64 + lane >= l && 64 + lane < n - 1) : 0ull;
                    // (readfirstlane: control flow and lane indices of the chase must be SCALAR for the compiler too)
                    m = __builtin_amdgcn_readfirstlane(b_lo ? __builtin_ctzll(b_lo) : b_hi ? 64 + __builtin_ctzll(b_hi) : n - 1);
                    go = (m != l) ? (iter < 40 ? 1 : 2) : 0;
                }
                if (go == 1) {
                    const double dl = d.get(l), el = e.get(l);
                    double g = (d.get(l + 1) - dl) / (2.0 * el);
                    double r = sqrt(fma(g, g, 1.0));
                    g = d.get(m) - dl + el / (g + copysign(r, g));
                    double sn = 1.0, c = 1.0, p = 0.0;
                    double d_ip1 = d.get(m);
                    // (no early exit: f^2 + g^2 = 0 needs an underflow — the pass starts with f = e[m - 1], which the convergence test has
                    // just found non-negligible, and s = f / r carries on — and a branch in here costs the loop a dozen register copies)
                    // one rotation; GH / PH: the half of the lane vectors that holds entry i / entry i + 1 (the chase runs in up to three stretches
                    // — both in the upper half, the crossing at i = 63, both in the lower half — so that no step touches both halves:
                    // 506 -> ~320 cycles per rotation at 75 dims)
                    auto rotation = [&](auto GH, auto PH, const int i) {
                        constexpr int gh = decltype(GH)::value, ph = decltype(PH)::value;
                        const double ei = e.template get_h<gh>(i - 64 * gh), di = d.template get_h<gh>(i - 64 * gh);
                        const double f = sn * ei, bq = c * ei;
                        const double h2 = fma(f, f, g * g);
                        const double ri = h2 > 0.0 ? rsqrt_nr(h2) : 0.0;
                        r = h2 * ri;
                        e.template put_h<ph>(i + 1 - 64 * ph, r);
                        sn = f * ri; c = g * ri;
                        g = d_ip1 - p;
                        r = fma(di - g, sn, 2.0 * c * bq);
                        p = sn * r;
                        d.template put_h<ph>(i + 1 - 64 * ph, g + p);
                        g = fma(c, r, -bq);
                        if (lane == 0) { cb[i] = c; sb[i] = sn; }      // rotation of columns i, i + 1 of V, for the rows' pass
                        d_ip1 = di;
                    };
                    using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
                    int i = m - 1;
                    if (HI) {
                        for (; i >= l && i >= 64; --i) rotation(H1{}, H1{}, i);
                        if (i >= l && i == 63) { rotation(H0{}, H1{}, i); --i; }
                    }
                    for (; i >= l; --i) rotation(H0{}, H0{}, i);
                    d.put(l, d_ip1 - p); e.put(l, g); e.put(m, 0.0);
                }
                if (lane == 0) { S.m[par] = m; S.go[par] = go; }
            } else if (pend_m >= 0 && tid - 64 < n) {      // a row of V takes the PREVIOUS pass's rotations, in the order they were made (m - 1 down to lo)
                const int row = tid - 64, m = pend_m, lo = pend_lo;
                const double* const pc = pend_buf ? S.d : S.c; const double* const ps = pend_buf ? S.e : S.s;
                double hi = V[(size_t)m * n + row];      // V[row][i + 1], carried
                int i = m - 1;
                for (; i - 3 >= lo; i -= 4) {      // four rotations' operands asked for at once: the loads do not depend on the carried value
                    double c4[4], s4[4], z4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { c4[q] = pc[i - q]; s4[q] = ps[i - q]; z4[q] = V[(size_t)(i - q) * n + row]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { V[(size_t)(i - q + 1) * n + row] = fma(s4[q], z4[q], c4[q] * hi); hi = fma(c4[q], z4[q], -s4[q] * hi); }
                }
                for (; i >= lo; --i) {
                    const double c = pc[i], sn = ps[i], z = V[(size_t)i * n + row];
                    V[(size_t)(i + 1) * n + row] = fma(sn, z, c * hi);
                    hi = fma(c, z, -sn * hi);
                }
                V[(size_t)lo * n + row] = hi;
            }
            __syncthreads();
            const int go = S.go[par], m = S.m[par];      // (the next pass writes the other pair)
            par ^= 1;
            pend_m = go == 1 ? m : -1; pend_lo = l; pend_buf = buf;
            if (go == 1) { buf ^= 1; ++n_pass; n_rot += m - l; }
            if (go == 2) ok = false;
            if (go != 1) break;
        }
    }
    if (dbg && tid == 0) { dbg[59] = (double)(__builtin_readcyclecounter() - q0); dbg[60] = 0.0; dbg[61] = n_pass; dbg[62] = n_rot; }
    return ok;
}
// stage 3: G <- V^T A' V (A' = the live part of the kept block, symmetrised, read again from global memory), exactly symmetric on exit.
// n <= 100 with 1024 threads: at most ten entries per thread, held in registers across the barrier that lets the result replace its operand.
MDEV void rotate_kept_block(double* G, const int lda, const double* V, const int n, const double* A, const int pos, const int m0, const int* s_live) {
    const int tid = threadIdx.x, T = blockDim.x;
    for (int t = tid; t < n * n; t += T) {
        const int c = t / n, r = t % n, gc = m0 + s_live[c], gr = m0 + s_live[r];
        G[(size_t)c * lda + r] = 0.5 * (A[(size_t)gr * pos + gc] + A[(size_t)gc * pos + gr]);
    }
    __syncthreads();
    double acc[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {      // W = A' V: W[i][j] = sum_k A'[k][i] V[k][j] (A' symmetric: column i, contiguous)
        const int t = tid + q * T;
        double a = 0.0;
        if (t < n * n) { const int j = t / n, i = t % n; const double* ai = G + (size_t)i * lda; const double* vj = V + (size_t)j * n; for (int k = 0; k < n; ++k) a = fma(ai[k], vj[k], a); }
        acc[q] = a;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 10; ++q) { const int t = tid + q * T; if (t < n * n) G[(size_t)(t / n) * lda + t % n] = acc[q]; }      // G holds W (column j contiguous)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 10; ++q) {      // G' = V^T W, lower triangle computed, mirrored on the way out
        const int t = tid + q * T;
        double a = 0.0;
        if (t < n * n) { const int j = t / n, i = t % n; if (i >= j) { const double* vi = V + (size_t)i * n; const double* wj = G + (size_t)j * lda; for (int k = 0; k < n; ++k) a = fma(vi[k], wj[k], a); } }
        acc[q] = a;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const int t = tid + q * T;
        if (t < n * n) { const int j = t / n, i = t % n; if (i >= j) { G[(size_t)j * lda + i] = acc[q]; G[(size_t)i * lda + j] = acc[q]; } }
    }
    __syncthreads();
}
MDEV double jacobi2_delta(const double* A, int lda, int n, double* s_part) {      // macheps |A|_F / 16  (call with the whole workgroup)
    double f = 0.0;
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) { const double v = A[(size_t)(t / n) * lda + t % n]; f += v * v; }
    f = wave_sum(f);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = f;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
    __syncthreads();
    return 1.1102230246251565e-16 / 16.0 * sqrt(tot);
}
// |g_p . g_q| <= JACOBI_TOL |g_p| |g_q| counts as orthogonal: an n-term fp64 dot product carries ~ n * 1.1e-16 of relative
// rounding, so 1e-15 (round 1) was never reached and every call ran all its sweeps; 1e-13 leaves eigenvalues good to ~1e-13
constexpr double JACOBI_TOL = 1e-13;
constexpr double JACOBI2_TOL = 1e-15;     // two-sided: |a_pq| against sqrt(|a_pp a_qq|), three exact scalars — no dot-product rounding to stay above
// 2 delta^2 of the comment above, from the Frobenius norm of the n x n matrix held in G (call with the whole workgroup)
MDEV double jacobi_noise2(const double* G, int n, double* s_part) {
    double f = 0.0;
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) f += G[t] * G[t];
    f = wave_sum(f);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = f;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
    __syncthreads();
    const double delta = 4.0 * n * 1.1102230246251565e-16;
    return 2.0 * delta * delta * tot;
}
constexpr int JLDS_MAX_N = 100;       // A' and V both in LDS: (100 * 101 + 100 * 100) doubles = 160,800 of the CU's 163,840 bytes
constexpr int JLDS_MAX_N2 = 140;      // A' alone in LDS (140 * 141 doubles = 157,920 bytes), V in global memory: the reference's 12-keyframe
                                      // window keeps up to 11 x 9 + 6 = 105 dims (src/mapHandler.cpp:6109-6188)
// eigen pseudo-inverse of the dropped keyframe block (<= 15 dims) of the system the landmarks have been eliminated from
__global__ __launch_bounds__(256) void k_pose_pinv(const double* A, int pos, int o, int sz, double eps, double* Pinv) {
    __shared__ double G[MAXB * MAXB], V[MAXB * MAXB], lam[MAXB];
    __shared__ int rot;
    for (int t = threadIdx.x; t < sz * sz; t += blockDim.x) {
        const int c = t / sz, r = t % sz;
        G[t] = 0.5 * (A[(size_t)(o + r) * pos + o + c] + A[(size_t)(o + c) * pos + o + r]);   // cpp:351
        V[t] = (r == c) ? 1.0 : 0.0;
    }
    __shared__ double s_part[16];
    __syncthreads();
    const double delta = jacobi2_delta(G, sz, sz, s_part);
    jacobi2_lds<1>(G, sz, V, sz, JACOBI2_TOL, delta, 60, &rot);
    __syncthreads();
    if ((int)threadIdx.x < sz) lam[threadIdx.x] = G[threadIdx.x * sz + threadIdx.x];
    __syncthreads();
    for (int t = threadIdx.x; t < sz * sz; t += blockDim.x) {
        const int i = t / sz, j = t % sz;
        double acc = 0.0;
        for (int k = 0; k < sz; ++k) if (lam[k] > eps) acc += V[k * sz + i] * V[k * sz + j] / lam[k];
        Pinv[i * MAXB + j] = acc;
    }
}
// the kept block: A' out, eigen square root J0 = sqrt(S) V^T (column-major), r0 = sqrt(S^-1) V^T b'   (cpp:364-372)
// outp = [Ar n*n | br n | J0 n*n | r0 n]
// Round 4: the decomposition runs on the LIVE part of A' only.  Columns that are exactly zero — the velocity / bias dims of keyframes no
// selected factor constrains: 15 of 60 at the configs[3] shape, 30 of 105 on the reference's own 12-keyframe window — are eigenvectors with
// eigenvalue 0 whatever the rest does (J0 rows and r0 entries 0); they already left the rotation schedule, but they still counted for
// the STORAGE: n = 105 > 100 put V into global memory and the reference's own window paid 3.2 ms per slide where the configs[3] shape
// paid 1.2.  The live columns are compacted into an nl x nl problem (75 x 75 there: both A' and V in LDS, 74 instead of 104 rounds per
// sweep); Vg is only used when even the live part exceeds the in-LDS limit.
__global__ __launch_bounds__(1024) void k_marg_finish(const double* A, const double* b, int pos, int m, int n, double eps, double* outp, double* Vg, double* dbg, int dyn_bytes, int prerotate) {
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    __shared__ int rot, s_nl;
    __shared__ int s_live[JLDS_MAX_N2];
    __shared__ unsigned char s_nz[JLDS_MAX_N2];
    double* Ar = outp; double* br = outp + (size_t)n * n; double* J0 = br + n; double* r0 = J0 + (size_t)n * n;
    for (int j = threadIdx.x; j < n; j += blockDim.x) s_nz[j] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) {
        const int c = t / n, r = t % n;
        const double v = 0.5 * (A[(size_t)(m + r) * pos + m + c] + A[(size_t)(m + c) * pos + m + r]);
        Ar[t] = v;
        J0[t] = 0.0;
        if (v != 0.0) { s_nz[c] = 1; s_nz[r] = 1; }      // (every writer stores the same value)
    }
    for (int t = threadIdx.x; t < n; t += blockDim.x) { br[t] = b[m + t]; r0[t] = 0.0; }
    __syncthreads();
    if (threadIdx.x == 0) { int k = 0; for (int j = 0; j < n; ++j) if (s_nz[j]) s_live[k++] = j; s_nl = k; }      // index order: deterministic
    __syncthreads();
    const int nl = s_nl, lda = nl | 1;
    const bool v_lds = nl <= JLDS_MAX_N;
    double* G = s_dyn;
    double* Vl = s_dyn + (size_t)nl * lda;      // only when v_lds
    double* V = v_lds ? Vl : Vg;
    for (int t = threadIdx.x; t < nl * nl; t += blockDim.x) {
        const int c = t / nl, r = t % nl, gc = m + s_live[c], gr = m + s_live[r];
        G[(size_t)c * lda + r] = 0.5 * (A[(size_t)gr * pos + gc] + A[(size_t)gc * pos + gr]);
        V[t] = (r == c) ? 1.0 : 0.0;
    }
    __shared__ double s_part[16];
    __syncthreads();
    const double delta = jacobi2_delta(G, lda, nl, s_part);
    if (dbg && threadIdx.x == 0) { for (int q = 0; q < 48; ++q) dbg[q] = -1.0; dbg[40] = delta; dbg[41] = (double)nl; }
    // two instances behind one uniform branch: with V in LDS its accesses stay ds_ instructions; in global memory (live part beyond 100) a
    // column pair written in one round is read by another half-wave of this workgroup after the barrier of that round
    // the one-barrier form when its row tables fit behind A' and V (always up to 96 live dims)
    const size_t used = ((size_t)nl * lda + (size_t)nl * nl) * sizeof(double);
    Jac2sScratch* sc = reinterpret_cast<Jac2sScratch*>(reinterpret_cast<char*>(s_dyn) + used);
    const bool one_barrier = v_lds && used + sizeof(Jac2sScratch) <= (size_t)dyn_bytes;
    if (one_barrier && nl >= 8 && prerotate) {      // (the scratch of the pre-rotation overlays the row tables, which the Jacobi initialises itself)
        TriScratch& ts = *reinterpret_cast<TriScratch*>(sc);
        const unsigned long long q0 = __builtin_readcyclecounter();
        tridiag_householder(G, lda, Vl, nl, ts);
        const unsigned long long q1 = __builtin_readcyclecounter();
        if (nl <= 64) tridiag_ql<false>(Vl, nl, ts, dbg); else tridiag_ql<true>(Vl, nl, ts, dbg);
        const unsigned long long q2 = __builtin_readcyclecounter();
        rotate_kept_block(G, lda, Vl, nl, A, pos, m, s_live);
        const unsigned long long q3 = __builtin_readcyclecounter();
        if (dbg && threadIdx.x == 0) { dbg[56] = (double)(q1 - q0); dbg[57] = (double)(q2 - q1); dbg[58] = (double)(q3 - q2); }
    }
    if (one_barrier && nl <= 64) jacobi2s_lds<1, 2>(G, lda, Vl, nl, JACOBI2_TOL, delta, 60, &rot, *sc, dbg);
    else if (one_barrier && nl <= 96) jacobi2s_lds<2, 3>(G, lda, Vl, nl, JACOBI2_TOL, delta, 60, &rot, *sc, dbg);      // (the reference's 12-keyframe window: 75 live dims)
    else if (one_barrier) jacobi2s_lds<2, 4>(G, lda, Vl, nl, JACOBI2_TOL, delta, 60, &rot, *sc, dbg);
    else if (v_lds) jacobi2_lds<2>(G, lda, Vl, nl, JACOBI2_TOL, delta, 60, &rot, dbg);
    else if (nl <= 128) jacobi2_lds<2>(G, lda, Vg, nl, JACOBI2_TOL, delta, 60, &rot, dbg);
    else jacobi2_lds<3>(G, lda, Vg, nl, JACOBI2_TOL, delta, 60, &rot, dbg);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wave; j < nl; j += nw) {
        const double* v = V + (size_t)j * nl;
        const double l = G[(size_t)j * lda + j];
        double vb = 0.0;
        for (int t = lane; t < nl; t += 64) vb += v[t] * b[m + s_live[t]];
        vb = wave_sum(vb);
        const double S = l > eps ? l : 0.0, Si = l > eps ? 1.0 / l : 0.0;
        const double ss = sqrt(S);
        for (int c = lane; c < nl; c += 64) J0[(size_t)s_live[c] * n + j] = ss * v[c];      // row j of J0 = sqrt(S_j) v_j^T, zero outside the live columns
        if (lane == 0) r0[j] = sqrt(Si) * vb;
    }
    if (threadIdx.x == 0) r0[n] = (nl < 2) ? 0.0 : (double)rot;      // columns still live when the last sweep began: >= 2 means the sweep limit was hit
}

// ---- multi-launch Jacobi for blocks that do not fit one workgroup's LDS --------------------------------------------------------
// (a) The dense eigen pseudo-inverse of the whole dropped block Amm (IMU/marginalization.cpp:351-353; m <= 474 at the call site).
//     Amm = Jm^T Jm with Jm the first m columns of the stacked Jacobian (R x m, column-major), so its eigen-decomposition is the
//     singular value decomposition of Jm: one-sided Jacobi (Hestenes) on the COLUMNS OF Jm — rotating column pairs until they are
//     orthogonal gives V (accumulated) and lambda_j = |g_j|^2 with high RELATIVE accuracy even where lambda_j / |Amm| ~ 1e-15,
//     which is where the 1e-8 threshold cuts (one-sided Jacobi on Amm itself only reaches n macheps |Amm| ~ 1e-7 absolute:
//     measured, the discarded subspace then differs from the oracle's).  One workgroup per pair, one launch per round.
//     A pair is left alone when |g_p . g_q| <= tol |g_p| |g_q|, or when one of the two columns has sunk to the rounding floor
//     (|g|^2 <= nfloor = (64 macheps)^2 |Jm|_F^2: a numerically null direction — every rank-deficient landmark block has some — whose
//     content is noise and can never pass the relative test).
// (b) Kept blocks beyond the in-LDS limit: two-sided Jacobi on A' itself, two launches per round (column pairs, then row pairs).
__global__ __launch_bounds__(1024) void k_fro_floor(const double* G, size_t count, double* nfloor) {
    __shared__ double s_part[16];
    double f = 0.0;
    for (size_t t = threadIdx.x; t < count; t += blockDim.x) f += G[t] * G[t];
    f = wave_sum(f);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
        const double c = 64.0 * 1.1102230246251565e-16;
        nfloor[0] = c * c * tot;      // one-sided: against |g|^2
        nfloor[1] = 1.1102230246251565e-16 / 16.0 * sqrt(tot);      // two-sided: the starting floor against |a_pq| (jacobi2_lds)
    }
}
MDEV void round_robin_pair(int i, int round, int npad, int& p, int& q) {
    const int mm = npad - 1;
    if (i == 0) { p = mm; q = round % mm; }
    else { p = (round + i) % mm; q = (round - i + mm) % mm; }
    if (p > q) { const int t = p; p = q; q = t; }
}
// G: rows x n column-major (leading dimension rows), V: n x n
__global__ __launch_bounds__(256) void k_hestenes_round(double* G, int rows, double* V, int n, int npad, int round, double tol2, const double* nfloor, int* rotated) {
    __shared__ double s4[3][4];
    int p, q;
    round_robin_pair(blockIdx.x, round, npad, p, q);
    if (q >= n) return;                                  // padding column: bye
    double* gp = G + (size_t)p * rows; double* gq = G + (size_t)q * rows;
    double a = 0.0, b = 0.0, g = 0.0;
    for (int t = threadIdx.x; t < rows; t += 256) { const double x = gp[t], y = gq[t]; a += x * x; b += y * y; g += x * y; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); g += __shfl_down(g, o, 64); }
    if ((threadIdx.x & 63) == 0) { s4[0][threadIdx.x >> 6] = a; s4[1][threadIdx.x >> 6] = b; s4[2][threadIdx.x >> 6] = g; }
    __syncthreads();
    a = (s4[0][0] + s4[0][1]) + (s4[0][2] + s4[0][3]);
    b = (s4[1][0] + s4[1][1]) + (s4[1][2] + s4[1][3]);
    g = (s4[2][0] + s4[2][1]) + (s4[2][2] + s4[2][3]);
    if (!(g * g > tol2 * a * b) || !(fmin(a, b) > nfloor[0]) || g == 0.0) return;
    if (threadIdx.x == 0) atomicAdd(rotated, 1);
    // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with zeta = (b - a) / (2 g), written without the division by g
    const double dd = b - a;
    const double t = ((dd * g >= 0.0) ? 2.0 : -2.0) * fabs(g) / (fabs(dd) + sqrt(dd * dd + 4.0 * g * g));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    for (int k = threadIdx.x; k < rows; k += 256) { const double x = gp[k], y = gq[k]; gp[k] = c * x - s * y; gq[k] = s * x + c * y; }
    double* vp = V + (size_t)p * n; double* vq = V + (size_t)q * n;
    for (int k = threadIdx.x; k < n; k += 256) { const double u = vp[k], w = vq[k]; vp[k] = c * u - s * w; vq[k] = s * u + c * w; }
}
// two-sided, phase 1: c, s of pair i from (a_pp, a_qq, a_pq) exactly as jacobi2_lds; rotate columns p, q of A (n x n, symmetric storage) and of V
__global__ __launch_bounds__(256) void k_j2_cols(double* A, double* V, int n, int npad, int round, double tol2, const double* nfloor, double fscale, double* cs, int* rotated) {
    int p, q;
    round_robin_pair(blockIdx.x, round, npad, p, q);
    double c = 1.0, sn = 0.0;
    if (q < n) {
        const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q], apq = A[(size_t)q * n + p];
        if (apq != 0.0 && apq * apq > tol2 * fabs(app * aqq) && fabs(apq) > fscale * nfloor[1] + JACOBI2_ANG * fabs(aqq - app)) {
            const double a = aqq - app, bb = 2.0 * apq, fa = fabs(a);
            const double h = sqrt(fma(a, a, bb * bb));
            const double r = 1.0 / sqrt(2.0 * h * (h + fa));
            c = (fa + h) * r;
            sn = ((a * bb >= 0.0) ? fabs(bb) : -fabs(bb)) * r;
        }
    }
    __syncthreads();      // every thread has read the three scalars before any column element changes
    if (threadIdx.x == 0) { cs[2 * blockIdx.x] = c; cs[2 * blockIdx.x + 1] = sn; if (sn != 0.0) atomicAdd(rotated, 1); }
    if (q >= n || sn == 0.0) return;
    double* ap = A + (size_t)p * n; double* aq = A + (size_t)q * n;
    double* vp = V + (size_t)p * n; double* vq = V + (size_t)q * n;
    for (int r = threadIdx.x; r < n; r += 256) {
        const double x = ap[r], y = aq[r], u = vp[r], w = vq[r];
        ap[r] = c * x - sn * y; aq[r] = sn * x + c * y;
        vp[r] = c * u - sn * w; vq[r] = sn * u + c * w;
    }
}
// phase 2: rows p, q of A with the same c, s
__global__ __launch_bounds__(256) void k_j2_rows(double* A, int n, int npad, int round, const double* cs) {
    int p, q;
    round_robin_pair(blockIdx.x, round, npad, p, q);
    const double c = cs[2 * blockIdx.x], sn = cs[2 * blockIdx.x + 1];
    if (q >= n || sn == 0.0) return;
    for (int k = threadIdx.x; k < n; k += 256) {
        double* col = A + (size_t)k * n;
        const double x = col[p], y = col[q];
        col[p] = c * x - sn * y; col[q] = sn * x + c * y;
    }
}
// two-sided result: eigenvalues on the diagonal of A.  J0 = diag(sqrt(S)) V^T (column-major), r0 = diag(sqrt(S^-1)) V^T b'
__global__ void k_eigen_sqrt(const double* A, const double* V, const double* bq, int n, double eps, double* J0, double* r0) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double* v = V + (size_t)j * n;
    const double lam = A[(size_t)j * n + j];
    double vb = 0.0;
    for (int t = 0; t < n; ++t) vb += v[t] * bq[t];
    const double S = lam > eps ? lam : 0.0, Si = lam > eps ? 1.0 / lam : 0.0;
    const double ss = sqrt(S);
    for (int c = 0; c < n; ++c) J0[(size_t)c * n + j] = ss * v[c];
    r0[j] = sqrt(Si) * vb;
}
// ---- dense pseudo-inverse path: A' = Arr - Arm V diag(1 / lambda > eps) V^T Amr,  b' likewise (cpp:351-362) ------------------
// lambda_j = |g_j|^2 (the converged columns of Jm are orthogonal), winv_j = 1 / lambda_j where it exceeds eps, else 0: one wave per column
__global__ __launch_bounds__(256) void k_eig_winv(const double* G, int rows, int n, double eps, double* lam, double* winv) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    const double* g = G + (size_t)j * rows;
    double l = 0.0;
    for (int t = lane; t < rows; t += 64) l += g[t] * g[t];
    l = wave_sum(l);
    if (lane == 0) { lam[j] = l; winv[j] = l > eps ? 1.0 / l : 0.0; }
}
// Y[j][c] = sum_t V[t][j] A[t][m + c]  (c < n),  Y[j][n] = sum_t V[t][j] b[t]      (j < m: eigenvector j against Amr | bmm)
__global__ __launch_bounds__(64) void k_vt_amr(const double* V, const double* A, const double* b, int pos, int m, int n, double* Y) {
    const int j = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
    if (c > n) return;
    const double* v = V + (size_t)j * m;
    double acc = 0.0;
    if (c < n) for (int t = 0; t < m; ++t) acc += v[t] * A[(size_t)t * pos + m + c];
    else for (int t = 0; t < m; ++t) acc += v[t] * b[t];
    Y[(size_t)j * (n + 1) + c] = acc;
}
// A[m + r][m + c] -= sum_j Y[j][r] winv_j Y[j][c];  b[m + r] -= sum_j Y[j][r] winv_j Y[j][n]
__global__ __launch_bounds__(64) void k_dense_schur(double* A, double* b, int pos, int m, int n, const double* Y, const double* winv) {
    const int r = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
    if (c > n) return;
    double acc = 0.0;
    for (int j = 0; j < m; ++j) acc += Y[(size_t)j * (n + 1) + r] * winv[j] * Y[(size_t)j * (n + 1) + c];
    if (c < n) A[(size_t)(m + r) * pos + m + c] -= acc;
    else b[m + r] -= acc;
}
// ---- certificate of marg_exact = 1: is the block-wise pseudo-inverse the dense one? ---------------------------------------------
// In the eigenbasis of the landmark blocks Amm = [[P, C~], [C~^T, Lambda]] (P: the dropped keyframe, <= 15 dims; Lambda diagonal).
// The block-wise path discards the landmark directions with Lambda_i <= eps and then the eigen-directions of
// P' = P - C~ Lambda^+ C~^T below eps; the dense path discards the eigenvectors of Amm below eps.  They agree (to second order in
// the tilt |c_i| / lambda_min of the rest) when
//   (1) every discarded landmark direction is an eigenvector of the WHOLE matrix: its column A[:, block] u_i vanishes
//       (w_max = max |A[:, block] u_i|, rows outside the block, kept parameters included), and
//   (2) nothing else of Amm lies at or below the threshold: with the kept Lambda_r > hi = 2 eps, the number of eigenvalues of the
//       remaining matrix below hi is the negative inertia of S(hi) = P - hi I - C~_r (Lambda_r - hi)^-1 C~_r^T (Sylvester), so
//       S(hi) - tau I must be positive definite (Cholesky), tau = max(1e4 w_max, 1e3 macheps trace P): then the tilt of a discarded
//       direction is <= w_max / tau <= 1e-4 and A' moves by <= 1e-8 relative.
// Anything else — far landmarks whose depth information falls through the threshold, an eigenvalue in (eps, hi], a keyframe block
// with a null direction of its own — takes the dense path.
__global__ __launch_bounds__(64) void k_cert_w(const double* A, int pos, const int* boff, const int* bsize, CertBuf cb) {
    __shared__ double su[36];
    const int bi = blockIdx.x, lane = threadIdx.x;
    const int nd = cb.ndrop[bi];
    if (nd == 0) return;
    const int s = bsize[bi], o = boff[bi];
    if (lane < 36) su[lane] = cb.udrop[(size_t)bi * 36 + lane];
    __syncthreads();
    for (int i = 0; i < nd; ++i) {
        double acc2 = 0.0;
        for (int r = lane; r < pos; r += 64) {
            if (r >= o && r < o + s) continue;
            double dsum = 0.0;
            for (int t = 0; t < s; ++t) dsum += A[(size_t)(o + t) * pos + r] * su[i * 6 + t];      // A is symmetric: row o + t read along r
            acc2 += dsum * dsum;
        }
        acc2 = wave_sum(acc2);
        if (lane == 0) atomicMax(cb.w2max_bits, (unsigned long long)__double_as_longlong(acc2));
    }
}
// out: diag = [w_max, smallest kept landmark eigenvalue, tau, smallest pivot, 1 when the block-wise path may be taken] — ONE read-back
// One wave per entry of S(hi) (round 4: a lane per landmark block, summed in a fixed order; one thread walked all ~100 blocks per entry
// before, 0.1 ms of dependent loads), then the 15 x 15 Cholesky in lane 0 of the workgroup that arrives last.
__global__ __launch_bounds__(64) void k_cert_final(const double* A, int pos, int po, int ps, const int* boff, const int* bsize, int nblk, double hi,
                                                  CertBuf cb, int* cert, double* diag, double* Sg, unsigned* arrive) {
    __shared__ double S[MAXB * MAXB];
    __shared__ int s_last;
    const int t = blockIdx.x, lane = threadIdx.x;
    {
        const int i = ps ? t / ps : 0, j = ps ? t % ps : 0;
        const double* ai = A + (size_t)(po + i) * pos; const double* aj = A + (size_t)(po + j) * pos;      // symmetric: row j for column j
        double acc = 0.0;
        if (ps > 0)
        for (int l = lane; l < nblk; l += 64) {
            const int s = bsize[l], o = boff[l];
            const double* ph = cb.pinv_hi + (size_t)l * 36;
            for (int tt = 0; tt < s; ++tt) {
                const double a = ai[o + tt];
                if (a == 0.0) continue;
                double z = 0.0;
                for (int c = 0; c < s; ++c) z += ph[tt * 6 + c] * aj[o + c];
                acc -= a * z;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (ps > 0) __hip_atomic_store(&Sg[t], ai[po + j] + acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // release / acquire on the arrival counter: the Sg stores happen-before the last workgroup's loads by the memory
            // model, not by what gfx950's L2 happens to do.  (A 15 x 15 grid, once per slide: the L2 write-back a release costs is nothing
            // here — unlike in the LM loop's hand-offs)
            const unsigned prev = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (prev == gridDim.x - 1) ? 1 : 0;
            if (s_last) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int q = lane; q < ps * ps; q += 64) S[q] = __hip_atomic_load(&Sg[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (lane != 0) return;
    const double wmax = sqrt(__longlong_as_double((long long)*cb.w2max_bits));
    const double lmin = __longlong_as_double((long long)*cb.lam_min_bits);
    double tr = 0.0;
    for (int i = 0; i < ps; ++i) tr += fabs(A[(size_t)(po + i) * pos + po + i]);
    const double tau = fmax(1e4 * wmax, 1e3 * 1.1102230246251565e-16 * tr);
    bool ok = (*cb.band == 0) && (lmin > tau);
    double minpiv = 1e300;
    for (int k = 0; k < ps && ok; ++k) {
        double d = S[k * ps + k] - hi - tau;
        for (int q = 0; q < k; ++q) d -= S[k * ps + q] * S[k * ps + q];
        minpiv = fmin(minpiv, d);
        if (!(d > 0.0)) { ok = false; break; }
        const double rd = 1.0 / sqrt(d);
        S[k * ps + k] = sqrt(d);
        for (int r = k + 1; r < ps; ++r) {
            double v = S[r * ps + k];
            for (int q = 0; q < k; ++q) v -= S[r * ps + q] * S[k * ps + q];
            S[r * ps + k] = v * rd;
        }
    }
    cert[0] = ok ? 1 : 0;
    diag[0] = wmax; diag[1] = lmin; diag[2] = tau; diag[3] = minpiv; diag[6] = ok ? 1.0 : 0.0;      // ([4], [5]: the atomically reduced lam_min / w_max^2 bits)
}
__global__ void k_set_identity(double* V, int n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)n * n) V[i] = (i / n == i % n) ? 1.0 : 0.0;
}
__global__ void k_extract_cm(const double* A, int pos, int m, int n, double* G) {   // G (col-major n x n) = A[m.., m..]
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * n) return;
    const int c = (int)(i / n), r = (int)(i % n);
    G[i] = 0.5 * (A[(size_t)(m + r) * pos + m + c] + A[(size_t)(m + c) * pos + m + r]);
}

__global__ void k_cert_init(CertBuf cb) {
    *cb.band = 0;
    *cb.lam_min_bits = (unsigned long long)__double_as_longlong(1e300);
    *cb.w2max_bits = 0ull;
}

// (a) above: G = Jm (rows x n, column-major) is rotated in place until its columns are orthogonal; V (n x n) accumulates.
// (b) above: A (n x n symmetric) is diagonalised in place, V accumulates.
// The host reads the rotation count of a sweep back (a blocking 4-byte copy) from the sweep on at which such a matrix can first
// have converged.
int jacobi_hbm(plba_problem* p, bool two_sided, double* G, int rows, double* V, int n, hipStream_t s) {
    DArr<double> dfloor, dcs; DArr<int> drot;
    PLBA_HIPCK(p, dfloor.alloc(2, false)); PLBA_HIPCK(p, drot.alloc(1));
    hipLaunchKernelGGL(k_set_identity, dim3((int)(((size_t)n * n + 255) / 256)), dim3(256), 0, s, V, n);
    hipLaunchKernelGGL(k_fro_floor, dim3(1), dim3(1024), 0, s, G, (size_t)rows * n, dfloor.p);
    if (n < 2) return PLBA_OK;
    const int npad = (n % 2) ? n + 1 : n;
    if (two_sided) PLBA_HIPCK(p, dcs.alloc(npad, false));
    int rotated = 1;
    double fscale = 1.0;      // the absolute floor of the two-sided form doubles from the 15th sweep on (jacobi2_lds)
    for (int sweep = 0; sweep < 60 && rotated; ++sweep) {
        if (sweep >= 15) fscale *= 2.0;
        PLBA_HIPCK(p, hipMemsetAsync(drot.p, 0, sizeof(int), s));
        for (int round = 0; round < npad - 1; ++round) {
            if (two_sided) {
                hipLaunchKernelGGL(k_j2_cols, dim3(npad / 2), dim3(256), 0, s, G, V, n, npad, round, JACOBI2_TOL * JACOBI2_TOL, dfloor.p, fscale, dcs.p, drot.p);
                hipLaunchKernelGGL(k_j2_rows, dim3(npad / 2), dim3(256), 0, s, G, n, npad, round, dcs.p);
            } else
                hipLaunchKernelGGL(k_hestenes_round, dim3(npad / 2), dim3(256), 0, s, G, rows, V, n, npad, round, JACOBI_TOL * JACOBI_TOL, dfloor.p, drot.p);
        }
        if (sweep >= 4) PLBA_HIPCK(p, plba_d2h(p, &rotated, drot.p, sizeof(int)));
    }
    if (rotated) PLBA_FAIL(p, PLBA_ERR_NUMERIC, "marginalize: the Jacobi eigen-decomposition of a %d-column block did not converge", n);
    return PLBA_OK;
}

struct Param { int pid, size, drop, kf, isbias; };

void est_pvr(const double* s, double* o) {   // GetEstData: P, V, Quaterniond(Rwb) as (x,y,z,w)   (IMU/g2otypes.h:127-168)
    memcpy(o, s, 24); memcpy(o + 3, s + 3, 24);
    Q4 q; q.x = s[6]; q.y = s[7]; q.z = s[8]; q.w = s[9];
    const Q4 c = R_to_q(q_to_R(q));
    o[6] = c.x; o[7] = c.y; o[8] = c.z; o[9] = c.w;
}
void est_bias(const double* s, double* o) { for (int c = 0; c < 3; ++c) { o[c] = s[10 + c] + s[16 + c]; o[3 + c] = s[13 + c] + s[19 + c]; } }

}  // namespace

#define PID_PT(i) ((1 << 28) + (i))
#define PID_LN(i) ((1 << 29) + (i))

// General form: explicit factor lists.  imu_edges[]: each contributes its PVR edge and its bias edge;
// pt_edges / ln_edges: indices into the uploaded point / line observation arrays; drop_vid[]: keyframe vertices
// (ids) to marginalize out.  The landmark of every listed observation is always dropped (drop_set {0} at the call site).
int marginalize_factors_device(plba_problem* p, const std::vector<int>& imu_edges, const std::vector<int>& pt_edges,
                               const std::vector<int>& ln_edges, bool use_prior, const std::vector<int>& drop_vid, plba_prior* out) {
    memset(out, 0, sizeof *out);
    const DevBuf& d = p->dv;
    hipStream_t s = p->stream;
    auto dropped = [&](int vid) { return std::find(drop_vid.begin(), drop_vid.end(), vid) != drop_vid.end(); };
    std::map<int, Param> params;     // keyed by pid: ascending order == SURVEY B-Q6 decision
    auto touch = [&](int pid, int size, bool drop, int kf, int isb) {
        auto it = params.find(pid);
        if (it == params.end()) params[pid] = Param{pid, size, drop ? 1 : 0, kf, isb};
        else if (drop) it->second.drop = 1;
    };
    int R = 0;
    std::vector<int> imu_rows;
    for (int m : imu_edges) {
        if (m < 0 || m >= p->M) PLBA_FAIL(p, PLBA_ERR_INVALID, "marginalize: imu edge %d out of range", m);
        const int ki = p->imu_i[m], kj = p->imu_j[m];
        touch(p->vid_pvr[ki], 9, dropped(p->vid_pvr[ki]), ki, 0); touch(p->vid_pvr[kj], 9, dropped(p->vid_pvr[kj]), kj, 0);
        touch(p->vid_bias[ki], 6, dropped(p->vid_bias[ki]), ki, 1); touch(p->vid_bias[kj], 6, dropped(p->vid_bias[kj]), kj, 1);
        imu_rows.push_back(R); R += 15;
    }
    std::vector<MargObs> obs;
    for (int e : pt_edges) {
        if (e < 0 || e >= p->Ep) PLBA_FAIL(p, PLBA_ERR_INVALID, "marginalize: point edge %d out of range", e);
        const int l = p->po_pt[e], k = p->po_kf[e];
        touch(PID_PT(l), 3, true, -1, 0);
        touch(p->vid_pvr[k], 9, dropped(p->vid_pvr[k]), k, 0);
        obs.push_back(MargObs{e, R, PID_PT(l), p->vid_pvr[k]});
        R += 2;
    }
    for (int e : ln_edges) {
        if (e < 0 || e >= p->El) PLBA_FAIL(p, PLBA_ERR_INVALID, "marginalize: line edge %d out of range", e);
        const int l = p->lo_ln[e], k = p->lo_kf[e];
        touch(PID_LN(l), 6, true, -1, 0);
        touch(p->vid_pvr[k], 9, dropped(p->vid_pvr[k]), k, 0);
        obs.push_back(MargObs{p->Ep + e, R, PID_LN(l), p->vid_pvr[k]});
        R += 2;
    }
    int prior_row = -1;
    if (use_prior && p->pr_nv > 0) {
        std::map<int, std::pair<int, int>> by_vid;
        for (int k = 0; k < p->K; ++k) { by_vid[p->vid_pvr[k]] = {k, 0}; if (p->vid_bias[k] >= 0) by_vid[p->vid_bias[k]] = {k, 1}; }
        for (int i = 0; i < p->pr_nv; ++i) {
            const auto& kv = by_vid[p->pr_vid[i]];
            touch(p->pr_vid[i], p->pr_size[i], dropped(p->pr_vid[i]), kv.first, kv.second);
        }
        prior_row = R; R += p->pr_n;
    }
    if (params.empty() || R == 0) PLBA_FAIL(p, PLBA_ERR_STATE, "marginalize: empty factor set");
    // ---- parameter order: dropped first (keyframe block, then landmark blocks), then kept; ascending id inside ---------
    std::map<int, int> col;          // pid -> column offset
    int pos = 0;
    std::vector<int> blk_off, blk_size;      // landmark blocks
    int pose_off = 0, pose_size = 0;
    for (auto& kv : params) if (kv.second.drop && kv.second.kf >= 0) { col[kv.first] = pos; pos += kv.second.size; }
    pose_size = pos;
    if (pose_size > MAXB) PLBA_FAIL(p, PLBA_ERR_INVALID, "marginalize: dropped keyframe block of %d dims", pose_size);
    for (auto& kv : params) if (kv.second.drop && kv.second.kf < 0) { col[kv.first] = pos; blk_off.push_back(pos); blk_size.push_back(kv.second.size); pos += kv.second.size; }
    const int m = pos;
    std::vector<const Param*> kept;
    for (auto& kv : params) if (!kv.second.drop) {
        if (kv.second.kf < 0) PLBA_FAIL(p, PLBA_ERR_INVALID, "marginalize: a landmark would be kept");
        col[kv.first] = pos; pos += kv.second.size; kept.push_back(&kv.second);
    }
    const int n = pos - m;
    for (auto& o : obs) { o.col_lm = col[o.col_lm]; o.col_kf = col[o.col_kf]; }
    // ---- device work: every index list goes up before the first launch, one stream synchronisation at the end ----------------
    DArr<double> dJ, dr, dA, db, dZ, dPinvL, dPinvP, dOut, dG, dV, dVm, dLam, dY, dCertD;
    DArr<MargObs> dobs;
    DArr<int> dboff, dbsize, delimL, delimP, dvcol, dimu, dCertI;
    DArr<uint8_t> drestL, drestP;
    struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } sync_on_exit{s};   // declared after the buffers: runs before they go back to the pool
    DArrStreamScope staged(s, p->have_ctx ? p->ctx.stage : nullptr);      // prepare() has synchronised: the staging area is free again
    const int state = p->cur;
    const double eps = p->opt.marg_eps;
    // host-side index vectors (alive until the final synchronisation: without a staging area the copies read them directly)
    std::vector<int> imu_desc, vcol, elimL, elimP;
    std::vector<uint8_t> restL(pos, 1), restP;
    for (size_t t = 0; t < imu_edges.size(); ++t) {
        const int mI = imu_edges[t], ki = p->imu_i[mI], kj = p->imu_j[mI];
        for (int v : {mI, imu_rows[t], col[p->vid_pvr[ki]], col[p->vid_pvr[kj]], col[p->vid_bias[ki]], col[p->vid_bias[kj]]}) imu_desc.push_back(v);
    }
    if (prior_row >= 0) { vcol.resize(p->pr_nv); for (int i = 0; i < p->pr_nv; ++i) vcol[i] = col[p->pr_vid[i]]; }
    for (size_t bq = 0; bq < blk_off.size(); ++bq) for (int t = 0; t < blk_size[bq]; ++t) { elimL.push_back(blk_off[bq] + t); restL[blk_off[bq] + t] = 0; }
    restP = restL;
    for (int t = 0; t < pose_size; ++t) { elimP.push_back(pose_off + t); restP[pose_off + t] = 0; }
    for (size_t bq = 0; bq < blk_size.size(); ++bq) if (blk_size[bq] != 3 && blk_size[bq] != 6) PLBA_FAIL(p, PLBA_ERR_INVALID, "marginalize: landmark block of %d dims", blk_size[bq]);
    PLBA_HIPCK(p, dJ.alloc((size_t)R * pos)); PLBA_HIPCK(p, dr.alloc(R)); PLBA_HIPCK(p, dA.alloc((size_t)pos * pos, false)); PLBA_HIPCK(p, db.alloc(pos, false));
    PLBA_HIPCK(p, dZ.alloc((size_t)pos * pos)); PLBA_HIPCK(p, dobs.upload(obs)); PLBA_HIPCK(p, dimu.upload(imu_desc)); PLBA_HIPCK(p, dvcol.upload(vcol));
    PLBA_HIPCK(p, dboff.upload(blk_off)); PLBA_HIPCK(p, dbsize.upload(blk_size)); PLBA_HIPCK(p, delimL.upload(elimL)); PLBA_HIPCK(p, delimP.upload(elimP));
    PLBA_HIPCK(p, drestL.upload(restL)); PLBA_HIPCK(p, drestP.upload(restP));
    PLBA_HIPCK(p, dPinvL.alloc(std::max<size_t>(blk_off.size(), 1) * MAXB * MAXB, false)); PLBA_HIPCK(p, dPinvP.alloc(MAXB * MAXB, false));
    const size_t nout = 2 * (size_t)n * n + 2 * (size_t)n + 1;      // [A' | b' | J0 | r0 | convergence word of k_marg_finish]
    PLBA_HIPCK(p, dOut.alloc(nout));
    // (1) factors -> stacked Jacobian
    if (prior_row >= 0) launch_pose_edges(d, state, false, p->rob, true, s);       // refreshes pr_err = EdgeMarginalization::computeError at the final estimate
    {
        const int nobs = (int)obs.size(), nimu = (int)imu_edges.size();
        const int ob = (nobs + 63) / 64, pb = prior_row >= 0 ? 16 : 0;
        if (ob + nimu + pb > 0)
            hipLaunchKernelGGL(k_marg_factors, dim3(ob + nimu + pb), dim3(64), 0, s, d, state, dobs.p, nobs, ob, dimu.p, nimu, prior_row, dvcol.p, dJ.p, dr.p, R);
    }
    if (p->opt.diag & PLBA_DIAG_MARG_DUMP) {      // the stacked Jacobian and residual, for an extended-precision evaluation of the whole step
        p->marg_dbg.assign(4 + (size_t)R * pos + R, 0.0);
        p->marg_dbg[0] = R; p->marg_dbg[1] = pos; p->marg_dbg[2] = m; p->marg_dbg[3] = n;
        PLBA_HIPCK(p, plba_d2h(p, p->marg_dbg.data() + 4, dJ.p, (size_t)R * pos * 8));
        PLBA_HIPCK(p, plba_d2h(p, p->marg_dbg.data() + 4 + (size_t)R * pos, dr.p, (size_t)R * 8));
    }
    // (2) A = J^T J, b = J^T r
    launch_ata(dJ.p, R, pos, dA.p, pos, s);
    hipLaunchKernelGGL(k_jt_r, dim3((pos + 3) / 4), dim3(256), 0, s, dJ.p, dr.p, R, pos, db.p);
    // (3) pseudo-inverse Schur elimination of the dropped block: block by block, or the dense eigen-decomposition of Amm
    const int nb = (int)blk_off.size();
    const double hi = 2.0 * eps;
    CertBuf cb{};
    int mode = p->opt.marg_exact;
    if (mode < 0 || mode > 2) PLBA_FAIL(p, PLBA_ERR_INVALID, "marg_exact = %d (0 block-wise, 1 certified block-wise else dense, 2 dense)", mode);
    bool blockwise = mode != 2;
    const bool cert = mode == 1 && m > 0;
    double cert_diag[7] = {0, 0, 0, 0, 0, 0, 0};
    constexpr int CERT_HDR = 8 + MAXB * MAXB + 8;      // [diag (8) | S(hi) (15 x 15) | arrival counter (8)] ahead of the per-block tables
    if (cert) {
        PLBA_HIPCK(p, dCertD.alloc((size_t)std::max(nb, 1) * 72 + CERT_HDR)); PLBA_HIPCK(p, dCertI.alloc((size_t)std::max(nb, 1) + 8));
        cb.pinv_hi = dCertD.p + CERT_HDR; cb.udrop = cb.pinv_hi + (size_t)std::max(nb, 1) * 36;
        cb.ndrop = dCertI.p + 8; cb.band = dCertI.p + 1;
        cb.lam_min_bits = reinterpret_cast<unsigned long long*>(dCertD.p + 4); cb.w2max_bits = reinterpret_cast<unsigned long long*>(dCertD.p + 5);
        hipLaunchKernelGGL(k_cert_init, dim3(1), dim3(1), 0, s, cb);
    }
    if (nb > 0 && blockwise)
        hipLaunchKernelGGL(k_block_pinv_small, dim3((nb + 63) / 64), dim3(64), 0, s, dA.p, pos, dboff.p, dbsize.p, nb, eps, dPinvL.p, cert, hi, cb);
    if (cert) {
        if (nb > 0) hipLaunchKernelGGL(k_cert_w, dim3(nb), dim3(64), 0, s, dA.p, pos, dboff.p, dbsize.p, cb);
        hipLaunchKernelGGL(k_cert_final, dim3(std::max(pose_size * pose_size, 1)), dim3(64), 0, s, dA.p, pos, pose_off, pose_size, dboff.p, dbsize.p, nb, hi, cb, dCertI.p, dCertD.p,
                           dCertD.p + 8, reinterpret_cast<unsigned*>(dCertD.p + 8 + MAXB * MAXB));
        PLBA_HIPCK(p, plba_d2h(p, cert_diag, dCertD.p, sizeof cert_diag));      // (blocking: the one decision this call takes on the host)
        blockwise = cert_diag[6] != 0.0;
    }
    p->marg_path[0] = blockwise ? 0.0 : 1.0;
    for (int t = 0; t < 4; ++t) p->marg_path[1 + t] = cert_diag[t];
    if (blockwise) {
        // landmark blocks, then the keyframe block of the reduced system
        if (nb > 0) {
            hipLaunchKernelGGL(k_block_Z, dim3((pos + 63) / 64, nb), dim3(64), 0, s, dA.p, pos, dboff.p, dbsize.p, nb, dPinvL.p, dZ.p);
            hipLaunchKernelGGL(k_schur_apply, dim3((pos + 1 + 15) / 16, (pos + 15) / 16), dim3(16, 16), 0, s, dA.p, db.p, pos, dZ.p, delimL.p, (int)elimL.size(), drestL.p);
        }
        if (pose_size > 0) {
            hipLaunchKernelGGL(k_pose_pinv, dim3(1), dim3(256), 0, s, dA.p, pos, pose_off, pose_size, eps, dPinvP.p);
            hipLaunchKernelGGL(k_block_Z1, dim3((pos * pose_size + 255) / 256), dim3(256), 0, s, dA.p, pos, pose_off, pose_size, dPinvP.p, dZ.p);
            hipLaunchKernelGGL(k_schur_apply, dim3((pos + 1 + 15) / 16, (pos + 15) / 16), dim3(16, 16), 0, s, dA.p, db.p, pos, dZ.p, delimP.p, (int)elimP.size(), drestP.p);
        }
    } else if (m > 0) {
        // the reference's own form: eigen-decomposition of the whole Amm, eigenvalues <= eps discarded (cpp:351-362)
        // Amm = Jm^T Jm: the first m columns of the stacked Jacobian (no longer needed as such: A and b are formed) are rotated in place
        PLBA_HIPCK(p, dVm.alloc((size_t)m * m, false));
        PLBA_HIPCK(p, dLam.alloc(2 * (size_t)m, false)); PLBA_HIPCK(p, dY.alloc((size_t)m * (n + 1), false));
        if (int rc = jacobi_hbm(p, false, dJ.p, R, dVm.p, m, s)) return rc;
        hipLaunchKernelGGL(k_eig_winv, dim3((m + 3) / 4), dim3(256), 0, s, dJ.p, R, m, eps, dLam.p, dLam.p + m);
        hipLaunchKernelGGL(k_vt_amr, dim3((n + 1 + 63) / 64, m), dim3(64), 0, s, dVm.p, dA.p, db.p, pos, m, n, dY.p);
        if (n > 0) hipLaunchKernelGGL(k_dense_schur, dim3((n + 1 + 63) / 64, n), dim3(64), 0, s, dA.p, db.p, pos, m, n, dY.p, dLam.p + m);
    }
    // (4) eigen square root of the kept block
    double* oAr = dOut.p; double* obr = oAr + (size_t)n * n; double* oJ0 = obr + n; double* or0 = oJ0 + (size_t)n * n;
    if (n <= JLDS_MAX_N2) {
        // the live part of A' (its non-zero columns, known to the kernel only) decides whether V fits LDS next to it: the launch asks for the
        // larger of the two layouts, and V gets a global buffer whenever n itself is beyond the in-LDS limit
        const bool v_lds = n <= JLDS_MAX_N;
        const size_t nl_max = std::min(n, JLDS_MAX_N);
        size_t sh = std::max((size_t)n * (n | 1), nl_max * (nl_max | 1) + nl_max * nl_max) * sizeof(double);
        // room for the one-barrier Jacobi's row tables behind A' and V, when the CU's LDS (160 KB less the kernel's static 4 KB) has it
        const size_t sh_fast = (nl_max * (nl_max | 1) + nl_max * nl_max) * sizeof(double) + sizeof(Jac2sScratch);
        if (std::max(sh, sh_fast) <= (size_t)163840 - 4096) sh = std::max(sh, sh_fast);
        PLBA_HIPCK(p, ensure_dyn_lds(reinterpret_cast<const void*>(k_marg_finish), (int)sh));
        if (!v_lds) PLBA_HIPCK(p, dV.alloc((size_t)n * n, false));
        hipLaunchKernelGGL(k_marg_finish, dim3(1), dim3(1024), sh, s, dA.p, db.p, pos, m, n, eps, dOut.p, v_lds ? nullptr : dV.p, d.dbgbuf, (int)sh, (p->opt.diag & PLBA_DIAG_NO_MARG_PREROTATE) ? 0 : 1);
    } else {
        // larger kept blocks: G and V in HBM, one launch per round
        PLBA_HIPCK(p, dG.alloc((size_t)n * n, false)); PLBA_HIPCK(p, dV.alloc((size_t)n * n, false));
        const int nn_blocks = (int)(((size_t)n * n + 255) / 256);
        hipLaunchKernelGGL(k_extract_cm, dim3(nn_blocks), dim3(256), 0, s, dA.p, pos, m, n, dG.p);
        PLBA_HIPCK(p, hipMemcpyAsync(oAr, dG.p, (size_t)n * n * 8, hipMemcpyDeviceToDevice, s));      // symmetric: col-major == row-major
        PLBA_HIPCK(p, hipMemcpyAsync(obr, db.p + m, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        if (int rc = jacobi_hbm(p, true, dG.p, n, dV.p, n, s)) return rc;
        hipLaunchKernelGGL(k_eigen_sqrt, dim3((n + 63) / 64), dim3(64), 0, s, dG.p, dV.p, db.p + m, n, eps, oJ0, or0);
    }
    // ---- results: two asynchronous copies into pinned memory (the staging area's free tail), ONE synchronisation -------------
    const size_t nkf = (size_t)p->K * KF_STRIDE;
    std::vector<double> pageable;
    double* hres = nullptr;
    {
        StageArea* st = darr_stage();
        const size_t need = (nout + nkf) * sizeof(double), at = st ? ((st->used + 255) & ~(size_t)255) : 0;
        if (st && st->base && at + need <= st->upload_cap()) hres = (double*)(st->base + at);
        else { pageable.resize(nout + nkf); hres = pageable.data(); }
    }
    PLBA_HIPCK(p, hipMemcpyAsync(hres, dOut.p, nout * 8, hipMemcpyDeviceToHost, s));
    PLBA_HIPCK(p, hipMemcpyAsync(hres + nout, d.kf[state], nkf * 8, hipMemcpyDeviceToHost, s));
    PLBA_HIPCK(p, plba_stream_wait(s));
    PLBA_HIPCK(p, hipGetLastError());
    if (hres[nout - 1] >= 2.0) PLBA_FAIL(p, PLBA_ERR_NUMERIC, "marginalize: the Jacobi eigen-decomposition of the kept %d x %d block hit its sweep limit", n, n);
    // ---- output (host buffers owned by the caller until plba_prior_free) ----------------------------------------------------------
    out->n = n; out->m = m; out->nv = (int)kept.size();
    out->vid = (int32_t*)calloc(kept.size() + 1, 4); out->size = (int32_t*)calloc(kept.size() + 1, 4); out->idx = (int32_t*)calloc(kept.size() + 1, 4);
    out->J0 = (double*)malloc(((size_t)n * n + 1) * 8); out->r0 = (double*)malloc((size_t)(n + 1) * 8);
    out->Ar = (double*)malloc(((size_t)n * n + 1) * 8); out->br = (double*)malloc((size_t)(n + 1) * 8);
    memcpy(out->Ar, hres, (size_t)n * n * 8); memcpy(out->br, hres + (size_t)n * n, (size_t)n * 8);
    memcpy(out->J0, hres + (size_t)n * n + n, (size_t)n * n * 8); memcpy(out->r0, hres + 2 * (size_t)n * n + n, (size_t)n * 8);
    const double* kfh = hres + nout;
    size_t nx = 0;
    for (auto* k : kept) nx += k->size == 9 ? 10 : 6;
    out->x0 = (double*)calloc(nx + 1, 8);
    nx = 0;
    for (size_t i = 0; i < kept.size(); ++i) {
        const Param* k = kept[i];
        out->vid[i] = k->pid; out->size[i] = k->size; out->idx[i] = col[k->pid] - m;
        const double* st = &kfh[(size_t)k->kf * KF_STRIDE];
        if (k->size == 9) { est_pvr(st, out->x0 + nx); nx += 10; } else { est_bias(st, out->x0 + nx); nx += 6; }
    }
    return PLBA_OK;
}

// Factor selection of the call site (src/mapHandler.cpp:6075-6188): first IMU edge, <= NUM+1 point edges and
// <= NUM+1 line edges whose landmark was first observed in the oldest keyframe, the old prior; drop that keyframe.
int marginalize_device(plba_problem* p, int first_kf, int max_edges, plba_prior* out) {
    if (first_kf < 0 || first_kf >= p->K) PLBA_FAIL(p, PLBA_ERR_INVALID, "marginalize: first_kf out of range");
    const int NUM = max_edges;
    std::vector<int> imu, pts, lns, drop;
    if (p->M > 0) imu.push_back(0);
    auto select = [&](const std::vector<int32_t>& lm_of, const std::vector<int32_t>& kf_of, int E, int N, std::vector<int>& outv) {
        int num = 0, e = 0;
        for (int l = 0; l < N && num <= NUM; ++l) {
            const int e0 = e;
            while (e < E && lm_of[e] == l) ++e;
            if (e0 == e || kf_of[e0] != first_kf) continue;      // kf_obs_list[0] == first_kf_idx
            for (int a = e0; a < e; ++a) { outv.push_back(a); if (++num > NUM) break; }   // `num>NUM` admits NUM+1 (B-Q10)
        }
    };
    select(p->po_pt, p->po_kf, p->Ep, p->Np, pts);
    select(p->lo_ln, p->lo_kf, p->El, p->Nl, lns);
    drop.push_back(p->vid_pvr[first_kf]);
    if (p->vid_bias[first_kf] >= 0) drop.push_back(p->vid_bias[first_kf]);
    return marginalize_factors_device(p, imu, pts, lns, true, drop, out);
}

}  // namespace plba

// plba_lba.hip — SURVEY §8f row 2: the pre-VIO-init, visual-only local BA of the reference,
// MapHandler::levMarquardtOptimizationLBA (src/mapHandler.cpp:1441-2098), on gfx950.
//
// The reference hand-rolls LM over X = [6 per local keyframe | 3 per point | 6 per line] with a SCALAR residual per
// observation (the norm of the reprojection error), Cauchy weights, multiplicative damping and a sparse LDL^T of the
// whole system.  A scalar residual makes every observation's Hessian contribution rank one: H_pl = w Jp Jl^T, so the
// landmark-block Schur complement of a landmark l seen from keyframes a, b is the rank-one update
//     S[a, b] -= (w_a w_b Jl_a^T D_l Jl_b) Jp_a Jp_b^T,        D_l = (Hll_l + lambda diag Hll_l)^-1,
// which is what k_lba_pairs forms (a wave per keyframe pair chunk over host-built pair entries); the 6 Nkf pose system then goes through the dense fp64 Cholesky of plba_dense.hip and
// landmarks are back-substituted.  The LM control (lambda schedule as coded, termination tests, the map-pose quirk of the
// line pass) stays on the host, two small read-backs per iteration: this path runs a handful of iterations on a small
// window before the IMU is initialised; it is not the north-star kernel and is not tuned like it.
//
// Kernel map (one iteration): k_lba_poses -> k_lba_landmarks (thread per landmark, observations landmark-major as
// the reference's lists are) -> k_lba_posesys (workgroup per local keyframe over its observations) -> k_lba_reduce
// -> [host: err, lambda] -> k_lba_sysinit -> k_lba_dinv -> k_lba_pairs -> k_lba_rhs -> Cholesky / back-substitution (plba_dense.hip)
// -> k_lba_backsub -> k_lba_update (|DX|^2; the update itself gated on solver_ok) -> [host: |DX|].
#include "plba_internal.h"
#include "plba_problem.h"
#include <chrono>
#include <cstddef>

#define HIPCK(p, call) PLBA_HIPCK(p, call)
#define FAIL(p, code, ...) PLBA_FAIL(p, code, __VA_ARGS__)

namespace plba {
namespace {

constexpr int REC = 14;      // per-observation record: Jp[6], Jl[6] (points use 3), w, |e|

// The pass-to-pass control of the optimiser (error comparison, lambda schedule, the three exits) lives on the device: a pass is enqueued
// without waiting for the one before — the host only polls a mapped mailbox one pass behind, to stop enqueuing — and every kernel
// of a pass returns at once when an earlier pass has ended the run (`done`).  Round 2 read `err` and `|DX|` back with two blocking
// copies per pass: 0.31 ms per pass of which 0.13 ms kernels.
struct LbaCtl {
    double err, err_prev, lambda, lambda_next, err_first, dx2;
    int iters, updates, do_update, done, failed, pad;
};
struct LbaMail {       // laid over the problem's mapped Mailbox (plba_internal.h): same size, `seq` at the same offset
    LbaCtl c;
    char pad[sizeof(Ctrl) - sizeof(LbaCtl)];
    unsigned long long seq;
    char pad2[sizeof(Mailbox) - sizeof(Ctrl) - sizeof(unsigned long long)];
};
static_assert(sizeof(LbaCtl) <= sizeof(Ctrl) && sizeof(LbaMail) == sizeof(Mailbox) && offsetof(LbaMail, seq) == offsetof(Mailbox, seq), "the LBA mailbox reuses the problem's mapped mailbox");
struct LbaDev {
    int K, Nkf, Np, Nl, Ep, El, P, Ppad, ld;
    double fx, fy, cx, cy, homog_th;
    double lambda_k, min_error, min_error_change;      // plba_lba_options: what the device-side control needs
    int variant;
    LbaCtl* ctl;
    LbaMail* mail;             // mapped host memory
    const double* Tmap;        // K x 16 row-major map poses (T_kf_w)
    const int32_t* kf_loc;     // K
    const int32_t* loc_kf;     // Nkf
    const int32_t* lm_start;   // Np + Nl + 1: first observation of each landmark (lines offset by Ep)
    const int32_t* obs_kf;     // Ep + El
    const double* obs_z;       // Ep x 2 then El x 3 (uv / line coefficients), packed at stride 3
    const int32_t* kf_start;   // Nkf + 1
    const int32_t* kf_obs;     // observations of each local keyframe
    double* Xp;                // 6 Nkf
    double* Xl;                // 3 Np + 6 Nl
    double* Tiw;               // 2 x K x 12: inverse poses used by the point pass / the line pass
    double* rec;               // (Ep + El) x REC
    double* Hll;               // (Np + Nl) x 21 (upper, row-major packed; points use the first 6)
    double* gl;                // (Np + Nl) x 6
    double* Dl;                // (Np + Nl) x 21: (Hll + lambda diag)^-1
    double* urec;              // (Ep + El) x 8: u = w D Jl (6), s = w Jl . (D gl), pad
    const int32_t* pair_ent;   // 2 x nent: the two observations (of keyframes la <= lb, one landmark) of each Schur pair entry, pair-major
    const int32_t* chunk;      // nchunk x 4: la, lb, first entry, end entry (<= PAIR_CHUNK entries of one keyframe pair)
    int nchunk;
    double* Hpp;               // Nkf x 21
    double* gp;                // Nkf x 6
    double* part;              // 2 x nblk per-block partials: error sums (later the landmark part of |DX|^2), max |Hll_ii|
    double* scal;
    double* DXl;               // 3 Np + 6 Nl
    double* sys;               // (Ppad + 64) x ld
    double* x;                 // dense solution
    Ctrl* ctrl;
    double* Tout;              // K x 16
};

struct T12 { double R[9]; double t[3]; };

__device__ __forceinline__ void hat9(const double* w, double* s) { s[0] = 0; s[1] = -w[2]; s[2] = w[1]; s[3] = w[2]; s[4] = 0; s[5] = -w[0]; s[6] = -w[1]; s[7] = w[0]; s[8] = 0; }
__device__ __forceinline__ void mm3(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
// expmap_se3, stvo-pl/src/auxiliar.cpp:124-141 (x = (t, w))
__device__ void se3_exp(const double* x, T12& T) {
    const double w0 = x[3], w1 = x[4], w2 = x[5];
    const double th = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
#pragma unroll
    for (int i = 0; i < 9; ++i) T.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    T.t[0] = x[0]; T.t[1] = x[1]; T.t[2] = x[2];
    if (!(th < 0.000001)) {
        const double wn[3] = {w0 / th, w1 / th, w2 / th};
        double s[9], ss[9], V[9];
        hat9(wn, s); mm3(s, s, ss);
        const double sn = sin(th), cs = cos(th);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double I = (i % 4 == 0) ? 1.0 : 0.0;
            T.R[i] = I + s[i] * sn + ss[i] * (1.0 - cs);
            V[i] = I + s[i] * (1.0 - cs) / th + ss[i] * (th - sn) / th;
        }
        const double t0 = x[0], t1 = x[1], t2 = x[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) T.t[i] = V[i * 3] * t0 + V[i * 3 + 1] * t1 + V[i * 3 + 2] * t2;
    }
}
__device__ __forceinline__ void se3_inv(const T12& T, T12& Ti) {      // inverse_se3, auxiliar.cpp:113-122
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Ti.R[i * 3 + j] = T.R[j * 3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Ti.t[i] = -(Ti.R[i * 3] * T.t[0] + Ti.R[i * 3 + 1] * T.t[1] + Ti.R[i * 3 + 2] * T.t[2]);
}
__device__ __forceinline__ void se3_mul(const T12& A, const T12& B, T12& C) {
    mm3(A.R, B.R, C.R);
#pragma unroll
    for (int i = 0; i < 3; ++i) C.t[i] = A.R[i * 3] * B.t[0] + A.R[i * 3 + 1] * B.t[1] + A.R[i * 3 + 2] * B.t[2] + A.t[i];
}
__device__ bool inv3(const double* A, double* Ai) {      // general 3 x 3 inverse by partial-pivot elimination (the oracle's lu_inverse)
    double M[3][6];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { M[i][j] = A[i * 3 + j]; M[i][3 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 3; ++c) {
        int pv = c;
        for (int r = c + 1; r < 3; ++r) if (fabs(M[r][c]) > fabs(M[pv][c])) pv = r;
        if (M[pv][c] == 0.0) return false;
        if (pv != c) for (int j = 0; j < 6; ++j) { const double t = M[c][j]; M[c][j] = M[pv][j]; M[pv][j] = t; }
        const double d = 1.0 / M[c][c];
        for (int j = 0; j < 6; ++j) M[c][j] *= d;
        for (int r = 0; r < 3; ++r) if (r != c) { const double f = M[r][c]; for (int j = 0; j < 6; ++j) M[r][j] -= f * M[c][j]; }
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ai[i * 3 + j] = M[i][3 + j];
    return true;
}
// logmap_se3, auxiliar.cpp:143-173
__device__ void se3_log(const T12& T, double* x) {
    const double* R = T.R;
    double cosine = (R[0] + R[4] + R[8] - 1.0) / 2.0;
    cosine = cosine > 1.0 ? 1.0 : (cosine < -1.0 ? -1.0 : cosine);
    double sine = sqrt(1.0 - cosine * cosine);
    if (sine > 1.0) sine = 1.0;
    const double theta = acos(cosine);
    double w[3] = {0, 0, 0}, V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta > 0.000001) {
        const double f = theta / (2.0 * sine);
        w[0] = f * (R[7] - R[5]); w[1] = f * (R[2] - R[6]); w[2] = f * (R[3] - R[1]);
        const double wn[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
        double s[9], ss[9];
        hat9(wn, s); mm3(s, s, ss);
#pragma unroll
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s[i] * (1.0 - cosine) / theta + ss[i] * (theta - sine) / theta;
    }
    double Vi[9];
    inv3(V, Vi);
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = Vi[i * 3] * T.t[0] + Vi[i * 3 + 1] * T.t[1] + Vi[i * 3 + 2] * T.t[2];
    x[3] = w[0]; x[4] = w[1]; x[5] = w[2];
}
__device__ __forceinline__ void load_T16(const double* m, T12& T) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { T.R[i * 3] = m[i * 4]; T.R[i * 3 + 1] = m[i * 4 + 1]; T.R[i * 3 + 2] = m[i * 4 + 2]; T.t[i] = m[i * 4 + 3]; }
}

// the initial X of the local keyframes: x_kf_w = logmap(T_kf_w)
__global__ void k_lba_init(LbaDev d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.Nkf) return;
    T12 T; load_T16(d.Tmap + 16 * d.loc_kf[i], T);
    se3_log(T, d.Xp + 6 * i);
}
// inverse poses of this pass: points take the iterate for local keyframes after the first pass (:1726-1730), lines take the
// map pose throughout (:1790) unless use_iter
__global__ void k_lba_poses(LbaDev d, int later_pass, int use_iter) {
    if (d.ctl->done) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d.K) return;
    const int loc = d.kf_loc[k];
    T12 Tm, Tx, Ti;
    load_T16(d.Tmap + 16 * k, Tm);
    const bool it = later_pass && loc >= 0;
    if (it) se3_exp(d.Xp + 6 * loc, Tx);
    se3_inv(it ? Tx : Tm, Ti);
#pragma unroll
    for (int i = 0; i < 9; ++i) d.Tiw[12 * k + i] = Ti.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) d.Tiw[12 * k + 9 + i] = Ti.t[i];
    if (!(it && use_iter)) se3_inv(Tm, Ti);
    double* o = d.Tiw + (size_t)12 * d.K + 12 * k;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = Ti.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[9 + i] = Ti.t[i];
}

// :1490-1512 — the six pose-Jacobian terms for a camera-frame point g and the pair (a, b) = (fx e0, fy e1); the first
// three are also the landmark row before its rotation into the world frame
__device__ __forceinline__ void jac_pieces(const double* g, double a, double b, double homog_th, double* J) {
    double gz2 = g[2] * g[2];
    gz2 = 1.0 / (homog_th > gz2 ? homog_th : gz2);
    J[0] = gz2 * a * g[2];
    J[1] = gz2 * b * g[2];
    J[2] = -gz2 * (a * g[0] + b * g[1]);
    J[3] = -gz2 * (a * g[0] * g[1] + b * g[1] * g[1] + b * g[2] * g[2]);
    J[4] = gz2 * (a * g[0] * g[0] + a * g[2] * g[2] + b * g[0] * g[1]);
    J[5] = gz2 * (b * g[0] * g[2] - a * g[1] * g[2]);
}
__device__ __forceinline__ void to_cam(const double* Tiw, const double* X, double* g) {
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = Tiw[i * 3] * X[0] + Tiw[i * 3 + 1] * X[1] + Tiw[i * 3 + 2] * X[2] + Tiw[9 + i];
}
__device__ __forceinline__ void row_R(const double* v, const double* Tiw, double* o) {      // v^T R
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = v[0] * Tiw[c] + v[1] * Tiw[3 + c] + v[2] * Tiw[6 + c];
}

constexpr int LM_NT = 128;
__device__ __forceinline__ int diag_q(int a) { return a * 6 - a * (a - 1) / 2; }      // packed index of (a, a)
// thread per landmark: its observations' records, Hll (packed upper), gl, and the error sum
__global__ void __launch_bounds__(LM_NT) k_lba_landmarks(LbaDev d) {
    if (d.ctl->done) return;
    const int l = blockIdx.x * LM_NT + threadIdx.x, L = d.Np + d.Nl;
    double err = 0.0, hm = 0.0;
    if (l < L) {
        const bool is_pt = l < d.Np;
        const int dim = is_pt ? 3 : 6;
        const double* X = is_pt ? d.Xl + 3 * l : d.Xl + 3 * d.Np + 6 * (l - d.Np);
        double H[21], gv[6];
#pragma unroll
        for (int i = 0; i < 21; ++i) H[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) gv[i] = 0.0;
        for (int e = d.lm_start[l]; e < d.lm_start[l + 1]; ++e) {
            const int k = d.obs_kf[e];
            const double* Tiw = d.Tiw + (is_pt ? 0 : (size_t)12 * d.K) + 12 * k;
            const double* z = d.obs_z + 3 * (size_t)e;
            double Jp[6], Jl[6] = {0, 0, 0, 0, 0, 0}, n;
            if (is_pt) {
                double g[3];
                to_cam(Tiw, X, g);
                const double e0 = z[0] - (d.cx + d.fx * g[0] / g[2]), e1 = z[1] - (d.cy + d.fy * g[1] / g[2]);
                n = sqrt(e0 * e0 + e1 * e1);
                const double dn = d.homog_th > n ? d.homog_th : n;
                jac_pieces(g, d.fx * e0, d.fy * e1, d.homog_th, Jp);
                row_R(Jp, Tiw, Jl);
#pragma unroll
                for (int i = 0; i < 3; ++i) Jl[i] /= dn;
#pragma unroll
                for (int i = 0; i < 6; ++i) Jp[i] /= dn;
            } else {
                double gp[3], gq[3], JP[6], JQ[6];
                to_cam(Tiw, X, gp); to_cam(Tiw, X + 3, gq);
                const double e0 = z[0] * (d.cx + d.fx * gp[0] / gp[2]) + z[1] * (d.cy + d.fy * gp[1] / gp[2]) + z[2];
                const double e1 = z[0] * (d.cx + d.fx * gq[0] / gq[2]) + z[1] * (d.cy + d.fy * gq[1] / gq[2]) + z[2];
                n = sqrt(e0 * e0 + e1 * e1);
                const double dn = d.homog_th > n ? d.homog_th : n;
                jac_pieces(gp, d.fx * e0, d.fy * e1, d.homog_th, JP);      // both end points with (fx lx, fy ly), :1580-1583 / :1612
                jac_pieces(gq, d.fx * e0, d.fy * e1, d.homog_th, JQ);
                row_R(JP, Tiw, Jl); row_R(JQ, Tiw, Jl + 3);
#pragma unroll
                for (int i = 0; i < 3; ++i) { Jl[i] = Jl[i] * e0 / dn; Jl[3 + i] = Jl[3 + i] * e1 / dn; }
#pragma unroll
                for (int i = 0; i < 6; ++i) Jp[i] = (JP[i] * e0 + JQ[i] * e1) / dn;
            }
            const double w = 1.0 / (1.0 + n * n);      // robustWeightCauchy, auxiliar.cpp:556-559
            err += n * n * w;
            double* r = d.rec + (size_t)REC * e;
#pragma unroll
            for (int i = 0; i < 6; ++i) { r[i] = Jp[i]; r[6 + i] = Jl[i]; }
            r[12] = w; r[13] = n;
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                gv[a] += Jl[a] * n * w;
#pragma unroll
                for (int b = a; b < 6; ++b, ++q) H[q] += Jl[a] * Jl[b] * w;
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) if (a < dim) hm = fmax(hm, fabs(H[diag_q(a)]));
#pragma unroll
        for (int i = 0; i < 21; ++i) d.Hll[(size_t)21 * l + i] = H[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) d.gl[(size_t)6 * l + i] = gv[i];
    }
    __shared__ double sh[LM_NT], shm[LM_NT];
    sh[threadIdx.x] = err; shm[threadIdx.x] = hm;
    __syncthreads();
    for (int s = LM_NT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[threadIdx.x] += sh[threadIdx.x + s]; shm[threadIdx.x] = fmax(shm[threadIdx.x], shm[threadIdx.x + s]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { d.part[blockIdx.x] = sh[0]; d.part[gridDim.x + blockIdx.x] = shm[0]; }
}
// one workgroup per local keyframe: Hpp (packed upper 21) and gp over the keyframe's observations, in a fixed order
constexpr int KF_NT = 256;
template <int NV>
__device__ __forceinline__ void block_sum(double* acc, double* sh /* KF_NT / 64 x NV */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) sh[wv * NV + q] = v;
    }
    __syncthreads();
    if (threadIdx.x < NV) { double v = 0.0; for (int w = 0; w < KF_NT / 64; ++w) v += sh[w * NV + threadIdx.x]; acc[0] = v; }
}
__global__ void __launch_bounds__(KF_NT) k_lba_posesys(LbaDev d) {
    if (d.ctl->done) return;
    const int i = blockIdx.x;
    __shared__ double sh[(KF_NT / 64) * 27];
    double acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.0;
    for (int t = d.kf_start[i] + threadIdx.x; t < d.kf_start[i + 1]; t += KF_NT) {
        const double* r = d.rec + (size_t)REC * d.kf_obs[t];
        const double w = r[12], n = r[13];
        double Jp[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) Jp[a] = r[a];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b, ++q) acc[q] += Jp[a] * Jp[b] * w;
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += Jp[a] * n * w;
    }
    block_sum<27>(acc, sh);
    if (threadIdx.x < 21) d.Hpp[21 * i + threadIdx.x] = acc[0];
    else if (threadIdx.x < 27) d.gp[6 * i + threadIdx.x - 21] = acc[0];
}
// scal[0] = sum of the error partials (fixed order), scal[1] = max |H_ii| over the whole diagonal (:1653-1658)
__device__ void lba_deliver(const LbaDev& d, unsigned long long seq) {      // control block -> mapped mailbox, then the sequence number the host polls
    d.mail->c = *d.ctl;
    __threadfence_system();
    __hip_atomic_store(&d.mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// + the first half of the pass's control (levMarquardtOptimizationLBA :1650-1659, :1882-1900): normalisation of the error as coded, the two
// exits on the error, whether the step of this pass will be applied, the next lambda
__global__ void __launch_bounds__(256) k_lba_reduce(LbaDev d, int nblk, int it, unsigned long long seq) {
    if (d.ctl->done) return;
    __shared__ double sh[256], shm[256];
    double hm = 0.0, e = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) { e += d.part[b]; hm = fmax(hm, d.part[nblk + b]); }
    for (int i = threadIdx.x; i < d.Nkf * 6; i += 256) hm = fmax(hm, fabs(d.Hpp[21 * (i / 6) + diag_q(i % 6)]));
    sh[threadIdx.x] = e; shm[threadIdx.x] = hm;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[threadIdx.x] += sh[threadIdx.x + s]; shm[threadIdx.x] = fmax(shm[threadIdx.x], shm[threadIdx.x + s]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d.scal[0] = sh[0]; d.scal[1] = shm[0];
        LbaCtl* c = d.ctl;
        double err = sh[0];
        if (it == 0) {
            c->err_first = err / (double)(d.Ep + d.El);      // reported only
            err /= 0.0;                    // :1650 as coded (both counters are still 0): +inf, which makes the first comparison of :1894 a "success" 
            c->lambda *= d.variant == 1 ? (double)(int)shm[0] : shm[0];      // :1653-1659; GBA declares `int Hmax` (:2468)
        } else {
            if (d.variant == 1) err /= 0.0;      // GBA :2744 divides by the zero counters in every pass
            else err /= (double)(d.Np + d.Nl);   // :1882 as coded
            if (fabs(err - c->err_prev) < d.min_error_change || err < d.min_error) { c->err = err; c->iters = it; c->done = 1; lba_deliver(d, seq); return; }
        }
        int do_update = 1;
        double lambda_next = c->lambda;
        if (it > 0) { if (err > c->err_prev) { lambda_next = c->lambda / d.lambda_k; do_update = 0; } else lambda_next = c->lambda * d.lambda_k; }
        c->err = err; c->do_update = do_update; c->lambda_next = lambda_next;
    }
}
// pose system before the Schur complement: damped diagonal blocks, unit padding, solver flag
__global__ void k_lba_sysinit(LbaDev d) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (d.ctl->done) {      // the run has ended: the (ungated) factorisation launches behind this one get the identity, not a cleared matrix
        if (t == 0) d.ctrl->solver_ok = 1;
        if (t < d.Ppad) d.sys[(size_t)t * d.ld + t] = 1.0;
        return;
    }
    const double lambda = d.ctl->lambda;
    if (t == 0) { d.ctrl->solver_ok = 1; }
    if (t < d.Nkf * 36) {
        const int i = t / 36, a = (t % 36) / 6, b = t % 6;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        double v = d.Hpp[21 * i + diag_q(lo) + (hi - lo)];
        if (a == b) v += lambda * v;
        d.sys[(size_t)(6 * i + a) * d.ld + 6 * i + b] = v;
    }
    if (t >= d.P && t < d.Ppad) d.sys[(size_t)t * d.ld + t] = 1.0;
}
// 3 x 3 / 6 x 6 SPD inverse from the packed upper triangle (Cholesky, then L^-1, then L^-T L^-1); false if not positive
template <int N>
__device__ bool spd_inv_packed(const double* Hp, double* Dp) {
    double Lm[N][N];
    int q = 0;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = a; b < N; ++b, ++q) Lm[b][a] = Hp[q];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double dj = Lm[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= Lm[j][k] * Lm[j][k];
        if (!(dj > 0.0)) ok = false;
        dj = sqrt(dj);
        Lm[j][j] = dj;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double s = Lm[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= Lm[i][k] * Lm[j][k];
            Lm[i][j] = s / dj;
        }
    }
    double Li[N][N];      // L^-1 (lower)
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
            if (r < c) { Li[r][c] = 0.0; continue; }
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = c; k < r; ++k) s -= Lm[r][k] * Li[k][c];
            Li[r][c] = s / Lm[r][r];
        }
    }
    q = 0;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = a; b < N; ++b, ++q) {
            double s = 0.0;
#pragma unroll
            for (int k = b; k < N; ++k) s += Li[k][a] * Li[k][b];
            Dp[q] = s;
        }
    return ok;
}
template <int N>
__device__ __forceinline__ void sym_mul_packed(const double* Dp, const double* v, double* o) {
    double M[N][N];
    int q = 0;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = a; b < N; ++b, ++q) { M[a][b] = Dp[q]; M[b][a] = Dp[q]; }
#pragma unroll
    for (int a = 0; a < N; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) s += M[a][b] * v[b];
        o[a] = s;
    }
}
// gathers the N-wide packed triangle out of the 6-wide packed storage
template <int N>
__device__ __forceinline__ void load_packed(const double* H21, double lambda, double* Hp) {
    int q = 0;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = a; b < N; ++b, ++q) { const double v = H21[diag_q(a) + (b - a)]; Hp[q] = (a == b) ? v + lambda * v : v; }
}
// thread per landmark: D = (Hll + lambda diag Hll)^-1 and, per observation, u = w D Jl and s = w Jl . (D gl): everything the
// pair kernel and the right-hand side need from the landmark block
template <int N>
__device__ void dinv_landmark(const LbaDev& d, int l, double lambda) {
    double Hp[N * (N + 1) / 2], Dp[N * (N + 1) / 2];
    load_packed<N>(d.Hll + (size_t)21 * l, lambda, Hp);
    if (!spd_inv_packed<N>(Hp, Dp)) d.ctrl->solver_ok = 0;
#pragma unroll
    for (int i = 0; i < N * (N + 1) / 2; ++i) d.Dl[(size_t)21 * l + i] = Dp[i];
    double gv[N], tv[N];
#pragma unroll
    for (int i = 0; i < N; ++i) gv[i] = d.gl[(size_t)6 * l + i];
    sym_mul_packed<N>(Dp, gv, tv);
    for (int e = d.lm_start[l]; e < d.lm_start[l + 1]; ++e) {
        const double* r = d.rec + (size_t)REC * e;
        double Ja[N], ua[N];
#pragma unroll
        for (int i = 0; i < N; ++i) Ja[i] = r[6 + i];
        const double wa = r[12];
        sym_mul_packed<N>(Dp, Ja, ua);
        double sg = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) sg += Ja[i] * tv[i];
        double* u = d.urec + (size_t)8 * e;
#pragma unroll
        for (int i = 0; i < 6; ++i) u[i] = i < N ? wa * ua[i] : 0.0;
        u[6] = wa * sg; u[7] = 0.0;
    }
}
__global__ void __launch_bounds__(LM_NT) k_lba_dinv(LbaDev d) {
    if (d.ctl->done) return;
    const double lambda = d.ctl->lambda;
    const int l = blockIdx.x * LM_NT + threadIdx.x;
    if (l >= d.Np + d.Nl) return;
    if (l < d.Np) dinv_landmark<3>(d, l, lambda); else dinv_landmark<6>(d, l, lambda);
}
// One wave per chunk of a keyframe pair's entries.  A scalar residual makes H_pl rank one, so an entry (observations a, b of
// one landmark) contributes  -(u_a . Jl_b) w_b  Jp_a Jp_b^T  to S[la, lb]: 36 sums per lane, a wave reduction, then one
// add per element and chunk (the only atomics of the iteration; a pair has few chunks).
constexpr int PAIR_CHUNK = 256;
__global__ void __launch_bounds__(64) k_lba_pairs(LbaDev d) {
    if (d.ctl->done) return;
    const int32_t* ck = d.chunk + 4 * (size_t)blockIdx.x;
    const int la = ck[0], lb = ck[1], lane = threadIdx.x;
    double acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.0;
    for (int t = ck[2] + lane; t < ck[3]; t += 64) {
        const int ea = d.pair_ent[2 * (size_t)t], eb = d.pair_ent[2 * (size_t)t + 1];
        const double* ra = d.rec + (size_t)REC * ea;
        const double* rb = d.rec + (size_t)REC * eb;
        const double* ua = d.urec + (size_t)8 * ea;
        double c = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) c += ua[i] * rb[6 + i];      // points carry zeros in the upper three
        c *= rb[12];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double ci = c * ra[i];
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i * 6 + j] += ci * rb[j];
        }
    }
#pragma unroll
    for (int q = 0; q < 36; ++q) {
        double v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        acc[q] = v;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                atomicAdd(d.sys + (size_t)(6 * la + i) * d.ld + 6 * lb + j, -acc[i * 6 + j]);
                if (la != lb) atomicAdd(d.sys + (size_t)(6 * lb + j) * d.ld + 6 * la + i, -acc[i * 6 + j]);
            }
    }
}
// right-hand side of the reduced system: gp - sum over the keyframe's observations of Jp s  (workgroup per local keyframe, fixed order)
__global__ void __launch_bounds__(KF_NT) k_lba_rhs(LbaDev d) {
    if (d.ctl->done) return;
    const int i = blockIdx.x;
    __shared__ double sh[(KF_NT / 64) * 6];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int t = d.kf_start[i] + threadIdx.x; t < d.kf_start[i + 1]; t += KF_NT) {
        const int e = d.kf_obs[t];
        const double* r = d.rec + (size_t)REC * e;
        const double sg = d.urec[(size_t)8 * e + 6];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[a] += r[a] * sg;
    }
    block_sum<6>(acc, sh);
    if (threadIdx.x < 6) d.sys[(size_t)d.Ppad * d.ld + 6 * i + threadIdx.x] = d.gp[6 * i + threadIdx.x] - acc[0];
}
// dx_l = D (gl - sum_a w_a Jl_a (Jp_a . dx_pose(a)))
template <int N>
__device__ void backsub_landmark(const LbaDev& d, int l, double* out) {
    double v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = d.gl[(size_t)6 * l + i];
    for (int e = d.lm_start[l]; e < d.lm_start[l + 1]; ++e) {
        const int la = d.kf_loc[d.obs_kf[e]];
        if (la < 0) continue;
        const double* r = d.rec + (size_t)REC * e;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) s += r[i] * d.x[6 * la + i];
        s *= r[12];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] -= r[6 + i] * s;
    }
    double Dp[N * (N + 1) / 2];
#pragma unroll
    for (int i = 0; i < N * (N + 1) / 2; ++i) Dp[i] = d.Dl[(size_t)21 * l + i];
    sym_mul_packed<N>(Dp, v, out);
}
__global__ void __launch_bounds__(LM_NT) k_lba_backsub(LbaDev d) {
    if (d.ctl->done) return;
    const int l = blockIdx.x * LM_NT + threadIdx.x;
    double nn = 0.0;
    if (l < d.Np) {
        double o[3]; backsub_landmark<3>(d, l, o);
        for (int i = 0; i < 3; ++i) { d.DXl[3 * (size_t)l + i] = o[i]; nn += o[i] * o[i]; }
    } else if (l < d.Np + d.Nl) {
        double o[6]; backsub_landmark<6>(d, l, o);
        for (int i = 0; i < 6; ++i) { d.DXl[3 * (size_t)d.Np + 6 * (size_t)(l - d.Np) + i] = o[i]; nn += o[i] * o[i]; }
    }
    __shared__ double sh[LM_NT];
    sh[threadIdx.x] = nn;
    __syncthreads();
    for (int s = LM_NT / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) d.part[blockIdx.x] = sh[0];
}
// |DX|^2 (:1911) and, when the step is taken and the solve succeeded, the update of :1901-1910
__global__ void __launch_bounds__(256) k_lba_update(LbaDev d, int nblk) {
    if (d.ctl->done) return;
    const int do_update = d.ctl->do_update;
    const bool ok = d.ctrl->solver_ok != 0;
    if (blockIdx.x == 0) {
        __shared__ double sh[256];
        double nn = 0.0;
        for (int b = threadIdx.x; b < nblk; b += 256) nn += d.part[b];
        for (int i = threadIdx.x; i < d.P; i += 256) nn += d.x[i] * d.x[i];
        sh[threadIdx.x] = nn;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
        if (threadIdx.x == 0) d.scal[2] = sh[0];
    }
    if (!ok || !do_update) return;
    const int t = blockIdx.x * 256 + threadIdx.x, nl = 3 * d.Np + 6 * d.Nl;
    if (t < nl) d.Xl[t] += d.DXl[t];
    if (t < d.Nkf) {
        T12 Tp, Td, Tdi, Tc;
        se3_exp(d.Xp + 6 * t, Tp); se3_exp(d.x + 6 * t, Td); se3_inv(Td, Tdi); se3_mul(Tp, Tdi, Tc);
        se3_log(Tc, d.Xp + 6 * t);
    }
}
// the second half of the pass's control (:1901-1932), in a launch of its own: every workgroup of k_lba_update must have applied the step
// before `done` may stop anything
__global__ void k_lba_post(LbaDev d, int it, unsigned long long seq) {
    LbaCtl* c = d.ctl;
    if (c->done) return;
    if (!d.ctrl->solver_ok) { c->failed = 1; c->iters = it; c->done = 1; lba_deliver(d, seq); return; }      // the reference's LDL^T has no such exit: a non-positive pivot ends the run here
    c->dx2 = d.scal[2];
    c->lambda = c->lambda_next;
    if (c->do_update) c->updates += 1;
    c->err_prev = c->err;
    if (it > 0 && sqrt(d.scal[2]) < d.min_error_change) { c->iters = it + 1; c->done = 1; }
    lba_deliver(d, seq);
}
__global__ void k_lba_final(LbaDev d) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d.K) return;
    double* o = d.Tout + 16 * k;
    const int loc = d.kf_loc[k];
    if (loc < 0) { for (int i = 0; i < 16; ++i) o[i] = d.Tmap[16 * k + i]; return; }
    T12 T; se3_exp(d.Xp + 6 * loc, T);
    for (int i = 0; i < 3; ++i) { o[i * 4] = T.R[i * 3]; o[i * 4 + 1] = T.R[i * 3 + 1]; o[i * 4 + 2] = T.R[i * 3 + 2]; o[i * 4 + 3] = T.t[i]; }
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}

}  // namespace
}  // namespace plba

using namespace plba;

extern "C" {

void plba_lba_default_options(plba_lba_options* o) {
    if (!o) return;
    o->lambda_lm = 1e-5;      // SlamConfig::lambdaLbaLM, src/slamConfig.cpp
    o->lambda_k = 10.0;       // SlamConfig::lambdaLbaK
    o->max_iters = 15;        // SlamConfig::maxItersLba
    o->homog_th = 1e-7;       // SlamConfig::homogTh
    o->min_error = 1e-7;      // Config::minError
    o->min_error_change = 1e-7;      // Config::minErrorChange
    o->use_iterate_poses = 0;
    o->variant = 0;
}

int plba_lba_visual(plba_problem* p, const plba_lba_options* opt, int K, const double* T_kf_w16, const int32_t* kf_loc,
                    int Np, double* xyz3, int Nl, double* pq6,
                    int Ep, const int32_t* po_pt, const int32_t* po_kf, const double* uv2,
                    int El, const int32_t* lo_ln, const int32_t* lo_kf, const double* l3,
                    double fx, double fy, double cx, double cy, double* T_out16, uint8_t* pt_moved, uint8_t* ln_moved, plba_lba_stats* st) {
    if (!p) return PLBA_ERR_INVALID;
    if (!opt || K <= 0 || !T_kf_w16 || !kf_loc || Np < 0 || Nl < 0 || Ep < 0 || El < 0 || !T_out16 || !st) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: null or negative argument");
    if (Ep + El == 0) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: no observation");
    if ((Np && !xyz3) || (Nl && !pq6) || (Ep && (!po_pt || !po_kf || !uv2)) || (El && (!lo_ln || !lo_kf || !l3))) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: missing array");
    // local keyframes: indices 0 .. Nkf-1, each once
    int Nkf = 0;
    for (int k = 0; k < K; ++k) if (kf_loc[k] >= 0) ++Nkf;
    std::vector<int32_t> loc_kf(std::max(Nkf, 1), -1);
    for (int k = 0; k < K; ++k) {
        if (kf_loc[k] < 0) continue;
        if (kf_loc[k] >= Nkf || loc_kf[kf_loc[k]] != -1) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: kf_loc must number the local keyframes 0..%d once each", Nkf - 1);
        loc_kf[kf_loc[k]] = k;
    }
    if (Nkf == 0) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: no local keyframe");
    // observation lists are landmark-major, as localBundleAdjustment builds them (src/mapHandler.cpp:1360-1420)
    const int L = Np + Nl, E = Ep + El;
    std::vector<int32_t> lm_start(L + 1, 0), obs_kf(E), kf_start(Nkf + 1, 0), kf_obs;
    std::vector<double> obs_z((size_t)3 * E, 0.0);
    for (int e = 0; e < Ep; ++e) {
        if (po_pt[e] < 0 || po_pt[e] >= Np || po_kf[e] < 0 || po_kf[e] >= K) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: point observation %d out of range", e);
        if (e && po_pt[e] < po_pt[e - 1]) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: point observations must be ordered by point");
        ++lm_start[po_pt[e] + 1]; obs_kf[e] = po_kf[e]; obs_z[3 * (size_t)e] = uv2[2 * e]; obs_z[3 * (size_t)e + 1] = uv2[2 * e + 1];
    }
    for (int e = 0; e < El; ++e) {
        if (lo_ln[e] < 0 || lo_ln[e] >= Nl || lo_kf[e] < 0 || lo_kf[e] >= K) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: line observation %d out of range", e);
        if (e && lo_ln[e] < lo_ln[e - 1]) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: line observations must be ordered by line");
        ++lm_start[Np + lo_ln[e] + 1]; obs_kf[Ep + e] = lo_kf[e];
        for (int i = 0; i < 3; ++i) obs_z[3 * (size_t)(Ep + e) + i] = l3[3 * e + i];
    }
    for (int l = 0; l < L; ++l) lm_start[l + 1] += lm_start[l];
    for (int e = 0; e < E; ++e) if (kf_loc[obs_kf[e]] >= 0) ++kf_start[kf_loc[obs_kf[e]] + 1];
    for (int i = 0; i < Nkf; ++i) kf_start[i + 1] += kf_start[i];
    kf_obs.assign(std::max(kf_start[Nkf], 1), 0);
    { std::vector<int32_t> cur(kf_start.begin(), kf_start.end() - 1); for (int e = 0; e < E; ++e) { const int lc = kf_loc[obs_kf[e]]; if (lc >= 0) kf_obs[cur[lc]++] = e; } }

    // Schur pair entries: for every landmark, every pair of its observations in local keyframes la <= lb (an observation with
    // itself included), grouped by keyframe pair and cut into chunks of <= PAIR_CHUNK entries; constant over the iterations
    std::vector<int32_t> pair_ent, chunk;
    {
        std::vector<int64_t> cnt((size_t)Nkf * Nkf + 1, 0);
        auto for_pairs = [&](auto&& fn) {
            for (int l = 0; l < L; ++l)
                for (int ea = lm_start[l]; ea < lm_start[l + 1]; ++ea) {
                    const int la = kf_loc[obs_kf[ea]];
                    if (la < 0) continue;
                    for (int eb = lm_start[l]; eb < lm_start[l + 1]; ++eb) {
                        const int lb = kf_loc[obs_kf[eb]];
                        if (lb < la) continue;      // the (lb, la) block is the transpose; la == lb keeps every ordered pair (normally just the observation with itself)
                        fn(la, lb, ea, eb);
                    }
                }
        };
        for_pairs([&](int la, int lb, int, int) { ++cnt[(size_t)la * Nkf + lb + 1]; });
        for (size_t i = 0; i < (size_t)Nkf * Nkf; ++i) cnt[i + 1] += cnt[i];
        const int64_t nent = cnt[(size_t)Nkf * Nkf];
        if (nent > INT32_MAX / 2) FAIL(p, PLBA_ERR_INVALID, "plba_lba_visual: too many co-observation pairs");
        pair_ent.assign((size_t)2 * std::max<int64_t>(nent, 1), 0);
        std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
        for_pairs([&](int la, int lb, int ea, int eb) { const int64_t t = cur[(size_t)la * Nkf + lb]++; pair_ent[2 * t] = ea; pair_ent[2 * t + 1] = eb; });
        for (int la = 0; la < Nkf; ++la)
            for (int lb = la; lb < Nkf; ++lb)
                for (int64_t t0 = cnt[(size_t)la * Nkf + lb], t1 = cnt[(size_t)la * Nkf + lb + 1]; t0 < t1; t0 += PAIR_CHUNK) {
                    chunk.push_back(la); chunk.push_back(lb); chunk.push_back((int32_t)t0); chunk.push_back((int32_t)std::min<int64_t>(t1, t0 + PAIR_CHUNK));
                }
    }
    const int nchunk = (int)(chunk.size() / 4);
    if (chunk.empty()) chunk.assign(4, 0);

    HIPCK(p, hipSetDevice(p->device));
    hipStream_t s = p->stream;
    const int P = 6 * Nkf, Ppad = std::max(TILE, (P + TILE - 1) / TILE * TILE), ld = Ppad;
    const size_t sysn = (size_t)(Ppad + TILE) * ld, nl = (size_t)3 * Np + (size_t)6 * Nl;
    const int nblk = (L + LM_NT - 1) / LM_NT;
    DArr<double> dT, dZ, dXp, dXl, dTiw, dRec, dHll, dgl, dDl, dHpp, dgp, dpart, dscal, dDXl, sys, Lfac, xx, Linv, LT32, rd32, Ninv, dTout;
    DArr<int32_t> dloc, dlockf, dlms, dokf, dkfs, dkfo, dpent, dchunk;
    DArr<double> dUrec;
    DArr<int> flags, cflags;
    DArr<Ctrl> ctrl;
    std::vector<double> hT(T_kf_w16, T_kf_w16 + (size_t)16 * K), hXl(std::max<size_t>(nl, 1), 0.0);
    if (Np) memcpy(hXl.data(), xyz3, (size_t)3 * Np * 8);
    if (Nl) memcpy(hXl.data() + 3 * (size_t)Np, pq6, (size_t)6 * Nl * 8);
    std::vector<int32_t> hloc(kf_loc, kf_loc + K);
    {
        DArrStreamScope staged(s, p->have_ctx ? p->ctx.stage : nullptr);      // host vectors above stay alive until the wait below
        HIPCK(p, dT.upload(hT)); HIPCK(p, dZ.upload(obs_z)); HIPCK(p, dXl.upload(hXl)); HIPCK(p, dloc.upload(hloc)); HIPCK(p, dlockf.upload(loc_kf));
        HIPCK(p, dlms.upload(lm_start)); HIPCK(p, dokf.upload(obs_kf)); HIPCK(p, dkfs.upload(kf_start)); HIPCK(p, dkfo.upload(kf_obs));
        HIPCK(p, dpent.upload(pair_ent)); HIPCK(p, dchunk.upload(chunk)); HIPCK(p, dUrec.alloc((size_t)8 * E));
        HIPCK(p, dXp.alloc(P)); HIPCK(p, dTiw.alloc((size_t)24 * K)); HIPCK(p, dRec.alloc((size_t)REC * E)); HIPCK(p, dHll.alloc((size_t)21 * L)); HIPCK(p, dgl.alloc((size_t)6 * L));
        HIPCK(p, dDl.alloc((size_t)21 * L)); HIPCK(p, dHpp.alloc((size_t)21 * Nkf)); HIPCK(p, dgp.alloc(P)); HIPCK(p, dpart.alloc((size_t)2 * nblk)); HIPCK(p, dscal.alloc(4)); HIPCK(p, dDXl.alloc(nl));
        HIPCK(p, sys.alloc(sysn)); HIPCK(p, Lfac.alloc(sysn)); HIPCK(p, xx.alloc(ld)); HIPCK(p, ctrl.alloc(1)); HIPCK(p, dTout.alloc((size_t)16 * K));
        HIPCK(p, Linv.alloc((size_t)(Ppad / TILE) * TILE * TILE)); HIPCK(p, flags.alloc(Ppad / TILE)); HIPCK(p, LT32.alloc((size_t)Ppad * 64)); HIPCK(p, rd32.alloc(Ppad));
        HIPCK(p, cflags.alloc((size_t)(Ppad / 32 + 2) * (Ppad / 32)));
        if (Ppad / 32 <= NINV_MAX_T) HIPCK(p, Ninv.alloc((size_t)2 * Ppad * ld));
        HIPCK(p, plba_stream_wait(s));
    }
    LbaDev d; memset(&d, 0, sizeof d);
    d.K = K; d.Nkf = Nkf; d.Np = Np; d.Nl = Nl; d.Ep = Ep; d.El = El; d.P = P; d.Ppad = Ppad; d.ld = ld;
    d.fx = fx; d.fy = fy; d.cx = cx; d.cy = cy; d.homog_th = opt->homog_th;
    d.Tmap = dT.p; d.kf_loc = dloc.p; d.loc_kf = dlockf.p; d.lm_start = dlms.p; d.obs_kf = dokf.p; d.obs_z = dZ.p; d.kf_start = dkfs.p; d.kf_obs = dkfo.p;
    d.Xp = dXp.p; d.Xl = dXl.p; d.Tiw = dTiw.p; d.rec = dRec.p; d.Hll = dHll.p; d.gl = dgl.p; d.Dl = dDl.p; d.Hpp = dHpp.p; d.gp = dgp.p;
    d.urec = dUrec.p; d.pair_ent = dpent.p; d.chunk = dchunk.p; d.nchunk = nchunk;
    d.part = dpart.p; d.scal = dscal.p; d.DXl = dDXl.p; d.sys = sys.p; d.x = xx.p; d.ctrl = ctrl.p; d.Tout = dTout.p;
    DevBuf dd; memset(&dd, 0, sizeof dd);
    dd.P = P; dd.Ppad = Ppad; dd.ld = ld; dd.sys = sys.p; dd.Lfac = Lfac.p; dd.x = xx.p; dd.ctrl = ctrl.p; dd.Linv = Linv.p; dd.flow_flags = flags.p; dd.LTblk = LT32.p; dd.Linv32 = LT32.p; dd.rdblk = rd32.p;
    dd.fb = (p->opt.factor_block == 64) ? 64 : 32; dd.chol_flags = cflags.p; dd.flow = p->opt.factor_flow != 0; dd.wide = p->opt.wide_steps != 0 && !dd.flow;
    if (Ninv.p) { dd.Ninv = Ninv.p; dd.Nwork = Ninv.p + (size_t)Ppad * ld; }

    // ---- device-side control block + the problem's mapped mailbox ---------------------------------------------------------------------
    DArr<LbaCtl> dctl;
    HIPCK(p, dctl.alloc(1));
    LbaCtl c0; memset(&c0, 0, sizeof c0);
    c0.err_prev = 999999999.9; c0.lambda = opt->lambda_lm; c0.lambda_next = opt->lambda_lm;
    HIPCK(p, plba_h2d(p, dctl.p, &c0, sizeof c0));
    d.ctl = dctl.p; d.mail = reinterpret_cast<LbaMail*>(p->d_mail);
    d.lambda_k = opt->lambda_k; d.min_error = opt->min_error; d.min_error_change = opt->min_error_change; d.variant = opt->variant;
    volatile LbaMail* hm = reinterpret_cast<volatile LbaMail*>(p->h_mail);
    const unsigned long long seq0 = p->mail_seq;      // sequence numbers go on from the problem's: never one the mailbox has held before
    p->mail_seq += (unsigned long long)std::max(opt->max_iters, 0) + 1;

    const bool ltime = (p->opt.diag & PLBA_DIAG_TIMING) != 0;
    if (ltime) HIPCK(p, plba_stream_wait(s));
    const auto lt0 = std::chrono::steady_clock::now();
    auto grid = [](size_t n, int b) { return dim3((unsigned)((n + b - 1) / b)); };
    hipLaunchKernelGGL(k_lba_init, grid(Nkf, 64), dim3(64), 0, s, d);
    memset(st, 0, sizeof *st);
    bool stop = false;
    for (int it = 0; it < opt->max_iters && !stop; ++it) {
        const unsigned long long seq = seq0 + (unsigned long long)it + 1;
        hipLaunchKernelGGL(k_lba_poses, grid(K, 64), dim3(64), 0, s, d, it > 0 ? 1 : 0, opt->use_iterate_poses);
        hipLaunchKernelGGL(k_lba_landmarks, dim3(nblk), dim3(LM_NT), 0, s, d);
        hipLaunchKernelGGL(k_lba_posesys, dim3(Nkf), dim3(KF_NT), 0, s, d);
        hipLaunchKernelGGL(k_lba_reduce, dim3(1), dim3(256), 0, s, d, nblk, it, seq);
        HIPCK(p, hipMemsetAsync(sys.p, 0, sysn * 8, s));
        hipLaunchKernelGGL(k_lba_sysinit, grid(std::max(Nkf * 36, Ppad), 256), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k_lba_dinv, dim3(nblk), dim3(LM_NT), 0, s, d);
        if (nchunk) hipLaunchKernelGGL(k_lba_pairs, dim3(nchunk), dim3(64), 0, s, d);
        hipLaunchKernelGGL(k_lba_rhs, dim3(Nkf), dim3(KF_NT), 0, s, d);
        launch_cholesky(dd, p->opt.use_mfma != 0, it + 1, s);      // (not gated: after the run has ended it factors the unit padding of a cleared system, at most one pass of it)
        launch_trsv_back(dd, p->opt.use_mfma != 0, it + 1, s);
        hipLaunchKernelGGL(k_lba_backsub, dim3(nblk), dim3(LM_NT), 0, s, d);
        hipLaunchKernelGGL(k_lba_update, grid(std::max<size_t>(nl, Nkf), 256), dim3(256), 0, s, d, nblk);
        hipLaunchKernelGGL(k_lba_post, dim3(1), dim3(1), 0, s, d, it, seq);
        // one pass behind: has pass it - 1 ended the run?  (pass `it` is in the queue already, the device never waits for this)
        if (it > 0) {
            const unsigned long long want = seq - 1;
            long spins = 0;
            while (__atomic_load_n(const_cast<const unsigned long long*>(&hm->seq), __ATOMIC_ACQUIRE) < want) {      // (what the mailbox held before is <= seq0)
                if (++spins > (1L << 22)) { HIPCK(p, plba_stream_wait(s)); break; }
            }
            if (hm->c.done) stop = true;
        }
    }
    hipLaunchKernelGGL(k_lba_final, grid(K, 64), dim3(64), 0, s, d);
    HIPCK(p, plba_d2h(p, T_out16, dTout.p, (size_t)16 * K * 8));
    HIPCK(p, plba_d2h(p, hXl.data(), dXl.p, nl * 8));
    LbaCtl cend;
    HIPCK(p, plba_d2h(p, &cend, dctl.p, sizeof cend));
    if (ltime) fprintf(stderr, "[lba] passes + read-back %.3f ms (%d passes)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lt0).count(), cend.done ? cend.iters : opt->max_iters);
    HIPCK(p, hipGetLastError());
    const int iters = cend.done ? cend.iters : opt->max_iters, updates = cend.updates;
    const double err = cend.err, lambda = cend.lambda;
    const bool failed = cend.failed != 0;
    st->err_first = cend.err_first;
    // :1944-1970: a landmark that moved more than 1 cm is flagged (the reference clears its `inlier`)
    for (int i = 0; i < Np; ++i) {
        double n2 = 0.0;
        for (int c = 0; c < 3; ++c) { const double dlt = hXl[3 * (size_t)i + c] - xyz3[3 * (size_t)i + c]; n2 += dlt * dlt; }
        if (pt_moved) pt_moved[i] = sqrt(n2) > 0.01;
    }
    for (int i = 0; i < Nl; ++i) {
        double n2 = 0.0;
        for (int c = 0; c < 6; ++c) { const double dlt = hXl[3 * (size_t)Np + 6 * (size_t)i + c] - pq6[6 * (size_t)i + c]; n2 += dlt * dlt; }
        if (ln_moved) ln_moved[i] = sqrt(n2) > 0.01;
    }
    if (Np) memcpy(xyz3, hXl.data(), (size_t)3 * Np * 8);
    if (Nl) memcpy(pq6, hXl.data() + 3 * (size_t)Np, (size_t)6 * Nl * 8);
    st->iterations = iters; st->updates = updates; st->err_last = err; st->lambda = lambda; st->solver_failed = failed ? 1 : 0;
    return PLBA_OK;
}

}  // extern "C"

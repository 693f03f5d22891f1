// plba_dense.hip — K7: dense fp64 solve of the reduced camera system on gfx950.
//
// Replaces g2o::LinearSolverEigen (sparse simplicial Cholesky of Hschur) by an exact dense LL^T on
// the padded (Ppad x Ppad) symmetric matrix `sys`, right-looking with NB-wide blocks, ONE launch per block step k.
// Workgroup (r,c), k < c <= r <= T (r == T is the right-hand-side block row), forms the panel blocks
// X_r = A(r,k) L(k,k)^-T and X_c, updates A(r,c) -= X_r X_c^T on the matrix cores (v_mfma_f64_16x16x4_f64), and the
// workgroup of (k+1,k+1) factors its freshly updated tile (look-ahead), so the next launch starts at once.
// The right-hand side rides along as an extra block row (row Ppad = bschur), so the forward solve L y = b is a
// by-product of the factorisation; k_trsv_flow finishes with L^T x = y as a single dataflow launch.
// A pivot <= 0 (or NaN) clears ctrl->solver_ok, which g2o reports as a failed linear solve.
//
// Two implementations of a step:
//   k_chol32 (default: factor_block = 32, use_mfma = 1)  panels as products with the published inverse L(k,k)^-1,
//       look-ahead tile factored AND inverted by a four-wave LDS pipeline (see "Look-ahead factorisation" below)
//   k_chol_step<MFMA, NB> (factor_block = 64, or use_mfma = 0: the VALU check path)  panels by substitution against
//       L(k,k)^T staged in LDS, look-ahead by a single-wave potrf
//
// Why it looks like this (rocprofv3 + cycle stamps, MI355X): the path is a chain of P dependent pivots, so everything
// is about the latency of one dependent step.  LDS + s_barrier per pivot column cost ~1300 cycles; one in-order
// wavefront issues an instruction every ~5-8 cycles whatever it is, so the pivot wave must execute as few
// instructions as possible and everything that is not the pivot recurrence belongs to another wave.
#include "plba_internal.h"
#include "plba_factor32_dev.h"

namespace plba {

typedef double double4v __attribute__((ext_vector_type(4)));

template <int NB> struct Blk {
    static constexpr int XS = 2 * NB + 16;   // LDS row stride of the transposed panel tiles XT[k][row]: rows [0,NB) = X_r, [NB,2NB) = X_c
    static constexpr int CS = NB + 2;        // LDS row stride of the look-ahead tile (even: keeps every sub-array 16-byte aligned)
    static constexpr size_t lds_bytes = (size_t)(NB * XS + NB * CS + NB * NB + NB + 2 * NB) * sizeof(double);
};

// x L^T = a for the row held in this lane's registers, column oriented: once x[j] is final every later entry
// is updated independently.  sLT[j*NB + t] = L[t][j] and srd[j] = 1/L[j][j] sit in LDS; every lane reads the same
// address (broadcast), and since nothing writes them the reads pipeline freely.
template <int NB>
__device__ __forceinline__ void trsm_row_lds(double* x, const double* sLT, const double* srd) {
    // software pipelined by hand: the broadcast reads of column j+1 are issued before the FMAs of column j; all reads
    // are 16-byte aligned pairs (ds_read_b128, immediate offsets): a row is read from the even index at or below j+1
    double lc[NB], ln[NB];
    const double2* L2 = reinterpret_cast<const double2*>(sLT);
#pragma unroll
    for (int t2 = 0; t2 < NB / 2; ++t2) { const double2 v = L2[t2]; lc[2 * t2] = v.x; lc[2 * t2 + 1] = v.y; }
    double rd = srd[0];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double rdn = 0.0;
        if (j + 1 < NB) {
            rdn = srd[j + 1];
#pragma unroll
            for (int t2 = (j + 2) / 2; t2 < NB / 2; ++t2) { const double2 v = L2[(j + 1) * (NB / 2) + t2]; ln[2 * t2] = v.x; ln[2 * t2 + 1] = v.y; }
        }
        x[j] *= rd;
        const double xj = x[j];
#pragma unroll
        for (int t = j + 1; t < NB; ++t) x[t] = fma(-xj, lc[t], x[t]);
#pragma unroll
        for (int t = ((j + 2) / 2) * 2; t < NB; ++t) lc[t] = ln[t];
        rd = rdn;
    }
}

// Right-looking Cholesky of an NB x NB tile inside ONE wavefront, lane i (< NB) holding the full SYMMETRIC row i in
// registers.  Column j of the current matrix (A[c][j], all c) is register j across the lanes: one ds_write_b64
// publishes it, broadcast reads fetch it back, and every lane applies  a[i][c] -= a[i][j] * a[c][j] / a[j][j].
// Software pipelined: column j+1 is updated and published first, then the rest of column j's update runs while that
// write travels; the pivot is taken from lane j's register (no LDS wait).  sbuf: 2*NB doubles (double-buffered by
// column parity; single wave, LDS operations of a wave execute in order).  On return a[j] = L[i][j] for j <= i.
template <int NB>
__device__ __forceinline__ bool potrf_inwave(double* a, int lane, double* sbuf) {
    bool bad = false;
    sbuf[lane] = a[0];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double* buf = sbuf + (j & 1) * NB;
        double* nbuf = sbuf + ((j + 1) & 1) * NB;
        // the two values the NEXT column depends on come straight from registers (v_readlane): the pivot A[j][j] and
        // A[j+1][j]; the LDS copy of column j only feeds the bulk update below, off the dependency chain
        const double sj = bcast_lane(a[j], j);
        const bool bj = !(sj > 0.0);
        bad = bad || bj;
        const double f = a[j] * fast_rcp(bj ? 1.0 : sj);
        if (j + 1 < NB) {
            const double s1 = bcast_lane(a[j], j + 1);
            a[j + 1] = fma(-f, s1, a[j + 1]);
            nbuf[lane] = a[j + 1];
        }
        {   // bulk update from the LDS copy of column j, read as aligned pairs
            const double2* b2 = reinterpret_cast<const double2*>(buf);
#pragma unroll
            for (int c2 = (j + 2) / 2; c2 < NB / 2; ++c2) {
                const double2 v = b2[c2];
                if (2 * c2 >= j + 2) a[2 * c2] = fma(-f, v.x, a[2 * c2]);
                a[2 * c2 + 1] = fma(-f, v.y, a[2 * c2 + 1]);
            }
        }
    }
    // L[i][j] = a[j] / sqrt(pivot_j); the pivots are the diagonal entries a[i][i] left by the sweep
    double dg = 1.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) if (lane == j) dg = a[j];
    sbuf[lane] = 1.0 / sqrt(dg > 0.0 ? dg : 1.0);
#pragma unroll
    for (int j = 0; j < NB; ++j) a[j] *= sbuf[j];
    return bad;
}

// store the factor of diagonal block kb: L rows into Lfac (zero above the diagonal), the transposed copy and the
// reciprocal diagonal for the TRSMs of the next step
template <int NB>
__device__ __forceinline__ void store_factor(const DevBuf& d, int kb, const double* a, int lane, bool bad) {
    const int ld = d.ld;
    double* Lrow = d.Lfac + (size_t)(kb * NB + lane) * ld + kb * NB;
    double* LT = d.LTblk + (size_t)kb * NB * NB;
#pragma unroll
    for (int j = 0; j < NB; j += 2) {
        const double v0 = (j <= lane) ? a[j] : 0.0, v1 = (j + 1 <= lane) ? a[j + 1] : 0.0;
        *reinterpret_cast<double2*>(Lrow + j) = make_double2(v0, v1);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) LT[j * NB + lane] = (lane > j) ? a[j] : 0.0;
    double dg = 1.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) if (lane == j) dg = a[j];
    d.rdblk[kb * NB + lane] = 1.0 / dg;
    if (bad && lane == 0) d.ctrl->solver_ok = 0;
}

template <int NB>
__global__ __launch_bounds__(64) void k_potrf0(DevBuf d) {
    __shared__ __attribute__((aligned(16))) double sbuf[2 * NB];
    const int lane = threadIdx.x;
    if (lane >= NB) return;
    double a[NB];
    const double* row = d.sys + (size_t)lane * d.ld;
#pragma unroll
    for (int j = 0; j < NB; j += 2) { const double2 v = *reinterpret_cast<const double2*>(row + j); a[j] = v.x; a[j + 1] = v.y; }
    const bool bad = potrf_inwave<NB>(a, lane, sbuf);
    store_factor<NB>(d, 0, a, lane, bad);
}

template <bool MFMA, int NB>
__global__ __launch_bounds__(256) void k_chol_step(DevBuf d, int k, int T) {
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // 16-byte alignment: ds_read_b128 with immediate offsets
    constexpr int XS = Blk<NB>::XS, CS = Blk<NB>::CS;
    double* sXT = s_dyn;                    // NB x XS
    double* sC = sXT + NB * XS;             // NB x CS
    double* sLT = sC + NB * CS;             // NB x NB
    double* srd = sLT + NB * NB;            // NB
    double* sbuf = srd + NB;                // 2 NB
    const int ld = d.ld;
    const int nt = T - k - 1;
    const int b = blockIdx.x;
    const int ntri = nt * (nt + 1) / 2;
    int rr, cc;
    if (b < ntri) {
        rr = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
        while ((rr + 1) * (rr + 2) / 2 <= b) ++rr;
        while (rr * (rr + 1) / 2 > b) --rr;
        cc = b - rr * (rr + 1) / 2;
    } else {
        rr = nt;
        cc = b - ntri;
    }
    const int r = k + 1 + rr, c = k + 1 + cc;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool have_update = (c < T);
#ifdef PLBA_STAMPS   // diagnostic build only: cycle stamps of the look-ahead workgroup into maxd_part[0..7] (never in the product build)
    const bool stamp = (blockIdx.x == 0 && threadIdx.x == 0 && k == 5);
    unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) do { if (blockIdx.x == 0 && k == 5 && wv == 0) ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
    STAMP(0);
    // stage L(k,k)^T and its reciprocal diagonal in LDS (coalesced)
    {
        const double4* LTg = reinterpret_cast<const double4*>(d.LTblk + (size_t)k * NB * NB);
        for (int i = threadIdx.x; i < NB * NB / 4; i += 256) reinterpret_cast<double4*>(sLT)[i] = LTg[i];
        if (threadIdx.x < NB) srd[threadIdx.x] = d.rdblk[k * NB + threadIdx.x];
    }
    // panel rows: NB == 32: wave 0 solves both panels (lanes 0-31 -> X_r, 32-63 -> X_c); NB == 64: wave 0 -> X_r, wave 1 -> X_c
    const bool second = (NB == 32) ? (lane >= NB) : (wv == 1);
    const bool solver_wave = (NB == 32) ? (wv == 0) : (wv < 2);
    const int rl = lane & (NB - 1);
    const bool active = solver_wave && (!second || (have_update && c != r));
    double x[NB];
    if (active) {   // fetched while the L tile lands in LDS
        const int br = second ? c : r;
        const double* grow = d.sys + (size_t)(br * NB + rl) * ld + k * NB;
#pragma unroll
        for (int j = 0; j < NB; j += 2) { const double2 v = *reinterpret_cast<const double2*>(grow + j); x[j] = v.x; x[j + 1] = v.y; }
    }
    // the C tile this workgroup updates is fetched now, in the shadow of the TRSM
    constexpr int TPD = NB / 16, TPW = TPD * TPD / 4;
    double cold[TPW * 4];
    if (MFMA && have_update) {
        const double* Cg = d.sys + (size_t)(r * NB) * ld + c * NB;
        const int li = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int t = wv * TPW + q, tr = t / TPD, tc = t % TPD;
#pragma unroll
            for (int v = 0; v < 4; ++v) cold[q * 4 + v] = Cg[(size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li];
        }
    }
    __syncthreads();
    STAMP(1);
    if (active) {
        trsm_row_lds<NB>(x, sLT, srd);
        const int xrow = second ? NB + rl : rl;
#pragma unroll
        for (int j = 0; j < NB; ++j) sXT[j * XS + xrow] = x[j];
        if (!second && c == k + 1) {
            // the solved panel block is the final L(r,k); it goes to Lfac, never back into sys: other
            // workgroups of this launch still read the unsolved panel from sys
            double* gout = d.Lfac + (size_t)(r * NB + rl) * ld + k * NB;
#pragma unroll
            for (int j = 0; j < NB; j += 2) *reinterpret_cast<double2*>(gout + j) = make_double2(x[j], x[j + 1]);
        }
    }
    if (!have_update) return;
    __syncthreads();
    STAMP(2);
    const int cb = (c == r) ? 0 : NB;
    double* C = d.sys + (size_t)(r * NB) * ld + c * NB;
    const bool lookahead = (r == k + 1 && c == k + 1);
    if (MFMA) {
        const int li = lane & 15, lk = lane >> 4;
        double4v acc[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q) acc[q] = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int kk = 0; kk < NB / 4; ++kk) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int t = wv * TPW + q, tr = t / TPD, tc = t % TPD;
                const double av = sXT[(kk * 4 + lk) * XS + tr * 16 + li];
                const double bv = sXT[(kk * 4 + lk) * XS + cb + tc * 16 + li];
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[q], 0, 0, 0);
            }
        }
        // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int t = wv * TPW + q, tr = t / TPD, tc = t % TPD;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = tr * 16 + lk + 4 * v, col = tc * 16 + li;
                const double nv = cold[q * 4 + v] - acc[q][v];
                if (lookahead) sC[row * CS + col] = nv;
                else C[(size_t)row * ld + col] = nv;
            }
        }
    } else {
        constexpr int PER = NB * NB / 256;           // outputs per thread, consecutive in a row
        const int row = (threadIdx.x * PER) / NB, c0 = (threadIdx.x * PER) % NB;
        double acc[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) acc[q] = 0.0;
        for (int kk = 0; kk < NB; ++kk) {
            const double av = sXT[kk * XS + row];
#pragma unroll
            for (int q = 0; q < PER; ++q) acc[q] += av * sXT[kk * XS + cb + c0 + q];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const double nv = C[(size_t)row * ld + c0 + q] - acc[q];
            if (lookahead) sC[row * CS + c0 + q] = nv;
            else C[(size_t)row * ld + c0 + q] = nv;
        }
    }
    if (!lookahead) return;
    __syncthreads();
    STAMP(3);
    if (wv == 0 && lane < NB) {
        double a[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) a[j] = sC[lane * CS + j];
        STAMP(4);
        const bool bad = potrf_inwave<NB>(a, lane, sbuf);
        STAMP(5);
        store_factor<NB>(d, k + 1, a, lane, bad);
        STAMP(6);
#ifdef PLBA_STAMPS
        if (stamp) for (int i = 0; i < 7; ++i) d.maxd_part[i] = (double)(ts[i] - ts[0]);
#endif
    }
}

// -------------------------------------------------------------------------------------------------
// 32-wide steps with inverse-based panels (the default path: factor_block = 32, use_mfma = 1)
//
// Same right-looking schedule as k_chol_step, but nothing in a step is a substitution any more:
//   * the look-ahead wavefront factors its 32x32 tile with ALL 64 lanes (lane = (row, column parity), 16 registers)
//     and immediately forms L(k,k)^-1: two 16x16 triangular inverses in the lanes, the off-diagonal block
//     -I22 L21 I11 as eight v_mfma_f64_16x16x4_f64;
//   * every panel block is X = A(r,k) L(k,k)^-T as a 32x32x32 product on the matrix cores, operands straight from
//     global memory (each lane reads 64 contiguous bytes), result scattered into LDS in the operand layout of
//   * the trailing update A(r,c) -= X_r X_c^T (matrix cores again).
// -------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_potrf0_32(DevBuf d) {
    __shared__ __attribute__((aligned(16))) double sC[32 * LS];
    __shared__ __attribute__((aligned(16))) Look32 S;
    for (int idx = threadIdx.x; idx < 1024; idx += 256) sC[(idx >> 5) * LS + (idx & 31)] = d.sys[(size_t)(idx >> 5) * d.ld + (idx & 31)];
    factor32_reset(S, threadIdx.x);
    __syncthreads();
    factor32_tile<false>(d, 0, sC, S, threadIdx.x >> 6, threadIdx.x & 63);
}

// One tile workgroup of block step k.  AUG = false: the tiles of the factorisation proper (trailing update, right-hand-side
// row, look-ahead).  AUG = true: the identity rows, below.  Two instantiations behind one uniform branch, so the code of
// the critical look-ahead workgroup is scheduled exactly as if the identity rows did not exist.
template <bool AUG>
__device__ __forceinline__ void chol32_tile(const DevBuf& d, const int k, const int T, const int r, const int c, const int aj, const int tflags, const bool first_col,
                                            double* sX, double* sC, Look32& S) {
    const bool to_alt = (tflags & 1) != 0, nolook = (tflags & 2) != 0, add_alt = (tflags & 8) != 0, look_here = (tflags & 16) != 0;
    const int ld = d.ld;
    // AUG workgroups (launched when d.Ninv is set) carry IDENTITY rows appended to the augmented system:
    // block row j of  N = I L^-T = L^-T  goes through exactly the panel product / trailing update of the right-hand-side
    // row (X_r = R(j,k) L(k,k)^-T,  R(j,c) -= X_r L(c,k)^T), one short tile each, so the explicit inverse is finished by
    // the same launches that finish the factor and the back-substitution becomes a matrix-vector product (k_back_gemv)
    // instead of a chain of cross-workgroup hops.  Row j starts at launch j (R(j,j) = I, R(j,c>j) = 0: never stored).
    constexpr bool aug = AUG;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const bool diag = (r == c);
    const bool aug_first = aug && aj == k;            // the identity block itself: nothing stored yet
    const double* rowsrc = aug ? d.Nwork + (size_t)(aj * 32) * ld : d.sys + (size_t)(r * 32) * ld;
#ifdef PLBA_STAMPS
    unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0};
#define STAMP32(i) do { if (blockIdx.x == 0 && k == 5) ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP32(i) do {} while (0)
#endif
    STAMP32(0);
    // panel products: wave (p, th) forms rows [16 th, 16 th + 16) of X_r (p = 0) or X_c (p = 1)
    const int p = wv >> 1, th = wv & 1;
    const bool act = !(p == 1 && diag);
    double4v x0 = (double4v){0.0, 0.0, 0.0, 0.0}, x1 = x0;
    if (act) {
        const double2* Ag = reinterpret_cast<const double2*>((p ? d.sys + (size_t)(c * 32) * ld : rowsrc) + (size_t)(th * 16 + li) * ld + k * 32 + lk * 8);
        const double2* B0 = reinterpret_cast<const double2*>(d.Linv32 + (size_t)k * 1024 + li * 32 + lk * 8);
        const double2* B1 = B0 + 16 * 32 / 2;
        double av[8], b0[8], b1[8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double2 va = Ag[s], v0 = B0[s], v1 = B1[s];
            av[2 * s] = va.x; av[2 * s + 1] = va.y; b0[2 * s] = v0.x; b0[2 * s + 1] = v0.y; b1[2 * s] = v1.x; b1[2 * s + 1] = v1.y;
        }
        if (aug_first && p == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = (th * 16 + li == lk * 8 + e) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {     // k index enumerated as 8 * (lane >> 4) + s on both operands
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], b0[s], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], b1[s], x1, 0, 0, 0);
        }
    }
    // the C tile this workgroup updates, fetched in the shadow of the panel products
    const int tr = wv >> 1, tc = wv & 1;
    double* const altb = (tflags & 32) ? d.alt2 : d.alt;
    double* C = (aug ? d.Nwork + (size_t)(aj * 32) * ld : (to_alt ? altb : d.sys) + (size_t)(r * 32) * ld) + c * 32;
    const bool have_update = (c < T);      // false only for the last step's right-hand-side block
    double cold[4] = {0.0, 0.0, 0.0, 0.0};
    if (have_update && !aug_first) {
#pragma unroll
        for (int v = 0; v < 4; ++v) cold[v] = C[(size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li];
        if (!aug && add_alt) {      // the other chain's finished part of this middle tile (twin factorisation)
            const double* Ca = altb + (size_t)(r * 32) * ld + c * 32;
#pragma unroll
            for (int v = 0; v < 4; ++v) cold[v] += Ca[(size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li];
        }
    }
    if (act) {
        const int xr = p * 32 + th * 16 + lk;
#pragma unroll
        for (int v = 0; v < 4; ++v) { sX[(xr + 4 * v) * LS + li] = x0[v]; sX[(xr + 4 * v) * LS + 16 + li] = x1[v]; }
        if (p == 0 && first_col) {      // first_col: c is the step's first trailing column (k + 1 in the ordinary enumeration)
            // the finished panel block L(r,k) goes to Lfac, never back into sys: other workgroups still read the unsolved panel
            double* g = (aug ? d.Ninv + (size_t)(aj * 32) * ld : d.Lfac + (size_t)(r * 32) * ld) + (size_t)(th * 16 + lk) * ld + k * 32 + li;
#pragma unroll
            for (int v = 0; v < 4; ++v) { g[(size_t)(4 * v) * ld] = x0[v]; g[(size_t)(4 * v) * ld + 16] = x1[v]; }
        }
    }
    if (!have_update) return;
    __syncthreads();
    STAMP32(1);
    const bool lookahead = diag && ((r == k + 1 && !nolook) || look_here);
    {
        const int cb = diag ? 0 : 32;
        double4v acc = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sX[(tr * 16 + li) * LS + kk * 4 + lk], sX[(cb + tc * 16 + li) * LS + kk * 4 + lk], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = tr * 16 + lk + 4 * v, col = tc * 16 + li;
            const double nv = cold[v] - acc[v];
            if (lookahead) sC[row * LS + col] = nv;
            else C[(size_t)row * ld + col] = nv;
        }
    }
    if (AUG || !lookahead) return;
    factor32_reset(S, threadIdx.x);
    __syncthreads();
    STAMP32(2);
    factor32_tile<false>(d, r, sC, S, wv, lane);
    STAMP32(3);
#ifdef PLBA_STAMPS
    if (lane == 0 && blockIdx.x == 0 && k == 5) for (int q = 0; q < 4; ++q) d.maxd_part[4 * wv + q] = (double)(ts[q] - ts[0]);
    __syncthreads();
    if (threadIdx.x < 16 && blockIdx.x == 0 && k == 5) d.maxd_part[16 + threadIdx.x] = (double)(long long)(g_lstamp[threadIdx.x] - ts[0]);
#endif
}


__global__ __launch_bounds__(256) void k_chol32(DevBuf d, int k, int T) {
    __shared__ __attribute__((aligned(16))) double sX[64 * LS];      // rows [0,32) = X_r, [32,64) = X_c, [row][k]
    __shared__ __attribute__((aligned(16))) double sC[32 * LS];      // look-ahead tile
    __shared__ __attribute__((aligned(16))) Look32 S;
    // the workgroup's tile: (rr, cc) over the trailing lower triangle, then the right-hand-side block row, then the identity rows
    const int nt = T - k - 1, ntri = nt * (nt + 1) / 2, ntiles = ntri + nt;
    const int nnormal = ntiles > 0 ? ntiles : 1;
    const int b = blockIdx.x;
    int rr, cc, aj = 0;
    const bool aug = b >= nnormal;
    if (aug) {
        const int w = nt > 0 ? nt : 1;
        aj = (b - nnormal) / w; cc = (b - nnormal) % w; rr = nt;
    } else if (b < ntri) {
        rr = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
        while ((rr + 1) * (rr + 2) / 2 <= b) ++rr;
        while (rr * (rr + 1) / 2 > b) --rr;
        cc = b - rr * (rr + 1) / 2;
    } else {
        rr = nt;
        cc = b - ntri;
    }
    if (aug) chol32_tile<true>(d, k, T, k + 1 + rr, k + 1 + cc, aj, 0, cc == 0, sX, sC, S);
    else chol32_tile<false>(d, k, T, k + 1 + rr, k + 1 + cc, 0, 0, cc == 0, sX, sC, S);
}

// -------------------------------------------------------------------------------------------------
// Multi-chain ("twin") factorisation of a block-banded system.  g2o's solver is a SPARSE Cholesky; the reduced camera system is
// banded in keyframe order (tracks span a few keyframes), so stretches of the band that are further apart than its width can
// be eliminated independently.  The band is cut into chains separated by separators at least as wide as the band — two chains
// (the two ends, one separator in the middle) for short systems, four chains and three separators for long ones — and stored
// permuted as [chain 0 | chain 1 | ... | separators] (k_chain_schur writes through d.perm; the last chain is turned around so
// that it, too, is eliminated towards its separator).  The chains are then stretches of ONE right-looking Cholesky whose steps
// do not read each other's data: launch t runs step t of every chain side by side, each workgroup taking its tile from a
// host-built list (the chain's remaining tiles, its adjacent separators, the right-hand-side row, its identity rows).
// Separator blocks have two writers; every second chain therefore accumulates its separator updates in d.alt (zeroed by the
// producer) and is one tile shorter, so that the other chains' LAST step — alone in its launch — folds those in (cold += alt)
// and factors the first separator tile by look-ahead, as any other step does.  Ordinary steps finish the separator region.
// With four chains the separator region is block-tridiagonal in its turn and is taken the same way once more (a second stage of
// two chains, accumulating in d.alt2).  T - 1 dependent launches become  (longest chain) [+ second-stage chain] + (final block - 1):
// 11 -> 7 at configs[2], 43 -> 15 at configs[4].
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol32_list(DevBuf d, int T, const TwinTile* list) {
    __shared__ __attribute__((aligned(16))) double sX[64 * LS];
    __shared__ __attribute__((aligned(16))) double sC[32 * LS];
    __shared__ __attribute__((aligned(16))) Look32 S;
    const TwinTile e = list[blockIdx.x];
    const int k = e.k;
    if (e.aj >= 0) chol32_tile<true>(d, k, T, e.r, e.c, e.aj, 0, (e.flags & 4) != 0, sX, sC, S);
    else chol32_tile<false>(d, k, T, e.r, e.c, 0, e.flags, (e.flags & 4) != 0, sX, sC, S);
}


// -------------------------------------------------------------------------------------------------
// 64-column block steps out of the same 32-wide pieces (wide_steps = 1, the default with factor_block 32 + use_mfma).
// Of the ~21k cycles a 32-wide step costs only ~10k are the pivot sweep; panel products, trailing update, write-out and
// the kernel boundary are paid per launch.  Here a launch retires 64 columns:
//   * panel  X = [A(r,k0) A(r,k1)] M^T  with the 64 x 64 inverse  M = [I11 0; I21 I22]  of the diagonal factor
//     (three 32 x 32 x 32 products per 32-row block, operands straight from global memory),
//   * trailing update  A(r,c) -= X_r X_c^T  with inner dimension 64,
//   * workgroup 0 forms X for the two block rows of the NEXT 64 x 64 diagonal tile, updates its three 32 x 32 tiles into
//     LDS and factors it there as two pipelined sweeps (factor64_lds).
// Right-hand-side row and identity rows (N = L^-T) ride along exactly as in k_chol32.
// -------------------------------------------------------------------------------------------------
constexpr int LX = 66;      // LDS row stride of the 64-wide panels (132 dwords = 4 mod 64: conflict-free operand reads)

struct Wide64Lds {
    double sX[64 * LX];                 // rows [0,32) = X_r, [32,64) = X_c; after the updates: I11 | I22 | L21 scratch
    double sD11[32 * LS], sD21[32 * LS], sD22[32 * LS];
    Look32 S;
};
static_assert(64 * LX >= 3 * 32 * LS, "the inverse / L21 scratch of factor64_lds aliases the panel buffer");

// the panel of one 32-row block: x[0..1] = A0 I11^T (columns 0-31), x[2..3] = A0 I21^T + A1 I22^T (columns 32-63);
// this wave's 16 rows are rows [16 th, 16 th + 16) of the block that starts at `rows`
__device__ __forceinline__ void panel64(const DevBuf& d, const int K, const double* rows, const int th, const int li, const int lk, const int ident /*0: stored rows, 1: [I 0], 2: [0 I]*/, double4v* x) {
    const int ld = d.ld, k0 = 2 * K;
    double a0[8], a1[8], b11[2][8], b21[2][8], b22[2][8];
    const double2* A0 = reinterpret_cast<const double2*>(rows + (size_t)(th * 16 + li) * ld + k0 * 32 + lk * 8);
    const double2* A1 = A0 + 16;
    const double* I11 = d.Linv32 + (size_t)k0 * 1024;
    const double* I22 = I11 + 1024;
    const double* I21 = d.Linv + (size_t)K * TILE * TILE + 32 * TILE;     // rows 32-63, columns 0-31 of the 64 x 64 inverse (row stride 64)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double2 v0 = A0[s], v1 = A1[s];
        a0[2 * s] = v0.x; a0[2 * s + 1] = v0.y; a1[2 * s] = v1.x; a1[2 * s + 1] = v1.y;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const double2 u11 = reinterpret_cast<const double2*>(I11 + (hh * 16 + li) * 32 + lk * 8)[s];
            const double2 u21 = reinterpret_cast<const double2*>(I21 + (hh * 16 + li) * TILE + lk * 8)[s];
            const double2 u22 = reinterpret_cast<const double2*>(I22 + (hh * 16 + li) * 32 + lk * 8)[s];
            b11[hh][2 * s] = u11.x; b11[hh][2 * s + 1] = u11.y; b21[hh][2 * s] = u21.x; b21[hh][2 * s + 1] = u21.y; b22[hh][2 * s] = u22.x; b22[hh][2 * s + 1] = u22.y;
        }
    }
    if (ident) {      // an identity row block that is entering the factorisation: nothing of it is stored yet
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double one = (th * 16 + li == lk * 8 + e) ? 1.0 : 0.0;
            a0[e] = ident == 1 ? one : 0.0; a1[e] = ident == 2 ? one : 0.0;
        }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {     // k index enumerated as 8 * (lane >> 4) + s on both operands
        x[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b11[0][s], x[0], 0, 0, 0);
        x[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b11[1][s], x[1], 0, 0, 0);
        x[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b21[0][s], x[2], 0, 0, 0);
        x[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b21[1][s], x[3], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        x[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b22[0][s], x[2], 0, 0, 0);
        x[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b22[1][s], x[3], 0, 0, 0);
    }
}
// C/D layout -> LDS rows [xrow0 + lk + 4 v] and, if g != null, the finished panel rows in global memory (row stride ld)
__device__ __forceinline__ void panel64_store(const double4v* x, double* sX, const int xrow0, double* g, const int ld, const int li, const int lk) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sX[(xrow0 + lk + 4 * v) * LX + 16 * q + li] = x[q][v];
            if (g) g[(size_t)(lk + 4 * v) * ld + 16 * q + li] = x[q][v];
        }
}

template <bool AUG>
__device__ __forceinline__ void chol64_tile(const DevBuf& d, const int K, const int r, const int c, const int aj, const bool diag, Wide64Lds& W, const int wv, const int lane) {
    const int ld = d.ld, T32 = d.Ppad / 32, c0 = 2 * K + 2;
    const int li = lane & 15, lk = lane >> 4, p = wv >> 1, th = wv & 1, tr = wv >> 1, tc = wv & 1;
    const int ident = AUG ? (aj == 2 * K ? 1 : aj == 2 * K + 1 ? 2 : 0) : 0;
    const double* rowsrc = AUG ? d.Nwork + (size_t)(aj * 32) * ld : d.sys + (size_t)(r * 32) * ld;
    const bool act = !(p == 1 && diag);
    const bool have_update = c < T32;
    double4v x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = (double4v){0.0, 0.0, 0.0, 0.0};
    if (act) panel64(d, K, p ? d.sys + (size_t)(c * 32) * ld : rowsrc, th, li, lk, p ? 0 : ident, x);
    double* C = (AUG ? d.Nwork + (size_t)(aj * 32) * ld : d.sys + (size_t)(r * 32) * ld) + c * 32;
    double cold[4] = {0.0, 0.0, 0.0, 0.0};
    if (have_update && !ident) {
#pragma unroll
        for (int v = 0; v < 4; ++v) cold[v] = C[(size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li];
    }
    if (act) {
        // the finished panel block goes to Lfac / Ninv (never back into sys: other workgroups still read the unsolved panel)
        double* g = nullptr;
        if (p == 0 && c == c0) g = (AUG ? d.Ninv + (size_t)(aj * 32) * ld : d.Lfac + (size_t)(r * 32) * ld) + (size_t)(th * 16) * ld + 2 * K * 32;
        panel64_store(x, W.sX, p * 32 + th * 16, g, ld, li, lk);
    }
    if (!have_update) return;
    __syncthreads();
    const int cb = diag ? 0 : 32;
    double4v acc = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(W.sX[(tr * 16 + li) * LX + kk * 4 + lk], W.sX[(cb + tc * 16 + li) * LX + kk * 4 + lk], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) C[(size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li] = cold[v] - acc[v];
}

// workgroup 0 of block step K: the next 64 x 64 diagonal tile (32-blocks a = 2K+2, b = a+1)
__device__ __forceinline__ void chol64_lookahead(const DevBuf& d, const int K, Wide64Lds& W, const int wv, const int lane) {
    const int ld = d.ld, a = 2 * K + 2, b = a + 1;
    const int li = lane & 15, lk = lane >> 4, p = wv >> 1, th = wv & 1, tr = wv >> 1, tc = wv & 1;
    double4v x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = (double4v){0.0, 0.0, 0.0, 0.0};
    // rows [0,32) of sX = X_b, rows [32,64) = X_a; both are finished panel blocks of L
    panel64(d, K, d.sys + (size_t)((p ? a : b) * 32) * ld, th, li, lk, 0, x);
    double caa[4], cba[4], cbb[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const size_t rr = (size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li;
        caa[v] = d.sys[(size_t)(a * 32) * ld + a * 32 + rr];
        cba[v] = d.sys[(size_t)(b * 32) * ld + a * 32 + rr];
        cbb[v] = d.sys[(size_t)(b * 32) * ld + b * 32 + rr];
    }
    panel64_store(x, W.sX, p * 32 + th * 16, d.Lfac + (size_t)((p ? a : b) * 32 + th * 16) * ld + 2 * K * 32, ld, li, lk);
    __syncthreads();
    double4v maa = (double4v){0.0, 0.0, 0.0, 0.0}, mba = maa, mbb = maa;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const double xa_r = W.sX[(32 + tr * 16 + li) * LX + kk * 4 + lk], xa_c = W.sX[(32 + tc * 16 + li) * LX + kk * 4 + lk];
        const double xb_r = W.sX[(tr * 16 + li) * LX + kk * 4 + lk], xb_c = W.sX[(tc * 16 + li) * LX + kk * 4 + lk];
        maa = __builtin_amdgcn_mfma_f64_16x16x4f64(xa_r, xa_c, maa, 0, 0, 0);
        mba = __builtin_amdgcn_mfma_f64_16x16x4f64(xb_r, xa_c, mba, 0, 0, 0);
        mbb = __builtin_amdgcn_mfma_f64_16x16x4f64(xb_r, xb_c, mbb, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int o = (tr * 16 + lk + 4 * v) * LS + tc * 16 + li;
        W.sD11[o] = caa[v] - maa[v]; W.sD21[o] = cba[v] - mba[v]; W.sD22[o] = cbb[v] - mbb[v];
    }
    __syncthreads();      // tiles complete; the panel buffer is free for the factorisation's scratch
    factor64_lds(d, a, W.sD11, W.sD21, W.sD22, W.S, W.sX, W.sX + 32 * LS, W.sX + 64 * LS, wv, lane);
}

__global__ __launch_bounds__(256) void k_potrf0_64(DevBuf d) {
    __shared__ __attribute__((aligned(16))) Wide64Lds W;
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
        const int rw = idx >> 5, cl = idx & 31;
        W.sD11[rw * LS + cl] = d.sys[(size_t)rw * d.ld + cl];
        W.sD21[rw * LS + cl] = d.sys[(size_t)(32 + rw) * d.ld + cl];
        W.sD22[rw * LS + cl] = d.sys[(size_t)(32 + rw) * d.ld + 32 + cl];
    }
    __syncthreads();
    factor64_lds(d, 0, W.sD11, W.sD21, W.sD22, W.S, W.sX, W.sX + 32 * LS, W.sX + 64 * LS, threadIdx.x >> 6, threadIdx.x & 63);
}

__global__ __launch_bounds__(256) void k_chol64(DevBuf d, int K) {
    __shared__ __attribute__((aligned(16))) Wide64Lds W;
    const int T32 = d.Ppad / 32, c0 = 2 * K + 2, nt = T32 - c0, ntri = nt * (nt + 1) / 2;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int bi = blockIdx.x;
    if (nt > 0) {
        if (bi == 0) { chol64_lookahead(d, K, W, wv, lane); return; }
        bi -= 1;
        if (bi < ntri - 3) {                 // lower-triangular tiles in row-major order; the first three belong to workgroup 0
            const int b = bi + 3;
            int rr = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
            while ((rr + 1) * (rr + 2) / 2 <= b) ++rr;
            while (rr * (rr + 1) / 2 > b) --rr;
            const int cc = b - rr * (rr + 1) / 2;
            chol64_tile<false>(d, K, c0 + rr, c0 + cc, 0, rr == cc, W, wv, lane);
        } else if (bi < ntri - 3 + nt) {     // right-hand-side row
            chol64_tile<false>(d, K, T32, c0 + (bi - (ntri - 3)), 0, false, W, wv, lane);
        } else {                             // identity rows (launched when d.Ninv is set)
            const int e = bi - (ntri - 3 + nt);
            chol64_tile<true>(d, K, T32, c0 + e % nt, e / nt, false, W, wv, lane);
        }
    } else {                                 // last step: only the panels of the right-hand side and of the identity rows
        if (bi == 0) chol64_tile<false>(d, K, T32, T32, 0, true, W, wv, lane);
        else chol64_tile<true>(d, K, T32, T32, bi - 1, true, W, wv, lane);
    }
}

// -------------------------------------------------------------------------------------------------
// Single-launch dataflow factorisation (factor_flow = 1; experimental, NOT the default): the whole LL^T in ONE kernel.
// Measured on MI355X at P = 735: the chain step itself drops to ~16k cycles (from 21.7k with a launch per step), but
// the two tiles the next step needs arrive through three cross-workgroup hops of ~10k cycles each (sc1 store drain,
// flag, poll, sc1 loads: ~4 us), i.e. later than the ~14k-cycle pivot sweep that should hide them, so the chain waits
// and a step costs ~35k cycles (351 us per factorisation against 227 us).  Kept as an option: it is correct (parity
// tests run both) and it is the starting point for a same-XCD / L2-coherent variant with cheaper hops.
//
// Left-looking by tiles: a workgroup per 32 x 32 tile (r,c) of the lower triangle (plus the right-hand-side row
// r = T) accumulates A(r,c) - sum_{k<c} L(r,k) L(c,k)^T on the matrix cores as the L tiles of earlier columns are
// published, then finishes with the inverse of its column's diagonal block, L(r,c) = (...) L(c,c)^-T, and publishes it.
// The chain of diagonal factorisations never leaves ONE workgroup (block 0):
//   step c:  X = pre(c,c-1) L(c-1,c-1)^-T   (L^-1 still in LDS from the previous step; publishes L(c,c-1))
//            A(c,c) = diagpre(c) - X X^T,   then the four-wave pivot pipeline (lookahead_factor32) -> L(c,c)^-1
//   where pre(c,c-1) and diagpre(c) are the two tiles with every OLDER panel already applied; their helper workgroups
//   publish them instead of finishing, and wave 2 of the chain workgroup (the 1/sqrt(pivot) producer, mostly idle)
//   prefetches them into LDS while the pivots of step c-1 run.  So one chain step costs two small products plus the pivot sweep: no kernel boundary, no
//   global-memory latency, and the panel solves and trailing updates of all other tiles run in its shadow.
// Hand-off between workgroups: payload with sc1 (write-through) stores, s_waitcnt vmcnt(0), barrier, then an
// epoch-stamped flag word; consumers poll the flag and read the payload with sc1 loads (no fences, no resets:
// MI355X_MICROARCH.md 'valid forms').  Workgroups are enumerated column-major behind the chain workgroup, so every
// dependency points to a lower block index: with in-order dispatch a waiting workgroup only ever waits for one that is
// resident or finished, whatever the grid size.  Every poll is bounded (SPIN_LIMIT_G): a logic error ends as a failed
// solve, never as a hung device.
// -------------------------------------------------------------------------------------------------
constexpr int SPIN_LIMIT_G = 1 << 18;
constexpr int FLOW_THREADS = 256;

__device__ __forceinline__ bool wait_flag_ge(const int* f, int want) {
    int spins = 0;
    while (__hip_atomic_load(const_cast<int*>(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_LIMIT_G) return false;
    }
    return true;
}
__device__ __forceinline__ void set_flag(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// tflag[r * T + c]: 2 e + 1 = "pre" (older panels applied, published into sys), 2 e + 2 = L(r,c) final in Lfac.
// iflag[c]: 2 e + 2 = L(c,c)^-1 in Linv32.
__global__ __launch_bounds__(FLOW_THREADS) void k_chol_flow(DevBuf d, int T, int epoch) {
    __shared__ __attribute__((aligned(16))) double sX[32 * LS];       // X of the chain step / staging tile of a helper
    __shared__ __attribute__((aligned(16))) double sC[32 * LS];
    __shared__ __attribute__((aligned(16))) double sLinv[32 * LS];
    __shared__ __attribute__((aligned(16))) double sPreA[32 * LS], sPreD[32 * LS];
    __shared__ __attribute__((aligned(16))) Look32 S;
    __shared__ int s_bad;
    const int ld = d.ld;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int tr = wv >> 1, tc = wv & 1;
    int* tflag = d.chol_flags;
    int* iflag = d.chol_flags + (size_t)(T + 1) * T;
    const int PRE = 2 * epoch + 1, FIN = 2 * epoch + 2;

    if (blockIdx.x == 0) {
        // ------------------------------------------------ the chain workgroup ------------------------------------------------
        if (threadIdx.x == 0) s_bad = 0;
        for (int c = 0; c < T; ++c) {
            CSTAMP(16);
#ifdef PLBA_STAMPS
            if (threadIdx.x == 0 && c == 6) g_lstamp[20] = __builtin_readcyclecounter();
#endif
            if (c == 0) {
                for (int idx = threadIdx.x; idx < 1024; idx += 256) sC[(idx >> 5) * LS + (idx & 31)] = d.sys[(size_t)(idx >> 5) * ld + (idx & 31)];
            } else {
                // X = pre(c,c-1) Linv(c-1)^T: tile (tr, tc), 8 k-steps, both operands already in LDS
                double4v x = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(sPreA[(tr * 16 + li) * LS + 4 * s + lk], sLinv[(tc * 16 + li) * LS + 4 * s + lk], x, 0, 0, 0);
                double* Lg = d.Lfac + (size_t)(c * 32) * ld + (c - 1) * 32;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = tr * 16 + lk + 4 * v, col = tc * 16 + li;
                    sX[row * LS + col] = x[v];
                    gstore_sc1(&Lg[(size_t)row * ld + col], x[v]);
                }
                __syncthreads();
                double4v acc = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sX[(tr * 16 + li) * LS + 4 * s + lk], sX[(tc * 16 + li) * LS + 4 * s + lk], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = tr * 16 + lk + 4 * v, col = tc * 16 + li;
                    sC[row * LS + col] = sPreD[row * LS + col] - acc[v];
                }
                drain_stores();      // L(c,c-1) is out before the flag below
            }
            CSTAMP(17);
            look32_reset(S, threadIdx.x);
            __syncthreads();
            CSTAMP(18);
            if (c > 0 && threadIdx.x == 0) set_flag(&tflag[(size_t)c * T + c - 1], FIN);
            NextTiles nt;
            nt.active = (c + 1 < T); nt.done = false; nt.failed = false; nt.want = PRE; nt.ld = ld; nt.stamp = (c == 5);
            nt.fa = &tflag[(size_t)(c + 1) * T + c]; nt.fd = &tflag[(size_t)(c + 1) * T + c + 1];
            nt.A = d.sys + (size_t)((c + 1) * 32) * ld + c * 32; nt.D = nt.A + 32;
            nt.sA = sPreA; nt.sD = sPreD;
            lookahead_factor32<true>(d, c, sC, S, wv, lane, sLinv, &nt);
            if (nt.failed && lane == 0) s_bad = 1;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            CSTAMP(19);
            if (threadIdx.x == 0) set_flag(&iflag[c], FIN);
        }
#ifdef PLBA_STAMPS
        __syncthreads();
        if (threadIdx.x < 32) d.maxd_part[threadIdx.x] = (double)(long long)(g_lstamp[threadIdx.x] - g_lstamp[16]);
#endif
        if (threadIdx.x == 0 && s_bad) d.ctrl->solver_ok = 0;
        return;
    }

    // ---------------------------------------------------- helper workgroups ----------------------------------------------------
    int c = 0, off = (int)blockIdx.x - 1;
    while (off >= T - c + 1) { off -= T - c + 1; ++c; }       // column-major enumeration: column c holds rows c..T
    const int r = c + off;
    if (r == 0) return;                                        // (0,0) belongs to the chain
    const bool diagpre = (r == c), pre = (r == c + 1 && r < T);
    const int kend = diagpre ? c - 1 : c;
    bool ok = true;
    double4v acc = (double4v){0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < kend; ++k) {
        if (lane == 0) ok = wait_flag_ge(&tflag[(size_t)r * T + k], FIN) && (diagpre || wait_flag_ge(&tflag[(size_t)c * T + k], FIN)) && ok;
        ok = __all(ok);
        const double* Lr = d.Lfac + (size_t)(r * 32 + tr * 16 + li) * ld + k * 32 + lk * 8;
        const double* Lc = d.Lfac + (size_t)(c * 32 + tc * 16 + li) * ld + k * 32 + lk * 8;
        double a[8], b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { a[s] = gload_sc1(Lr + s); b[s] = gload_sc1(Lc + s); }
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);   // k enumerated as 8 (lane >> 4) + s
    }
    double val[4];
    {
        const double* A = d.sys + (size_t)(r * 32) * ld + c * 32;
#pragma unroll
        for (int v = 0; v < 4; ++v) val[v] = A[(size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li] - acc[v];
    }
    if (diagpre || pre) {
        double* A = d.sys + (size_t)(r * 32) * ld + c * 32;
#pragma unroll
        for (int v = 0; v < 4; ++v) gstore_sc1(&A[(size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li], val[v]);
        drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) { set_flag(&tflag[(size_t)r * T + c], PRE); if (!ok) d.ctrl->solver_ok = 0; }
        return;
    }
    // full tile: L(r,c) = val Linv(c)^T
#pragma unroll
    for (int v = 0; v < 4; ++v) sX[(tr * 16 + lk + 4 * v) * LS + tc * 16 + li] = val[v];
    if (lane == 0) ok = wait_flag_ge(&iflag[c], FIN) && ok;
    ok = __all(ok);
    __syncthreads();
    {
        const double* Ig = d.Linv32 + (size_t)c * 1024 + (size_t)(tc * 16 + li) * 32 + lk * 8;
        double b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = gload_sc1(Ig + s);
        double4v x = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) x = __builtin_amdgcn_mfma_f64_16x16x4f64(sX[(tr * 16 + li) * LS + lk * 8 + s], b[s], x, 0, 0, 0);
        double* Lg = d.Lfac + (size_t)(r * 32) * ld + c * 32;
#pragma unroll
        for (int v = 0; v < 4; ++v) gstore_sc1(&Lg[(size_t)(tr * 16 + lk + 4 * v) * ld + tc * 16 + li], x[v]);
    }
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) { set_flag(&tflag[(size_t)r * T + c], FIN); if (!ok) d.ctrl->solver_ok = 0; }
}

// Linv[k] (64 x 64, for the back-substitution) from the 32 x 32 inverses the factorisation published:
//   [A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1  C^-1]
__global__ __launch_bounds__(256) void k_inv_diag32(DevBuf d) {
    constexpr int FB = 32, CS = 34;
    __shared__ double sAi[FB * CS], sCi[FB * CS], sB[FB * CS], sT[FB * CS];
    const int k = blockIdx.x, ld = d.ld;
    double* out = d.Linv + (size_t)k * TILE * TILE;
    for (int idx = threadIdx.x; idx < FB * FB; idx += 256) {
        const int rw = idx >> 5, cl = idx & 31;
        sAi[rw * CS + cl] = d.Linv32[(size_t)(2 * k) * 1024 + idx];
        sCi[rw * CS + cl] = d.Linv32[(size_t)(2 * k + 1) * 1024 + idx];
        sB[rw * CS + cl] = d.Lfac[(size_t)(k * TILE + FB + rw) * ld + k * TILE + cl];
    }
    __syncthreads();
    const int row = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    {
        double acc[4] = {0, 0, 0, 0};
        for (int q = 0; q < FB; ++q) { const double bv = sB[row * CS + q];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += bv * sAi[q * CS + c0 + e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) sT[row * CS + c0 + e] = acc[e];
    }
    __syncthreads();
    double acc[4] = {0, 0, 0, 0};
    for (int q = 0; q < FB; ++q) { const double cv = sCi[row * CS + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += cv * sT[q * CS + c0 + e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        out[(FB + row) * TILE + c0 + e] = -acc[e];
        out[row * TILE + c0 + e] = sAi[row * CS + c0 + e];
        out[(FB + row) * TILE + FB + c0 + e] = sCi[row * CS + c0 + e];
        out[row * TILE + FB + c0 + e] = 0.0;
    }
}

// Linv[k] = L(k,k)^-1 for every 64x64 diagonal tile of the factor (for the back-substitution).
// NB == 64: one solve against the staged tile (lane c = column c).  NB == 32: assembled from the two 32-blocks,
//   [A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1  C^-1].
template <int NB>
__global__ __launch_bounds__(256) void k_inv_diag(DevBuf d) {
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    const int k = blockIdx.x, ld = d.ld;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* out = d.Linv + (size_t)k * TILE * TILE;
    if (NB == 64) {
        double* sLT = s_dyn;                 // 64 x 64
        double* srd = sLT + 64 * 64;
        const double4* LTg = reinterpret_cast<const double4*>(d.LTblk + (size_t)k * 64 * 64);
        for (int i = threadIdx.x; i < 64 * 64 / 4; i += 256) reinterpret_cast<double4*>(sLT)[i] = LTg[i];
        if (threadIdx.x < 64) srd[threadIdx.x] = d.rdblk[k * 64 + threadIdx.x];
        __syncthreads();
        if (wv == 0) {
            double y[64];
#pragma unroll
            for (int t = 0; t < 64; ++t) y[t] = (t == lane) ? 1.0 : 0.0;
            trsm_row_lds<64>(y, sLT, srd);
#pragma unroll
            for (int t = 0; t < 64; ++t) out[t * TILE + lane] = y[t];      // Linv[t][c]
        }
    } else {
        constexpr int FB = 32, CS = 34;
        double* sAi = s_dyn; double* sCi = sAi + FB * CS; double* sB = sCi + FB * CS; double* sT = sB + FB * CS;
        double* sLT = sT + FB * CS;          // 2 x 32 x 32
        double* srd = sLT + 2 * FB * FB;     // 2 x 32
        const double4* LTg = reinterpret_cast<const double4*>(d.LTblk + (size_t)(2 * k) * FB * FB);
        for (int i = threadIdx.x; i < 2 * FB * FB / 4; i += 256) reinterpret_cast<double4*>(sLT)[i] = LTg[i];
        if (threadIdx.x < 2 * FB) srd[threadIdx.x] = d.rdblk[2 * k * FB + threadIdx.x];
        for (int idx = threadIdx.x; idx < FB * FB; idx += 256) {
            const int rw = idx >> 5, cl = idx & 31;
            sB[rw * CS + cl] = d.Lfac[(size_t)(k * TILE + FB + rw) * ld + k * TILE + cl];
        }
        __syncthreads();
        if (wv == 0) {
            const int hi = lane >= FB ? 1 : 0, cidx = lane & (FB - 1);
            double y[FB];
#pragma unroll
            for (int t = 0; t < FB; ++t) y[t] = (t == cidx) ? 1.0 : 0.0;
            trsm_row_lds<FB>(y, sLT + hi * FB * FB, srd + hi * FB);
            double* dst = hi ? sCi : sAi;          // dst[t][c] = (L^-1)(t, c)
#pragma unroll
            for (int t = 0; t < FB; ++t) dst[t * CS + cidx] = y[t];
        }
        __syncthreads();
        const int row = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
        {
            double acc[4] = {0, 0, 0, 0};
            for (int q = 0; q < FB; ++q) { const double bv = sB[row * CS + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += bv * sAi[q * CS + c0 + e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) sT[row * CS + c0 + e] = acc[e];
        }
        __syncthreads();
        double acc[4] = {0, 0, 0, 0};
        for (int q = 0; q < FB; ++q) { const double cv = sCi[row * CS + q];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += cv * sT[q * CS + c0 + e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            out[(FB + row) * TILE + c0 + e] = -acc[e];
            out[row * TILE + c0 + e] = sAi[row * CS + c0 + e];
            out[(FB + row) * TILE + FB + c0 + e] = sCi[row * CS + c0 + e];
            out[row * TILE + FB + c0 + e] = 0.0;
        }
    }
}

// L^T x = y as a dataflow over tile columns, one workgroup per column c (launched in descending c):
//   x_c = L(c,c)^-T ( y_c - sum_{k>c} L(k,c)^T x_k )
// Workgroup c folds in x_k as soon as workgroup k has published it (sc1 stores + epoch-stamped flag word per
// column, sc1 loads on the consumer: no fences, no reset); tiles of one column are streamed by their own CU.
__global__ __launch_bounds__(256) void k_trsv_flow(DevBuf d, int T, int epoch) {
    __shared__ double z[TILE];
    __shared__ double sx[TILE];
    __shared__ double part[4][TILE];
    __shared__ int s_fail;
    const int c = T - 1 - blockIdx.x;
    const int ld = d.ld;
    const int t = threadIdx.x & 63, jg = threadIdx.x >> 6;
    if (threadIdx.x < TILE) z[threadIdx.x] = d.Lfac[(size_t)d.Ppad * ld + c * TILE + threadIdx.x];
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();
    for (int k = T - 1; k > c; --k) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(&d.flow_flags[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { s_fail = 1; break; }     // never hang the device on a logic error
            }
        }
        __syncthreads();
        if (s_fail) break;
        // agent-scope (sc1) loads on top of the acquire: never served from a stale L1 line
        if (threadIdx.x < TILE) sx[threadIdx.x] = __hip_atomic_load(&d.x[k * TILE + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const double* Lt = d.Lfac + (size_t)(k * TILE + jg * 16) * ld + c * TILE + t;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += Lt[(size_t)j * ld] * sx[jg * 16 + j];
        part[jg][t] = s;
        __syncthreads();
        if (threadIdx.x < TILE) z[threadIdx.x] -= (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        __syncthreads();
    }
    // x_c[t] = sum_r Linv[r][t] z[r]
    {
        const double* Li = d.Linv + (size_t)c * TILE * TILE + (size_t)(jg * 16) * TILE + t;
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += Li[r * TILE] * z[jg * 16 + r];
        part[jg][t] = s;
    }
    __syncthreads();
    if (threadIdx.x < TILE) {
        double v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (s_fail) v = 0.0;
        __hip_atomic_store(&d.x[c * TILE + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_fail) d.ctrl->solver_ok = 0;
        // x_c went out with sc1 (write-through) stores and this wave drained them (vmcnt(0) above) before the barrier;
        // consumers read it with sc1 loads only, so no release / acquire fence is needed (MI355X_MICROARCH.md,
        // 'Valid forms': 8-byte agent atomics on both sides, flag after the storing wave's wait)
        __hip_atomic_store(&d.flow_flags[c], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NB>
static void launch_cholesky_nb(const DevBuf& d, bool use_mfma, hipStream_t s) {
    const int T = d.Ppad / NB;
    const size_t sh = Blk<NB>::lds_bytes;
    if (ensure_dyn_lds(reinterpret_cast<const void*>(k_chol_step<true, NB>), (int)sh) != hipSuccess ||
        ensure_dyn_lds(reinterpret_cast<const void*>(k_chol_step<false, NB>), (int)sh) != hipSuccess) return;      // surfaces at the caller's hipGetLastError
    hipLaunchKernelGGL(k_potrf0<NB>, dim3(1), dim3(64), 0, s, d);
    for (int k = 0; k < T; ++k) {
        const int nt = T - k - 1;
        const int tiles = nt * (nt + 1) / 2 + nt;
        const int grid = tiles > 0 ? tiles : 1;
        if (use_mfma) hipLaunchKernelGGL((k_chol_step<true, NB>), dim3(grid), dim3(256), sh, s, d, k, T);
        else hipLaunchKernelGGL((k_chol_step<false, NB>), dim3(grid), dim3(256), sh, s, d, k, T);
    }
}
static bool inverse_panels(const DevBuf& d, bool use_mfma) { return d.fb == 32 && use_mfma; }
void launch_twin_cholesky(const DevBuf& d, const TwinView& tv, hipStream_t s) {
    const int T = tv.T, m0 = tv.m0;
    for (int t = 0; t < tv.nlaunch; ++t) {
        const int n = tv.off[t + 1] - tv.off[t];
        if (n > 0) hipLaunchKernelGGL(k_chol32_list, dim3(n), dim3(256), 0, s, d, T, tv.list + tv.off[t]);
    }
    for (int k = m0; k < T - 1; ++k) {      // the last step (panels only) is folded into k_back_gemv, as in launch_cholesky
        const int nt = T - k - 1;
        const int tiles = nt * (nt + 1) / 2 + nt;
        hipLaunchKernelGGL(k_chol32, dim3((tiles > 0 ? tiles : 1) + (k + 1) * (nt > 0 ? nt : 1)), dim3(256), 0, s, d, k, T);
    }
}
void launch_cholesky(const DevBuf& d, bool use_mfma, int epoch, hipStream_t s, bool tile0_done) {
    if (inverse_panels(d, use_mfma) && d.flow) {
        const int T = d.Ppad / 32;
        const int tiles = (T + 1) * (T + 2) / 2 - 1;        // sum_{c<T} (T - c + 1): rows c..T of every column
        hipLaunchKernelGGL(k_chol_flow, dim3(1 + tiles), dim3(FLOW_THREADS), 0, s, d, T, epoch);
        return;
    }
    if (inverse_panels(d, use_mfma) && d.wide) {
        const int T32 = d.Ppad / 32, T64 = d.Ppad / TILE;
        hipLaunchKernelGGL(k_potrf0_64, dim3(1), dim3(256), 0, s, d);
        for (int K = 0; K < T64; ++K) {
            const int nt = T32 - (2 * K + 2);
            const int normal = nt > 0 ? 1 + (nt * (nt + 1) / 2 - 3) + nt : 1;
            const int aug = d.Ninv ? (2 * K + 2) * (nt > 0 ? nt : 1) : 0;
            hipLaunchKernelGGL(k_chol64, dim3(normal + aug), dim3(256), 0, s, d, K);
        }
        return;
    }
    if (inverse_panels(d, use_mfma)) {
        const int T = d.Ppad / 32;
        if (!tile0_done) hipLaunchKernelGGL(k_potrf0_32, dim3(1), dim3(256), 0, s, d);
        const int kend = (d.Ninv && !d.wide) ? T - 1 : T;      // with the explicit inverse the last step (panels only) is folded into k_back_gemv
        for (int k = 0; k < kend; ++k) {
            const int nt = T - k - 1;
            const int tiles = nt * (nt + 1) / 2 + nt;
            hipLaunchKernelGGL(k_chol32, dim3((tiles > 0 ? tiles : 1) + (d.Ninv ? (k + 1) * (nt > 0 ? nt : 1) : 0)), dim3(256), 0, s, d, k, T);
        }
        return;
    }
    if (d.fb == 64) launch_cholesky_nb<64>(d, use_mfma, s);
    else launch_cholesky_nb<32>(d, use_mfma, s);
}

// x = L^-T y = N y with the explicit inverse the factorisation launches left in Ninv: a wave per row, no dependencies.
// The LAST block step of the factorisation (no trailing matrix left: only the panels  y_T = b_T M^T  and  N(:,T) = R(:,T) M^T,
// M = L(T,T)^-1) is not launched when this kernel follows; it is folded in here: with  z = M^T (M b_T)  (32 values, formed
// once per workgroup from the published inverse) row c of the product ends with  R(c,T) . z,  and the last 32 rows ARE z.
__global__ __launch_bounds__(256) void k_back_gemv(DevBuf d, int fold_last) {
    __shared__ double sM[32 * 33], sr[32], sy[32], sz[32];
    const int c0 = fold_last ? d.Ppad - 32 : d.Ppad;             // first column of the last block (fold_last = 0: every step was launched)
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // the grid covers Ppad rows exactly
    const double* Nr = d.Ninv + (size_t)c * d.ld;
    const double* y = d.Lfac + (size_t)d.Ppad * d.ld;
    double s0 = 0.0, s1 = 0.0;
    int r = (c & ~31) + lane;
    for (; r + 64 < c0; r += 128) { s0 = fma(Nr[r], y[r], s0); s1 = fma(Nr[r + 64], y[r + 64], s1); }
    if (r < c0) s0 = fma(Nr[r], y[r], s0);
    const double tw = (fold_last && c < c0 && lane < 32) ? d.Nwork[(size_t)c * d.ld + c0 + lane] : 0.0;      // R(c, T): the row's unsolved last block
    if (fold_last) {
        for (int idx = threadIdx.x; idx < 1024; idx += 256) sM[(idx >> 5) * 33 + (idx & 31)] = d.Linv32[(size_t)(c0 >> 5) * 1024 + idx];
        if (threadIdx.x < 32) sr[threadIdx.x] = d.sys[(size_t)d.Ppad * d.ld + c0 + threadIdx.x];
    } else if (threadIdx.x < 32) sz[threadIdx.x] = 0.0;
    __syncthreads();
    if (fold_last && threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) t = fma(sM[threadIdx.x * 33 + q], sr[q], t);
        sy[threadIdx.x] = t;
    }
    __syncthreads();
    if (fold_last && threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) t = fma(sM[i * 33 + threadIdx.x], sy[i], t);
        sz[threadIdx.x] = t;
    }
    __syncthreads();
    double v = s0 + s1 + (lane < 32 ? tw * sz[lane] : 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) d.x[d.xmap ? d.xmap[c] : c] = c < c0 ? v : sz[c - c0];      // xmap: the system is stored permuted (twin factorisation)
}

void launch_trsv_back(const DevBuf& d, bool use_mfma, int epoch, hipStream_t s) {
    const int T = d.Ppad / TILE;
    if (inverse_panels(d, use_mfma) && !d.flow && d.Ninv) {
        hipLaunchKernelGGL(k_back_gemv, dim3((d.Ppad + 3) / 4), dim3(256), 0, s, d, d.wide ? 0 : 1);
        return;
    }
    if (inverse_panels(d, use_mfma)) {
        hipLaunchKernelGGL(k_inv_diag32, dim3(T), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k_trsv_flow, dim3(T), dim3(256), 0, s, d, T, epoch);
        return;
    }
    const size_t sh64 = (size_t)(64 * 64 + 64) * sizeof(double), sh32 = (size_t)(4 * 32 * 34 + 2 * 32 * 32 + 64) * sizeof(double);
    if (ensure_dyn_lds(reinterpret_cast<const void*>(k_inv_diag<64>), (int)sh64) != hipSuccess ||
        ensure_dyn_lds(reinterpret_cast<const void*>(k_inv_diag<32>), (int)sh32) != hipSuccess) return;      // surfaces at the caller's hipGetLastError
    if (d.fb == 64) hipLaunchKernelGGL(k_inv_diag<64>, dim3(T), dim3(256), sh64, s, d);
    else hipLaunchKernelGGL(k_inv_diag<32>, dim3(T), dim3(256), sh32, s, d);
    hipLaunchKernelGGL(k_trsv_flow, dim3(T), dim3(256), 0, s, d, T, epoch);
}

// out (cols x cols, row-major, leading dimension ldo) = A^T A for a column-major rows x cols matrix
__global__ void k_ata(const double* __restrict__ A, int rows, int cols, double* __restrict__ out, int ldo) {
    __shared__ double sA[16][17], sB[16][17];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int ca = blockIdx.y * 16 + ty, cb = blockIdx.x * 16 + tx;
    double acc = 0.0;
    for (int r0 = 0; r0 < rows; r0 += 16) {
        const int ra = r0 + tx;
        const int colA = blockIdx.y * 16 + ty, colB = blockIdx.x * 16 + ty;
        sA[ty][tx] = (ra < rows && colA < cols) ? A[(size_t)colA * rows + ra] : 0.0;
        sB[ty][tx] = (ra < rows && colB < cols) ? A[(size_t)colB * rows + ra] : 0.0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += sA[ty][q] * sB[tx][q];
        __syncthreads();
    }
    if (ca < cols && cb < cols) out[(size_t)ca * ldo + cb] = acc;
}
void launch_ata(const double* A, int rows, int cols, double* out, int ldo, hipStream_t s) {
    dim3 grid((cols + 15) / 16, (cols + 15) / 16), block(16, 16);
    hipLaunchKernelGGL(k_ata, grid, block, 0, s, A, rows, cols, out, ldo);
}

}  // namespace plba

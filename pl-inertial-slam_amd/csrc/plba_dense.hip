// plba_dense.hip — K7: dense fp64 solve of the reduced camera system on gfx950.
//
// Replaces g2o::LinearSolverEigen (sparse simplicial Cholesky of Hschur) by an exact
// dense LL^T on the padded (Ppad x Ppad, 64-wide tiles) symmetric matrix `sys`, right-looking:
//   step k:  k_chol_diag   one workgroup factors the 64x64 diagonal tile in LDS
//            k_chol_step   one workgroup per trailing tile (r,c): both panel tiles are solved by
//                          substitution (a lane per row, row in registers), then the tile update
//                          C -= X_r X_c^T runs on the matrix cores (v_mfma_f64_16x16x4_f64), or on
//                          the VALU when use_mfma = 0 (cross-check path for the tests)
// The right-hand side rides along as an extra tile row (row Ppad = bschur), so the forward solve
// L y = b is a by-product of the factorisation; k_trsv_back finishes with L^T x = y.
// A pivot <= 0 (or NaN) clears ctrl->solver_ok, which g2o reports as a failed linear solve.
#include "plba_internal.h"

namespace plba {

typedef double double4v __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(4))) cdouble;   // constant address space: uniform loads take the scalar path

constexpr int FB = 32;    // factorisation block (in-wave potrf / trsm); the back-substitution works on 64-wide tiles
constexpr int XS = 80;    // LDS row stride of the transposed panel tiles XT[k][row]  (rows 0..31 = X_r, 32..63 = X_c)
constexpr int CS = FB + 1;

// x L^T = a for the row held in this lane's registers, column oriented: once x[j] is final every later entry
// is updated independently (no dependent accumulation chain).  sLT[j*NB + t] = L[t][j] and srd[j] = 1/L[j][j] sit in
// LDS; every lane reads the same address (broadcast), and since nothing writes them the reads pipeline freely.
template <int NB>
__device__ __forceinline__ void trsm_row_lds(double* x, const double* sLT, const double* srd) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        x[j] *= srd[j];
        const double xj = x[j];
#pragma unroll
        for (int t = j + 1; t < NB; ++t) x[t] = fma(-xj, sLT[j * NB + t], x[t]);
    }
}
// scalar-path variant (operands through s_load / SGPRs) used where the L block is not staged in LDS
template <int NB>
__device__ __forceinline__ void trsm_row_scalar(double* x, const double* LTg, const double* rdg) {
    cdouble* LT = (cdouble*)(uintptr_t)LTg;
    cdouble* rd = (cdouble*)(uintptr_t)rdg;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        x[j] *= rd[j];
        const double xj = x[j];
#pragma unroll
        for (int t = j + 1; t < NB; ++t) x[t] = fma(-xj, LT[j * NB + t], x[t]);
    }
}

// Right-looking Cholesky of an NB x NB tile inside ONE wavefront, lane i (< NB) holding the full SYMMETRIC row i in
// registers.  Because the tile stays symmetric, the column needed at step j (A[c][j] for all c) is register j across
// the lanes: one ds_write_b64 publishes it, broadcast ds_reads fetch it back, and every lane applies
//   a[i][c] -= a[i][j] * a[c][j] / a[j][j].
// No barrier (single wave, LDS operations of a wave execute in order), no cross-lane VALU traffic, only a reciprocal
// on the dependency chain.  sbuf: 2*NB doubles of LDS (double-buffered by column parity).  On return a[j] = L[i][j], j <= i.
template <int NB>
__device__ __forceinline__ bool potrf_inwave(double* a, int lane, double* sbuf) {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double* buf = sbuf + (j & 1) * NB;
        buf[lane] = a[j];
        const double sj = buf[j];
        const bool bj = !(sj > 0.0);
        bad = bad || bj;
        const double rinv = 1.0 / (bj ? 1.0 : sj);
        const double f = a[j] * rinv;
#pragma unroll
        for (int c = j + 1; c < NB; ++c) a[c] = fma(-f, buf[c], a[c]);
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
// reciprocal diagonal for the scalar-path TRSM of the next step
__device__ __forceinline__ void store_factor(const DevBuf& d, int kb, const double* a, int lane, bool bad) {
    const int ld = d.ld;
    double* Lrow = d.Lfac + (size_t)(kb * FB + lane) * ld + kb * FB;
    double* LT = d.LT32 + (size_t)kb * FB * FB;
#pragma unroll
    for (int j = 0; j < FB; j += 2) {
        const double v0 = (j <= lane) ? a[j] : 0.0, v1 = (j + 1 <= lane) ? a[j + 1] : 0.0;
        *reinterpret_cast<double2*>(Lrow + j) = make_double2(v0, v1);
    }
#pragma unroll
    for (int j = 0; j < FB; ++j) LT[j * FB + lane] = (lane > j) ? a[j] : 0.0;
    double dg = 1.0;
#pragma unroll
    for (int j = 0; j < FB; ++j) if (lane == j) dg = a[j];
    d.rd32[kb * FB + lane] = 1.0 / dg;
    if (bad && lane == 0) d.ctrl->solver_ok = 0;
}

__global__ __launch_bounds__(64) void k_potrf0(DevBuf d) {
    __shared__ double sbuf[2 * FB];
    const int lane = threadIdx.x;
    if (lane >= FB) return;
    double a[FB];
    const double* row = d.sys + (size_t)lane * d.ld;
#pragma unroll
    for (int j = 0; j < FB; j += 2) { const double2 v = *reinterpret_cast<const double2*>(row + j); a[j] = v.x; a[j + 1] = v.y; }
    const bool bad = potrf_inwave<FB>(a, lane, sbuf);
    store_factor(d, 0, a, lane, bad);
}

// One launch per block step k (right-looking, 32-wide blocks).  Workgroup (r,c), k < c <= r <= T (r == T is the
// right-hand-side block row):
//   wave 0, lanes 0-31 : X_r = A(r,k) L(k,k)^-T     lanes 32-63 : X_c = A(c,k) L(k,k)^-T     (scalar-path TRSM)
//   all 4 waves        : A(r,c) -= X_r X_c^T  on the matrix cores (v_mfma_f64_16x16x4_f64), one 16x16 tile per wave
//   look-ahead         : the workgroup of (k+1,k+1) factors its freshly updated tile in-wave and publishes
//                        L(k+1,k+1), so the next launch can start its TRSMs immediately.
template <bool MFMA>
__global__ __launch_bounds__(256) void k_chol_step(DevBuf d, int k, int T) {
    __shared__ double sXT[FB * XS];
    __shared__ double sC[FB * CS];
    __shared__ double sLT[FB * FB];
    __shared__ double srd[FB];
    __shared__ double sbuf[2 * FB];
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
    // stage L(k,k)^T and its reciprocal diagonal in LDS (4 doubles per thread, coalesced)
    {
        const double* LTg = d.LT32 + (size_t)k * FB * FB;
        const double4 v = reinterpret_cast<const double4*>(LTg)[threadIdx.x];
        reinterpret_cast<double4*>(sLT)[threadIdx.x] = v;
        if (threadIdx.x < FB) srd[threadIdx.x] = d.rd32[k * FB + threadIdx.x];
    }
    const bool upper = lane >= FB;
    const int rl = lane & (FB - 1);
    const bool active = (wv == 0) && (!upper || (have_update && c != r));
    double x[FB];
    if (active) {   // panel rows are fetched while the L tile lands in LDS
        const int br = upper ? c : r;
        const double* grow = d.sys + (size_t)(br * FB + rl) * ld + k * FB;
#pragma unroll
        for (int j = 0; j < FB; j += 2) { const double2 v = *reinterpret_cast<const double2*>(grow + j); x[j] = v.x; x[j + 1] = v.y; }
    }
    __syncthreads();
    if (wv == 0) {
        if (active) {
            trsm_row_lds<FB>(x, sLT, srd);
#pragma unroll
            for (int j = 0; j < FB; ++j) sXT[j * XS + lane] = x[j];
            if (!upper && c == k + 1) {
                // the solved panel block is the final L(r,k); it goes to Lfac, never back into sys: other
                // workgroups of this launch still read the unsolved panel from sys
                double* gout = d.Lfac + (size_t)(r * FB + rl) * ld + k * FB;
#pragma unroll
                for (int j = 0; j < FB; j += 2) *reinterpret_cast<double2*>(gout + j) = make_double2(x[j], x[j + 1]);
            }
        }
    }
    if (!have_update) return;
    __syncthreads();
    const int cb = (c == r) ? 0 : FB;
    double* C = d.sys + (size_t)(r * FB) * ld + c * FB;
    const bool lookahead = (r == k + 1 && c == k + 1);
    if (MFMA) {
        const int tr = wv >> 1, tc = wv & 1;
        const int li = lane & 15, lk = lane >> 4;
        double4v acc = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < FB / 4; ++kk) {
            const double av = sXT[(kk * 4 + lk) * XS + tr * 16 + li];
            const double bv = sXT[(kk * 4 + lk) * XS + cb + tc * 16 + li];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
        // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = tr * 16 + lk + 4 * v, col = tc * 16 + li;
            const double nv = C[(size_t)row * ld + col] - acc[v];
            if (lookahead) sC[row * CS + col] = nv;
            else C[(size_t)row * ld + col] = nv;
        }
    } else {
        const int row = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < FB; ++kk) {
            const double av = sXT[kk * XS + row];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += av * sXT[kk * XS + cb + c0 + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double nv = C[(size_t)row * ld + c0 + q] - acc[q];
            if (lookahead) sC[row * CS + c0 + q] = nv;
            else C[(size_t)row * ld + c0 + q] = nv;
        }
    }
    if (!lookahead) return;
    __syncthreads();
    if (wv == 0 && lane < FB) {
        double a[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j) a[j] = sC[lane * CS + j];
        const bool bad = potrf_inwave<FB>(a, lane, sbuf);
        store_factor(d, k + 1, a, lane, bad);
    }
}

// Linv[k] = L(k,k)^-1 for every 64x64 diagonal tile of the factor, assembled from its two 32-blocks:
//   [A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1  C^-1]   (A^-1, C^-1 by scalar-path TRSM on the identity)
__global__ __launch_bounds__(256) void k_inv_diag(DevBuf d) {
    __shared__ double sAi[FB * CS], sCi[FB * CS], sB[FB * CS], sT[FB * CS];
    const int k = blockIdx.x, ld = d.ld;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wv == 0) {
        const int blk = 2 * k + (lane >= FB ? 1 : 0), cidx = lane & (FB - 1);
        double y[FB];
#pragma unroll
        for (int t = 0; t < FB; ++t) y[t] = (t == cidx) ? 1.0 : 0.0;
        trsm_row_scalar<FB>(y, d.LT32 + (size_t)blk * FB * FB, d.rd32 + blk * FB);
        double* dst = (lane >= FB) ? sCi : sAi;          // dst[t][c] = (L^-1)(t, c)
#pragma unroll
        for (int t = 0; t < FB; ++t) dst[t * CS + cidx] = y[t];
    }
    for (int idx = threadIdx.x; idx < FB * FB; idx += 256) {
        const int rw = idx >> 5, cl = idx & 31;
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
    double* out = d.Linv + (size_t)k * TILE * TILE;
    {
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

void launch_cholesky(const DevBuf& d, bool use_mfma, hipStream_t s) {
    const int T = d.Ppad / FB;
    hipLaunchKernelGGL(k_potrf0, dim3(1), dim3(64), 0, s, d);
    for (int k = 0; k < T; ++k) {
        const int nt = T - k - 1;
        const int tiles = nt * (nt + 1) / 2 + nt;
        const int grid = tiles > 0 ? tiles : 1;
        if (use_mfma) hipLaunchKernelGGL(k_chol_step<true>, dim3(grid), dim3(256), 0, s, d, k, T);
        else hipLaunchKernelGGL(k_chol_step<false>, dim3(grid), dim3(256), 0, s, d, k, T);
    }
}

void launch_trsv_back(const DevBuf& d, int epoch, hipStream_t s) {
    const int T = d.Ppad / TILE;
    hipLaunchKernelGGL(k_inv_diag, dim3(T), dim3(256), 0, s, d);
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

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

constexpr int LS = 65;    // LDS row stride of the L tile (conflict-free column walks)
constexpr int XS = 80;    // LDS row stride of the transposed panel tiles XT[k][row]

__global__ __launch_bounds__(256) void k_chol_diag(DevBuf d, int k) {
    __shared__ double sA[TILE * LS];
    const int ld = d.ld;
    const double* A = d.sys + (size_t)(k * TILE) * ld + k * TILE;
    double* Lo = d.Lfac + (size_t)(k * TILE) * ld + k * TILE;
    for (int idx = threadIdx.x; idx < TILE * TILE; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        sA[r * LS + c] = A[(size_t)r * ld + c];
    }
    const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
    bool bad = false;
    for (int j = 0; j < TILE; ++j) {
        __syncthreads();
        const double djj = sA[j * LS + j];
        const bool bj = !(djj > 0.0);
        bad = bad || bj;
        const double ljj = bj ? 1.0 : sqrt(djj);
        const double lij = sA[i * LS + j] / ljj;
        __syncthreads();
        if (g == 0) {
            if (i > j) sA[i * LS + j] = lij;
            else if (i == j) sA[j * LS + j] = ljj;
        }
        __syncthreads();
        if (i > j)
            for (int c = j + 1 + g; c <= i; c += 4) sA[i * LS + c] -= lij * sA[c * LS + j];
    }
    __syncthreads();
    if (bad && threadIdx.x == 0) d.ctrl->solver_ok = 0;
    for (int idx = threadIdx.x; idx < TILE * TILE; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        Lo[(size_t)r * ld + c] = (c <= r) ? sA[r * LS + c] : 0.0;
    }
}

// X L^T = A  for one 64-row panel tile: lane `row` keeps its row in registers.
__device__ __forceinline__ void trsm_row(const double* __restrict__ grow, const double* sL, double* sXT, int row, double* __restrict__ gout) {
    double x[TILE];
#pragma unroll
    for (int j = 0; j < TILE; j += 2) {
        const double2 v = *reinterpret_cast<const double2*>(grow + j);
        x[j] = v.x; x[j + 1] = v.y;
    }
#pragma unroll
    for (int j = 0; j < TILE; ++j) {
        double s = x[j];
#pragma unroll
        for (int t = 0; t < j; ++t) s -= x[t] * sL[j * LS + t];
        x[j] = s / sL[j * LS + j];
    }
#pragma unroll
    for (int j = 0; j < TILE; ++j) sXT[j * XS + row] = x[j];
    if (gout) {
#pragma unroll
        for (int j = 0; j < TILE; j += 2) *reinterpret_cast<double2*>(gout + j) = make_double2(x[j], x[j + 1]);
    }
}

template <bool MFMA>
__global__ __launch_bounds__(256) void k_chol_step(DevBuf d, int k, int T) {
    extern __shared__ double s_dyn[];
    double* sL = s_dyn;                       // 64 x LS
    double* sXr = sL + TILE * LS;             // 64 x XS  (transposed: [kk][row])
    double* sXc = sXr + TILE * XS;
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
    const int r = k + 1 + rr, c = k + 1 + cc;   // r == T is the right-hand-side tile row
    const double* Lkk = d.Lfac + (size_t)(k * TILE) * ld + k * TILE;
    for (int idx = threadIdx.x; idx < TILE * TILE; idx += 256) {
        const int rw = idx >> 6, cl = idx & 63;
        sL[rw * LS + cl] = Lkk[(size_t)rw * ld + cl];
    }
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool have_update = (c < T);
    if (wv == 0) {
        // the solved panel tile is the final L(r,k); it goes to Lfac, never back into sys, because other
        // workgroups of this launch still read the unsolved panel from sys
        const double* grow = d.sys + (size_t)(r * TILE + lane) * ld + k * TILE;
        double* gout = d.Lfac + (size_t)(r * TILE + lane) * ld + k * TILE;
        trsm_row(grow, sL, sXr, lane, (c == k + 1) ? gout : nullptr);
    } else if (wv == 1 && have_update && c != r) {
        const double* grow = d.sys + (size_t)(c * TILE + lane) * ld + k * TILE;
        trsm_row(grow, sL, sXc, lane, nullptr);
    }
    if (!have_update) return;
    __syncthreads();
    const double* XC = (c == r) ? sXr : sXc;
    double* C = d.sys + (size_t)(r * TILE) * ld + c * TILE;
    if (MFMA) {
        // wave wv owns rows 16wv..16wv+15; 4 column tiles of 16
        double4v acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = (double4v){0.0, 0.0, 0.0, 0.0};
        const int li = lane & 15, lk = lane >> 4;
#pragma unroll 4
        for (int kk = 0; kk < TILE / 4; ++kk) {
            const double a = sXr[(kk * 4 + lk) * XS + wv * 16 + li];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const double bv = XC[(kk * 4 + lk) * XS + n * 16 + li];
                acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[n], 0, 0, 0);
            }
        }
        // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = wv * 16 + lk + 4 * v, col = n * 16 + li;
                C[(size_t)row * ld + col] -= acc[n][v];
            }
    } else {
        const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[a][q] = 0.0;
        for (int kk = 0; kk < TILE; ++kk) {
            double ar[4], bc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { ar[a] = sXr[kk * XS + ty * 4 + a]; bc[a] = XC[kk * XS + tx * 4 + a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[a][q] += ar[a] * bc[q];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) C[(size_t)(ty * 4 + a) * ld + tx * 4 + q] -= acc[a][q];
    }
}

// L^T x = y, y = row Ppad of the factored augmented system.  One 1024-thread workgroup, blocks from last to first.
__global__ __launch_bounds__(1024) void k_trsv_back(DevBuf d, int T) {
    extern __shared__ double s_dyn[];
    double* z = s_dyn;                     // Ppad
    double* sD = z + d.Ppad;               // 64 x LS diagonal tile
    __shared__ double sx[TILE];
    const int ld = d.ld, n = d.Ppad;
    const double* y = d.Lfac + (size_t)n * ld;
    for (int i = threadIdx.x; i < n; i += 1024) z[i] = y[i];
    for (int kb = T - 1; kb >= 0; --kb) {
        __syncthreads();
        const double* Lkk = d.Lfac + (size_t)(kb * TILE) * ld + kb * TILE;
        for (int idx = threadIdx.x; idx < TILE * TILE; idx += 1024) {
            const int r = idx >> 6, c = idx & 63;
            sD[r * LS + c] = Lkk[(size_t)r * ld + c];
        }
        __syncthreads();
        if (threadIdx.x < 64) {            // wave 0: in-tile back substitution, column oriented
            const int t = threadIdx.x;
            double zt = z[kb * TILE + t];
            for (int j = TILE - 1; j >= 0; --j) {
                const double xj = __shfl(zt, j, 64) / sD[j * LS + j];
                if (t == j) zt = xj;
                else if (t < j) zt -= sD[j * LS + t] * xj;
            }
            sx[t] = zt;
            d.x[kb * TILE + t] = zt;
        }
        __syncthreads();
        // z[c] -= sum_j L[kb*64 + j][c] * x[j]  for every column c left of the diagonal tile
        const int ncol = kb * TILE;
        for (int c = threadIdx.x; c < ncol; c += 1024) {
            const double* Lr = d.Lfac + (size_t)(kb * TILE) * ld + c;
            double s = 0.0;
#pragma unroll 8
            for (int j = 0; j < TILE; ++j) s += Lr[(size_t)j * ld] * sx[j];
            z[c] -= s;
        }
    }
}

void launch_cholesky(const DevBuf& d, bool use_mfma, hipStream_t s) {
    const int T = d.Ppad / TILE;
    const size_t sh = (size_t)(TILE * LS + 2 * TILE * XS) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_step<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_step<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_set = true;
    }
    for (int k = 0; k < T; ++k) {
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, s, d, k);
        const int nt = T - k - 1;
        const int tiles = nt * (nt + 1) / 2 + nt;
        const int grid = tiles > 0 ? tiles : 1;
        if (use_mfma) hipLaunchKernelGGL(k_chol_step<true>, dim3(grid), dim3(256), sh, s, d, k, T);
        else hipLaunchKernelGGL(k_chol_step<false>, dim3(grid), dim3(256), sh, s, d, k, T);
    }
}

void launch_trsv_back(const DevBuf& d, hipStream_t s) {
    const int T = d.Ppad / TILE;
    const size_t sh = (size_t)(d.Ppad + TILE * LS) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv_back), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_trsv_back, dim3(1), dim3(1024), sh, s, d, T);
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

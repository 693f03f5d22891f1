// plba_band.hip — K7 for BANDED reduced camera systems: a twisted (two-ended) block Cholesky that never leaves LDS.
//
// What g2o hands to LinearSolverEigen is a SPARSE matrix (simplicial Cholesky with AMD ordering, IMU/g2otypes.h:18, SURVEY
// App. A.6).  In keyframe order the reduced camera system of a sliding window is block-BANDED: a landmark is tracked over a
// few consecutive keyframes, and the chain elimination (plba_chain.hip) only couples the keyframes of one segment window.
// With HB = 3 sub-diagonal 32 x 32 tiles the band holds every non-zero of the 375-dim system of BASELINE configs[2]
// (50 keyframes) and of the 1392-dim system of configs[4] (200 keyframes); prepare() measures the band from the structure
// and keeps the dense multi-launch path (plba_dense.hip) for anything wider.
//
// The dense path pays one launch per 32 columns (12 / 44 dependent launches of ~9-10 us).  Here ONE workgroup walks down the
// band with the active window (10 tiles) resident in LDS — no launch boundary, no global round trip between the pivot
// sweep, the panel products and the trailing update — and a SECOND workgroup walks UP from the other end at the same time
// (the reversed matrix is banded too: "twisted factorisation").
// Measured (MI355X, round 2): a step costs 29k cycles here (pivot pipeline 17.3k — 300 scalar spills sit in its chain once it
// is inlined into the sweep loop —, panels 3.7k, trailing + rhs 5.3k: ONE compute unit's fp64 rate, 64 FMA / clock, is the
// floor for the step's 295k FMAs, whatever the number of waves), against 22k cycles per launch of the dense path whose
// trailing update is spread over many compute units.  Two sweeps in parallel therefore win where the chain is long —
// configs[4], 44 tiles: 0.92 instead of 1.07 ms per LM iteration — and lose at 12 tiles (configs[2]: 142 vs 107 us), so
// prepare() takes this path from BAND_MIN_TILES tiles on.  The sweeps stop three tiles apart; the 96 x 96 middle block
// has received the Schur complements of both sides and is solved by both workgroups redundantly, after which each
// back-substitutes its own half:
//
//   k_band_fwd   grid 2: top-down / bottom-up forward sweep over n tiles: per step the look-ahead pipeline of plba_dense_dev.h
//                (L(k,k) and L(k,k)^-1 in LDS), panels L(i,k) = A(i,k) L(k,k)^-T and the window's trailing update on
//                v_mfma_f64_16x16x4_f64, right-hand side carried along; L band + y to HBM for the back-substitution, the
//                updated middle window to HBM for the hand-over
//   k_band_back  grid 2: middle = A_mid + (top window - A_mid) + (bottom window - A_mid), three more steps in LDS, x_mid, then
//                x(k) = L(k,k)^-T (y(k) - sum_d L(k+d,k)^T x(k+d)) outwards, tiles prefetched a step ahead
//
// Same arithmetic as the dense path up to the order of the trailing updates, so the parity tests are unchanged.
#include "plba_internal.h"
#include "plba_factor32_dev.h"

namespace plba {

namespace {

constexpr int HB = BAND_HB;         // sub-diagonal tiles the window holds
constexpr int NSLOT_B = 10;         // live tiles of the window: 4 + 3 + 2 + 1
constexpr int TSZ = 32 * LS;        // doubles of one LDS tile (row stride LS)
constexpr int NT = 256;             // four waves: the look-ahead pipeline needs them all and ~256 VGPRs each (8 or 16 waves were built: the
                                    // register cap makes the pipeline spill, and a CU's fp64 rate — 64 FMA / clock, 4.6k cycles for a
                                    // step's panel + trailing products — is the same however many waves share it)
constexpr int NW = NT / 64;

// Time-consistent slot of band tile (column block j, sub-diagonal d): a tile keeps its slot from the step its row enters
// the window (k = j + d - 3) to the step its column is eliminated (k = j), and the four tiles of the row entering at step
// k + 1 take exactly the slots column k frees (slot(j,3) = slot(j-1,0), slot(j,2) = slot(j-2,1), slot(j,1) = slot(j-3,2),
// slot(j,0) = slot(j-4,3)).
__device__ __forceinline__ int bslot(int j, int d) {
    switch (d) {
        case 0: return j % 5;
        case 3: return (j + 4) % 5;
        case 1: return 5 + j % 5;
        default: return 5 + (j + 3) % 5;
    }
}

struct BandCtx {
    const double* sys; int ld, Pdpad, T, dir;
    __device__ __forceinline__ double at(int r, int c) const {        // element (r, c) of the matrix this direction sees
        const int P1 = Pdpad - 1;
        return dir ? sys[(size_t)(P1 - r) * ld + (P1 - c)] : sys[(size_t)r * ld + c];
    }
    __device__ __forceinline__ double rhs(int c) const { return sys[(size_t)Pdpad * ld + (dir ? Pdpad - 1 - c : c)]; }
};

// tile (i, j) of the matrix -> LDS tile (row stride LS)
__device__ __forceinline__ void load_tile(const BandCtx& bc, int i, int j, double* dst) {
    for (int idx = threadIdx.x; idx < 1024; idx += NT) { const int r = idx >> 5, c = idx & 31; dst[r * LS + c] = bc.at(32 * i + r, 32 * j + c); }
}

// C(tr, tc quadrant) = sum_k A[r][k] B[c][k]   (A B^T) for two LDS tiles; result m[v] at (tr*16 + lk + 4v, tc*16 + li)
__device__ __forceinline__ double4v abt_quadrant(const double* A, const double* B, int tr, int tc, int li, int lk) {
    double4v m = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) m = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(tr * 16 + li) * LS + kk * 4 + lk], B[(tc * 16 + li) * LS + kk * 4 + lk], m, 0, 0, 0);
    return m;
}

// One forward step on the window held in LDS: factor tile (k,k), y(k), panels, right-hand side and trailing update.
// rows: the window's row limit (tiles with row index >= rows do not exist for this sweep).  Lb (may be null): where the
// step's inverse and panels go in HBM, [4][1024] row-major.  keep_inv: also leave L(k,k)^-1 in the diagonal tile's slot.
__device__ __forceinline__ void band_step(const DevBuf& dd, int k, int rows, double* tiles, double* sLinv, Look32& S, double* bvec, double* yk,
                                          double* Lb, double* yout, bool keep_inv) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    double* Akk = tiles + bslot(k, 0) * TSZ;
#ifdef PLBA_BAND_STAMPS
    const bool stamp = (k == 1 && blockIdx.x == 0 && tid == 0 && Lb != nullptr);
    if (stamp) dd.dbgbuf[0] = (double)__builtin_readcyclecounter();
#define BSTAMP(i) do { if (stamp) dd.dbgbuf[i] = (double)__builtin_readcyclecounter(); } while (0)
#else
#define BSTAMP(i) do {} while (0)
#endif
    factor32_reset(S, tid);
    __syncthreads();
    factor32_tile<true>(dd, k, Akk, S, wv, lane, sLinv);
    __syncthreads();
    BSTAMP(1);
    if (tid < 32) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < 32; ++c) t = fma(sLinv[tid * LS + c], bvec[32 * k + c], t);
        yk[tid] = t;
        if (yout) yout[32 * k + tid] = t;
    }
    if (Lb) for (int idx = tid; idx < 1024; idx += NT) Lb[idx] = sLinv[(idx >> 5) * LS + (idx & 31)];
    // panels L(k+d,k) = A(k+d,k) L(k,k)^-T: 3 tiles x 4 quadrants = 12 jobs over the NW waves
    constexpr int PR = (4 * HB + NW - 1) / NW;
    double4v pan[PR];
#pragma unroll
    for (int rnd = 0; rnd < PR; ++rnd) {
        const int jb = wv + NW * rnd, pd = 1 + jb / 4, pq = jb & 3;
        pan[rnd] = (jb < 4 * HB && k + pd < rows) ? abt_quadrant(tiles + bslot(k, pd) * TSZ, sLinv, pq >> 1, pq & 1, li, lk) : (double4v){0.0, 0.0, 0.0, 0.0};
    }
    BSTAMP(2);
    __syncthreads();                                   // every wave has read the A(i,k) tiles it needs: they may be overwritten
#pragma unroll
    for (int rnd = 0; rnd < PR; ++rnd) {
        const int jb = wv + NW * rnd, pd = 1 + jb / 4, pq = jb & 3;
        if (jb >= 4 * HB || k + pd >= rows) continue;
        double* Lt = tiles + bslot(k, pd) * TSZ;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = (pq >> 1) * 16 + lk + 4 * v, c = (pq & 1) * 16 + li;
            Lt[r * LS + c] = pan[rnd][v];
            if (Lb) Lb[pd * 1024 + r * 32 + c] = pan[rnd][v];
        }
    }
    if (keep_inv) for (int idx = tid; idx < 1024; idx += NT) Akk[(idx >> 5) * LS + (idx & 31)] = sLinv[(idx >> 5) * LS + (idx & 31)];
    __syncthreads();
    BSTAMP(3);
    // right-hand side of the rows below: b(k+d) -= L(k+d,k) y(k): the last HB waves, a lane per row
    if (wv >= NW - HB && k + (wv - (NW - HB) + 1) < rows && lane < 32) {
        const int d = wv - (NW - HB) + 1;
        const double* Lt = tiles + bslot(k, d) * TSZ;
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < 32; ++c) t = fma(Lt[lane * LS + c], yk[c], t);
        bvec[32 * (k + d) + lane] -= t;
    }
    // trailing update of the window: A(i,j) -= L(i,k) L(j,k)^T for k < j <= i <= k + HB: 6 tiles x 4 quadrants = 24 jobs over the NW waves
#pragma unroll
    for (int rnd = 0; rnd < (24 + NW - 1) / NW; ++rnd) {
        const int jb = wv + NW * rnd;
        if (jb >= 24) continue;
        const int tl = jb >> 2, q = jb & 3;
        // tile list (dj, di): (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
        const int dj = tl < 3 ? 1 : (tl < 5 ? 2 : 3), di = tl < 3 ? 1 + tl : (tl < 5 ? tl - 1 : 3);
        if (k + di >= rows) continue;
        const double4v m = abt_quadrant(tiles + bslot(k, di) * TSZ, tiles + bslot(k, dj) * TSZ, q >> 1, q & 1, li, lk);
        double* At = tiles + bslot(k + dj, di - dj) * TSZ;
#pragma unroll
        for (int v = 0; v < 4; ++v) At[((q >> 1) * 16 + lk + 4 * v) * LS + (q & 1) * 16 + li] -= m[v];
    }
    BSTAMP(4);
    __syncthreads();
    BSTAMP(5);
}

// x(k) = L(k,k)^-T (y(k) - sum_d L(k+d,k)^T x(k+d)); tp[0] = L(k,k)^-1, tp[d] = L(k+d,k) as LDS tiles (row stride LS).
// yv / xv: right-hand side and solution in this direction's indexing.  All 256 threads; s_part: 544 doubles of scratch.
__device__ __forceinline__ void band_back_step(int k, int rows, const double* const* tp, const double* yv, double* xv, double* s_part) {
    const int tid = threadIdx.x, c = tid & 31, part = (tid >> 5) & 7;     // threads 0-255: 8 parts x 12 of the 96 (d, r) terms
    const bool act = tid < 256;
    double t = 0.0;
    if (act) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int e = part * 12 + q, d = 1 + e / 32, r = e % 32;
            if (k + d < rows) t = fma(tp[d][r * LS + c], xv[32 * (k + d) + r], t);
        }
        s_part[part * 32 + c] = t;
    }
    __syncthreads();
    if (tid < 32) {
        double sacc = yv[32 * k + tid];
#pragma unroll
        for (int p = 0; p < 8; ++p) sacc -= s_part[p * 32 + tid];
        s_part[256 + tid] = sacc;
    }
    __syncthreads();
    if (act) {   // x(k)[c] = sum_r Linv[r][c] t[r]
        double u = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int r = part * 4 + q; u = fma(tp[0][r * LS + c], s_part[256 + r], u); }
        s_part[288 + part * 32 + c] = u;
    }
    __syncthreads();
    if (tid < 32) {
        double sacc = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) sacc += s_part[288 + p * 32 + tid];
        xv[32 * k + tid] = sacc;
    }
    __syncthreads();
}

struct BandLdsMap {                 // carving of the dynamic LDS
    double *tiles, *sLinv, *bvec, *yk, *part, *ymid, *bmid;
    Look32* S;
    __device__ BandLdsMap(double* base, int Pdpad) {
        tiles = base; base += NSLOT_B * TSZ;
        sLinv = base; base += TSZ;
        S = reinterpret_cast<Look32*>(base); base += (sizeof(Look32) + 15) / 16 * 2;
        bvec = base; base += Pdpad + 32;
        yk = base; base += 32;
        part = base; base += 544;
        ymid = base; base += 96;
        bmid = base;                   // 96 doubles
    }
};

__global__ __launch_bounds__(NT) void k_band_fwd(DevBuf dd, BandView bv) {
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    BandLdsMap M(s_dyn, dd.Ppad);
    const int dir = blockIdx.x, tid = threadIdx.x;
    const int n = dir ? bv.nB : bv.nA, rows = n + HB;              // the sweep's window never reaches past the middle block
    BandCtx bc{dd.sys, dd.ld, dd.Ppad, dd.Ppad / 32, dir};
    double* Lb = bv.Lband + (size_t)dir * bv.T * 4 * 1024;
    double* yout = bv.y + (size_t)dir * dd.Ppad;
    for (int c = tid; c < dd.Ppad; c += NT) M.bvec[c] = bc.rhs(c);
    for (int j = 0; j <= HB; ++j) for (int i = j; i <= HB; ++i) if (i < rows) load_tile(bc, i, j, M.tiles + bslot(j, i - j) * TSZ);
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        // the row entering the window after this step: tiles (k+4, k+1 .. k+4), fetched now, stored once column k's slots are free
        constexpr int EPT = 1024 / NT;
        double pre[HB + 1][EPT];
        const bool enter = k + HB + 1 < rows;
        if (enter) {
#pragma unroll
            for (int t = 0; t <= HB; ++t)
#pragma unroll
                for (int q = 0; q < EPT; ++q) { const int idx = q * NT + tid; pre[t][q] = bc.at(32 * (k + HB + 1) + (idx >> 5), 32 * (k + 1 + t) + (idx & 31)); }
        }
        band_step(dd, k, rows, M.tiles, M.sLinv, *M.S, M.bvec, M.yk, Lb + (size_t)k * 4 * 1024, yout, false);
        if (enter) {
#pragma unroll
            for (int t = 0; t <= HB; ++t)
#pragma unroll
                for (int q = 0; q < EPT; ++q) { const int idx = q * NT + tid; M.tiles[bslot(k + 1 + t, HB - t) * TSZ + (idx >> 5) * LS + (idx & 31)] = pre[t][q]; }
        }
        // (band_step ends in a barrier; the next one starts with look32_reset + barrier)
    }
    __syncthreads();
    // hand-over: the middle window (tiles (n+a, n+b), a >= b) with both-sided... this side's Schur complement applied, and its rhs
    double* mid = bv.mid + (size_t)dir * (9 * 1024 + 96);
    for (int a = 0; a < HB; ++a) for (int b = 0; b <= a; ++b) {
        const double* src = M.tiles + bslot(n + b, a - b) * TSZ;
        for (int idx = tid; idx < 1024; idx += NT) mid[(a * 3 + b) * 1024 + idx] = src[(idx >> 5) * LS + (idx & 31)];
    }
    if (tid < 96) mid[9 * 1024 + tid] = M.bvec[32 * n + tid];
}

__global__ __launch_bounds__(NT) void k_band_back(DevBuf dd, BandView bv) {
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    BandLdsMap M(s_dyn, dd.Ppad);
    const int dir = blockIdx.x, tid = threadIdx.x;
    const int T = dd.Ppad / 32, m0 = bv.nA, n = dir ? bv.nB : bv.nA, P1 = dd.Ppad - 1;
    const double* midA = bv.mid; const double* midB = bv.mid + (9 * 1024 + 96);
    double* xv = M.bvec;                 // solution in this direction's indexing
    // ---- middle block in the ORIGINAL orientation: A_mid + (top - A_mid) + (bottom - A_mid) -----------------------------------
    BandCtx b0{dd.sys, dd.ld, dd.Ppad, T, 0};
    for (int a = 0; a < HB; ++a) for (int b = 0; b <= a; ++b) {
        double* dst = M.tiles + bslot(b, a - b) * TSZ;
        for (int idx = tid; idx < 1024; idx += NT) {
            const int r = idx >> 5, c = idx & 31;
            // the bottom sweep's share of the same block: its tile (2-b, 2-a) (lower there: 2-b >= 2-a), transposed and reversed
            const double vb = midB[((2 - b) * 3 + (2 - a)) * 1024 + (31 - c) * 32 + (31 - r)];
            dst[r * LS + c] = midA[(a * 3 + b) * 1024 + idx] + vb - b0.at(32 * (m0 + a) + r, 32 * (m0 + b) + c);
        }
    }
    if (tid < 96) {
        const int a = tid >> 5, r = tid & 31;
        M.bmid[tid] = midA[9 * 1024 + tid] + midB[9 * 1024 + 32 * (2 - a) + (31 - r)] - b0.rhs(32 * m0 + tid);
    }
    __syncthreads();
    for (int k = 0; k < HB; ++k) band_step(dd, k, HB, M.tiles, M.sLinv, *M.S, M.bmid, M.yk, nullptr, M.ymid, true);
    double* xmid = M.bmid;               // the right-hand side has been consumed: the solution takes its place
    for (int k = HB - 1; k >= 0; --k) {
        const double* tp[4] = {M.tiles + bslot(k, 0) * TSZ, M.tiles + bslot(k, 1) * TSZ, M.tiles + bslot(k, 2) * TSZ, M.tiles + bslot(k, 3) * TSZ};
        band_back_step(k, HB, tp, M.ymid, xmid, M.part);
    }
    if (dir == 0 && tid < 96) dd.x[32 * m0 + tid] = xmid[tid];
    // ---- this half, outwards from the middle (whose tiles are n .. n+2 in this direction's indexing) ----------------------------
    if (tid < 96) {
        const int a = tid >> 5, r = tid & 31;
        xv[32 * n + tid] = dir ? xmid[32 * (2 - a) + (31 - r)] : xmid[tid];
    }
    __syncthreads();
    const double* Lb = bv.Lband + (size_t)dir * bv.T * 4 * 1024;
    const double* yv = bv.y + (size_t)dir * dd.Ppad;
    double* stage = M.tiles;                                     // four tiles of staging: the middle's window is done with
    const double* tp[4] = {stage, stage + TSZ, stage + 2 * TSZ, stage + 3 * TSZ};
    constexpr int EPT = 1024 / NT;
    double pre[4][EPT];
    double ypre = 0.0;
    if (n > 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int q = 0; q < EPT; ++q) pre[d][q] = Lb[((size_t)(n - 1) * 4 + d) * 1024 + q * NT + tid];
        if (tid < 32) ypre = yv[32 * (n - 1) + tid];
    }
    for (int k = n - 1; k >= 0; --k) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int q = 0; q < EPT; ++q) { const int idx = q * NT + tid; stage[d * TSZ + (idx >> 5) * LS + (idx & 31)] = pre[d][q]; }
        if (tid < 32) M.yk[tid] = ypre;
        __syncthreads();
        if (k > 0) {                                             // next step's tiles travel while this one is computed
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int q = 0; q < EPT; ++q) pre[d][q] = Lb[((size_t)(k - 1) * 4 + d) * 1024 + q * NT + tid];
            if (tid < 32) ypre = yv[32 * (k - 1) + tid];
        }
        band_back_step(k, n + HB, tp, M.yk - 32 * k, xv, M.part);
        if (tid < 32) { const int g = 32 * k + tid; dd.x[dir ? P1 - g : g] = xv[g]; }
    }
}

}  // namespace

size_t band_lds_bytes(int Pdpad) {
    return ((size_t)NSLOT_B * TSZ + TSZ + (sizeof(Look32) + 15) / 16 * 2 + (size_t)Pdpad + 32 + 32 + 544 + 96 + 96 + 16) * sizeof(double);
}

void launch_band_solve(const DevBuf& dd, const BandView& bv, hipStream_t s) {
    const size_t sh = band_lds_bytes(dd.Ppad);
    if (ensure_dyn_lds(reinterpret_cast<const void*>(k_band_fwd), 160 * 1024) != hipSuccess ||
        ensure_dyn_lds(reinterpret_cast<const void*>(k_band_back), 160 * 1024) != hipSuccess) return;      // the sticky HIP error surfaces at the caller's hipGetLastError
    hipLaunchKernelGGL(k_band_fwd, dim3(2), dim3(NT), sh, s, dd, bv);
    hipLaunchKernelGGL(k_band_back, dim3(2), dim3(NT), sh, s, dd, bv);
}

}  // namespace plba

// plba_chain.hip — elimination of the velocity / bias ("chain") variables ahead of the dense factorisation.
//
// Of the 15 dimensions a keyframe contributes to the reduced camera system only 6 (position, rotation) are coupled to
// landmarks.  The other 9 (velocity, gyro / accel bias deltas) are touched by the IMU edges alone: in keyframe order
// their 9 x 9 blocks form a block-TRIDIAGONAL matrix, coupled to the pose dimensions of the neighbouring keyframes only.
// One level of nested dissection makes that structure parallel: every (SEG+1)-th chain block is kept as a SEPARATOR
// and joins the dense variables; the runs of SEG blocks in between ("segments") no longer see each other and are
// eliminated by one workgroup each, all at once:
//
//     [ C   B ] [x_c]   [b_c]     C = blockdiag over segments of block-tridiagonal C_g = L_g L_g^T
//     [ B^T A ] [x_d] = [b_d]     W = L^-1 [B | b_c];   (A - W_B^T W_B) x_d = b_d - W_B^T w_b;   x_c = L^-T (w_b - W_B x_d)
//
// with x_d = pose dimensions + separator chain dimensions (348 instead of 735 at the headline size: 12 block steps of
// the dense factorisation instead of 24) and at most SEG sequential 9 x 9 steps per workgroup instead of one per keyframe.
// B only has columns for the keyframes next to a segment, so each workgroup carries ~150 columns, and W^T W only
// touches the diagonal band of the dense system.
//
//   k_chain_elim   one workgroup per segment.  wave 0: per chain block  C_ii -= L_i,i-1 L_i,i-1^T, 9 x 9 Cholesky and
//                  inverse in registers (v_readlane broadcasts), L_i+1,i = C_i+1,i L_ii^-T, published through LDS;
//                  waves 1-3: a lane per coupled column:  w_i = L_ii^-1 (B_i - L_i,i-1 w_i-1).  Everything a step reads
//                  was staged in LDS up front.
//   k_chain_schur  tile (a,b) of  A - W^T W  (v_mfma_f64_16x16x4_f64) over the rows of the segments whose column window
//                  meets the tile; right-hand side row; identity padding
//   k_chain_back   one workgroup per segment:  v = w_b - W_B x_d  over its column window, backward block substitution,
//                  scatter of x into system order
//
// Applicable when no marginalization prior is attached (it couples chain variables of several keyframes) and every IMU
// edge joins neighbouring chain blocks; otherwise the solver falls back to the dense path on the full system.
// Everything is read from the LOWER triangle of `sys` (in a sharded run only that part is globally summed).
#include "plba_internal.h"

namespace plba {

typedef double double4v __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ double lane_bcast(double v, int l) {   // lane l (compile time) -> uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_full(double d) {          // v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    return y;
}
template <typename T> __device__ __forceinline__ void lds_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename T> __device__ __forceinline__ T lds_load(const T* p) { return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ double sym_at(const double* sys, int ld, int i, int j) { return i >= j ? sys[(size_t)i * ld + j] : sys[(size_t)j * ld + i]; }
constexpr int SPIN_MAX = 1 << 20;
constexpr int SEGMAX = CHAIN_SEG;          // block steps per workgroup; also the depth of the LDS ring (no wrap, no back-pressure)
constexpr int NSLOT = CHAIN_NSLOT;         // dense columns a keyframe can own: 6 pose + 9 separator chain dimensions
constexpr int COLW = 3;                    // column waves: <= 192 coupled columns per segment (window of SEG + 2 keyframes + rhs)
constexpr int ELIM_THREADS = 64 * (1 + COLW);
}  // namespace

__global__ __launch_bounds__(ELIM_THREADS) void k_chain_elim(DevBuf d, ChainView cv) {
    __shared__ __attribute__((aligned(16))) double sLsub[SEGMAX][90], sLinv[SEGMAX][90];   // published chain factors; rows of 9 padded to 10: aligned pairs
    __shared__ double sA[81];
    __shared__ double sCg[SEGMAX][162];                 // C_ii (81) | C_{i+1,i} (81)
    __shared__ double sBg[SEGMAX][3 * NSLOT * 9];       // B_i against the dense columns of positions p-1, p, p+1
    __shared__ double sRhs[SEGMAX][9];
    __shared__ int s_step, s_bad;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = d.ld;
    const int g = blockIdx.x;
    const int i0 = cv.seg_start[g], i1 = cv.seg_start[g + 1], n = i1 - i0;      // eliminated blocks of this segment
    if (threadIdx.x == 0) { s_step = 0; s_bad = 0; }
    // ---- stage everything the steps read (one gather per entry, all in flight) --------------------------------------------
    for (int idx = threadIdx.x; idx < n * 162; idx += ELIM_THREADS) {
        const int bi = idx / 162, e = idx % 162;
        const int32_t* ci = cv.cidx + (i0 + bi) * 9;
        const bool nxt = bi + 1 < n;
        const int ga = e < 81 ? ci[e / 9] : (nxt ? ci[9 + (e - 81) / 9] : -1);
        const int gb = e < 81 ? ci[e % 9] : ci[(e - 81) % 9];
        const bool ok = ga >= 0 && gb >= 0;
        const double v = sym_at(d.sys, ld, ok ? ga : 0, ok ? gb : 0);
        sCg[bi][e] = ok ? v : ((e < 81 && e / 9 == e % 9) ? 1.0 : 0.0);
    }
    for (int idx = threadIdx.x; idx < n * 3 * NSLOT * 9; idx += ELIM_THREADS) {
        const int bi = idx / (3 * NSLOT * 9), e = idx % (3 * NSLOT * 9);
        const int dl = e / (NSLOT * 9), sl = (e / 9) % NSLOT, r = e % 9;
        const int pos = cv.epos[i0 + bi] + dl - 1;
        const int col = (pos >= 0 && pos < cv.npos) ? cv.slotcol[pos * NSLOT + sl] : -1;
        const int gi = cv.cidx[(i0 + bi) * 9 + r];
        const bool ok = col >= 0 && gi >= 0;
        const double v = sym_at(d.sys, ld, ok ? gi : 0, ok ? cv.pidx[ok ? col : 0] : 0);
        sBg[bi][e] = ok ? v : 0.0;
    }
    for (int idx = threadIdx.x; idx < n * 9; idx += ELIM_THREADS) {
        const int gi = cv.cidx[i0 * 9 + idx];
        sRhs[idx / 9][idx % 9] = gi < 0 ? 0.0 : d.sys[(size_t)d.Ppad * ld + gi];
    }
    __syncthreads();
    if (wv == 0) {
        // ---- the chain wave ----------------------------------------------------------------------------------------------
        bool bad = false;
        const int e0 = lane, e1 = lane + 64;              // the (up to) two entries of a 9 x 9 block this lane owns
        const int r0 = e0 / 9, c0 = e0 % 9, r1 = (e1 < 81 ? e1 : 0) / 9, c1 = (e1 < 81 ? e1 : 0) % 9;
        for (int i = 0; i < n; ++i) {
            const bool has_next = i + 1 < n;
            const double* Cg = sCg[i];
            double v0 = Cg[e0], v1 = Cg[e1 < 81 ? e1 : 0];
            double cs0[9], cs1[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) { cs0[t] = Cg[81 + r0 * 9 + t]; cs1[t] = Cg[81 + r1 * 9 + t]; }
            if (i > 0) {       // (a) C_ii - L_{i,i-1} L_{i,i-1}^T
                const double* Lp = sLsub[i - 1];
#pragma unroll
                for (int t = 0; t < 9; ++t) { v0 = fma(-Lp[r0 * 10 + t], Lp[c0 * 10 + t], v0); v1 = fma(-Lp[r1 * 10 + t], Lp[c1 * 10 + t], v1); }
            }
            sA[e0] = v0;
            if (e1 < 81) sA[e1] = v1;
            // (b) 9 x 9 Cholesky, lane = row (lanes >= 9 shadow row 8; their values are never used)
            const int row = lane < 9 ? lane : 8;
            double a[9], rs[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) a[c] = sA[row * 9 + c];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double pj = lane_bcast(a[j], j);
                const bool bj = !(pj > 0.0);
                bad = bad || bj;
                const double r = rsqrt_full(bj ? 1.0 : pj);
                rs[j] = r;
                const double lij = a[j] * r;
                a[j] = lij;
#pragma unroll
                for (int c = j + 1; c < 9; ++c) a[c] = fma(-lij, lane_bcast(lij, c), a[c]);
            }
            // (c) L_ii^-1, lane = column (forward substitution on the identity)
            double x[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) x[t] = (t == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                x[j] *= rs[j];
#pragma unroll
                for (int t = j + 1; t < 9; ++t) x[t] = fma(-lane_bcast(a[j], t), x[j], x[t]);     // L[t][j] lives in lane t, register j
            }
            double* Li = sLinv[i];
            if (lane < 9) {
#pragma unroll
                for (int t = 0; t < 9; ++t) { Li[t * 10 + lane] = x[t]; cv.Ldinv[(size_t)(i0 + i) * 81 + t * 9 + lane] = x[t]; }
                Li[lane * 10 + 9] = 0.0;
            }
            // (d) L_{i+1,i} = C_{i+1,i} L_ii^-T
            if (has_next) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int t = 0; t < 9; ++t) { s0 = fma(cs0[t], Li[c0 * 10 + t], s0); s1 = fma(cs1[t], Li[c1 * 10 + t], s1); }
                double* Lo = sLsub[i];
                Lo[r0 * 10 + c0] = s0;
                cv.Lsub[(size_t)(i0 + i) * 81 + e0] = s0;
                if (e1 < 81) { Lo[r1 * 10 + c1] = s1; cv.Lsub[(size_t)(i0 + i) * 81 + e1] = s1; }
                if (lane < 9) Lo[lane * 10 + 9] = 0.0;
            }
            asm volatile("" ::: "memory");
            lds_store(&s_step, i + 1);        // a wave's LDS operations execute in order: the data above is visible first
        }
        if (bad && lane == 0) d.ctrl->solver_ok = 0;
        return;
    }
    // ---- column lanes: w_i = L_ii^-1 (B_i - L_{i,i-1} w_{i-1}) over the dense columns of the segment's window + the rhs -------
    const int wlo = cv.seg_col[2 * g], whi = cv.seg_col[2 * g + 1];      // dense columns [wlo, whi) can couple to this segment
    const int lc = (wv - 1) * 64 + lane;                                  // local column; the one behind the window is the rhs
    const bool rhs = (lc == whi - wlo);
    const bool act = lc <= whi - wlo;
    const int col = rhs ? cv.Pd : wlo + (act ? lc : 0);
    const int cpos = (act && !rhs) ? cv.ppos[col] : 0, cslot = (act && !rhs) ? cv.pslot[col] : 0;
    double wp[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) wp[r] = 0.0;
    for (int i = 0; i < n; ++i) {
        // B is sparse: chain block i only couples to the dense columns of the neighbouring keyframe positions
        const int dl = cpos - cv.epos[i0 + i] + 1;
        const bool near = act && !rhs && dl >= 0 && dl <= 2;
        const double* src = rhs ? sRhs[i] : sBg[i] + ((near ? dl : 0) * NSLOT + cslot) * 9;
        double t[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) { const double v = src[r]; t[r] = (near || (rhs && act)) ? v : 0.0; }
        int spins = 0;
        while (lds_load(&s_step) <= i) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_MAX) { s_bad = 1; break; } }
        asm volatile("" ::: "memory");
        const double2* Li2 = reinterpret_cast<const double2*>(sLinv[i]);
        if (i > 0) {
            const double2* Lp2 = reinterpret_cast<const double2*>(sLsub[i - 1]);      // row stride 5 pairs, pad column 0
#pragma unroll
            for (int r = 0; r < 9; ++r)
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) {
                    const double2 l = Lp2[r * 5 + q2];
                    t[r] = fma(-l.x, wp[2 * q2], t[r]);
                    if (2 * q2 + 1 < 9) t[r] = fma(-l.y, wp[2 * q2 + 1], t[r]);
                }
        }
        double wn[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            wn[r] = 0.0;
#pragma unroll
            for (int q2 = 0; q2 <= r / 2; ++q2) {
                const double2 l = Li2[r * 5 + q2];            // L^-1 is lower triangular: entries beyond the diagonal are 0
                wn[r] = fma(l.x, t[2 * q2], wn[r]);
                if (2 * q2 + 1 <= r) wn[r] = fma(l.y, t[2 * q2 + 1], wn[r]);
            }
        }
        if (act) {
#pragma unroll
            for (int r = 0; r < 9; ++r) { wp[r] = wn[r]; cv.W[(size_t)((i0 + i) * 9 + r) * cv.Wld + col] = wn[r]; }
        }
    }
    if (s_bad && threadIdx.x == 64) d.ctrl->solver_ok = 0;
}

// dd.sys tile (ta, tb), ta >= tb, of the dense system  A - W_B^T W_B  (32 x 32, matrix cores); the tiles of block row
// ta == Pdpad / 32 carry the right-hand side  b_d - W_B^T w_b  in their first row.  Only the rows of W that belong to
// segments whose column window meets both tiles are read (everything else in those columns is zero).
__global__ __launch_bounds__(256) void k_chain_schur(DevBuf d, ChainView cv, DevBuf dd) {
    const int T = cv.Pdpad / 32;
    const int b = blockIdx.x, ntri = T * (T + 1) / 2;
    int ta, tb;
    if (b < ntri) {
        ta = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
        while ((ta + 1) * (ta + 2) / 2 <= b) ++ta;
        while (ta * (ta + 1) / 2 > b) --ta;
        tb = b - ta * (ta + 1) / 2;
    } else { ta = T; tb = b - ntri; }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int tr = wv >> 1, tc = wv & 1;
    const int ldd = dd.ld;
    const bool rhs_row = (ta == T);
    const int acol = rhs_row ? cv.Pd : ta * 32 + tr * 16 + li;     // rhs row: every output row uses w_b; only row 0 is kept
    const int bcol = tb * 32 + tc * 16 + li;
    const bool a_ok = rhs_row || acol < cv.Pd, b_ok = bcol < cv.Pd;
    const double* Wa = cv.W + (a_ok ? acol : cv.Wld - 1);          // column Wld - 1 is zero padding (Wld >= Pd + 2)
    const double* Wb = cv.W + (b_ok ? bcol : cv.Wld - 1);
    double4v acc = (double4v){0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < cv.nseg; ++g) {
        const int wlo = cv.seg_col[2 * g], whi = cv.seg_col[2 * g + 1];
        const bool meets_b = tb * 32 < whi && tb * 32 + 32 > wlo;
        const bool meets_a = rhs_row || (ta * 32 < whi && ta * 32 + 32 > wlo);
        if (!(meets_a && meets_b)) continue;
        const int r0 = cv.seg_start[g] * 9, r1 = cv.seg_start[g + 1] * 9;          // W has 4 zero rows behind the last one
        for (int s0 = r0; s0 < r1; s0 += 32) {
            double av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = s0 + 4 * u + lk;
                const bool in = r < r1;
                av[u] = in ? Wa[(size_t)r * cv.Wld] : 0.0;
                bv[u] = in ? Wb[(size_t)r * cv.Wld] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) + 4 v
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int lr = tr * 16 + lk + 4 * v, lc = tc * 16 + li;
        const int cb = tb * 32 + lc;
        double out;
        if (rhs_row) {
            out = (lr == 0 && cb < cv.Pd) ? d.sys[(size_t)d.Ppad * d.ld + cv.pidx[cb]] - acc[v] : 0.0;
            dd.sys[(size_t)(cv.Pdpad + lr) * ldd + cb] = out;
        } else {
            const int ca = ta * 32 + lr;
            if (ca < cv.Pd && cb < cv.Pd) out = sym_at(d.sys, d.ld, cv.pidx[ca], cv.pidx[cb]) - acc[v];
            else out = (ca == cb) ? 1.0 : 0.0;
            dd.sys[(size_t)ca * ldd + cb] = out;
            if (ta != tb) dd.sys[(size_t)cb * ldd + ca] = out;     // keep the matrix symmetric (debug readers)
        }
    }
}

// per segment: x_c = L^-T (w_b - W_B x_d), then x (system order); workgroup 0 also scatters the dense solution dd.x
constexpr int BACK_THREADS = 256;
__global__ __launch_bounds__(BACK_THREADS) void k_chain_back(DevBuf d, ChainView cv, DevBuf dd) {
    __shared__ double sv[SEGMAX * 9];
    __shared__ double sxw[192];            // dense solution over the segment's column window
    __shared__ double sM[SEGMAX][162];     // L_ii^-1 | L_{i+1,i}
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x;
    const int i0 = cv.seg_start[g], n = cv.seg_start[g + 1] - i0;
    const int wlo = cv.seg_col[2 * g], wn = cv.seg_col[2 * g + 1] - wlo;
    if (g == 0) for (int c = threadIdx.x; c < cv.Pd; c += BACK_THREADS) d.x[cv.pidx[c]] = dd.x[c];
    for (int c = threadIdx.x; c < wn; c += BACK_THREADS) sxw[c] = dd.x[wlo + c];
    for (int idx = threadIdx.x; idx < n * 81; idx += BACK_THREADS) {
        sM[idx / 81][idx % 81] = cv.Ldinv[(size_t)i0 * 81 + idx];
        sM[idx / 81][81 + idx % 81] = cv.Lsub[(size_t)i0 * 81 + idx];
    }
    __syncthreads();
    // v = w_b - W_B x_d over the window: a wave per row
    for (int r = wv; r < n * 9; r += BACK_THREADS / 64) {
        const double* Wr = cv.W + (size_t)(i0 * 9 + r) * cv.Wld;
        double s = 0.0;
        for (int c = lane; c < wn; c += 64) s = fma(Wr[wlo + c], sxw[c], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) sv[r] = Wr[cv.Pd] - s;
    }
    __syncthreads();
    if (wv == 0) {
        const int r = lane < 9 ? lane : 8;     // lanes 0..8 = components; x_{i+1} is kept in lanes 0..8 of `xn`
        double xn = 0.0;
        for (int i = n - 1; i >= 0; --i) {
            const double* Li = sM[i];              // L_ii^-1
            const double* Ls = Li + 81;            // L_{i+1,i}
            double t = sv[i * 9 + r];
            if (i + 1 < n) {
#pragma unroll
                for (int q = 0; q < 9; ++q) t = fma(-Ls[q * 9 + r], lane_bcast(xn, q), t);     // (L_{i+1,i}^T x_{i+1})_r
            }
            double xi = 0.0;
#pragma unroll
            for (int q = 0; q < 9; ++q) xi = fma(Li[q * 9 + r], lane_bcast(t, q), xi);         // (L_ii^-T t)_r ; L^-1 is lower: zeros where q < r
            xn = xi;
            const int gi = cv.cidx[(i0 + i) * 9 + r];
            if (lane < 9 && gi >= 0) d.x[gi] = xi;
        }
    }
}

void launch_chain_elim(const DevBuf& d, const ChainView& cv, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_elim, dim3(cv.nseg), dim3(ELIM_THREADS), 0, s, d, cv);
}
void launch_chain_schur(const DevBuf& d, const ChainView& cv, const DevBuf& dd, hipStream_t s) {
    const int T = cv.Pdpad / 32;
    hipLaunchKernelGGL(k_chain_schur, dim3(T * (T + 1) / 2 + T), dim3(256), 0, s, d, cv, dd);
}
void launch_chain_back(const DevBuf& d, const ChainView& cv, const DevBuf& dd, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_back, dim3(cv.nseg), dim3(BACK_THREADS), 0, s, d, cv, dd);
}

}  // namespace plba

// plba_chain.hip — elimination of the velocity / bias ("chain") variables ahead of the dense factorisation.
//
// Of the 15 dimensions a keyframe contributes to the reduced camera system only 6 (position, rotation) are coupled to
// landmarks.  The other 9 (velocity, gyro / accel bias deltas) are touched by the IMU edges alone: in keyframe order
// their 9 x 9 blocks form a block-TRIDIAGONAL matrix C, coupled to the pose dimensions by a sparse B:
//
//     [ C   B ] [x_c]   [b_c]        C = L_c L_c^T (block bidiagonal),  W = L_c^-1 [B | b_c]
//     [ B^T A ] [x_p] = [b_p]        (A - W_B^T W_B) x_p = b_p - W_B^T w_b,   x_c = L_c^-T (w_b - W_B x_p)
//
// C is factored by ONE workgroup at the in-wave pivot rate (no kernel boundary per block step, no trailing matrix in
// memory) while the ~300 coupled columns ride along in the other wavefronts of the same workgroup, one block step
// behind; a matrix-core SYRK then forms the 6K x 6K dense system, which goes through the ordinary blocked
// factorisation (plba_dense.hip) with 10 instead of 24 block steps at the headline size.
//
//   k_chain_elim   wave 0: per chain block  C_ii -= L_i,i-1 L_i,i-1^T, 9 x 9 Cholesky and inverse in registers
//                          (v_readlane broadcasts), L_i+1,i = C_i+1,i L_ii^-T;  published through LDS (4-deep ring)
//                  waves 1-5: a lane per coupled column:  w_i = L_ii^-1 (B_i - L_i,i-1 w_i-1)
//   k_chain_schur  tile (a,b) of  A - W^T W  (v_mfma_f64_16x16x4_f64), right-hand side row, identity padding
//   k_chain_back   v = w_b - W_B x_p (a wave per row), backward block substitution, scatter of x into system order
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
constexpr int RING = 8;               // LDS ring of published chain factors; the chain wave checks the column waves' progress every 4th step
constexpr int CHUNK = 32;             // block steps whose C blocks are staged in LDS at a time (41 KB)
constexpr int COLW_MAX = 15;          // k_chain_elim: wave 0 = chain, up to 15 waves of column lanes (1024 threads)
}  // namespace

// NCS: columns per lane (1 up to 960 coupled columns, 2 up to 1920); colw: number of column waves
// MAXT: launch bound (384 keeps the register budget of the chain wave at the headline size)
#ifdef PLBA_STAMPS
#define ESTAMP(slot) do { if (lane == 0 && i == 20) d.maxd_part[40 + slot] = (double)__builtin_readcyclecounter(); } while (0)
#else
#define ESTAMP(slot) do {} while (0)
#endif
template <int NCS, int MAXT>
__global__ __launch_bounds__(MAXT) void k_chain_elim(DevBuf d, ChainView cv, int colw) {
    __shared__ __attribute__((aligned(16))) double sLsub[RING][90], sLinv[RING][90];     // ring indexed by block step & (RING - 1); rows of 9 padded to 10: aligned pairs
    __shared__ double sA[81];
    __shared__ int s_step, s_prog[COLW_MAX], s_bad;   // chain steps published; steps completed per column wave
    // staged per chunk of CHUNK block steps (dynamic LDS, 86 KB): nothing inside a step touches global memory for input
    extern __shared__ double s_stage[];
    double (*sCg)[162] = reinterpret_cast<double (*)[162]>(s_stage);                       // C_ii (81) | C_{i+1,i} (81)
    double (*sBg)[162] = reinterpret_cast<double (*)[162]>(s_stage + CHUNK * 162);         // B_i against the pose dims of blocks i-1, i, i+1: [3][6][9]
    double* sRhs = s_stage + 2 * CHUNK * 162;                                              // b_c of the chunk: [CHUNK][9]
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = d.ld, n = cv.nblk;
    if (threadIdx.x == 0) { s_step = 0; s_bad = 0; }
    if (threadIdx.x < COLW_MAX) s_prog[threadIdx.x] = 0;
    // wave 0 (chain) state
    bool bad = false;
    const int e0 = lane, e1 = lane + 64;              // the (up to) two entries of a 9 x 9 block a chain lane owns
    const int r0 = e0 / 9, c0 = e0 % 9, r1 = (e1 < 81 ? e1 : 0) / 9, c1 = (e1 < 81 ? e1 : 0) % 9;
    // column-lane state (waves 1..colw): w_i = L_ii^-1 (B_i - L_{i,i-1} w_{i-1}), NCS columns per lane
    const int stride = colw * 64;
    const int col0 = (wv - 1) * 64 + lane;
    double wp[NCS][9];
    int pb[NCS], pc[NCS];      // chain block of the column's keyframe, index (0..5) of the column among that keyframe's pose dimensions
    bool act[NCS], rhs[NCS];
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
        const int col = col0 + cs * stride;
        rhs[cs] = (col == cv.Pd);
        act[cs] = wv > 0 && col <= cv.Pd;
        pb[cs] = (act[cs] && !rhs[cs]) ? cv.pblk[col] : 0;
        pc[cs] = (act[cs] && !rhs[cs]) ? col - cv.pcol0[pb[cs]] : 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) wp[cs][r] = 0.0;
    }
    for (int ch0 = 0; ch0 < n; ch0 += CHUNK) {
        const int ch1 = min(ch0 + CHUNK, n);
        __syncthreads();                               // everybody is done with the previous chunk's sCg
        // stage this chunk's C blocks: one scattered gather per entry, all in flight at once
        for (int idx = threadIdx.x; idx < (ch1 - ch0) * 162; idx += blockDim.x) {
            const int blk = ch0 + idx / 162, e = idx % 162;
            const int32_t* ci = cv.cidx + blk * 9;
            double v;
            if (e < 81) {
                const int gi = ci[e / 9], gj = ci[e % 9];
                v = (gi < 0 || gj < 0) ? (e / 9 == e % 9 ? 1.0 : 0.0) : sym_at(d.sys, ld, gi, gj);
            } else {
                const int ga = ci[9 + (e - 81) / 9], gt = ci[(e - 81) % 9];      // row of block blk + 1 (sentinel row of -1 behind the last block)
                v = (ga < 0 || gt < 0) ? 0.0 : sym_at(d.sys, ld, ga, gt);
            }
            sCg[idx / 162][e] = v;
            // coupling of block `blk` with the pose dimensions of the keyframes of blocks blk-1, blk, blk+1
            const int dl = e / 54, kc = (e % 54) / 9, rr = e % 9;
            const int nb = blk + dl - 1;
            const int c0 = (nb >= 0 && nb < n) ? cv.pcol0[nb] : -1;
            const int gi = ci[rr];
            double bvv = 0.0;
            if (gi >= 0 && c0 >= 0) bvv = sym_at(d.sys, ld, gi, cv.pidx[c0 + kc]);
            sBg[idx / 162][e] = bvv;
            if (e < 9) sRhs[(idx / 162) * 9 + e] = ci[e] < 0 ? 0.0 : d.sys[(size_t)d.Ppad * ld + ci[e]];
        }
        __syncthreads();
        if (wv == 0) {
            for (int i = ch0; i < ch1; ++i) {
                if (i >= 4 && (i & 3) == 0) {      // ring of 8: before writing slots i..i+3 the column waves must be done with step i - 4
                    int spins = 0;
                    while (true) {
                        int mn = lds_load(&s_prog[0]);
                        for (int q = 1; q < colw; ++q) mn = min(mn, lds_load(&s_prog[q]));
                        if (mn >= i - 3) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_MAX) { bad = true; break; }
                    }
                }
                ESTAMP(0);
#ifdef PLBA_STAMPS
                if (lane == 0 && i >= 20 && i < 24) d.maxd_part[60 + 2 * (i - 20)] = (double)__builtin_readcyclecounter();
#endif
                const bool has_next = i + 1 < n;
                const double* Cg = sCg[i - ch0];
                double v0 = Cg[e0], v1 = Cg[e1 < 81 ? e1 : 0];
                double cs0[9], cs1[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) { cs0[t] = Cg[81 + r0 * 9 + t]; cs1[t] = Cg[81 + r1 * 9 + t]; }
                if (i > 0) {       // (a) C_ii - L_{i,i-1} L_{i,i-1}^T
                    const double* Lp = sLsub[(i - 1) & (RING - 1)];
#pragma unroll
                    for (int t = 0; t < 9; ++t) { v0 = fma(-Lp[r0 * 10 + t], Lp[c0 * 10 + t], v0); v1 = fma(-Lp[r1 * 10 + t], Lp[c1 * 10 + t], v1); }
                }
                sA[e0] = v0;
                if (e1 < 81) sA[e1] = v1;
                ESTAMP(1);
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
                ESTAMP(2);
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
                ESTAMP(3);
                double* Li = sLinv[i & (RING - 1)];
                if (lane < 9) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) { Li[t * 10 + lane] = x[t]; cv.Ldinv[(size_t)i * 81 + t * 9 + lane] = x[t]; }
                    Li[lane * 10 + 9] = 0.0;
                }
                // (d) L_{i+1,i} = C_{i+1,i} L_ii^-T
                if (has_next) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int t = 0; t < 9; ++t) { s0 = fma(cs0[t], Li[c0 * 10 + t], s0); s1 = fma(cs1[t], Li[c1 * 10 + t], s1); }
                    double* Lo = sLsub[i & (RING - 1)];
                    Lo[r0 * 10 + c0] = s0;
                    cv.Lsub[(size_t)i * 81 + e0] = s0;
                    if (e1 < 81) { Lo[r1 * 10 + c1] = s1; cv.Lsub[(size_t)i * 81 + e1] = s1; }
                    if (lane < 9) Lo[lane * 10 + 9] = 0.0;
                }
                asm volatile("" ::: "memory");
                lds_store(&s_step, i + 1);        // a wave's LDS operations execute in order: the data above is visible first
                ESTAMP(4);
#ifdef PLBA_STAMPS
                if (lane == 0 && i >= 20 && i < 24) d.maxd_part[61 + 2 * (i - 20)] = (double)__builtin_readcyclecounter();
#endif
#ifdef PLBA_STAMPS
                if (lane == 0 && i == 21) d.maxd_part[45] = (double)__builtin_readcyclecounter();
#endif
            }
        } else {
            for (int i = ch0; i < ch1; ++i) {
                // W is block lower-trapezoidal: columns of keyframes beyond block i + 1 are still zero.  A wave whose
                // lanes are all in that state has nothing to do in this step (W keeps its zeros from allocation).
                bool idle = true;
#pragma unroll
                for (int cs = 0; cs < NCS; ++cs) idle = idle && (!act[cs] || (!rhs[cs] && pb[cs] > i + 1));
                if (__all(idle)) {
                    int spins0 = 0;
                    while (lds_load(&s_step) <= i) { __builtin_amdgcn_s_sleep(1); if (++spins0 > SPIN_MAX) { s_bad = 1; break; } }
                    if (lane == 0) lds_store(&s_prog[wv - 1], i + 1);
                    continue;
                }
                double t[NCS][9];
#pragma unroll
                for (int cs = 0; cs < NCS; ++cs) {
                    // B is sparse: chain block i only couples to the pose dimensions of the neighbouring keyframes
                    const int dl = pb[cs] - i + 1;                      // 0, 1, 2 where there is coupling
                    const bool near = act[cs] && !rhs[cs] && dl >= 0 && dl <= 2;
                    const double* src = rhs[cs] ? sRhs + (i - ch0) * 9 : sBg[i - ch0] + ((near ? dl : 0) * 6 + pc[cs]) * 9;
#pragma unroll
                    for (int r = 0; r < 9; ++r) { const double v = src[r]; t[cs][r] = (near || (rhs[cs] && act[cs])) ? v : 0.0; }
                }
                if (wv == 1) ESTAMP(8);
                int spins = 0;
                while (lds_load(&s_step) <= i) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_MAX) { s_bad = 1; break; } }
                asm volatile("" ::: "memory");
                if (wv == 1) ESTAMP(9);
                const double2* Lp2 = reinterpret_cast<const double2*>(sLsub[(i + RING - 1) & (RING - 1)]);      // step i - 1; row stride 5 pairs
                const double2* Li2 = reinterpret_cast<const double2*>(sLinv[i & (RING - 1)]);
                if (i > 0) {
#pragma unroll
                    for (int r = 0; r < 9; ++r)
#pragma unroll
                        for (int q2 = 0; q2 < 5; ++q2) {
                            const double2 l = Lp2[r * 5 + q2];        // the pad column holds 0
#pragma unroll
                            for (int cs = 0; cs < NCS; ++cs) {
                                t[cs][r] = fma(-l.x, wp[cs][2 * q2], t[cs][r]);
                                if (2 * q2 + 1 < 9) t[cs][r] = fma(-l.y, wp[cs][2 * q2 + 1], t[cs][r]);
                            }
                        }
                }
                double wn[NCS][9];
#pragma unroll
                for (int r = 0; r < 9; ++r) {
#pragma unroll
                    for (int cs = 0; cs < NCS; ++cs) wn[cs][r] = 0.0;
#pragma unroll
                    for (int q2 = 0; q2 <= r / 2; ++q2) {
                        const double2 l = Li2[r * 5 + q2];            // L^-1 is lower triangular: entries beyond the diagonal are 0
#pragma unroll
                        for (int cs = 0; cs < NCS; ++cs) {
                            wn[cs][r] = fma(l.x, t[cs][2 * q2], wn[cs][r]);
                            if (2 * q2 + 1 <= r) wn[cs][r] = fma(l.y, t[cs][2 * q2 + 1], wn[cs][r]);
                        }
                    }
                }
#pragma unroll
                for (int cs = 0; cs < NCS; ++cs) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) wp[cs][r] = wn[cs][r];
                    const int col = col0 + cs * stride;
                    if (col < cv.Wld) {
#pragma unroll
                        for (int r = 0; r < 9; ++r) cv.W[(size_t)(i * 9 + r) * cv.Wld + col] = act[cs] ? wp[cs][r] : 0.0;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS reads of step i are complete
                if (lane == 0) lds_store(&s_prog[wv - 1], i + 1);
                if (wv == 1) ESTAMP(10);
#ifdef PLBA_STAMPS
                if (lane == 0 && i == 20) d.maxd_part[70 + wv] = (double)__builtin_readcyclecounter();
#endif
            }
        }
    }
    if (wv == 0 && bad && lane == 0) d.ctrl->solver_ok = 0;
    if (s_bad && threadIdx.x == 64) d.ctrl->solver_ok = 0;
}

// dd.sys tile (ta, tb), ta >= tb, of the dense system  A - W_B^T W_B  (32 x 32, matrix cores); the tiles of block row
// ta == Pdpad / 32 carry the right-hand side  b_p - W_B^T w_b  in their first row
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
    const int R = cv.nblk * 9;
    const int ldd = dd.ld;
    const bool rhs_row = (ta == T);
    // operand columns: A operand = column a of W (or the w_b column for the rhs row), B operand = column b
    const int acol = rhs_row ? cv.Pd : ta * 32 + tr * 16 + li;     // rhs row: every output row uses w_b; only row 0 is kept
    const int bcol = tb * 32 + tc * 16 + li;
    const bool a_ok = acol <= cv.Pd && (rhs_row || acol < cv.Pd), b_ok = bcol < cv.Pd;
    double4v acc = (double4v){0.0, 0.0, 0.0, 0.0};
    // W has 4 zero rows behind row R - 1 and unused columns are zero: no bounds checks; 8 k-steps of loads in flight
    const double* Wa = cv.W + (a_ok ? acol : cv.Wld - 1);      // column Wld - 1 is zero padding (Wld >= Pd + 2)
    const double* Wb = cv.W + (b_ok ? bcol : cv.Wld - 1);
    const int Rp = (R + 3) & ~3;
    for (int s0 = 0; s0 < Rp; s0 += 32) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = s0 + 4 * u + lk;
            const bool in = r < Rp;
            av[u] = in ? Wa[(size_t)r * cv.Wld] : 0.0;
            bv[u] = in ? Wb[(size_t)r * cv.Wld] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
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

// x_c = L_c^-T (w_b - W_B x_p), then x (system order) from x_c and the dense solution dd.x
constexpr int BACK_THREADS = 1024;
__global__ __launch_bounds__(BACK_THREADS) void k_chain_back(DevBuf d, ChainView cv, DevBuf dd) {
    extern __shared__ double s_dyn[];
    const int R = cv.nblk * 9;
    double* sv = s_dyn;                    // R:  w_b - W_B x_p
    double* sxp = sv + R;                  // Pd: dense solution
    double* sM = sxp + cv.Pd + (cv.Pd & 1);   // CHUNK x 162: L_ii^-1 | L_{i+1,i} of the current chunk of block steps
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < cv.Pd; c += BACK_THREADS) sxp[c] = dd.x[c];
    __syncthreads();
    // v = w_b - W_B x_p: a wave per row, four rows at a time so that their loads are all in flight together
    constexpr int NW = BACK_THREADS / 64;
    for (int rb = wv * 4; rb < R; rb += NW * 4) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = lane; c < cv.Pd; c += 64) {
            const double xc = sxp[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = fma(cv.W[(size_t)(rb + u) * cv.Wld + c], xc, acc[u]);     // rows R..R+3 of W exist and are zero
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double s = acc[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
            if (lane == 0 && rb + u < R) sv[rb + u] = cv.W[(size_t)(rb + u) * cv.Wld + cv.Pd] - s;
        }
    }
    // backward block substitution, the factors staged in LDS a chunk of block steps at a time
    const int r = lane < 9 ? lane : 8;     // lanes 0..8 of wave 0 = components; x_{i+1} is kept in lanes 0..8 of `xn`
    double xn = 0.0;
    for (int ch1 = cv.nblk; ch1 > 0; ch1 -= CHUNK) {
        const int ch0 = ch1 > CHUNK ? ch1 - CHUNK : 0;
        __syncthreads();
        for (int idx = threadIdx.x; idx < (ch1 - ch0) * 81; idx += BACK_THREADS) {
            const int bi = idx / 81, e = idx % 81;
            sM[bi * 162 + e] = cv.Ldinv[(size_t)ch0 * 81 + idx];
            sM[bi * 162 + 81 + e] = cv.Lsub[(size_t)ch0 * 81 + idx];
        }
        __syncthreads();
        if (wv == 0) {
            for (int i = ch1 - 1; i >= ch0; --i) {
                const double* Li = sM + (i - ch0) * 162;       // L_ii^-1
                const double* Ls = Li + 81;                    // L_{i+1,i}
                double t = sv[i * 9 + r];
                if (i + 1 < cv.nblk) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) t = fma(-Ls[q * 9 + r], lane_bcast(xn, q), t);     // (L_{i+1,i}^T x_{i+1})_r
                }
                double xi = 0.0;
#pragma unroll
                for (int q = 0; q < 9; ++q) xi = fma(Li[q * 9 + r], lane_bcast(t, q), xi);         // (L_ii^-T t)_r ; L^-1 is lower: zeros where q < r
                xn = xi;
                const int gi = cv.cidx[i * 9 + r];
                if (lane < 9 && gi >= 0) d.x[gi] = xi;
            }
        }
    }
    for (int c = threadIdx.x; c < cv.Pd; c += BACK_THREADS) d.x[cv.pidx[c]] = sxp[c];
}

bool chain_elim_supported(int Pd) { return Pd + 1 <= 2 * COLW_MAX * 64; }
void launch_chain_elim(const DevBuf& d, const ChainView& cv, hipStream_t s) {
    const int cols = cv.Pd + 1;
    int colw = (cols + 63) / 64;
    const size_t sh = (size_t)(2 * CHUNK * 162 + CHUNK * 9) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_elim<1, 384>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_elim<1, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_elim<2, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_set = true;
    }
    if (colw <= 5) {
        hipLaunchKernelGGL((k_chain_elim<1, 384>), dim3(1), dim3(64 * (1 + colw)), sh, s, d, cv, colw);
    } else if (colw <= COLW_MAX) {
        hipLaunchKernelGGL((k_chain_elim<1, 1024>), dim3(1), dim3(64 * (1 + colw)), sh, s, d, cv, colw);
    } else {
        colw = (cols + 127) / 128;
        hipLaunchKernelGGL((k_chain_elim<2, 1024>), dim3(1), dim3(64 * (1 + colw)), sh, s, d, cv, colw);
    }
}
void launch_chain_schur(const DevBuf& d, const ChainView& cv, const DevBuf& dd, hipStream_t s) {
    const int T = cv.Pdpad / 32;
    hipLaunchKernelGGL(k_chain_schur, dim3(T * (T + 1) / 2 + T), dim3(256), 0, s, d, cv, dd);
}
void launch_chain_back(const DevBuf& d, const ChainView& cv, const DevBuf& dd, hipStream_t s) {
    const size_t sh = (size_t)(cv.nblk * 9 + cv.Pd + 2 + CHUNK * 162) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_back), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); attr_set = true; }
    hipLaunchKernelGGL(k_chain_back, dim3(1), dim3(BACK_THREADS), sh, s, d, cv, dd);
}

}  // namespace plba

// plba_factor32_dev.h — Cholesky factor AND inverse of a 32 x 32 tile held in LDS, as two 16-column sweeps of ONE wavefront with the
// 16 x 16 coupling blocks on the matrix cores between them (round 5).
//
// The look-ahead pipeline of plba_dense_dev.h (lookahead_factor32) sweeps 32 columns on one wave that holds 16 registers of the tile per
// lane — 58 instructions per column, issue-bound, ~300 cycles — while three follower waves build L^-1 from the published columns and a tail
// of products and scalings runs after the last pivot: 14.5 k cycles per tile (stamps of round 5), the largest item of every
// dependent launch of the reduced-camera solve.  Here, with  A = [A11 .; A21 A22]  (16 x 16 blocks) and  A = Lu D Lu^T:
//
//   sweep 1   A11 = Lu11 D1 Lu11^T  AND  X1 = Lu11^-1 in the same pass (the row operations of the elimination applied to an identity that rides
//             in four more registers: no follower wave, no substitution afterwards)
//   W  = A21 X1^T = Lu21 D1,   Lu21 = W D1^-1,   S = A22 - W Lu21^T                          (two 16 x 16 x 16 products, four MFMAs each)
//   sweep 2   S = Lu22 D2 Lu22^T,  X2 = Lu22^-1
//   Lu^-1 = [X1 0; -X2 Lu21 X1, X2]                                                            (two more products)
//   L = Lu D^1/2,  L^-1 = D^-1/2 Lu^-1                                                         (scalings at write-out, all four waves)
//
// A sweep holds a 16 x 16 block as lane (R = lane / 16, i = lane % 16) <-> row i, columns 4 R .. 4 R + 3: FOUR registers.  Column j:
// the pivot by v_readlane, its reciprocal (v_rcp_f64 + two Newton steps), the multipliers  m_i = A[i][j] / d_j  from the 16 lanes that hold
// column j to all four 16-lane rows (v_permlane32_swap + v_permlane16_swap: gfx950), row j of the block to the lanes of each row by DPP
// row_newbcast (the one DPP control the 64-bit pipeline takes), four FMAs for A and four for X.  ~40 instructions per column, half of them
// 32-bit moves.
#pragma once
#include "plba_dense_dev.h"

namespace plba {

template <int J> __device__ __forceinline__ double row_bcast16(double v) {      // lane J of every 16-lane row -> all lanes of that row
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + J, 0xF, 0xF, true);      // row_newbcast:J
    hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + J, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// (v_fmac_f64_dpp with row_newbcast — the two moves and the FMA of an update as ONE instruction; the 64-bit pipeline takes this DPP control on its
// VOP2 forms — assembles and was tried through inline assembly: back to back it gives wrong factors, with an s_nop 0 in front of each it is right
// and exactly as fast as the three instructions, 3.3 k cycles per sweep: the wait states are the hardware's.)
template <int R> __device__ __forceinline__ double rows_bcast(double v) {       // 16-lane row R of the wave -> all four rows (same position)
    const int l0 = __double2loint(v), h0 = __double2hiint(v);
    // v_permlane32_swap a, a: first result = the low 32 lanes everywhere, second = the high 32 lanes everywhere
    const auto a = __builtin_amdgcn_permlane32_swap(l0, l0, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(h0, h0, false, false);
    const int lo = (R >> 1) ? a[1] : a[0], hi = (R >> 1) ? b[1] : b[0];      // rows now [rA rB rA rB]
    // v_permlane16_swap a, a: first result = [row0 row0 row2 row2], second = [row1 row1 row3 row3]
    const auto c = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto e = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((R & 1) ? e[1] : e[0], (R & 1) ? c[1] : c[0]);
}

constexpr int F32_ST = 18;      // row stride of the 16 x 16 staging blocks

// One column of the 16-column sweep (see above).  a: the block, x: Lu^-1 under construction, lu: finished columns of -Lu (register t of the
// lanes of row R: column 4 R + t).  Nothing is stored to LDS inside the sweep: every lane storing the same
// pivot to the same address (the first form, chosen to avoid exec masking) is a 64-way bank conflict per store — ~110 of a column's 230
// cycles went there.
template <int J>
__device__ __forceinline__ void sweep16_step(double (&a)[4], double (&x)[4], double (&lu)[4], double& dreg, const int R, const int i, const int lane, const int bp_addr) {
    constexpr int RJ = J / 4, TJ = J % 4;
    const double col = a[TJ];                                  // lanes of row RJ: A[i][J]
    const double d = bcast_lane(col, 16 * RJ + J);             // the pivot, uniform
    const double rinv = fast_rcp(d);
    if constexpr (J < 15) {
        // the multipliers -Lu[i][J] from the lanes of row RJ to lane i of every row: ds_bpermute (two issue slots; the permlane swaps take twelve)
        const double mneg = col * -rinv;
        const int lo = __builtin_amdgcn_ds_bpermute(bp_addr + 64 * RJ, __double2loint(mneg)), hi = __builtin_amdgcn_ds_bpermute(bp_addr + 64 * RJ, __double2hiint(mneg));
        const double mall = __hiloint2double(hi, lo);
        const double nm = (i > J) ? mall : 0.0;                // rows <= J are finished (rows of X) or dead (rows of A)
        lu[TJ] = (R == RJ) ? nm : lu[TJ];                      // (MINUS Lu: the sign is taken back where the block is read)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (12 + t > J) a[t] = fma(nm, row_bcast16<J>(a[t]), a[t]);        // columns 4 R + t > J somewhere
            if (t <= J) x[t] = fma(nm, row_bcast16<J>(x[t]), x[t]);            // row J of Lu^-1 ends at column J
        }
    }
}
// a[t] = block[i][4 R + t] on entry; on exit x[t] = (Lu^-1)[i][4 R + t], lu[t] = -Lu[i][4 R + t] below the diagonal (0 elsewhere), dreg = pivot i
__device__ __forceinline__ void sweep16(double (&a)[4], double (&x)[4], double (&lu)[4], double& dreg, const int lane) {
    const int R = lane >> 4, i = lane & 15, bp = i * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) { x[t] = (i == 4 * R + t) ? 1.0 : 0.0; lu[t] = 0.0; }
    dreg = 1.0;
    sweep16_step<0>(a, x, lu, dreg, R, i, lane, bp);   sweep16_step<1>(a, x, lu, dreg, R, i, lane, bp);
    sweep16_step<2>(a, x, lu, dreg, R, i, lane, bp);   sweep16_step<3>(a, x, lu, dreg, R, i, lane, bp);
    sweep16_step<4>(a, x, lu, dreg, R, i, lane, bp);   sweep16_step<5>(a, x, lu, dreg, R, i, lane, bp);
    sweep16_step<6>(a, x, lu, dreg, R, i, lane, bp);   sweep16_step<7>(a, x, lu, dreg, R, i, lane, bp);
    sweep16_step<8>(a, x, lu, dreg, R, i, lane, bp);   sweep16_step<9>(a, x, lu, dreg, R, i, lane, bp);
    sweep16_step<10>(a, x, lu, dreg, R, i, lane, bp);  sweep16_step<11>(a, x, lu, dreg, R, i, lane, bp);
    sweep16_step<12>(a, x, lu, dreg, R, i, lane, bp);  sweep16_step<13>(a, x, lu, dreg, R, i, lane, bp);
    sweep16_step<14>(a, x, lu, dreg, R, i, lane, bp);  sweep16_step<15>(a, x, lu, dreg, R, i, lane, bp);
    // the pivots are the diagonal of what is left in a: row J is not touched after column J (its multipliers are 0)
    const int t = i & 3;
    dreg = t == 0 ? a[0] : t == 1 ? a[1] : t == 2 ? a[2] : a[3];      // valid in the lanes with R == i / 4
    dreg = __shfl(dreg, (i >> 2) * 16 + i);                            // -> lane i of every row
}

struct Factor32Lds {            // (fits the look-ahead pipeline's Look32: the two are never live together)
    double Lu11[16 * F32_ST], X1[16 * F32_ST], W[16 * F32_ST], Lu21[16 * F32_ST], Lu22[16 * F32_ST], X2[16 * F32_ST], Y[16 * F32_ST], Z[16 * F32_ST];
    double d[32], rinv[32], rs[32];
    int half_ready, y_ready, bad;
};
static_assert(sizeof(Factor32Lds) <= sizeof(Look32), "factor32_dpp stages its blocks in the look-ahead pipeline's LDS");

// The tile in sC (32 x 32, row stride LS, complete and visible: call after a barrier); all four waves enter.  L(kb,kb) -> d.Lfac,
// L(kb,kb)^-1 -> d.Linv32[kb] (and, KEEP, -> sLinv: 32 x 32, row stride LS, for the same workgroup's next stage: visible after the caller's next barrier).
// A pivot <= 0 or NaN clears d.ctrl->solver_ok (LinearSolverEigen failure).  sC is overwritten.
#ifdef PLBA_F32STAMPS
static __device__ unsigned long long g_f32stamp[16];
#define F32ST(k) do { if (threadIdx.x == 0) g_f32stamp[k] = __builtin_readcyclecounter(); } while (0)
#else
#define F32ST(k) do {} while (0)
#endif
// One 32 x 32 element of the outputs (r, c) from the staged blocks: L^-1 = D^-1/2 Lu^-1 (row scaling), L = Lu D^1/2 (column scaling:
// d_c rs_c = sqrt(d_c)); the sweeps store Lu below its diagonal only.  rs_r, rs_c: 1 / sqrt(d) of the row's and the column's pivot.
template <bool KEEP>
__device__ __forceinline__ void factor32_emit(const DevBuf& d, const Factor32Lds& F, double* Ig, double* Lg, double* sLinv, const int r, const int c, const double rs_r, const double rs_c, const bool want_inv, const bool want_l) {
    const int rb = r & 15, cb = c & 15;
    double ui, ul;      // Lu^-1[r][c], Lu[r][c]
    if (r < 16) { ui = c < 16 ? F.X1[rb * F32_ST + cb] : 0.0; ul = c < 16 ? (rb > cb ? -F.Lu11[rb * F32_ST + cb] : rb == cb ? 1.0 : 0.0) : 0.0; }
    else if (c < 16) { ui = -F.Z[rb * F32_ST + cb]; ul = F.Lu21[rb * F32_ST + cb]; }
    else { ui = F.X2[rb * F32_ST + cb]; ul = rb > cb ? -F.Lu22[rb * F32_ST + cb] : rb == cb ? 1.0 : 0.0; }
    if (want_inv) { const double vi = ui * rs_r; Ig[r * 32 + c] = vi; if (KEEP) sLinv[r * LS + c] = vi; }
    if (want_l) Lg[(size_t)r * d.ld + c] = ul * (F.d[c] * rs_c);
}
__device__ __forceinline__ bool factor32_wait(int* flag, Factor32Lds& F) {
    int spins = 0;
    while (!vload(flag)) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { vstore(&F.bad, 1); return false; } }
    asm volatile("" ::: "memory");
    return true;
}
// call before the barrier that precedes factor32_dpp (any one thread set suffices; every thread may call)
__device__ __forceinline__ void factor32_reset(Look32& S, const int tid) {
    Factor32Lds& F = *reinterpret_cast<Factor32Lds*>(&S);
    if (tid == 0) { F.half_ready = 0; F.y_ready = 0; F.bad = 0; }
}
// Wave 0 carries the dependent chain (sweep 1, S, sweep 2, Z); once the first half is staged (one LDS flag) wave 1 forms Y = Lu21 X1 and
// waves 2, 3 write everything that only needs the first half — rows 0..15 of L^-1, columns 0..15 of L — while sweep 2 runs; after the
// closing barrier all four waves write the rest.
template <bool KEEP>
__device__ __forceinline__ void factor32_dpp(const DevBuf& d, const int kb, double* sC, Factor32Lds& F, const int wv, const int lane, double* sLinv = nullptr) {
    const int li = lane & 15, lk = lane >> 4;
    double* Ig = d.Linv32 + (size_t)kb * 1024;
    double* Lg = d.Lfac + (size_t)(kb * 32) * d.ld + kb * 32;
    if (wv == 0) {
        const int R = lane >> 4, i = lane & 15;
        double a[4], x[4], lu[4], dreg;
        F32ST(0);
        {
            const double2* p = reinterpret_cast<const double2*>(sC + i * LS + 4 * R);
            const double2 v0 = p[0], v1 = p[1];
            a[0] = v0.x; a[1] = v0.y; a[2] = v1.x; a[3] = v1.y;
        }
        sweep16(a, x, lu, dreg, lane);
        F32ST(1);
#pragma unroll
        for (int t = 0; t < 4; ++t) { F.X1[i * F32_ST + 4 * R + t] = x[t]; F.Lu11[i * F32_ST + 4 * R + t] = lu[t]; }
        if (lane < 16) { F.d[lane] = dreg; F.rinv[lane] = fast_rcp(dreg); }
        // W = A21 X1^T (= Lu21 D1): W[r][c] = sum_k A21[r][k] X1[c][k]
        double4v w = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w = __builtin_amdgcn_mfma_f64_16x16x4f64(sC[(16 + li) * LS + kk * 4 + lk], F.X1[li * F32_ST + kk * 4 + lk], w, 0, 0, 0);
        const double rc = F.rinv[li];
#pragma unroll
        for (int v = 0; v < 4; ++v) { F.W[(lk + 4 * v) * F32_ST + li] = w[v]; F.Lu21[(lk + 4 * v) * F32_ST + li] = w[v] * rc; }
        asm volatile("" ::: "memory");
        vstore(&F.half_ready, 1);      // (a wave's LDS operations execute in order: the flag is seen after the blocks)
        // S = A22 - W Lu21^T
        double4v s2 = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s2 = __builtin_amdgcn_mfma_f64_16x16x4f64(F.W[li * F32_ST + kk * 4 + lk], F.Lu21[li * F32_ST + kk * 4 + lk], s2, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) sC[(16 + lk + 4 * v) * LS + 16 + li] -= s2[v];
        F32ST(2);
        {
            const double2* p = reinterpret_cast<const double2*>(sC + (16 + i) * LS + 16 + 4 * R);
            const double2 v0 = p[0], v1 = p[1];
            a[0] = v0.x; a[1] = v0.y; a[2] = v1.x; a[3] = v1.y;
        }
        sweep16(a, x, lu, dreg, lane);
        F32ST(3);
#pragma unroll
        for (int t = 0; t < 4; ++t) { F.X2[i * F32_ST + 4 * R + t] = x[t]; F.Lu22[i * F32_ST + 4 * R + t] = lu[t]; }
        if (lane < 16) F.d[16 + lane] = dreg;
        // Z = X2 Y  (the (2,1) block of Lu^-1 is -Z); Y = Lu21 X1 came from wave 1 long ago
        factor32_wait(&F.y_ready, F);
        double4v z = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) z = __builtin_amdgcn_mfma_f64_16x16x4f64(F.X2[li * F32_ST + kk * 4 + lk], F.Y[(kk * 4 + lk) * F32_ST + li], z, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) F.Z[(lk + 4 * v) * F32_ST + li] = z[v];
        F32ST(4);
        if (lane < 16) {
            const double pv = F.d[16 + lane];
            const bool bad = !(pv > 0.0);
            F.rs[16 + lane] = fast_rsqrt(bad ? 1.0 : pv);
            if ((__any(bad) || vload(&F.bad)) && lane == 0) d.ctrl->solver_ok = 0;
        }
        F32ST(5);
    } else {
        factor32_wait(&F.half_ready, F);
        if (wv == 1) {
            double4v y = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) y = __builtin_amdgcn_mfma_f64_16x16x4f64(F.Lu21[li * F32_ST + kk * 4 + lk], F.X1[(kk * 4 + lk) * F32_ST + li], y, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) F.Y[(lk + 4 * v) * F32_ST + li] = y[v];
            asm volatile("" ::: "memory");
            vstore(&F.y_ready, 1);
            if (lane < 16) {      // the first half's pivots: 1 / sqrt for the closing write-out, and the check
                const double pv = F.d[lane];
                const bool bad = !(pv > 0.0);
                F.rs[lane] = fast_rsqrt(bad ? 1.0 : pv);
                if (__any(bad) && lane == 0) d.ctrl->solver_ok = 0;
            }
        } else {
            // 128 threads: L^-1 rows 0..15 (512 entries: [D1^-1/2 X1 | 0]) and L columns 0..15 of all rows plus the zero block above the
            // diagonal (rows 0..15, columns 16..31)
            const int t = (wv - 2) * 64 + lane;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = e * 128 + t, r = idx >> 5, c = idx & 31;      // r < 16
                const double pr = F.d[r], pc = F.d[c & 15];
                const double rs_r = fast_rsqrt(pr > 0.0 ? pr : 1.0), rs_c = fast_rsqrt(pc > 0.0 ? pc : 1.0);
                factor32_emit<KEEP>(d, F, Ig, Lg, sLinv, r, c, rs_r, rs_c, true, true);                  // rows 0..15: both outputs, all 32 columns
                factor32_emit<KEEP>(d, F, Ig, Lg, sLinv, 16 + r, c & 15, 0.0, rs_c, false, (c < 16));    // L21 (a lane with c >= 16 has nothing here)
            }
        }
    }
    __syncthreads();
    F32ST(6);
    // the second half, all four waves: L^-1 rows 16..31 (512 entries) and L22 (256)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = e * 256 + (int)threadIdx.x, r = 16 + (idx >> 5), c = idx & 31;
        factor32_emit<KEEP>(d, F, Ig, Lg, sLinv, r, c, F.rs[r], F.rs[c], true, c >= 16);
    }
    F32ST(7);
}

// the call sites hold a Look32 (the look-ahead pipeline's LDS, which the dataflow and 64-column forms of the factorisation still use)
template <bool KEEP>
__device__ __forceinline__ void factor32_tile(const DevBuf& d, const int kb, double* sC, Look32& S, const int wv, const int lane, double* sLinv = nullptr) {
    factor32_dpp<KEEP>(d, kb, sC, *reinterpret_cast<Factor32Lds*>(&S), wv, lane, sLinv);
}

}  // namespace plba

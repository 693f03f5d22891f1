// plba_lm_dev.h — fused landmark-major passes of one LM iteration (device bodies; included by plba_kernels.hip).
//
// Round 3.  The record-based path (k_linearize -> k_landmark_hll -> k_schur_pairs -> k_backsub) materialises a compact record per
// observation and gathers it three times; the pair-entry Schur pass alone re-reads every record once per co-observing pair
// (profiles/r02_pmc_traffic.json: 108 MB per iteration at configs[2] against 8.9 MB of algorithmic bytes).  Here nothing is
// materialised: a landmark's observations are evaluated where they are needed, from the observation arrays and the states alone.
//
//   k_lm_schur  (K1-K2 + K5 + K6: computeError / linearizeOplus of the point and line edges, constructQuadraticForm, the landmark
//               side of BlockSolver::solve; IMU/g2otypes.cpp:286-341, 1306-1359)
//               per landmark (8 lanes, one per observation): residuals, Jacobian rows, Huber weight; Hll, bl by a fixed-order DPP
//               reduction over the 8 lanes; (Hll + lambda I) = R R^T; A'_e = Hpl_e R^-T (6 x 3).  The landmark's whole contribution to
//               the reduced camera system is then  sum_e Hpp_e - A' A'^T  over its window of keyframes — a rank-3 update of a
//               48 x 48 matrix per landmark, accumulated over the landmarks of a GROUP (<= 8 keyframes in its window, built at
//               upload) on the matrix cores (v_mfma_f64_16x16x4_f64: a true symmetric rank-k update, k = 3 x landmarks), in a fixed
//               order: deterministic, no atomics.  Each group leaves its 36 pose-pair blocks + right-hand-side parts for
//   k_lm_gather  which adds the groups' parts of every pose-pair block in a fixed order together with the pose-side terms
//               (IMU / prior accumulators, lambda): the assembly of the reduced system.
//   k_lm_trial  (K8 + the trial's computeActiveErrors): back-substitution x_l = D (bl - sum Hpl^T x_p) with the same in-register
//               linearisation, landmark update into the trial buffer, residuals of the trial state, chi2 partials.
//
// Layout of a workgroup: 4 waves x 8 "units" x 8 lanes.  A unit is one 3-dim landmark block: a point (two residual rows per
// observation) or one END POINT of a line (one row per observation; a line is two neighbouring units — its Hll is exactly
// block-diagonal 3 + 3).  Lane `sub` of a unit owns the unit's sub-th observation.  Waves run decoupled (no workgroup barrier
// inside the loop over landmarks); a group's four partial sums are added in wave order at the end.
#pragma once

namespace plba {

typedef double double4v_lm __attribute__((ext_vector_type(4)));

constexpr int LMF_UNITS = 32;            // units per workgroup step (4 waves x 8)
constexpr int LMF_ROWS = 6 * LMF_W;      // 48: rows of a group's local system (slot-major, 6 per keyframe of the window)
constexpr int LMF_SCOLS = 3 * LMF_UNITS; // 96: k-columns of a step's operand of A' A'^T
constexpr int LMF_HCOLS = 2 * LMF_UNITS; // 64: ... of sum Jp^T w Jp (points; lines use half)
struct LmLds {
    double kc[2][LMF_W][KFCAM_STRIDE];   // camera blocks of the window keyframes: [0] linearisation state, [1] trial state (k_lm_trial)
    double xs[LMF_W][6];                 // pose step (dp, dphi) of the window keyframes (k_lm_trial)
    double red[4][8];
    int koff[LMF_W];
};
struct LmAcc {                           // k_lm_schur only
    double op[LMF_SCOLS * LMF_ROWS];     // operand of the rank-k update, [k-column][row] (36 KB); after the last step: the group's 11 output tiles
    double vh[LMF_HCOLS * LMF_ROWS];     // first the right-hand-side vectors [unit][slot][bp 6 | bs 6], then the operand of sum Jp^T w Jp (24 KB)
};

DEV double quad_or_sum(double v) { return quad_sum(v); }
DEV int quad_or_i(int v) {
    v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    return v;
}
DEV double shfl_xor8(double v) {      // the other unit of a line (lanes 8 apart inside a 16-lane row): row_ror:8
    return dpp_get<0x128, 0xf>(v);
}

// 1 / x and 1 / sqrt(x) for normal positive x: the hardware seed + two Newton steps (~1 ulp), a third of the instructions of the
// IEEE division / square root sequences the compiler would emit (no range handling is needed here)
DEV double lm_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
DEV double lm_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    return y * fma(-hx * y, y, 1.5);
}

// what one lane holds of its observation: NR residual rows (2: point, 1: one end point of a line)
template <int NR>
struct LmRows { double j[NR][6], l[NR][3], e[NR], wr, wj, chi, rho; bool has, act; };      // wj: the weight on the pose-side rows (0 for a fixed keyframe)

// residual + Jacobian rows of the lane's observation at one state.  j: rows of Jp (dp, dphi), zero for a fixed keyframe;
// l: rows of Jl on the unit's 3 coordinates; wr = inv_sigma2 * rho'(chi2) (0: no observation, or gated to level 1)
template <bool IS_LINE, int NR>
DEV void lm_eval(const DevBuf& d, const Robust& rb, const double* kc, bool kf_free, const double* L6, const double* meas, double wt, bool has, bool lvl0, int rowsel, bool jac, LmRows<NR>& o) {
    o.has = has; o.act = has && lvl0;
    double e2[2] = {0.0, 0.0}, Jp[12], Jl[6];
    bool dpos;
#pragma unroll
    for (int t = 0; t < 12; ++t) Jp[t] = 0.0;
#pragma unroll
    for (int t = 0; t < 6; ++t) Jl[t] = 0.0;
    if (o.act) {
        if (!IS_LINE) point_edge(d.cam, kc, v3(L6[0], L6[1], L6[2]), meas[0], meas[1], e2, Jp, Jl, dpos, jac);
        else line_edge(d.cam, kc, v3(L6[0], L6[1], L6[2]), v3(L6[3], L6[4], L6[5]), meas[0], meas[1], meas[2], d.fix_q1 != 0, e2, Jp, Jl, dpos, jac);
    }
    const double chi = o.act ? wt * (e2[0] * e2[0] + e2[1] * e2[1]) : 0.0;
    double r0 = chi, r1 = 1.0;
    const int kind = IS_LINE ? PLBA_EDGE_LINE : PLBA_EDGE_POINT;
    if (rb.on[kind]) huber(chi, rb.delta[kind], r0, r1);
    o.chi = chi; o.rho = o.act ? r0 : 0.0; o.wr = o.act ? wt * r1 : 0.0;
    o.wj = kf_free ? o.wr : 0.0;
#pragma unroll
    for (int a = 0; a < NR; ++a) {
        const int ra = IS_LINE ? rowsel : a;
        o.e[a] = (ra == 0) ? e2[0] : e2[1];
#pragma unroll
        for (int c = 0; c < 6; ++c) o.j[a][c] = (ra == 0) ? Jp[c] : Jp[6 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) o.l[a][c] = (ra == 0) ? Jl[c] : Jl[3 + c];
    }
}

// Hll (upper 6) and bl (3) of the unit, summed over its 8 lanes in a fixed order; every lane ends up with the totals
template <int NR>
DEV void lm_hll(const LmRows<NR>& r, double* h, double* b, int& nact) {
#pragma unroll
    for (int t = 0; t < 6; ++t) h[t] = 0.0;
#pragma unroll
    for (int t = 0; t < 3; ++t) b[t] = 0.0;
#pragma unroll
    for (int a = 0; a < NR; ++a) {
        const double* l = r.l[a];
        h[0] += r.wr * l[0] * l[0]; h[1] += r.wr * l[0] * l[1]; h[2] += r.wr * l[0] * l[2];
        h[3] += r.wr * l[1] * l[1]; h[4] += r.wr * l[1] * l[2]; h[5] += r.wr * l[2] * l[2];
        b[0] -= r.wr * l[0] * r.e[a]; b[1] -= r.wr * l[1] * r.e[a]; b[2] -= r.wr * l[2] * r.e[a];
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) h[t] = quad_sum(h[t]);
#pragma unroll
    for (int t = 0; t < 3; ++t) b[t] = quad_sum(b[t]);
    nact = quad_sum_i(r.act ? 1 : 0);
}
// (Hll + lambda I) = R R^T;  Li = R^-1 (lower: [00, 10, 11, 20, 21, 22]), so that D = (Hll + lambda I)^-1 = Li^T Li.  Zero when the
// unit is inactive or the block is not positive definite (the record-based path leaves D = 0 there as well)
DEV void lm_chol_inv(const double* h, double lambda, bool active, double* Li) {
    const double a = h[0] + lambda, b = h[1], c = h[2], e = h[3] + lambda, f = h[4], g = h[5] + lambda;
    const bool oka = active && a > 0.0;
    const double i0 = lm_rcp(oka ? a : 1.0);
    const double l10 = b * i0, l20 = c * i0;
    const double d1 = e - l10 * b;
    const bool okb = oka && d1 > 0.0;
    const double i1 = lm_rcp(okb ? d1 : 1.0);
    const double l21 = (f - l20 * b) * i1;
    const double d2 = g - l20 * c - l21 * l21 * d1;
    const bool ok = okb && d2 > 0.0;
    // unit lower L' with L' diag(a, d1, d2) L'^T: R = L' diag(sqrt), R^-1 = diag(1 / sqrt) L'^-1
    const double s0 = ok ? lm_rsqrt(a) : 0.0, s1 = ok ? lm_rsqrt(d1) : 0.0, s2 = ok ? lm_rsqrt(d2) : 0.0;
    const double m10 = -l10, m21 = -l21, m20 = l10 * l21 - l20;
    Li[0] = s0; Li[1] = ok ? s1 * m10 : 0.0; Li[2] = s1; Li[3] = ok ? s2 * m20 : 0.0; Li[4] = ok ? s2 * m21 : 0.0; Li[5] = s2;
}
DEV void lm_lower_mul(const double* Li, const double* v, double* o) {      // o = Li v
    o[0] = Li[0] * v[0];
    o[1] = Li[1] * v[0] + Li[2] * v[1];
    o[2] = Li[3] * v[0] + Li[4] * v[1] + Li[5] * v[2];
}
DEV void lm_lowerT_mul(const double* Li, const double* v, double* o) {     // o = Li^T v
    o[0] = Li[0] * v[0] + Li[1] * v[1] + Li[3] * v[2];
    o[1] = Li[2] * v[1] + Li[4] * v[2];
    o[2] = Li[5] * v[2];
}
struct LmStep {      // one lane's share of a step's inputs (prefetched a step ahead)
    int slot, k, e, ws, orig;      // ws: the window slot this lane writes in the operands — its observation's, or (a lane without one) one of the slots
                                   // none of the unit's observations uses: the 8 lanes of a unit cover the 8 slots, so the operand never needs clearing
    bool uvalid, has, lvl0, fixed;
    double meas[3], wt, L[6];
};
template <bool IS_LINE>
DEV void lm_load(const DevBuf& d, const LmView& lv, const LmGroup& g, int state, int step, int wv, int lane, LmStep& s) {
    const int unit = step * LMF_UNITS + wv * 8 + (lane >> 3), sub = lane & 7;
    const int n = IS_LINE ? (unit >> 1) : unit;
    s.uvalid = n < g.nlm;
    s.slot = 0; s.k = 0; s.e = 0; s.ws = sub; s.orig = 0; s.has = false; s.lvl0 = false; s.fixed = true; s.wt = 0.0;      // (a unit past the group's end: its 8 lanes zero the 8 slots)
#pragma unroll
    for (int t = 0; t < 3; ++t) s.meas[t] = 0.0;
#pragma unroll
    for (int t = 0; t < 6; ++t) s.L[t] = 0.0;
    if (!s.uvalid) return;
    const int gi = g.lm0 + n;
    s.slot = lv.lm_slot[gi];
    const int e0 = lv.lm_ob0[gi];
    s.k = lv.lm_ob0[gi + 1] - e0;
    s.has = sub < s.k;
    s.e = e0 + sub;
    s.fixed = lv.lm_fixed_g[gi] != 0;
    s.ws = lv.lm_ws8[(size_t)gi * LMF_W + sub];
    const double* Lp = d.lm[state] + (size_t)s.slot * 6;
    s.L[0] = Lp[0]; s.L[1] = Lp[1]; s.L[2] = Lp[2];
    if (IS_LINE) { s.L[3] = Lp[3]; s.L[4] = Lp[4]; s.L[5] = Lp[5]; }
    if (s.has) {
        s.orig = lv.ob_orig[s.e]; s.wt = lv.ob_wt[s.e];
        s.lvl0 = lv.ob_level_g[s.e] == 0;
        if (IS_LINE) { const double* m = lv.meas_ln + (size_t)(s.e - d.Ep) * 3; s.meas[0] = m[0]; s.meas[1] = m[1]; s.meas[2] = m[2]; }
        else { const double2 m = reinterpret_cast<const double2*>(lv.meas_pt)[s.e]; s.meas[0] = m.x; s.meas[1] = m.y; }
    }
}

// camera blocks (and kf_off_pvr) of the group's window at `state`
DEV void lm_stage_window(const DevBuf& d, const LmGroup& g, int state, LmLds& S) {
    if ((int)threadIdx.x < LMF_W) {
        const int p = threadIdx.x;
        if (p < g.nw) { kfcam_make(d.cam, d.kf[state] + (size_t)g.kf[p] * KF_STRIDE, S.kc[0][p]); S.koff[p] = g.off[p]; }
        else { for (int t = 0; t < KFCAM_STRIDE; ++t) S.kc[0][p][t] = 0.0; S.koff[p] = -1; }
    }
}

// ---- k_lm_schur: one group ------------------------------------------------------------------------------------------------------
// MODE 0: the Schur pass proper.  MODE 1 (first iteration of a call, before lambda exists): chi2 of the state, the diagonals
// computeLambdaInit needs (max |Hll_jj| per group; diag of sum Jp^T w Jp per window slot, left in the bp part of the group's
// output), the cached per-edge chi2 — no landmark inverse, no rank-k update.
//
// A step takes 32 units (8 per wave).  All lanes write their unit's 3 k-columns of the operand of A' A'^T (96 columns x 48 rows,
// shared by the workgroup) and their right-hand-side 6-vectors; after a barrier each WAVE owns some of the 16 x 16 output tiles
// and runs them over all 96 columns (so a wave carries 4 tiles' accumulators, not 11: what lets two workgroups share a CU);
// the waves with one tile add up the right-hand-side vectors meanwhile.  The same again for sum Jp^T w Jp with 2 (1) columns per
// unit, whose operand takes the place of the vector table.  Four workgroup barriers per step.
template <bool IS_LINE, int MODE>
DEV void lm_schur_group(const DevBuf& d, const LmView& lv, const int gidx, const int state, const Robust& rb, LmLds& S, LmAcc& A4) {
    constexpr int NR = IS_LINE ? 1 : 2;
    const LmGroup g = lv.grp[gidx];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 7, li = lane & 15, lk = lane >> 4;
    const double lambda = MODE == 0 ? d.ctrl->lambda : 0.0;
    lm_stage_window(d, g, state, S);
    const int nunits = IS_LINE ? 2 * g.nlm : g.nlm;
    const int nsteps = (nunits + LMF_UNITS - 1) / LMF_UNITS;
#ifdef PLBA_STAMPS_LMF
    unsigned long long fs[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; fs[0] = __builtin_readcyclecounter();
#define FSTAMP(i) do { if (step == 0) fs[i] = __builtin_readcyclecounter(); } while (0)
#define FSTAMP1(i) fs[i] = __builtin_readcyclecounter()
#else
#define FSTAMP(i) do {} while (0)
#define FSTAMP1(i) do {} while (0)
#endif
    LmStep cur;
    lm_load<IS_LINE>(d, lv, g, state, 0, wv, lane, cur);
    __syncthreads();
    // tiles of this wave (row offsets of the A and B fragments): A' A'^T lower tiles (0,0) (1,0) | (1,1) (2,0) | (2,1) | (2,2);
    // sum Jp^T w Jp tiles holding a slot's diagonal block (a slot's 6 rows may straddle two 16-row tiles): (2,2) | - | (0,0) (1,0) | (1,1) (2,1)
    const bool s2 = wv < 2, h1 = wv != 1, h2 = wv >= 2;
    double4v_lm accS0 = (double4v_lm){0.0, 0.0, 0.0, 0.0}, accS1 = accS0, accH0 = accS0, accH1 = accS0;
    double accV = 0.0, chi_acc = 0.0, maxd = 0.0;      // accV: wave 2, lanes (slot, dof): bp; wave 3: bs
    double* op = A4.op;
    double* vt = A4.vh;
    const int u32 = wv * 8 + (lane >> 3);      // unit inside the step
    for (int step = 0; step < nsteps; ++step) {
        LmStep nxt;
        if (step + 1 < nsteps) lm_load<IS_LINE>(d, lv, g, state, step + 1, wv, lane, nxt);      // in flight while this step computes
        FSTAMP(1);
        LmRows<NR> r;
        const int rowsel = IS_LINE ? ((lane >> 3) & 1) : 0;
        lm_eval<IS_LINE, NR>(d, rb, S.kc[0][cur.ws], cur.has && S.koff[cur.ws] >= 0, cur.L, cur.meas, cur.wt, cur.has, cur.lvl0, rowsel, true, r);
        FSTAMP(2);
        double h[6], b[3];
        int nact;
        lm_hll<NR>(r, h, b, nact);
        FSTAMP(3);
        const bool active = cur.uvalid && nact > 0 && !cur.fixed;
        if (!IS_LINE || rowsel == 0) chi_acc += r.rho;
        if (MODE == 1 && r.act && (!IS_LINE || rowsel == 0)) {
            d.ob_chi2[cur.orig] = r.chi;
            if (lv.ob_err) { lv.ob_err[2 * (size_t)cur.orig] = r.e[0]; }
        }
        if (MODE == 1 && lv.ob_err && r.act) lv.ob_err[2 * (size_t)cur.orig + (IS_LINE ? rowsel : 1)] = r.e[NR - 1];
        if (lv.dbg_out && cur.uvalid && sub == 0) {      // diagnostics of the parity tests: Hll, bl in the record-based path's layout
            double* ho = d.hll + (size_t)cur.slot * 12 + (IS_LINE ? 6 * rowsel : 0);
            double* bo = d.bl + (size_t)cur.slot * 6 + (IS_LINE ? 3 * rowsel : 0);
#pragma unroll
            for (int t = 0; t < 6; ++t) ho[t] = h[t];
#pragma unroll
            for (int t = 0; t < 3; ++t) bo[t] = b[t];
            if (!IS_LINE || rowsel == 0) d.lm_active[cur.slot] = active ? 1 : 0;
        }
        const int wslot = cur.ws;
        double* v = vt + (u32 * LMF_W + wslot) * 12;
        if (MODE == 1) {
            if (active && sub == 0) maxd = fmax(maxd, fmax(fmax(fabs(h[0]), fabs(h[3])), fabs(h[5])));
            // diag of Jp^T w Jp of this observation -> the vector table, summed per slot below
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double q = 0.0;
#pragma unroll
                for (int a = 0; a < NR; ++a) q += r.wj * r.j[a][c] * r.j[a][c];
                v[c] = q;
            }
            __syncthreads();
            if (wv == 2 && lane < LMF_ROWS) {
                const int p = lane / 6, c = lane % 6;
#pragma unroll 8
                for (int q = 0; q < LMF_UNITS; ++q) accV += vt[(q * LMF_W + p) * 12 + c];
            }
            __syncthreads();
        } else {
            double Li[6], y[3], u[NR][3];
            lm_chol_inv(h, lambda, active, Li);
            lm_lower_mul(Li, b, y);
#pragma unroll
            for (int a = 0; a < NR; ++a) lm_lower_mul(Li, r.l[a], u[a]);
            // A' = wr sum_rows j (x) u  (6 x 3) -> k-columns 3 u32 + m, rows 6 slot + c;  gs = A' y;  gp = -wr sum_rows j e
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double A3[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    double q = 0.0;
#pragma unroll
                    for (int a = 0; a < NR; ++a) q += r.j[a][c] * u[a][m];
                    A3[m] = r.wj * q;      // (a lane without observation or with a gated one: all zeros)
                    op[(3 * u32 + m) * LMF_ROWS + 6 * wslot + c] = A3[m];
                }
                double q = 0.0;
#pragma unroll
                for (int a = 0; a < NR; ++a) q += r.j[a][c] * r.e[a];
                v[c] = -r.wj * q;
                v[6 + c] = A3[0] * y[0] + A3[1] * y[1] + A3[2] * y[2];
            }
            FSTAMP(4);
            __syncthreads();
            // ---- rank-k update, this wave's tiles over the step's 96 k-columns; the one-tile waves also sum the right-hand-side vectors ----
            // (one loop per wave, with the fragments each tile pair shares read once: the operand is streamed from LDS by all four waves)
            if (wv == 0) {      // (0,0) (1,0)
#pragma unroll 4
                for (int s4 = 0; s4 < LMF_SCOLS / 4; ++s4) {
                    const double* col = op + (4 * s4 + lk) * LMF_ROWS + li;
                    const double f0 = col[0], f1 = col[16];
                    accS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, accS0, 0, 0, 0);
                    accS1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f0, accS1, 0, 0, 0);
                }
            } else if (wv == 1) {      // (1,1) (2,0)
#pragma unroll 4
                for (int s4 = 0; s4 < LMF_SCOLS / 4; ++s4) {
                    const double* col = op + (4 * s4 + lk) * LMF_ROWS + li;
                    const double f0 = col[0], f1 = col[16], f2 = col[32];
                    accS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, accS0, 0, 0, 0);
                    accS1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f0, accS1, 0, 0, 0);
                }
            } else if (wv == 2) {      // (2,1)
#pragma unroll 4
                for (int s4 = 0; s4 < LMF_SCOLS / 4; ++s4) {
                    const double* col = op + (4 * s4 + lk) * LMF_ROWS + li;
                    accS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(col[32], col[16], accS0, 0, 0, 0);
                }
            } else {      // (2,2)
#pragma unroll 4
                for (int s4 = 0; s4 < LMF_SCOLS / 4; ++s4) {
                    const double f2 = op[(4 * s4 + lk) * LMF_ROWS + li + 32];
                    accS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f2, accS0, 0, 0, 0);
                }
            }
            if (wv >= 2 && lane < LMF_ROWS) {
                const int p = lane / 6, c = lane % 6 + (wv == 3 ? 6 : 0);
#pragma unroll 8
                for (int q = 0; q < LMF_UNITS; ++q) accV += vt[(q * LMF_W + p) * 12 + c];
            }
            FSTAMP(5);
            __syncthreads();
            // ---- sum Jp^T w Jp: k-columns NR u32 + a hold sqrt(wr) j_a (the operand takes the vector table's place) ----------------------
            const double sw = r.wj > 0.0 ? r.wj * lm_rsqrt(r.wj) : 0.0;
#pragma unroll
            for (int a = 0; a < NR; ++a)
#pragma unroll
                for (int c = 0; c < 6; ++c) vt[(NR * u32 + a) * LMF_ROWS + 6 * wslot + c] = sw * r.j[a][c];
            __syncthreads();
            if (wv == 2) {      // (0,0) (1,0)
#pragma unroll 4
                for (int s4 = 0; s4 < (LMF_UNITS * NR) / 4; ++s4) {
                    const double* col = vt + (4 * s4 + lk) * LMF_ROWS + li;
                    const double f0 = col[0], f1 = col[16];
                    accH0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, accH0, 0, 0, 0);
                    accH1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f0, accH1, 0, 0, 0);
                }
            } else if (wv == 3) {      // (1,1) (2,1)
#pragma unroll 4
                for (int s4 = 0; s4 < (LMF_UNITS * NR) / 4; ++s4) {
                    const double* col = vt + (4 * s4 + lk) * LMF_ROWS + li;
                    const double f1 = col[16], f2 = col[32];
                    accH0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, accH0, 0, 0, 0);
                    accH1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f1, accH1, 0, 0, 0);
                }
            } else if (wv == 0) {      // (2,2)
#pragma unroll 4
                for (int s4 = 0; s4 < (LMF_UNITS * NR) / 4; ++s4) {
                    const double f2 = vt[(4 * s4 + lk) * LMF_ROWS + li + 32];
                    accH0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f2, accH0, 0, 0, 0);
                }
            }
            FSTAMP(6);
            __syncthreads();
        }
        FSTAMP(7);
        cur = nxt;
    }
    FSTAMP1(8);
    // ---- the group's parts go out: every tile lives in exactly one wave ------------------------------------------------------------------
    double* part = lv.part + (size_t)gidx * LMF_PART;
    double* cb = A4.op;      // 11 tiles x 256 doubles = 22.5 KB of the 36 KB operand (free after the loop's last barrier)
    if (MODE == 0) {
        // tile slots: A' A'^T lower tiles I (I + 1) / 2 + J = 0 .. 5; sum Jp^T w Jp tiles (0,0) (1,0) (1,1) (2,1) (2,2) = 6 .. 10
        const int tS0 = wv == 0 ? 0 : wv == 1 ? 2 : wv == 2 ? 4 : 5, tS1 = wv == 0 ? 1 : 3;
        const int tH0 = wv == 0 ? 10 : wv == 2 ? 6 : 8, tH1 = wv == 2 ? 7 : 9;
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            const int o = (lk + 4 * v4) * 16 + li;
            cb[tS0 * 256 + o] = accS0[v4];
            if (s2) cb[tS1 * 256 + o] = accS1[v4];
            if (h1) cb[tH0 * 256 + o] = accH0[v4];
            if (h2) cb[tH1 * 256 + o] = accH1[v4];
        }
    }
    chi_acc = wave_sum(chi_acc); maxd = wave_max(maxd);
    if (lane == 0) { S.red[wv][0] = chi_acc; S.red[wv][1] = maxd; }
    __syncthreads();
    FSTAMP1(9);
    if (MODE == 0) {
        // entry (r, c) of pair block (p <= q): rows a = 6 p + r, b = 6 q + c of the local system, read from the lower tiles
        for (int idx = threadIdx.x; idx < 36 * 36; idx += 256) {
            const int t = idx / 36, rc = idx - t * 36, r = rc / 6, c = rc - r * 6;
            const int q = (t >= 28) ? 7 : (t >= 21) ? 6 : (t >= 15) ? 5 : (t >= 10) ? 4 : (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0;
            const int p = t - q * (q + 1) / 2;
            int a = 6 * p + r, b2 = 6 * q + c;
            if (a < b2) { const int tmp = a; a = b2; b2 = tmp; }      // symmetric: take the lower-triangle copy
            const int ta = a >> 4, tb = b2 >> 4;
            const double sv = cb[(ta * (ta + 1) / 2 + tb) * 256 + (a & 15) * 16 + (b2 & 15)];
            double hv = 0.0;
            if (p == q) hv = cb[(6 + ((ta == tb) ? 2 * ta : 2 * ta - 1)) * 256 + (a & 15) * 16 + (b2 & 15)];      // tiles (0,0) (1,0) (1,1) (2,1) (2,2)
            part[idx] = hv - sv;
        }
    }
    if (lane < LMF_ROWS && (wv == 2 || (MODE == 0 && wv == 3))) part[36 * 36 + (lane / 6) * 12 + (wv == 3 ? 6 : 0) + lane % 6] = accV;
    if (threadIdx.x == 0) {
        d.chi_part[gidx] = (S.red[0][0] + S.red[1][0]) + (S.red[2][0] + S.red[3][0]);
        if (MODE == 1) d.maxd_part[gidx] = fmax(fmax(S.red[0][1], S.red[1][1]), fmax(S.red[2][1], S.red[3][1]));
    }
#ifdef PLBA_STAMPS_LMF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FSTAMP1(10);
    if (MODE == 0 && threadIdx.x == 0 && (gidx == 0 || gidx == lv.ngrp / 2 || gidx == lv.ngrp - 1)) {
        const int o = gidx == 0 ? 0 : gidx == lv.ngrp / 2 ? 16 : 32;
        for (int q = 0; q < 11; ++q) d.dbgbuf[o + q] = (double)(fs[q] - fs[0]);
        d.dbgbuf[o + 11] = (double)nsteps; d.dbgbuf[o + 12] = (double)(long long)__builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---- k_lm_gather: assembly of the landmark-coupled part of the reduced system -------------------------------------------------------
// 36 lanes per pose-pair block (i <= j): sys(i, j) = Himu + Hconst (+ lambda on the diagonal) + sum over the groups whose window holds
// both keyframes, in list order; both mirror images are written and the idle IMU accumulator is cleared, as assemble_part does for the
// rest of the system.  12 lanes per observed free keyframe: the two right-hand-side rows.
// One workgroup per block / per keyframe: the contributions (~30 .. 90 of them) are dealt round-robin to 7 (21) lane groups so that
// their loads are in flight together, then added up in a fixed order through LDS: deterministic.
DEV void lm_gather_part(const DevBuf& d, const LmView& lv, int add_lambda, int bid, int tid, double* s_red /* 256 */) {
    const double lambda = d.ctrl->lambda;
    const int ld = d.ld;
    if (bid < lv.nblk) {
        const int b = bid, rc = tid % 36, ch = tid / 36;      // 7 chunks of contributions (lanes 252 .. 255 idle)
        const int n0 = lv.blk_start[b], n1 = lv.blk_start[b + 1];
        double a0 = 0.0, a1 = 0.0;
        if (ch < 7) {
            int s = n0 + ch;
            for (; s + 7 < n1; s += 14) {
                const int src0 = lv.blk_src[s], src1 = lv.blk_src[s + 7];      // group * 36 + pair index
                a0 += lv.part[(size_t)(src0 / 36) * LMF_PART + (src0 % 36) * 36 + rc];
                a1 += lv.part[(size_t)(src1 / 36) * LMF_PART + (src1 % 36) * 36 + rc];
            }
            if (s < n1) { const int src0 = lv.blk_src[s]; a0 += lv.part[(size_t)(src0 / 36) * LMF_PART + (src0 % 36) * 36 + rc]; }
        }
        s_red[tid] = a0 + a1;
        __syncthreads();
        if (tid >= 36) return;
        const int ij = lv.blk_ij[b], i = ij & 0xffff, j = (ij >> 16) & 0xffff;
        const int r = rc / 6, c = rc % 6;
        const size_t e0 = (size_t)(d.kf_off_pvr[i] + pmap(r)) * ld + d.kf_off_pvr[j] + pmap(c);
        const size_t e1 = (size_t)(d.kf_off_pvr[j] + pmap(c)) * ld + d.kf_off_pvr[i] + pmap(r);
        double v = d.Himu[e0] + d.Hconst[e0];
        if (i == j && r == c && add_lambda) v += lambda;
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 7; ++q) sum += s_red[q * 36 + rc];
        v += sum;
        d.sys[e0] = v; d.Himu_alt[e0] = 0.0;
        if (i != j) { d.sys[e1] = v; d.Himu_alt[e1] = 0.0; }
        return;
    }
    const int k = bid - lv.nblk;
    if (k >= lv.nrow) return;
    const int t = tid % 12, ch = tid / 12;      // 21 chunks
    const int n0 = lv.row_start[k], n1 = lv.row_start[k + 1];
    double acc = 0.0;
    if (ch < 21)
        for (int s = n0 + ch; s < n1; s += 21) {
            const int src = lv.row_src[s];      // group * LMF_W + slot
            acc += lv.part[(size_t)(src / LMF_W) * LMF_PART + 36 * 36 + (src % LMF_W) * 12 + t];
        }
    s_red[tid] = acc;
    __syncthreads();
    if (tid >= 6) return;
    double bp = 0.0, bs = 0.0;
#pragma unroll
    for (int q = 0; q < 21; ++q) { bp += s_red[q * 12 + tid]; bs += s_red[q * 12 + 6 + tid]; }
    const int col = d.kf_off_pvr[lv.row_kf[k]] + pmap(tid);
    const double base = d.bimu[col] + d.bprior[col];
    d.sys[(size_t)d.Ppad * ld + col] = base + bp - bs;      // bschur = bp - sum Hpl D bl
    d.sys[(size_t)(d.Ppad + 1) * ld + col] = base + bp; d.bpg[col] = base + bp; d.bimu_alt[col] = 0.0;
}
__host__ __device__ inline int lm_gather_blocks(const LmView& lv) { return lv.nblk + lv.nrow; }
// first iteration: kfdiag[k][c] = sum over the groups of the diagonal parts k_lm_schur<MODE 1> left in their bp slots
DEV void lm_gather_diag(const DevBuf& d, const LmView& lv, int bid, int tid, double* s_red) {
    const int k = bid;
    if (k >= lv.nrow) return;
    const int t = tid % 12, ch = tid / 12;
    double acc = 0.0;
    if (ch < 21 && t < 6)
        for (int s = lv.row_start[k] + ch; s < lv.row_start[k + 1]; s += 21) {
            const int src = lv.row_src[s];
            acc += lv.part[(size_t)(src / LMF_W) * LMF_PART + 36 * 36 + (src % LMF_W) * 12 + t];
        }
    s_red[tid] = acc;
    __syncthreads();
    if (tid >= 6) return;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 21; ++q) v += s_red[q * 12 + tid];
    d.kfdiag[lv.row_kf[k] * 6 + tid] = v;
}
// the rest of the system (IMU / prior blocks and diagonal outside the gathered blocks, the other right-hand-side columns)
DEV void lm_assemble_rest(const DevBuf& d, const LmView& lv, int add_lambda, int bid, int nblocks, int tid, int nthreads) {
    const double lambda = d.ctrl->lambda;
    const int n = lv.nalist2;
    const size_t stride = (size_t)nblocks * nthreads;
    for (size_t k = (size_t)bid * nthreads + tid; k < (size_t)n + (size_t)d.ld; k += stride) {
        if (k < (size_t)n) {
            const int idx = lv.alist2[k];
            const int r = idx / d.ld, c = idx - r * d.ld;
            double v = d.Himu[idx] + d.Hconst[idx];
            if (r == c && add_lambda) v += (r < d.P) ? lambda : 1.0;
            d.Himu_alt[idx] = 0.0;
            d.sys[idx] = v;
        } else {
            const int c = (int)(k - n);
            if (lv.col_gather[c]) continue;
            const double v = d.bimu[c] + d.bprior[c];
            d.bpg[c] = v; d.bimu_alt[c] = 0.0;
            d.sys[(size_t)d.Ppad * d.ld + c] = v;
            d.sys[(size_t)(d.Ppad + 1) * d.ld + c] = v;
        }
    }
}

// ---- k_lm_trial: one group ----------------------------------------------------------------------------------------------------------
// xd / cv: the pose step is read from the dense solution when the chain segments ride in the same launch (they are still writing d.x)
template <bool IS_LINE>
DEV void lm_trial_group(const DevBuf& d, const LmView& lv, const int gidx, const int cur_state, const int trial, const Robust& rb, const ChainView& cv, const double* xd, const bool from_dense, LmLds& S) {
    constexpr int NR = IS_LINE ? 1 : 2;
    const LmGroup g = lv.grp[gidx];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 7;
    const double lambda = d.ctrl->lambda;
    const bool sok = d.ctrl->solver_ok != 0;
    if ((int)threadIdx.x < LMF_W) {
        const int p = threadIdx.x;
        double u9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (p < g.nw) {
            const int k = g.kf[p], o = g.off[p];
            const double* s = d.kf[cur_state] + (size_t)k * KF_STRIDE;
            if (o >= 0 && sok) {
                if (from_dense) { const int32_t* sc = cv.slotcol + cv.kfpos[k] * NSLOT; u9[0] = xd[sc[0]]; u9[1] = xd[sc[1]]; u9[2] = xd[sc[2]]; u9[6] = xd[sc[3]]; u9[7] = xd[sc[4]]; u9[8] = xd[sc[5]]; }
                else { u9[0] = d.x[o]; u9[1] = d.x[o + 1]; u9[2] = d.x[o + 2]; u9[6] = d.x[o + 6]; u9[7] = d.x[o + 7]; u9[8] = d.x[o + 8]; }
            }
            double st[KF_STRIDE];
#pragma unroll
            for (int t = 0; t < KF_STRIDE; ++t) st[t] = s[t];
            kfcam_make(d.cam, st, S.kc[0][p]);
            if (o >= 0 && sok) kf_oplus_pvr(s, u9, st);      // the pose part of update(): the same function, the same inputs as the keyframe update
            kfcam_make(d.cam, st, S.kc[1][p]);
            S.koff[p] = o;
        } else {
            for (int t = 0; t < KFCAM_STRIDE; ++t) { S.kc[0][p][t] = 0.0; S.kc[1][p][t] = 0.0; }
            S.koff[p] = -1;
        }
        S.xs[p][0] = u9[0]; S.xs[p][1] = u9[1]; S.xs[p][2] = u9[2]; S.xs[p][3] = u9[6]; S.xs[p][4] = u9[7]; S.xs[p][5] = u9[8];
    }
    const int nunits = IS_LINE ? 2 * g.nlm : g.nlm;
    const int nsteps = (nunits + LMF_UNITS - 1) / LMF_UNITS;
    LmStep cur;
    lm_load<IS_LINE>(d, lv, g, cur_state, 0, wv, lane, cur);
    __syncthreads();
    double chi_acc = 0.0, sc_acc = 0.0;
    for (int step = 0; step < nsteps; ++step) {
        LmStep nxt;
        if (step + 1 < nsteps) lm_load<IS_LINE>(d, lv, g, cur_state, step + 1, wv, lane, nxt);
        LmRows<NR> r;
        const int rowsel = IS_LINE ? ((lane >> 3) & 1) : 0;
        lm_eval<IS_LINE, NR>(d, rb, S.kc[0][cur.ws], cur.has && S.koff[cur.ws] >= 0, cur.L, cur.meas, cur.wt, cur.has, cur.lvl0, rowsel, true, r);
        double h[6], b[3];
        int nact;
        lm_hll<NR>(r, h, b, nact);
        const bool active = cur.uvalid && nact > 0 && !cur.fixed && sok;
        double Li[6];
        lm_chol_inv(h, lambda, active, Li);
        // c = sum_e w Jl^T (Jp x_p)
        double cv3[3] = {0.0, 0.0, 0.0};
        {
            const double* x = S.xs[cur.ws];
#pragma unroll
            for (int a = 0; a < NR; ++a) {
                double sdot = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) sdot += r.j[a][c] * x[c];
                const double ws = r.wr * sdot;
                cv3[0] += ws * r.l[a][0]; cv3[1] += ws * r.l[a][1]; cv3[2] += ws * r.l[a][2];
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) cv3[t] = quad_sum(cv3[t]);
        double rhs[3] = {b[0] - cv3[0], b[1] - cv3[1], b[2] - cv3[2]}, tt[3], xl[3];
        lm_lower_mul(Li, rhs, tt);
        lm_lowerT_mul(Li, tt, xl);
        if (sub == 0 && cur.uvalid) {
            if (active) sc_acc += xl[0] * (lambda * xl[0] + b[0]) + xl[1] * (lambda * xl[1] + b[1]) + xl[2] * (lambda * xl[2] + b[2]);
            double* Lt = d.lm[trial] + (size_t)cur.slot * 6 + (IS_LINE ? 3 * rowsel : 0);
            const double* Lc = cur.L + (IS_LINE ? 3 * rowsel : 0);
            Lt[0] = Lc[0] + xl[0]; Lt[1] = Lc[1] + xl[1]; Lt[2] = Lc[2] + xl[2];
            if (lv.dbg_out) { double* xo = d.xl + (size_t)cur.slot * 6 + (IS_LINE ? 3 * rowsel : 0); xo[0] = xl[0]; xo[1] = xl[1]; xo[2] = xl[2]; }
        }
        // residual of the trial state
        double Ltr[6];
        if (!IS_LINE) { Ltr[0] = cur.L[0] + xl[0]; Ltr[1] = cur.L[1] + xl[1]; Ltr[2] = cur.L[2] + xl[2]; Ltr[3] = Ltr[4] = Ltr[5] = 0.0; }
        else {
            const double o0 = shfl_xor8(xl[0]), o1 = shfl_xor8(xl[1]), o2 = shfl_xor8(xl[2]);      // the line's other end point
            const double* xa = rowsel == 0 ? xl : nullptr;
            Ltr[0] = cur.L[0] + (rowsel == 0 ? xl[0] : o0); Ltr[1] = cur.L[1] + (rowsel == 0 ? xl[1] : o1); Ltr[2] = cur.L[2] + (rowsel == 0 ? xl[2] : o2);
            Ltr[3] = cur.L[3] + (rowsel == 0 ? o0 : xl[0]); Ltr[4] = cur.L[4] + (rowsel == 0 ? o1 : xl[1]); Ltr[5] = cur.L[5] + (rowsel == 0 ? o2 : xl[2]);
            (void)xa;
        }
        LmRows<NR> rt;
        lm_eval<IS_LINE, NR>(d, rb, S.kc[1][cur.ws], false, Ltr, cur.meas, cur.wt, cur.has, cur.lvl0, rowsel, false, rt);
        if (!IS_LINE || rowsel == 0) {
            chi_acc += rt.rho;
            if (rt.act) d.ob_chi2[cur.orig] = rt.chi;
        }
        cur = nxt;
    }
    chi_acc = wave_sum(chi_acc); sc_acc = wave_sum(sc_acc);
    if (lane == 0) { S.red[wv][0] = chi_acc; S.red[wv][1] = sc_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double c = (S.red[0][0] + S.red[1][0]) + (S.red[2][0] + S.red[3][0]), s = (S.red[0][1] + S.red[1][1]) + (S.red[2][1] + S.red[3][1]);
        d.chi_part[gidx] = c; d.scale_part[gidx] = s;
    }
}

}  // namespace plba

// plba_lm_dev.h — fused landmark-major passes of one LM iteration (device bodies; included by plba_kernels.hip).
//
// Round 3.  The record-based path (k_linearize -> k_landmark_hll -> k_schur_pairs -> k_backsub) materialises a compact record per
// observation and gathers it three times; the pair-entry Schur pass alone re-reads every record once per co-observing pair
// (profiles/r02_pmc_traffic.json: 108 MB per iteration at configs[2] against 8.9 MB of algorithmic bytes).  Here nothing is
// materialised: a landmark's observations are evaluated where they are needed, from the observation arrays and the states alone.
//
//   k_lm_schur  (K1-K2 + K5 + K6: computeError / linearizeOplus of the point and line edges, constructQuadraticForm, the landmark
//               side of BlockSolver::solve; IMU/g2otypes.cpp:286-341, 1306-1359)
//               per landmark (8 lanes, one per window keyframe): residuals, Jacobian rows, Huber weight; Hll, bl by a fixed-order DPP
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
// block-diagonal 3 + 3).  The 8 lanes of a unit ARE the 8 keyframe slots of the group's window: lane w takes the unit's observation made
// from window keyframe w, or none (lm_ws8), so a lane's camera block, its rows of the local system and its register accumulators never
// move.  Waves run decoupled (no workgroup barrier inside the loop over landmarks); a group's four partial sums are added in wave order
// at the end.
#pragma once

namespace plba {

typedef double double4v_lm __attribute__((ext_vector_type(4)));

constexpr int LMF_UNITS = 32;            // units per workgroup step (4 waves x 8)
constexpr int LMF_ROWS = 6 * LMF_W;      // 48: rows of a group's local system (slot-major, 6 per keyframe of the window)
constexpr int LMF_SCOLS = 3 * LMF_UNITS; // 96: k-columns of a step's operand of A' A'^T
constexpr int LMF_HCOLS = 2 * LMF_UNITS; // 64: ... of sum Jp^T w Jp (points; lines use half)
struct LmLds {                           // (sized for a wide group's 16 window slots; a standard group uses the first 8)
    double kc[2][LMF_W2][KFCAM_STRIDE];  // camera blocks of the window keyframes: [0] linearisation state, [1] trial state (k_lm_trial)
    double xs[LMF_W2][6];                // pose step (dp, dphi) of the window keyframes (k_lm_trial)
    double red[4][8];
    int koff[LMF_W2];
};
struct LmAcc {                           // k_lm_schur only
    double op[4 * 24 * LMF_ROWS];        // per wave: the step's operand of A' A'^T, [k-column][row] (4 x 9 KB); after the last step: the waves' tiles and vectors
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

// Hll (upper 6) and bl (3) of the unit, summed over its 8 lanes (WIDE: over the 16 lanes of the landmark's two units) in a fixed order;
// every lane ends up with the totals
template <int NR, bool WIDE = false>
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
    for (int t = 0; t < 6; ++t) { h[t] = quad_sum(h[t]); if (WIDE) h[t] += shfl_xor8(h[t]); }
#pragma unroll
    for (int t = 0; t < 3; ++t) { b[t] = quad_sum(b[t]); if (WIDE) b[t] += shfl_xor8(b[t]); }
    nact = quad_sum_i(r.act ? 1 : 0);
    if (WIDE) nact += __builtin_amdgcn_update_dpp(0, nact, 0x128, 0xf, 0xf, false);      // row_ror:8: the landmark's other unit
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
    int slot, k, e, ws, orig;      // ws: the window slot of this lane (= lane & 7: the 8 lanes of a unit ARE the 8 slots, with or without an observation,
                                   // so the operand never needs clearing)
    bool uvalid, has, lvl0, fixed;
    double meas[3], wt, L[6];
};
// the loads of a step come in two dependent levels: the unit's indices (LmIdx), then its landmark and its lane's observation
struct LmIdx { int slot, e; bool uvalid, has, fixed; };
// WIDE: a landmark block takes TWO neighbouring units — window slots 0 - 7 and 8 - 15 — so a point is 2 units and a line 4
// (end point P: slots 0 - 7 | 8 - 15, then end point Q likewise); the lane's window slot is 8 (unit & 1) + (lane & 7)
// The group descriptor is NOT copied into a local: its window arrays are indexed by the thread (g.kf[p]), which put the whole struct into
// scratch memory — 80 (round 3) / 144 bytes per LANE written at the start of every workgroup, ~10 MB per launch at configs[2]: the
// "wasted writes" of profiles/r03_pmc_traffic.json (WRITE_SIZE 8.75 MB against 3.3 MB of outputs).  The scalars the loop needs travel
// in LmHead (registers); the window arrays are read from global memory where they are used, once.
struct LmHead { int lm0, nlm, nw; };
DEV LmHead lm_head(const LmGroup& g) { LmHead h; h.lm0 = g.lm0; h.nlm = g.nlm; h.nw = g.nw; return h; }
template <bool IS_LINE, bool WIDE = false>
DEV void lm_load_idx(const LmView& lv, const LmHead& g, int step, int wv, int lane, LmIdx& x) {
    const int unit = step * LMF_UNITS + wv * 8 + (lane >> 3), sub = WIDE ? 8 * ((lane >> 3) & 1) + (lane & 7) : (lane & 7);
    const int n = WIDE ? (IS_LINE ? (unit >> 2) : (unit >> 1)) : (IS_LINE ? (unit >> 1) : unit);
    x.uvalid = n < g.nlm; x.slot = 0; x.e = 0; x.has = false; x.fixed = true;
    if (!x.uvalid) return;
    const int gi = g.lm0 + n;
    x.slot = lv.lm_grouped ? gi : lv.lm_slot[gi];      // the landmark's place in DevBuf::lm (grouped storage: its place in the group tables)
    const int off = lv.lm_ws8[(size_t)gi * lv.wmax + sub];      // lane = window slot
    x.has = off != 0xFF;
    x.e = lv.lm_ob0[gi] + (x.has ? off : 0);
    x.fixed = lv.lm_fixed_g[gi] != 0;
}
template <bool IS_LINE, bool WIDE = false>
DEV void lm_load_data(const DevBuf& d, const LmView& lv, int state, int lane, const LmIdx& x, LmStep& s) {
    s.uvalid = x.uvalid; s.slot = x.slot; s.k = 0; s.e = x.e; s.ws = WIDE ? 8 * ((lane >> 3) & 1) + (lane & 7) : (lane & 7); s.orig = 0; s.has = x.has; s.lvl0 = false; s.fixed = x.fixed; s.wt = 0.0;
#pragma unroll
    for (int t = 0; t < 3; ++t) s.meas[t] = 0.0;
#pragma unroll
    for (int t = 0; t < 6; ++t) s.L[t] = 0.0;
    if (!s.uvalid) return;      // (a unit past the group's end: its 8 lanes zero the 8 slots)
    const double* Lp = d.lm[state] + (size_t)s.slot * 6;
    s.L[0] = Lp[0]; s.L[1] = Lp[1]; s.L[2] = Lp[2];
    if (IS_LINE) { s.L[3] = Lp[3]; s.L[4] = Lp[4]; s.L[5] = Lp[5]; }
    if (s.has) {
        s.orig = lv.ob_err ? lv.ob_orig[s.e] : 0; s.wt = lv.ob_wt[s.e];      // (the original index: only the parity tests' residual dump needs it)
        s.lvl0 = lv.ob_level_g[s.e] == 0;
        if (IS_LINE) { const double* m = lv.meas_ln + (size_t)(s.e - d.Ep) * 3; s.meas[0] = m[0]; s.meas[1] = m[1]; s.meas[2] = m[2]; }
        else { const double2 m = reinterpret_cast<const double2*>(lv.meas_pt)[s.e]; s.meas[0] = m.x; s.meas[1] = m.y; }
    }
}
template <bool IS_LINE, bool WIDE = false>
DEV void lm_load(const DevBuf& d, const LmView& lv, const LmHead& g, int state, int step, int wv, int lane, LmStep& s) {
    LmIdx x;
    lm_load_idx<IS_LINE, WIDE>(lv, g, step, wv, lane, x);
    lm_load_data<IS_LINE, WIDE>(d, lv, state, lane, x, s);
}

// camera blocks (and kf_off_pvr) of the group's window at `state`
DEV void lm_stage_window(const DevBuf& d, const LmGroup* gp, int nw, int state, LmLds& S) {
    if ((int)threadIdx.x < LMF_W2) {
        const int p = threadIdx.x;
        if (p < nw) { kfcam_make(d.cam, d.kf[state] + (size_t)gp->kf[p] * KF_STRIDE, S.kc[0][p]); S.koff[p] = gp->off[p]; }
        else { for (int t = 0; t < KFCAM_STRIDE; ++t) S.kc[0][p][t] = 0.0; S.koff[p] = -1; }
    }
}

// ---- k_lm_schur: one group ------------------------------------------------------------------------------------------------------
// MODE 0: the Schur pass proper.  MODE 1 (first iteration of a call, before lambda exists): chi2 of the state, the diagonals
// computeLambdaInit needs (max |Hll_jj| per group; diag of sum Jp^T w Jp per window slot, left in the bp part of the group's
// output), the cached per-edge chi2 — no landmark inverse, no rank-k update.
//
// The WAVES of a workgroup run decoupled: a step of a wave takes 8 units; the 8 lanes of a unit are the 8 window slots (lm_load), so
// a lane's keyframe never changes.  What stays inside the lane is accumulated in registers over the whole group: the diagonal block
// sum Jp^T w Jp of its slot (21 entries), the right-hand-side parts bp = -sum Jp^T w e and bs = sum A' y (6 + 6).  What couples the
// slots — A' A'^T — goes through a WAVE-PRIVATE operand panel in LDS (24 k-columns x 48 rows) and the wave's own six 16 x 16
// accumulator tiles (36 v_mfma_f64_16x16x4_f64 per step): no workgroup barrier inside the loop, only the in-order LDS queue of the
// wave itself.  At the end of the group the four waves' tiles and vectors are added in a fixed order through LDS.
constexpr int LMF_WCOLS = 24;                 // k-columns of a wave's operand panel (8 units x 3)
constexpr int LMF_NVEC = 33;                  // per-lane vector accumulators: 21 (H, lower by rows) + 6 (bp) + 6 (bs)
DEV void lm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool IS_LINE, int MODE>
DEV void lm_schur_group(const DevBuf& d, const LmView& lv, const int gidx, const int state, const Robust& rb, LmLds& S, LmAcc& A4) {
    constexpr int NR = IS_LINE ? 1 : 2;
    const LmGroup* gp = lv.grp + gidx;
    const LmHead g = lm_head(*gp);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 7, li = lane & 15, lk = lane >> 4, u8 = lane >> 3;
    const double lambda = MODE == 0 ? d.ctrl->lambda : 0.0;
    lm_stage_window(d, gp, g.nw, state, S);
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
    LmIdx ix1;      // indices of step + 1 (a step ahead of the data they lead to: the two levels of loads never wait for each other)
    ix1.uvalid = false; ix1.slot = 0; ix1.e = 0; ix1.has = false; ix1.fixed = true;
    if (nsteps > 1) lm_load_idx<IS_LINE>(lv, g, 1, wv, lane, ix1);
    __syncthreads();
    const double* kc = S.kc[0][sub];
    const bool kfree = S.koff[sub] >= 0;
    const double4v_lm z4 = (double4v_lm){0.0, 0.0, 0.0, 0.0};
    double4v_lm a00 = z4, a10 = z4, a11 = z4, a20 = z4, a21 = z4, a22 = z4;      // lower tiles of A' A'^T, this wave's units only
    double vec[LMF_NVEC];
#pragma unroll
    for (int t = 0; t < LMF_NVEC; ++t) vec[t] = 0.0;
    double chi_acc = 0.0, maxd = 0.0;
    double* op = A4.op + wv * (LMF_WCOLS * LMF_ROWS);
    for (int step = 0; step < nsteps; ++step) {
        FSTAMP(1);
        LmRows<NR> r;
        const int rowsel = IS_LINE ? (u8 & 1) : 0;
        lm_eval<IS_LINE, NR>(d, rb, kc, cur.has && kfree, cur.L, cur.meas, cur.wt, cur.has, cur.lvl0, rowsel, true, r);
        FSTAMP(2);
        double h[6], b[3];
        int nact;
        lm_hll<NR>(r, h, b, nact);
        FSTAMP(3);
        const bool active = cur.uvalid && nact > 0 && !cur.fixed;
        if (!IS_LINE || rowsel == 0) chi_acc += r.rho;
        if (MODE == 1 && r.act && (!IS_LINE || rowsel == 0)) {
            lv.ob_chi_g[cur.e] = r.chi;
            if (lv.ob_err) { lv.ob_err[2 * (size_t)cur.orig] = r.e[0]; }
        }
        if (MODE == 1 && lv.ob_err && r.act) lv.ob_err[2 * (size_t)cur.orig + (IS_LINE ? rowsel : 1)] = r.e[NR - 1];
        if (lv.dbg_out && cur.uvalid && sub == 0) {      // diagnostics of the parity tests: Hll, bl in the record-based path's layout
            const int dslot = lv.lm_grouped ? lv.lm_slot[cur.slot] : cur.slot;      // (the true landmark slot)
            double* ho = d.hll + (size_t)dslot * 12 + (IS_LINE ? 6 * rowsel : 0);
            double* bo = d.bl + (size_t)dslot * 6 + (IS_LINE ? 3 * rowsel : 0);
#pragma unroll
            for (int t = 0; t < 6; ++t) ho[t] = h[t];
#pragma unroll
            for (int t = 0; t < 3; ++t) bo[t] = b[t];
            if (!IS_LINE || rowsel == 0) d.lm_active[dslot] = active ? 1 : 0;
        }
        if (MODE == 1) {
            if (active && sub == 0) maxd = fmax(maxd, fmax(fmax(fabs(h[0]), fabs(h[3])), fabs(h[5])));
            // diag of Jp^T w Jp of this observation, per slot (left in the bp part of the output)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double q = 0.0;
#pragma unroll
                for (int a = 0; a < NR; ++a) q += r.wj * r.j[a][c] * r.j[a][c];
                vec[21 + c] += q;
            }
            LmStep nxt;
            if (step + 1 < nsteps) lm_load_data<IS_LINE>(d, lv, state, lane, ix1, nxt);
            if (step + 2 < nsteps) lm_load_idx<IS_LINE>(lv, g, step + 2, wv, lane, ix1);
            cur = nxt;
        } else {
            double Li[6], y[3], u[NR][3];
            lm_chol_inv(h, lambda, active, Li);
            lm_lower_mul(Li, b, y);
#pragma unroll
            for (int a = 0; a < NR; ++a) lm_lower_mul(Li, r.l[a], u[a]);
            // A' = wr sum_rows j (x) u  (6 x 3) -> k-columns 3 u8 + m, rows 6 slot + c;  bs += A' y;  bp -= wr sum_rows j e;  H += wr sum_rows j j^T
            // (with js = sqrt(wr) j, us = sqrt(wr) u, es = sqrt(wr) e: one scaled copy of the rows instead of j and wr j side by side)
            const double sw = r.wj > 0.0 ? r.wj * lm_rsqrt(r.wj) : 0.0;      // (a lane without observation or with a gated one: zero rows)
            double es[NR];
#pragma unroll
            for (int a = 0; a < NR; ++a) {
                es[a] = sw * r.e[a];
#pragma unroll
                for (int c = 0; c < 6; ++c) r.j[a][c] *= sw;
#pragma unroll
                for (int m = 0; m < 3; ++m) u[a][m] *= sw;
            }
            int hidx = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double A3[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    double q = 0.0;
#pragma unroll
                    for (int a = 0; a < NR; ++a) q += r.j[a][c] * u[a][m];
                    A3[m] = q;
                    op[(3 * u8 + m) * LMF_ROWS + 6 * sub + c] = q;
                }
                double q = 0.0;
#pragma unroll
                for (int a = 0; a < NR; ++a) q += r.j[a][c] * es[a];
                vec[21 + c] -= q;
                vec[27 + c] += A3[0] * y[0] + A3[1] * y[1] + A3[2] * y[2];
#pragma unroll
                for (int c2 = 0; c2 <= c; ++c2) {
                    double hq = 0.0;
#pragma unroll
                    for (int a = 0; a < NR; ++a) hq += r.j[a][c] * r.j[a][c2];
#ifndef PLBA_X_NOH
                    vec[hidx++] += hq;
#else
                    (void)hq; (void)hidx;
#endif
                }
            }
            FSTAMP(4);
            // the next step's inputs are requested HERE — this step's evaluation is over, so its registers are free — and arrive during the
            // matrix phase; the indices of the step after that ride along
            LmStep nxt;
            if (step + 1 < nsteps) lm_load_data<IS_LINE>(d, lv, state, lane, ix1, nxt);
            if (step + 2 < nsteps) lm_load_idx<IS_LINE>(lv, g, step + 2, wv, lane, ix1);
            lm_wave_sync();
            // ---- rank-24 update of the wave's six lower tiles ----
#pragma unroll 2
            for (int s4 = 0; s4 < LMF_WCOLS / 4; ++s4) {
                const double* col = op + (4 * s4 + lk) * LMF_ROWS + li;
                const double f0 = col[0], f1 = col[16], f2 = col[32];
#ifdef PLBA_X_NOMFMA
                a00[0] += f0 * f1 * f2;
                continue;
#endif
                a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, a00, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f0, a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, a11, 0, 0, 0);
                a20 = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f0, a20, 0, 0, 0);
                a21 = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f1, a21, 0, 0, 0);
                a22 = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f2, a22, 0, 0, 0);
            }
            FSTAMP(5);
            lm_wave_sync();      // (the next step's operand writes stay behind these reads)
            cur = nxt;
        }
        FSTAMP(7);
    }
    FSTAMP1(8);
    // ---- the group's parts go out: waves added in a fixed order ---------------------------------------------------------------------------
    // vectors: over the 8 units of the wave (lanes with the same slot), then over the waves through LDS
#pragma unroll
    for (int t = 0; t < LMF_NVEC; ++t) {
        double v = vec[t];
        v += shfl_xor8(v);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        vec[t] = v;
    }
    chi_acc = wave_sum(chi_acc); maxd = wave_max(maxd);
    __syncthreads();      // every wave is done with its operand panel: the region now takes the tiles and the vectors
    double* cb = A4.op;                                   // [2][6 tiles][256]
    double* vs = A4.op + 2 * 6 * 256;                     // [4 waves][8 slots][LMF_NVEC]
    if (lane < LMF_W) {
#pragma unroll
        for (int t = 0; t < LMF_NVEC; ++t) vs[(wv * LMF_W + lane) * LMF_NVEC + t] = vec[t];
    }
    if (lane == 0) { S.red[wv][0] = chi_acc; S.red[wv][1] = maxd; }
    if (MODE == 0) {
        double* mine = cb + (wv & 1) * (6 * 256);
        if (wv < 2) {
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const int o = (lk + 4 * v4) * 16 + li;
                mine[0 * 256 + o] = a00[v4]; mine[1 * 256 + o] = a10[v4]; mine[2 * 256 + o] = a11[v4];
                mine[3 * 256 + o] = a20[v4]; mine[4 * 256 + o] = a21[v4]; mine[5 * 256 + o] = a22[v4];
            }
        }
        __syncthreads();
        if (wv >= 2) {
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const int o = (lk + 4 * v4) * 16 + li;
                mine[0 * 256 + o] += a00[v4]; mine[1 * 256 + o] += a10[v4]; mine[2 * 256 + o] += a11[v4];
                mine[3 * 256 + o] += a20[v4]; mine[4 * 256 + o] += a21[v4]; mine[5 * 256 + o] += a22[v4];
            }
        }
    }
    __syncthreads();
    FSTAMP1(9);
    double* part = lv.part + (size_t)gidx * lv.part_stride;
    if (MODE == 0) {
        // entry (r, c) of pair block (p <= q): rows a = 6 p + r, b = 6 q + c of the local system, read from the lower tiles
        for (int idx = threadIdx.x; idx < 36 * 36; idx += 256) {
            const int t = idx / 36, rc = idx - t * 36, r = rc / 6, c = rc - r * 6;
            const int q = (t >= 28) ? 7 : (t >= 21) ? 6 : (t >= 15) ? 5 : (t >= 10) ? 4 : (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0;
            const int p = t - q * (q + 1) / 2;
            int a = 6 * p + r, b2 = 6 * q + c;
            if (a < b2) { const int tmp = a; a = b2; b2 = tmp; }      // symmetric: take the lower-triangle copy
            const int ta = a >> 4, tb = b2 >> 4;
            const int o = (ta * (ta + 1) / 2 + tb) * 256 + (a & 15) * 16 + (b2 & 15);
            const double sv = cb[o] + cb[6 * 256 + o];
            double hv = 0.0;
            if (p == q) {
                const int hr = r >= c ? r : c, hc = r >= c ? c : r, k = hr * (hr + 1) / 2 + hc;
                hv = (vs[(0 * LMF_W + p) * LMF_NVEC + k] + vs[(1 * LMF_W + p) * LMF_NVEC + k]) + (vs[(2 * LMF_W + p) * LMF_NVEC + k] + vs[(3 * LMF_W + p) * LMF_NVEC + k]);
            }
            part[idx] = hv - sv;
        }
    }
    if ((int)threadIdx.x < LMF_W * 12 && (MODE == 0 || threadIdx.x % 12 < 6)) {
        const int p = threadIdx.x / 12, k = 21 + threadIdx.x % 12;
        part[lv.npair * 36 + threadIdx.x] = (vs[(0 * LMF_W + p) * LMF_NVEC + k] + vs[(1 * LMF_W + p) * LMF_NVEC + k]) + (vs[(2 * LMF_W + p) * LMF_NVEC + k] + vs[(3 * LMF_W + p) * LMF_NVEC + k]);
    }
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

// ---- k_lm_schur: one WIDE group (round 4) -----------------------------------------------------------------------------------------
// Landmarks seen from 9 .. 16 keyframes: the reference's sliding window holds 12 (include/mapHandler.h:217) and its tracks span most
// of it (kf_obs_list, src/mapHandler.cpp:5296-5413) — with 8-slot groups alone such a window sent the WHOLE problem to the record-based
// passes.  A wide landmark block takes two neighbouring 8-lane units (window slots 0 - 7 | 8 - 15): 4 points or 2 lines per wave step.
// Everything per lane is as in lm_schur_group (evaluation, register accumulators of the lane's slot); what differs:
//   * Hll / bl are summed over the block's 16 lanes;
//   * the local system has 96 rows: 21 lower 16 x 16 tiles (15 while the window holds <= 13 keyframes: rows 78 .. 95 are empty).  A wave
//     cannot hold them all, so the four waves SHARE one operand panel (48 k-columns x 96 rows: wave w writes k-columns 12 w .. 12 w + 11,
//     one column triple per landmark block) and each wave owns a quarter of the tiles over ALL 48 k-columns: two workgroup barriers per
//     step (panel written | panel consumed).  The decoupled-waves form of the standard group is worth 1.6 x; it does not fit
//     here, and wide groups are the small windows' case — launch-bound, not flop-bound;
//   * a tile has ONE owner, so the group's output goes from the accumulator registers straight to the gather buffer.
__device__ const signed char LMW_TILE[2][4][6][2] = {
    {{{0, 0}, {1, 0}, {1, 1}, {2, 0}, {-1, -1}, {-1, -1}}, {{2, 1}, {2, 2}, {3, 0}, {3, 1}, {-1, -1}, {-1, -1}},
     {{3, 2}, {3, 3}, {4, 0}, {4, 1}, {-1, -1}, {-1, -1}}, {{4, 2}, {4, 3}, {4, 4}, {-1, -1}, {-1, -1}, {-1, -1}}},
    {{{0, 0}, {1, 0}, {1, 1}, {2, 0}, {2, 1}, {2, 2}}, {{3, 3}, {4, 3}, {4, 4}, {5, 3}, {5, 4}, {-1, -1}},
     {{3, 0}, {3, 1}, {3, 2}, {4, 0}, {5, 5}, {-1, -1}}, {{4, 1}, {4, 2}, {5, 0}, {5, 1}, {5, 2}, {-1, -1}}}};
constexpr int LMW_ROWS = 6 * LMF_W2;          // 96
constexpr int LMW_KCOLS = 48;                 // 4 waves x 4 landmark blocks x 3
template <bool IS_LINE, int MODE>
DEV void lm_schur_group_wide(const DevBuf& d, const LmView& lv, const int gidx, const int state, const Robust& rb, LmLds& S, LmAcc& A4) {
    constexpr int NR = IS_LINE ? 1 : 2;
    static_assert(sizeof(LmAcc) >= sizeof(double) * LMW_KCOLS * LMW_ROWS, "the wide panel shares the standard groups' dynamic LDS");
    const LmGroup* gp = lv.grp + gidx;
    const LmHead g = lm_head(*gp);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, u8 = lane >> 3;
    const int slot = 8 * (u8 & 1) + (lane & 7);      // the lane's window slot
    const int blk = u8 >> 1;                         // the landmark block inside the wave (0 .. 3)
    const double lambda = MODE == 0 ? d.ctrl->lambda : 0.0;
    lm_stage_window(d, gp, g.nw, state, S);
    const int nunits = (IS_LINE ? 4 : 2) * g.nlm;
    const int nsteps = (nunits + LMF_UNITS - 1) / LMF_UNITS;
    LmStep cur;
    lm_load<IS_LINE, true>(d, lv, g, state, 0, wv, lane, cur);
    LmIdx ix1;
    ix1.uvalid = false; ix1.slot = 0; ix1.e = 0; ix1.has = false; ix1.fixed = true;
    if (nsteps > 1) lm_load_idx<IS_LINE, true>(lv, g, 1, wv, lane, ix1);
    __syncthreads();
    const double* kc = S.kc[0][slot];
    const bool kfree = S.koff[slot] >= 0;
    // this wave's tiles
    const int var = g.nw > 13 ? 1 : 0;
    int ra[6], rb6[6], ntile = 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int ti = LMW_TILE[var][wv][t][0], tj = LMW_TILE[var][wv][t][1];
        ra[t] = 16 * (ti < 0 ? 0 : ti); rb6[t] = 16 * (tj < 0 ? 0 : tj);
        if (ti >= 0) ntile = t + 1;
    }
    const double4v_lm z4 = (double4v_lm){0.0, 0.0, 0.0, 0.0};
    double4v_lm acc[6] = {z4, z4, z4, z4, z4, z4};
    double vec[LMF_NVEC];
#pragma unroll
    for (int t = 0; t < LMF_NVEC; ++t) vec[t] = 0.0;
    double chi_acc = 0.0, maxd = 0.0;
    double* op = A4.op;      // [k-column][row]: 48 x 96, shared by the four waves
    for (int step = 0; step < nsteps; ++step) {
        LmRows<NR> r;
        const int rowsel = IS_LINE ? (blk & 1) : 0;
        lm_eval<IS_LINE, NR>(d, rb, kc, cur.has && kfree, cur.L, cur.meas, cur.wt, cur.has, cur.lvl0, rowsel, true, r);
        double h[6], b[3];
        int nact;
        lm_hll<NR, true>(r, h, b, nact);
        const bool active = cur.uvalid && nact > 0 && !cur.fixed;
        if (!IS_LINE || rowsel == 0) chi_acc += r.rho;
        if (MODE == 1 && r.act && (!IS_LINE || rowsel == 0)) {
            lv.ob_chi_g[cur.e] = r.chi;
            if (lv.ob_err) { lv.ob_err[2 * (size_t)cur.orig] = r.e[0]; }
        }
        if (MODE == 1 && lv.ob_err && r.act) lv.ob_err[2 * (size_t)cur.orig + (IS_LINE ? rowsel : 1)] = r.e[NR - 1];
        if (lv.dbg_out && cur.uvalid && (lane & 15) == 0) {      // diagnostics of the parity tests: Hll, bl in the record-based path's layout
            const int dslot = lv.lm_grouped ? lv.lm_slot[cur.slot] : cur.slot;      // (the true landmark slot)
            double* ho = d.hll + (size_t)dslot * 12 + (IS_LINE ? 6 * rowsel : 0);
            double* bo = d.bl + (size_t)dslot * 6 + (IS_LINE ? 3 * rowsel : 0);
#pragma unroll
            for (int t = 0; t < 6; ++t) ho[t] = h[t];
#pragma unroll
            for (int t = 0; t < 3; ++t) bo[t] = b[t];
            if (!IS_LINE || rowsel == 0) d.lm_active[dslot] = active ? 1 : 0;
        }
        if (MODE == 1) {
            if (active && (lane & 15) == 0) maxd = fmax(maxd, fmax(fmax(fabs(h[0]), fabs(h[3])), fabs(h[5])));
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double q = 0.0;
#pragma unroll
                for (int a = 0; a < NR; ++a) q += r.wj * r.j[a][c] * r.j[a][c];
                vec[21 + c] += q;
            }
            LmStep nxt;
            if (step + 1 < nsteps) lm_load_data<IS_LINE, true>(d, lv, state, lane, ix1, nxt);
            if (step + 2 < nsteps) lm_load_idx<IS_LINE, true>(lv, g, step + 2, wv, lane, ix1);
            cur = nxt;
            continue;
        }
        double Li[6], y[3], u[NR][3];
        lm_chol_inv(h, lambda, active, Li);
        lm_lower_mul(Li, b, y);
#pragma unroll
        for (int a = 0; a < NR; ++a) lm_lower_mul(Li, r.l[a], u[a]);
        const double sw = r.wj > 0.0 ? r.wj * lm_rsqrt(r.wj) : 0.0;
        double es[NR];
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            es[a] = sw * r.e[a];
#pragma unroll
            for (int c = 0; c < 6; ++c) r.j[a][c] *= sw;
#pragma unroll
            for (int m = 0; m < 3; ++m) u[a][m] *= sw;
        }
        int hidx = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double A3[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                double q = 0.0;
#pragma unroll
                for (int a = 0; a < NR; ++a) q += r.j[a][c] * u[a][m];
                A3[m] = q;
                op[(12 * wv + 3 * blk + m) * LMW_ROWS + 6 * slot + c] = q;
            }
            double q = 0.0;
#pragma unroll
            for (int a = 0; a < NR; ++a) q += r.j[a][c] * es[a];
            vec[21 + c] -= q;
            vec[27 + c] += A3[0] * y[0] + A3[1] * y[1] + A3[2] * y[2];
#pragma unroll
            for (int c2 = 0; c2 <= c; ++c2) {
                double hq = 0.0;
#pragma unroll
                for (int a = 0; a < NR; ++a) hq += r.j[a][c] * r.j[a][c2];
                vec[hidx++] += hq;
            }
        }
        LmStep nxt;
        if (step + 1 < nsteps) lm_load_data<IS_LINE, true>(d, lv, state, lane, ix1, nxt);
        if (step + 2 < nsteps) lm_load_idx<IS_LINE, true>(lv, g, step + 2, wv, lane, ix1);
        __syncthreads();      // the panel of this step is complete
#pragma unroll 2
        for (int s4 = 0; s4 < LMW_KCOLS / 4; ++s4) {
            const double* col = op + (4 * s4 + lk) * LMW_ROWS + li;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if (t < ntile) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(col[ra[t]], col[rb6[t]], acc[t], 0, 0, 0);
        }
        __syncthreads();      // ... and consumed: the next step may overwrite it
        cur = nxt;
    }
    // ---- the group's parts go out ---------------------------------------------------------------------------------------------------------
    // vectors: over the 4 landmark blocks of the wave (lanes with the same slot), then over the waves through LDS
#pragma unroll
    for (int t = 0; t < LMF_NVEC; ++t) {
        double v = vec[t];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        vec[t] = v;
    }
    chi_acc = wave_sum(chi_acc); maxd = wave_max(maxd);
    __syncthreads();
    double* vs = A4.op;                                   // [4 waves][16 slots][LMF_NVEC]
    if (lane < LMF_W2) {                                  // (lanes 0 .. 15 hold slots 0 .. 15)
#pragma unroll
        for (int t = 0; t < LMF_NVEC; ++t) vs[(wv * LMF_W2 + lane) * LMF_NVEC + t] = vec[t];
    }
    if (lane == 0) { S.red[wv][0] = chi_acc; S.red[wv][1] = maxd; }
    __syncthreads();
    double* part = lv.part + (size_t)gidx * lv.part_stride;
    if (MODE == 0) {
        // lower entry (a >= b) of the local system -> entry (r, c) of pair block (p <= q): a = 6 q + c, b = 6 p + r (lm_schur_group's layout)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (t >= ntile) continue;
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const int a = ra[t] + lk + 4 * v4, b2 = rb6[t] + li;
                if (a < b2) continue;      // (upper half of a diagonal tile: the mirror image is written below)
                const int q = a / 6, c = a - 6 * q, pq = b2 / 6, rr = b2 - 6 * pq;
                if (q >= g.nw) continue;
                double hv = 0.0;
                if (pq == q) {
                    const int k = c * (c + 1) / 2 + rr;      // (c >= rr)
                    hv = (vs[(0 * LMF_W2 + q) * LMF_NVEC + k] + vs[(1 * LMF_W2 + q) * LMF_NVEC + k]) + (vs[(2 * LMF_W2 + q) * LMF_NVEC + k] + vs[(3 * LMF_W2 + q) * LMF_NVEC + k]);
                }
                const double val = hv - acc[t][v4];
                double* blkp = part + (size_t)(q * (q + 1) / 2 + pq) * 36;
                blkp[rr * 6 + c] = val;
                if (pq == q && rr != c) blkp[c * 6 + rr] = val;
            }
        }
    }
    if ((int)threadIdx.x < LMF_W2 * 12 && (MODE == 0 || threadIdx.x % 12 < 6)) {
        const int pslot = threadIdx.x / 12, k = 21 + threadIdx.x % 12;
        part[lv.npair * 36 + threadIdx.x] = (vs[(0 * LMF_W2 + pslot) * LMF_NVEC + k] + vs[(1 * LMF_W2 + pslot) * LMF_NVEC + k]) + (vs[(2 * LMF_W2 + pslot) * LMF_NVEC + k] + vs[(3 * LMF_W2 + pslot) * LMF_NVEC + k]);
    }
    if (threadIdx.x == 0) {
        d.chi_part[gidx] = (S.red[0][0] + S.red[1][0]) + (S.red[2][0] + S.red[3][0]);
        if (MODE == 1) d.maxd_part[gidx] = fmax(fmax(S.red[0][1], S.red[1][1]), fmax(S.red[2][1], S.red[3][1]));
    }
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
                const int src0 = lv.blk_src[s], src1 = lv.blk_src[s + 7];      // group * npair + pair index
                a0 += lv.part[(size_t)(src0 / lv.npair) * lv.part_stride + (src0 % lv.npair) * 36 + rc];
                a1 += lv.part[(size_t)(src1 / lv.npair) * lv.part_stride + (src1 % lv.npair) * 36 + rc];
            }
            if (s < n1) { const int src0 = lv.blk_src[s]; a0 += lv.part[(size_t)(src0 / lv.npair) * lv.part_stride + (src0 % lv.npair) * 36 + rc]; }
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
            const int src = lv.row_src[s];      // group * wmax + slot
            acc += lv.part[(size_t)(src / lv.wmax) * lv.part_stride + lv.npair * 36 + (src % lv.wmax) * 12 + t];
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
            acc += lv.part[(size_t)(src / lv.wmax) * lv.part_stride + lv.npair * 36 + (src % lv.wmax) * 12 + t];
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
template <bool IS_LINE, bool WIDE = false>
DEV void lm_trial_group(const DevBuf& d, const LmView& lv, const int gidx, const int cur_state, const int trial, const Robust& rb, const ChainView& cv, const double* xd, const bool from_dense, LmLds& S) {
    constexpr int NR = IS_LINE ? 1 : 2;
    const LmGroup* gp = lv.grp + gidx;
    const LmHead g = lm_head(*gp);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = WIDE ? (lane & 15) : (lane & 7);      // sub == 0: the landmark block's first lane
    const double lambda = d.ctrl->lambda;
    const bool sok = d.ctrl->solver_ok != 0;
    if ((int)threadIdx.x < (WIDE ? LMF_W2 : LMF_W)) {
        const int p = threadIdx.x;
        double u9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (p < g.nw) {
            const int k = gp->kf[p], o = gp->off[p];
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
    const int nunits = (IS_LINE ? 2 * g.nlm : g.nlm) * (WIDE ? 2 : 1);
    const int nsteps = (nunits + LMF_UNITS - 1) / LMF_UNITS;
    LmStep cur;
    lm_load<IS_LINE, WIDE>(d, lv, g, cur_state, 0, wv, lane, cur);
    LmIdx ix1;      // indices of step + 1, a step ahead of the data they lead to (see lm_schur_group)
    ix1.uvalid = false; ix1.slot = 0; ix1.e = 0; ix1.has = false; ix1.fixed = true;
    if (nsteps > 1) lm_load_idx<IS_LINE, WIDE>(lv, g, 1, wv, lane, ix1);
    __syncthreads();
    double chi_acc = 0.0, sc_acc = 0.0;
    for (int step = 0; step < nsteps; ++step) {
        LmRows<NR> r;
        const int rowsel = IS_LINE ? ((lane >> (WIDE ? 4 : 3)) & 1) : 0;
        lm_eval<IS_LINE, NR>(d, rb, S.kc[0][cur.ws], cur.has && S.koff[cur.ws] >= 0, cur.L, cur.meas, cur.wt, cur.has, cur.lvl0, rowsel, true, r);
        LmStep nxt;
        if (step + 1 < nsteps) lm_load_data<IS_LINE, WIDE>(d, lv, cur_state, lane, ix1, nxt);      // in flight during the rest of the step
        if (step + 2 < nsteps) lm_load_idx<IS_LINE, WIDE>(lv, g, step + 2, wv, lane, ix1);
        double h[6], b[3];
        int nact;
        lm_hll<NR, WIDE>(r, h, b, nact);
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
        for (int t = 0; t < 3; ++t) { cv3[t] = quad_sum(cv3[t]); if (WIDE) cv3[t] += shfl_xor8(cv3[t]); }
        double rhs[3] = {b[0] - cv3[0], b[1] - cv3[1], b[2] - cv3[2]}, tt[3], xl[3];
        lm_lower_mul(Li, rhs, tt);
        lm_lowerT_mul(Li, tt, xl);
        if (sub == 0 && cur.uvalid) {
            if (active) sc_acc += xl[0] * (lambda * xl[0] + b[0]) + xl[1] * (lambda * xl[1] + b[1]) + xl[2] * (lambda * xl[2] + b[2]);
            double* Lt = d.lm[trial] + (size_t)cur.slot * 6 + (IS_LINE ? 3 * rowsel : 0);
            // (selects, not `cur.L + 3 * rowsel`: an index the compiler cannot resolve puts the whole prefetched step into scratch memory —
            // 104 bytes per lane and step, 6.6 MB per launch at configs[2] before round 4)
            const bool second = IS_LINE && rowsel != 0;
            const double c0 = second ? cur.L[3] : cur.L[0], c1 = second ? cur.L[4] : cur.L[1], c2 = second ? cur.L[5] : cur.L[2];
            Lt[0] = c0 + xl[0]; Lt[1] = c1 + xl[1]; Lt[2] = c2 + xl[2];
            if (lv.dbg_out) { const int dslot = lv.lm_grouped ? lv.lm_slot[cur.slot] : cur.slot; double* xo = d.xl + (size_t)dslot * 6 + (IS_LINE ? 3 * rowsel : 0); xo[0] = xl[0]; xo[1] = xl[1]; xo[2] = xl[2]; }
        }
        // residual of the trial state
        double Ltr[6];
        if (!IS_LINE) { Ltr[0] = cur.L[0] + xl[0]; Ltr[1] = cur.L[1] + xl[1]; Ltr[2] = cur.L[2] + xl[2]; Ltr[3] = Ltr[4] = Ltr[5] = 0.0; }
        else {
            const double o0 = WIDE ? __shfl_xor(xl[0], 16) : shfl_xor8(xl[0]), o1 = WIDE ? __shfl_xor(xl[1], 16) : shfl_xor8(xl[1]), o2 = WIDE ? __shfl_xor(xl[2], 16) : shfl_xor8(xl[2]);      // the line's other end point
            const double* xa = rowsel == 0 ? xl : nullptr;
            Ltr[0] = cur.L[0] + (rowsel == 0 ? xl[0] : o0); Ltr[1] = cur.L[1] + (rowsel == 0 ? xl[1] : o1); Ltr[2] = cur.L[2] + (rowsel == 0 ? xl[2] : o2);
            Ltr[3] = cur.L[3] + (rowsel == 0 ? o0 : xl[0]); Ltr[4] = cur.L[4] + (rowsel == 0 ? o1 : xl[1]); Ltr[5] = cur.L[5] + (rowsel == 0 ? o2 : xl[2]);
            (void)xa;
        }
        LmRows<NR> rt;
        lm_eval<IS_LINE, NR>(d, rb, S.kc[1][cur.ws], false, Ltr, cur.meas, cur.wt, cur.has, cur.lvl0, rowsel, false, rt);
        if (!IS_LINE || rowsel == 0) {
            chi_acc += rt.rho;
            if (rt.act) lv.ob_chi_g[cur.e] = rt.chi;      // (group order: the unit's lanes write neighbouring words; in original order these were 1 M scattered 8-byte stores at configs[4])
        }
        cur = nxt;
    }
    chi_acc = wave_sum(chi_acc); sc_acc = wave_sum(sc_acc);
    if (lane == 0) { S.red[wv][0] = chi_acc; S.red[wv][1] = sc_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double c = (S.red[0][0] + S.red[1][0]) + (S.red[2][0] + S.red[3][0]), s = (S.red[0][1] + S.red[1][1]) + (S.red[2][1] + S.red[3][1]);
        publish(&d.chi_part[gidx], c); publish(&d.scale_part[gidx], s);      // (read by the launch's last workgroup: trial_arrive)
    }
}

}  // namespace plba

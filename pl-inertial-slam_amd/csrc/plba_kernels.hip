// plba_kernels.hip — hand-written gfx950 kernels of the sparse part of one LM iteration.
//
//   K1/K2  k_linearize      one lane per observation: residual + Jacobians + Huber weight, keyframe
//                            camera blocks (Rcb*Rwb^T, Pwb) staged in LDS, coalesced SoA observation loads
//   K5     k_landmark_hll   per-landmark segmented reduction of Jl^T w Jl / Jl^T w e (landmark-major CSR,
//                            fixed order, no atomics)
//          k_landmark_dinv  (Hll + lambda I)^-1 and its product with bl
//   K6     k_schur_pairs    one workgroup per co-observing keyframe pair: sum of Jp_i^T Q Jp_j over the
//                            pair's shared landmarks, wavefront + LDS reduction, exclusive block writes
//          k_backsub        landmark back-substitution fused with the landmark update
//   K3/K4  k_pose_edges / k_prior   IMU PVR + bias edges (one wave per edge), marginalization prior edge
//   K8     k_update_kf, k_reduce, k_lambda_init, k_decide   state update, chi2 reductions, LM control
//
// g2o semantics reproduced: SURVEY.md Appendix A; reference formulas: see plba_math.h.
#include "plba_internal.h"

namespace plba {

#define DEV __device__ __forceinline__

// -------------------------------------------------------------------------------------------------
// reductions (wave = 64 lanes)
// -------------------------------------------------------------------------------------------------
DEV double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
DEV double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
// deterministic block sum for 256-thread blocks; result valid on thread 0
DEV double block_sum_256(double v, double* s4) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s4[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) r = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    return r;
}
DEV double block_max_256(double v, double* s4) {
    v = wave_max(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s4[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) r = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
    __syncthreads();
    return r;
}
DEV int pmap(int r) { return r < 3 ? r : r + 3; }   // (dp, dphi) -> position inside the 9-dim PVR block

// -------------------------------------------------------------------------------------------------
// K1/K2: per-observation residual / Jacobian / robust weight
// -------------------------------------------------------------------------------------------------
template <bool JAC>
__global__ __launch_bounds__(256) void k_linearize(DevBuf d, int state, Robust rb) {
    extern __shared__ double s_dyn[];
    double* s_kc = s_dyn;               // K x 12 staged camera blocks
    __shared__ double s4[4];
    const double* kf = d.kf[state];
    for (int k = threadIdx.x; k < d.K; k += 256) kfcam_make(d.cam, kf + (size_t)k * KF_STRIDE, s_kc + k * KFCAM_STRIDE);
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    double rho = 0.0;
    if (e < d.E) {
        if (d.ob_level[e] == 0) {
            const int k = d.ob_kf[e];
            const int slot = d.ob_slot[e];
            const double w0 = d.ob_w[e];
            const double* L = d.lm[state] + (size_t)slot * 6;
            const double* kc = s_kc + k * KFCAM_STRIDE;
            double e2[2], Jp[12], Jl[6];
            bool dpos;
            int kind;
            if (e < d.Ep) {
                kind = PLBA_EDGE_POINT;
                const double2 uv = reinterpret_cast<const double2*>(d.po_uv)[e];
                point_edge(d.cam, kc, v3(L[0], L[1], L[2]), uv.x, uv.y, e2, Jp, Jl, dpos, JAC);
            } else {
                kind = PLBA_EDGE_LINE;
                const int le = e - d.Ep;
                const double* l = d.lo_l + (size_t)le * 3;
                line_edge(d.cam, kc, v3(L[0], L[1], L[2]), v3(L[3], L[4], L[5]), l[0], l[1], l[2], d.fix_q1 != 0, e2, Jp, Jl, dpos, JAC);
            }
            const double chi = w0 * (e2[0] * e2[0] + e2[1] * e2[1]);
            double r0 = chi, r1 = 1.0;
            if (rb.on[kind]) huber(chi, rb.delta[kind], r0, r1);
            rho = r0;
            d.ob_chi2[e] = chi;
            if (JAC) {
                double4* rec = reinterpret_cast<double4*>(d.erec + (size_t)e * EREC);
                rec[0] = make_double4(Jp[0], Jp[1], Jp[2], Jp[3]);
                rec[1] = make_double4(Jp[4], Jp[5], Jp[6], Jp[7]);
                rec[2] = make_double4(Jp[8], Jp[9], Jp[10], Jp[11]);
                rec[3] = make_double4(Jl[0], Jl[1], Jl[2], Jl[3]);
                rec[4] = make_double4(Jl[4], Jl[5], w0 * r1, e2[0]);
                rec[5] = make_double4(e2[1], chi, 0.0, 0.0);
            }
        } else if (JAC) {
            double4* rec = reinterpret_cast<double4*>(d.erec + (size_t)e * EREC);
            const double4 z = make_double4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) rec[i] = z;
        }
    }
    double bs = block_sum_256(rho, s4);
    if (threadIdx.x == 0) d.chi_part[blockIdx.x] = bs;
}

// -------------------------------------------------------------------------------------------------
// K5: landmark blocks.  Thread per landmark slot, fixed edge order (deterministic).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_landmark_hll(DevBuf d) {
    __shared__ double s4[4];
    const int slot = blockIdx.x * 256 + threadIdx.x;
    double md = 0.0;
    if (slot < d.L) {
        const int s = d.lm_start[slot], en = d.lm_start[slot + 1];
        double h[12], b[6];
#pragma unroll
        for (int i = 0; i < 12; ++i) h[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] = 0.0;
        int nact = 0;
        const bool is_pt = slot < d.Np;
        for (int ed = s; ed < en; ++ed) {
            const double* r = d.erec + (size_t)ed * EREC;
            const double w = r[18];
            if (d.ob_level[ed] == 0) ++nact;
            if (w == 0.0) continue;
            const double a0 = r[12], a1 = r[13], a2 = r[14], b0 = r[15], b1 = r[16], b2 = r[17];
            const double e0 = r[19], e1 = r[20];
            if (is_pt) {   // Jl = [a; b] both on the same 3 coordinates
                h[0] += w * (a0 * a0 + b0 * b0); h[1] += w * (a0 * a1 + b0 * b1); h[2] += w * (a0 * a2 + b0 * b2);
                h[3] += w * (a1 * a1 + b1 * b1); h[4] += w * (a1 * a2 + b1 * b2); h[5] += w * (a2 * a2 + b2 * b2);
                b[0] -= w * (a0 * e0 + b0 * e1); b[1] -= w * (a1 * e0 + b1 * e1); b[2] -= w * (a2 * e0 + b2 * e1);
            } else {       // row0 on sP, row1 on eP: block-diagonal 6x6
                h[0] += w * a0 * a0; h[1] += w * a0 * a1; h[2] += w * a0 * a2; h[3] += w * a1 * a1; h[4] += w * a1 * a2; h[5] += w * a2 * a2;
                h[6] += w * b0 * b0; h[7] += w * b0 * b1; h[8] += w * b0 * b2; h[9] += w * b1 * b1; h[10] += w * b1 * b2; h[11] += w * b2 * b2;
                b[0] -= w * a0 * e0; b[1] -= w * a1 * e0; b[2] -= w * a2 * e0;
                b[3] -= w * b0 * e1; b[4] -= w * b1 * e1; b[5] -= w * b2 * e1;
            }
        }
        const bool active = (nact > 0) && !d.lm_fixed[slot];
        d.lm_active[slot] = active ? 1 : 0;
        double* ho = d.hll + (size_t)slot * 12;
        double* bo = d.bl + (size_t)slot * 6;
#pragma unroll
        for (int i = 0; i < 12; ++i) ho[i] = h[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) bo[i] = b[i];
        if (active) {
            md = fmax(fmax(fabs(h[0]), fabs(h[3])), fabs(h[5]));
            if (!is_pt) md = fmax(md, fmax(fmax(fabs(h[6]), fabs(h[9])), fabs(h[11])));
        }
    }
    double bm = block_max_256(md, s4);
    if (threadIdx.x == 0) d.maxd_part[blockIdx.x] = bm;
}

__global__ __launch_bounds__(256) void k_landmark_dinv(DevBuf d) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= d.L) return;
    double* D = d.dinv + (size_t)slot * 12;
    double* t = d.tv + (size_t)slot * 6;
    if (!d.lm_active[slot]) {
#pragma unroll
        for (int i = 0; i < 12; ++i) D[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i] = 0.0;
        return;
    }
    const double lambda = d.ctrl->lambda;
    const double* h = d.hll + (size_t)slot * 12;
    const double* b = d.bl + (size_t)slot * 6;
    double dd[6];
    sym3_inv(h, lambda, dd);
    V3 t0 = sym3_mul(dd, v3(b[0], b[1], b[2]));
#pragma unroll
    for (int i = 0; i < 6; ++i) D[i] = dd[i];
    t[0] = t0.x; t[1] = t0.y; t[2] = t0.z;
    if (slot >= d.Np) {
        sym3_inv(h + 6, lambda, dd);
        V3 t1 = sym3_mul(dd, v3(b[3], b[4], b[5]));
#pragma unroll
        for (int i = 0; i < 6; ++i) D[6 + i] = dd[i];
        t[3] = t1.x; t[4] = t1.y; t[5] = t1.z;
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) D[6 + i] = 0.0;
        t[3] = t[4] = t[5] = 0.0;
    }
}

// per-keyframe diagonal of sum Jp^T w Jp (only needed for lambda_init at iteration 0)
__global__ __launch_bounds__(256) void k_kfdiag(DevBuf d) {
    __shared__ double s4[4];
    const int p = blockIdx.x;
    const int i = d.pair_i[p];
    if (i != d.pair_j[p]) return;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int n = d.pair_start[p] + threadIdx.x; n < d.pair_start[p + 1]; n += 256) {
        const double* r = d.erec + (size_t)d.ent_ei[n] * EREC;
        const double w = r[18];
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] += w * (r[c] * r[c] + r[6 + c] * r[6 + c]);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double v = block_sum_256(acc[c], s4);
        if (threadIdx.x == 0) d.kfdiag[i * 6 + c] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// K6: Schur complement over keyframe pairs.
//   Hschur(i,j) = Hpp(i,j) - sum_l Hpl(i,l) D_l Hpl(j,l)^T  with Hpl(i,l) = Jp_i^T w_i Jl_i
//               = [pose-side edges + lambda] + sum_entries Jp_i^T Q Jp_j,   Q = [ei==ej] w I2 - w_i w_j Jl_i D Jl_j^T
//   bschur_i   = b_i - sum Hpl D bl = [pose-side] + sum_{e in kf i} -w Jp^T (e + Jl t_l),  t_l = D_l bl_l
// One 256-thread workgroup per pair; each lane accumulates a 6x6 (+2x6 for diagonal pairs) in registers,
// then wavefront shuffles + LDS; the owning workgroup read-modify-writes its exclusive blocks of `sys`.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_schur_pairs(DevBuf d) {
    __shared__ double s_red[4][48];
    const int p = blockIdx.x;
    const int i = d.pair_i[p], j = d.pair_j[p];
    const bool diag = (i == j);
    double acc[36];
    double gb[6], gp[6];
#pragma unroll
    for (int t = 0; t < 36; ++t) acc[t] = 0.0;
#pragma unroll
    for (int t = 0; t < 6; ++t) { gb[t] = 0.0; gp[t] = 0.0; }
    const int s = d.pair_start[p], en = d.pair_start[p + 1];
    for (int n = s + threadIdx.x; n < en; n += 256) {
        const int ei = d.ent_ei[n], ej = d.ent_ej[n];
        const double* ri = d.erec + (size_t)ei * EREC;
        const double wi = ri[18];
        if (wi == 0.0) continue;
        const double* rj = d.erec + (size_t)ej * EREC;
        const double wj = rj[18];
        if (wj == 0.0) continue;
        const int slot = d.ob_slot[ei];
        const double* D = d.dinv + (size_t)slot * 12;
        double Ji[12], Jj[12];
#pragma unroll
        for (int t = 0; t < 12; ++t) { Ji[t] = ri[t]; Jj[t] = rj[t]; }
        const V3 ai = v3(ri[12], ri[13], ri[14]), bi = v3(ri[15], ri[16], ri[17]);
        const V3 aj = v3(rj[12], rj[13], rj[14]), bj = v3(rj[15], rj[16], rj[17]);
        double q00, q01, q10, q11;
        const double ww = wi * wj;
        if (slot < d.Np) {
            const V3 Daj = sym3_mul(D, aj), Dbj = sym3_mul(D, bj);
            q00 = -ww * dot(ai, Daj); q01 = -ww * dot(ai, Dbj);
            q10 = -ww * dot(bi, Daj); q11 = -ww * dot(bi, Dbj);
        } else {
            q00 = -ww * dot(ai, sym3_mul(D, aj)); q11 = -ww * dot(bi, sym3_mul(D + 6, bj));
            q01 = 0.0; q10 = 0.0;
        }
        if (ei == ej) { q00 += wi; q11 += wi; }
        double T0[6], T1[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            T0[c] = q00 * Jj[c] + q01 * Jj[6 + c];
            T1[c] = q10 * Jj[c] + q11 * Jj[6 + c];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[r * 6 + c] += Ji[r] * T0[c] + Ji[6 + r] * T1[c];
        if (diag) {
            const double* t = d.tv + (size_t)slot * 6;
            const double e0 = ri[19], e1 = ri[20];
            double f0, f1;
            if (slot < d.Np) {
                const V3 tt = v3(t[0], t[1], t[2]);
                f0 = e0 + dot(ai, tt); f1 = e1 + dot(bi, tt);
            } else {
                f0 = e0 + dot(ai, v3(t[0], t[1], t[2])); f1 = e1 + dot(bi, v3(t[3], t[4], t[5]));
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                gp[r] -= wi * (Ji[r] * e0 + Ji[6 + r] * e1);
                gb[r] -= wi * (Ji[r] * f0 + Ji[6 + r] * f1);
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 36; ++t) {
        double v = wave_sum(acc[t]);
        if (lane == 0) s_red[wv][t] = v;
    }
    if (diag) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            double v = wave_sum(gb[t]);
            double u = wave_sum(gp[t]);
            if (lane == 0) { s_red[wv][36 + t] = v; s_red[wv][42 + t] = u; }
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    const int oi = d.kf_off_pvr[i], oj = d.kf_off_pvr[j];
    const int ld = d.ld;
    if (t < 36) {
        const double v = (s_red[0][t] + s_red[1][t]) + (s_red[2][t] + s_red[3][t]);
        const int r = pmap(t / 6), c = pmap(t % 6);
        d.sys[(size_t)(oi + r) * ld + oj + c] += v;
        if (!diag) d.sys[(size_t)(oj + c) * ld + oi + r] += v;
    } else if (diag && t < 48) {
        const double v = (s_red[0][t] + s_red[1][t]) + (s_red[2][t] + s_red[3][t]);
        const int r = pmap((t - 36) % 6);
        const int row = (t < 42) ? d.Ppad : d.Ppad + 1;     // bschur row / bp row of the augmented system
        d.sys[(size_t)row * ld + oi + r] += v;
    }
}

// sys = Himu (+ lambda I on the real diagonal, 1 on the padded diagonal) ; row Ppad = row Ppad+1 = pose-side gradient
__global__ __launch_bounds__(256) void k_assemble(DevBuf d, int add_lambda) {
    const size_t n = (size_t)(d.Ppad + TILE) * d.ld;
    const double lambda = d.ctrl->lambda;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / d.ld), c = (int)(idx % d.ld);
        double v = 0.0;
        if (r < d.Ppad) {
            v = d.Himu[idx];
            if (r == c && add_lambda) v += (r < d.P) ? lambda : 1.0;
        } else if (r <= d.Ppad + 1) {
            v = d.bimu[c];
        }
        d.sys[idx] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// landmark back-substitution + landmark update (+ landmark part of computeScale)
//   xl = D (bl - sum_e w Jl^T Jp x_kf)      trial_lm = cur_lm + xl
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_backsub(DevBuf d, int cur, int trial) {
    __shared__ double s4[4];
    const int slot = blockIdx.x * 256 + threadIdx.x;
    double sc = 0.0;
    if (slot < d.L) {
        const double* Lc = d.lm[cur] + (size_t)slot * 6;
        double* Lt = d.lm[trial] + (size_t)slot * 6;
        double xl[6] = {0, 0, 0, 0, 0, 0};
        const bool is_pt = slot < d.Np;
        if (d.lm_active[slot] && d.ctrl->solver_ok) {
            const double* b = d.bl + (size_t)slot * 6;
            double c[6] = {b[0], b[1], b[2], b[3], b[4], b[5]};
            for (int ed = d.lm_start[slot]; ed < d.lm_start[slot + 1]; ++ed) {
                const double* r = d.erec + (size_t)ed * EREC;
                const double w = r[18];
                if (w == 0.0) continue;
                const int o = d.kf_off_pvr[d.ob_kf[ed]];
                if (o < 0) continue;
                const double* xp = d.x + o;
                const double x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[6], x4 = xp[7], x5 = xp[8];
                const double s0 = r[0] * x0 + r[1] * x1 + r[2] * x2 + r[3] * x3 + r[4] * x4 + r[5] * x5;
                const double s1 = r[6] * x0 + r[7] * x1 + r[8] * x2 + r[9] * x3 + r[10] * x4 + r[11] * x5;
                if (is_pt) {
                    c[0] -= w * (r[12] * s0 + r[15] * s1); c[1] -= w * (r[13] * s0 + r[16] * s1); c[2] -= w * (r[14] * s0 + r[17] * s1);
                } else {
                    c[0] -= w * r[12] * s0; c[1] -= w * r[13] * s0; c[2] -= w * r[14] * s0;
                    c[3] -= w * r[15] * s1; c[4] -= w * r[16] * s1; c[5] -= w * r[17] * s1;
                }
            }
            const double* D = d.dinv + (size_t)slot * 12;
            V3 a = sym3_mul(D, v3(c[0], c[1], c[2]));
            xl[0] = a.x; xl[1] = a.y; xl[2] = a.z;
            if (!is_pt) { V3 e = sym3_mul(D + 6, v3(c[3], c[4], c[5])); xl[3] = e.x; xl[4] = e.y; xl[5] = e.z; }
            const double lambda = d.ctrl->lambda;
#pragma unroll
            for (int t = 0; t < 6; ++t) sc += xl[t] * (lambda * xl[t] + b[t]);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) { Lt[t] = Lc[t] + xl[t]; d.xl[(size_t)slot * 6 + t] = xl[t]; }
    }
    double bs = block_sum_256(sc, s4);
    if (threadIdx.x == 0) d.scale_part[blockIdx.x] = bs;
}

// SparseOptimizer::update for the keyframe vertices (oplusImpl of VertexNavStatePVR / VertexNavStateBias)
__global__ void k_update_kf(DevBuf d, int cur, int trial) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d.K) return;
    const double* s = d.kf[cur] + (size_t)k * KF_STRIDE;
    double* o = d.kf[trial] + (size_t)k * KF_STRIDE;
    double tmp[KF_STRIDE];
#pragma unroll
    for (int i = 0; i < KF_STRIDE; ++i) tmp[i] = s[i];
    const bool ok = d.ctrl->solver_ok != 0;
    const int op = d.kf_off_pvr[k], ob = d.kf_off_bias[k];
    if (ok && op >= 0) {
        double u[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) u[i] = d.x[op + i];
        kf_oplus_pvr(s, u, tmp);
    }
    if (ok && ob >= 0) {   // IMU/NavState.cpp:100-121
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[16 + i] = s[16 + i] + d.x[ob + i];
    }
#pragma unroll
    for (int i = 0; i < KF_STRIDE; ++i) o[i] = tmp[i];
}

// -------------------------------------------------------------------------------------------------
// K3: IMU edges.  One wavefront per (PVR edge, bias edge) pair of consecutive keyframes.
// Lane 0 evaluates the residuals and the three Jacobian blocks into LDS; all 64 lanes then form
// Omega' J and J^T Omega' J (24 x 24 over [PVR_i | PVR_j | Bias_i]) and add them into the dense
// pose-side system with fp64 atomics (a handful of edges share a destination block).
// -------------------------------------------------------------------------------------------------
template <bool JAC>
__global__ __launch_bounds__(64) void k_pose_edges(DevBuf d, int state, Robust rb) {
    __shared__ double sJ[9 * 24];     // [J0 | J1 | J2] row-major 9 x 24
    __shared__ double sOJ[9 * 24];
    __shared__ double sE[16];
    __shared__ double sW[2];
    const int m = blockIdx.x;
    const int lane = threadIdx.x;
    const int ki = d.imu_i[m], kj = d.imu_j[m];
    const double* si = d.kf[state] + (size_t)ki * KF_STRIDE;
    const double* sj = d.kf[state] + (size_t)kj * KF_STRIDE;
    const double* pre = d.imu_pre + (size_t)m * PRE_STRIDE;
    const double* Om = d.imu_info_pvr + (size_t)m * 81;
    const double* Ob = d.imu_info_bias + (size_t)m * 36;
    if (lane == 0) {
        double e9[9], e6[6];
        pvr_error(si, sj, pre, d.gw, e9);
        bias_error(si, sj, e6);
        double chi = 0.0;
        for (int r = 0; r < 9; ++r) { double t = 0.0; for (int c = 0; c < 9; ++c) t += Om[r * 9 + c] * e9[c]; chi += e9[r] * t; }
        double chib = 0.0;
        for (int r = 0; r < 6; ++r) { double t = 0.0; for (int c = 0; c < 6; ++c) t += Ob[r * 6 + c] * e6[c]; chib += e6[r] * t; }
        double r0 = chi, r1 = 1.0, b0 = chib, b1 = 1.0;
        if (rb.on[PLBA_EDGE_IMU_PVR]) huber(chi, rb.delta[PLBA_EDGE_IMU_PVR], r0, r1);
        if (rb.on[PLBA_EDGE_IMU_BIAS]) huber(chib, rb.delta[PLBA_EDGE_IMU_BIAS], b0, b1);
        double* eo = d.imu_err + (size_t)m * 16;
        for (int r = 0; r < 9; ++r) { eo[r] = e9[r]; sE[r] = e9[r]; }
        for (int r = 0; r < 6; ++r) { eo[9 + r] = e6[r]; sE[9 + r] = e6[r]; }
        double* co = d.imu_chi + (size_t)m * 4;
        co[0] = chi; co[1] = chib; co[2] = r0; co[3] = b0;
        sW[0] = r1; sW[1] = b1;
        if (JAC) {
            double J0[81], J1[81], J2[54];
            for (int t = 0; t < 81; ++t) { J0[t] = 0.0; J1[t] = 0.0; }
            for (int t = 0; t < 54; ++t) J2[t] = 0.0;
            pvr_jacobians(si, sj, pre, d.gw, e9, J0, J1, J2);
            for (int r = 0; r < 9; ++r) {
                for (int c = 0; c < 9; ++c) { sJ[r * 24 + c] = J0[r * 9 + c]; sJ[r * 24 + 9 + c] = J1[r * 9 + c]; }
                for (int c = 0; c < 6; ++c) sJ[r * 24 + 18 + c] = J2[r * 6 + c];
            }
        }
    }
    if (!JAC) return;
    __syncthreads();
    const double w = sW[0], wb = sW[1];
    // OJ = w * Omega * J   (9 x 24)
    for (int t = lane; t < 9 * 24; t += 64) {
        const int r = t / 24, c = t % 24;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += Om[r * 9 + k] * sJ[k * 24 + c];
        sOJ[t] = w * s;
    }
    __syncthreads();
    int off[24];
    {
        const int o0 = d.kf_off_pvr[ki], o1 = d.kf_off_pvr[kj], o2 = d.kf_off_bias[ki];
#pragma unroll
        for (int c = 0; c < 9; ++c) { off[c] = o0 < 0 ? -1 : o0 + c; off[9 + c] = o1 < 0 ? -1 : o1 + c; }
#pragma unroll
        for (int c = 0; c < 6; ++c) off[18 + c] = o2 < 0 ? -1 : o2 + c;
    }
    const int ld = d.ld;
    // H += J^T OJ (24 x 24), g += -J^T (w Omega e) = -OJ^T e
    for (int t = lane; t < 24 * 24; t += 64) {
        const int a = t / 24, b = t % 24;
        int oa = -1, ob = -1;
#pragma unroll
        for (int q = 0; q < 24; ++q) { if (q == a) oa = off[q]; if (q == b) ob = off[q]; }
        if (oa < 0 || ob < 0) continue;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += sJ[k * 24 + a] * sOJ[k * 24 + b];
        if (s != 0.0) atomicAdd(&d.Himu[(size_t)oa * ld + ob], s);
    }
    if (lane < 24) {
        int oa = -1;
#pragma unroll
        for (int q = 0; q < 24; ++q) if (q == lane) oa = off[q];
        if (oa >= 0) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) s += sOJ[k * 24 + lane] * sE[k];
            atomicAdd(&d.bimu[oa], -s);
        }
    }
    // bias edge: J = [-I, +I] on (Bias_i, Bias_j); H_ii += W, H_jj += W, H_ij = H_ji -= W; g_i += W e, g_j -= W e
    {
        const int oi = d.kf_off_bias[ki], oj = d.kf_off_bias[kj];
        if (lane < 36) {
            const int r = lane / 6, c = lane % 6;
            const double v = wb * Ob[r * 6 + c];
            if (v != 0.0) {
                if (oi >= 0) atomicAdd(&d.Himu[(size_t)(oi + r) * ld + oi + c], v);
                if (oj >= 0) atomicAdd(&d.Himu[(size_t)(oj + r) * ld + oj + c], v);
                if (oi >= 0 && oj >= 0) {
                    atomicAdd(&d.Himu[(size_t)(oi + r) * ld + oj + c], -v);
                    atomicAdd(&d.Himu[(size_t)(oj + r) * ld + oi + c], -v);
                }
            }
        } else if (lane < 42) {
            const int r = lane - 36;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) s += Ob[r * 6 + c] * sE[9 + c];
            s *= wb;
            if (oi >= 0) atomicAdd(&d.bimu[oi + r], s);
            if (oj >= 0) atomicAdd(&d.bimu[oj + r], -s);
        }
    }
}

// K4: marginalization prior edge (one workgroup): dx, e = r0 + J0 dx, chi2 = |e|^2, g += -J0^T e.
// Its Hessian J0^T J0 is constant and pre-scattered into Hconst at upload.
template <bool JAC>
__global__ __launch_bounds__(256) void k_prior(DevBuf d, int state) {
    __shared__ double s4[4];
    const int n = d.pr_n;
    for (int v = threadIdx.x; v < d.pr_nv; v += 256) {
        const double* s = d.kf[state] + (size_t)d.pr_kf[v] * KF_STRIDE;
        const double* x0 = d.pr_x0 + d.pr_x0off[v];
        double* dx = d.pr_dx + d.pr_idx[v];
        if (d.pr_isbias[v]) prior_dx_bias(s, x0, dx);
        else prior_dx_pvr(s, x0, dx);
    }
    __syncthreads();
    double chi = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) {
        double sacc = d.pr_r0[r];
        for (int c = 0; c < n; ++c) sacc += d.pr_J0[(size_t)c * n + r] * d.pr_dx[c];
        d.pr_err[r] = sacc;
        chi += sacc * sacc;
    }
    double tot = block_sum_256(chi, s4);
    if (threadIdx.x == 0) d.pr_chi[0] = tot;
    if (!JAC) return;
    __syncthreads();
    for (int v = 0; v < d.pr_nv; ++v) {
        const int o = d.pr_off[v];
        if (o < 0) continue;
        const int sz = d.pr_size[v], ix = d.pr_idx[v];
        for (int c = threadIdx.x; c < sz; c += 256) {
            const double* col = d.pr_J0 + (size_t)(ix + c) * n;
            double sacc = 0.0;
            for (int r = 0; r < n; ++r) sacc += col[r] * d.pr_err[r];
            atomicAdd(&d.bimu[o + c], -sacc);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// K8: reductions and LM control (single workgroup, fixed summation order)
// red[0] = activeRobustChi2 (local), red[1] = landmark part of computeScale (local), red[2] = max |Hll_jj| (local)
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce(DevBuf d, int nblk_edges, int nblk_lm, int pose_edges, double* red) {
    __shared__ double s4[4];
    double c = 0.0, sc = 0.0, md = 0.0;
    for (int i = threadIdx.x; i < nblk_edges; i += 256) c += d.chi_part[i];
    if (pose_edges) {
        for (int i = threadIdx.x; i < d.M; i += 256) c += d.imu_chi[(size_t)i * 4 + 2] + d.imu_chi[(size_t)i * 4 + 3];
        if (threadIdx.x == 0 && d.pr_nv > 0) c += d.pr_chi[0];
    }
    for (int i = threadIdx.x; i < nblk_lm; i += 256) { sc += d.scale_part[i]; md = fmax(md, d.maxd_part[i]); }
    double C = block_sum_256(c, s4);
    double S = block_sum_256(sc, s4);
    double Mx = block_max_256(md, s4);
    if (threadIdx.x == 0) { red[0] = C; red[1] = S; red[2] = Mx; }
}

// diagonal of the pose-side Hessian held by this rank: pose-side edges (Himu) + sum_e Jp^T w Jp (kfdiag);
// in a sharded run the vector is all-reduced (sum) before the max, so every rank derives the same lambda
__global__ __launch_bounds__(256) void k_posediag(DevBuf d) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < d.ld; r += gridDim.x * 256) d.posediag[r] = (r < d.P) ? d.Himu[(size_t)r * d.ld + r] : 0.0;
}
__global__ __launch_bounds__(256) void k_posediag_kf(DevBuf d) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= d.K * 6) return;
    const int o = d.kf_off_pvr[t / 6];
    if (o >= 0) d.posediag[o + pmap(t % 6)] += d.kfdiag[t];
}
// start of an outer iteration: currentChi, and on the first one computeLambdaInit (tau * max |H_jj|)
__global__ __launch_bounds__(256) void k_lambda_init(DevBuf d, LmParams lp, const double* red, int first_iter, int iteration) {
    __shared__ double s4[4];
    double md = 0.0;
    if (first_iter) for (int r = threadIdx.x; r < d.P; r += 256) md = fmax(md, fabs(d.posediag[r]));
    double mx = block_max_256(md, s4);
    if (threadIdx.x == 0) {
        Ctrl* c = d.ctrl;
        c->current_chi = red[0];
        c->iteration = iteration;
        c->trial = 0;
        if (first_iter) {
            mx = fmax(mx, red[2]);
            c->maxdiag = mx;
            c->lambda = lp.user_lambda > 0 ? lp.user_lambda : lp.tau * mx;
            c->ni = 2.0;
        }
    }
}

// end of a trial: rho test and lambda schedule of OptimizationAlgorithmLevenberg::solve
__global__ __launch_bounds__(256) void k_decide(DevBuf d, LmParams lp, const double* red) {
    __shared__ double s4[4];
    Ctrl* c = d.ctrl;
    const double lambda = c->lambda;
    double sp = 0.0;
    if (c->solver_ok) for (int j = threadIdx.x; j < d.P; j += 256) { const double xj = d.x[j]; sp += xj * (lambda * xj + d.bpg[j]); }
    double SP = block_sum_256(sp, s4);
    if (threadIdx.x != 0) return;
    double tempChi = red[0];
    if (!c->solver_ok) tempChi = 1.7976931348623157e308;
    double scale = SP + red[1];
    scale += 1e-3;
    const double rho = (c->current_chi - tempChi) / scale;
    const int n = *d.trace_n;
    if (n < d.trace_cap) {
        plba_trace_row* tr = d.trace + n;
        tr->iteration = c->iteration; tr->trial = c->trial; tr->solver_ok = c->solver_ok;
        tr->lambda = lambda; tr->chi2_current = c->current_chi; tr->chi2_trial = tempChi; tr->scale = scale; tr->rho = rho;
        tr->accepted = (rho > 0 && isfinite(tempChi)) ? 1 : 0;
    }
    *d.trace_n = n + 1;
    c->temp_chi = tempChi; c->scale = scale; c->rho = rho;
    if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, lp.upper);
        const double sf = fmax(lp.lower, alpha);
        c->lambda = lambda * sf;
        c->ni = 2.0;
        c->current_chi = tempChi;
        c->accepted = 1;
    } else {
        c->lambda = lambda * c->ni;
        c->ni *= 2.0;
        c->accepted = 0;
    }
    c->trial += 1;
    if (!c->solver_ok) c->n_fail += 1;
    c->solver_ok = 1;
}

// chi2() > thresh || !isDepthPositive()  =>  setLevel(1)   (mapHandler.cpp:6047-6066)
__global__ __launch_bounds__(256) void k_gate(DevBuf d, int state, double thresh, uint8_t* depth_out, int do_gate) {
    extern __shared__ double s_dyn[];
    double* s_kc = s_dyn;
    const double* kf = d.kf[state];
    for (int k = threadIdx.x; k < d.K; k += 256) kfcam_make(d.cam, kf + (size_t)k * KF_STRIDE, s_kc + k * KFCAM_STRIDE);
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= d.E) return;
    const double* L = d.lm[state] + (size_t)d.ob_slot[e] * 6;
    const double* kc = s_kc + d.ob_kf[e] * KFCAM_STRIDE;
    bool dpos = cam_Pc(d.cam, kc, v3(L[0], L[1], L[2])).z > 0.0;
    if (e >= d.Ep) dpos = dpos && (cam_Pc(d.cam, kc, v3(L[3], L[4], L[5])).z > 0.0);
    if (depth_out) depth_out[e] = dpos ? 1 : 0;
    if (do_gate && (d.ob_chi2[e] > thresh || !dpos)) {
        if (d.ob_level[e] == 0) atomicAdd(e < d.Ep ? &d.ctrl->n_gate_pt : &d.ctrl->n_gate_ln, 1);
        d.ob_level[e] = 1;
    }
}

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
int edge_blocks(const DevBuf& d) { return (d.E + 255) / 256; }
static int lm_blocks(const DevBuf& d) { return (d.L + 255) / 256; }

void launch_linearize(const DevBuf& d, int state, bool jac, const Robust& rb, hipStream_t s) {
    if (d.E == 0) return;
    const size_t sh = (size_t)d.K * KFCAM_STRIDE * sizeof(double);
    if (jac) hipLaunchKernelGGL(k_linearize<true>, dim3(edge_blocks(d)), dim3(256), sh, s, d, state, rb);
    else hipLaunchKernelGGL(k_linearize<false>, dim3(edge_blocks(d)), dim3(256), sh, s, d, state, rb);
}
void launch_pose_edges(const DevBuf& d, int state, bool jac, const Robust& rb, bool owns, hipStream_t s) {
    if (!owns) return;
    if (d.M > 0) {
        if (jac) hipLaunchKernelGGL(k_pose_edges<true>, dim3(d.M), dim3(64), 0, s, d, state, rb);
        else hipLaunchKernelGGL(k_pose_edges<false>, dim3(d.M), dim3(64), 0, s, d, state, rb);
    }
    if (d.pr_nv > 0) {
        if (jac) hipLaunchKernelGGL(k_prior<true>, dim3(1), dim3(256), 0, s, d, state);
        else hipLaunchKernelGGL(k_prior<false>, dim3(1), dim3(256), 0, s, d, state);
    }
}
void launch_landmark_hll(const DevBuf& d, hipStream_t s) {
    if (d.L) hipLaunchKernelGGL(k_landmark_hll, dim3(lm_blocks(d)), dim3(256), 0, s, d);
}
void launch_kfdiag(const DevBuf& d, hipStream_t s) {
    if (d.npairs) hipLaunchKernelGGL(k_kfdiag, dim3(d.npairs), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_posediag, dim3((d.ld + 255) / 256), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_posediag_kf, dim3((d.K * 6 + 255) / 256), dim3(256), 0, s, d);
}
void launch_landmark_dinv(const DevBuf& d, hipStream_t s) {
    if (d.L) hipLaunchKernelGGL(k_landmark_dinv, dim3(lm_blocks(d)), dim3(256), 0, s, d);
}
void launch_assemble(const DevBuf& d, bool add_lambda, hipStream_t s) {
    const size_t n = (size_t)(d.Ppad + TILE) * d.ld;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(256), 0, s, d, add_lambda ? 1 : 0);
}
void launch_schur_pairs(const DevBuf& d, hipStream_t s) {
    if (d.npairs) hipLaunchKernelGGL(k_schur_pairs, dim3(d.npairs), dim3(256), 0, s, d);
}
void launch_backsub(const DevBuf& d, int cur, int trial, hipStream_t s) {
    if (d.L) hipLaunchKernelGGL(k_backsub, dim3(lm_blocks(d)), dim3(256), 0, s, d, cur, trial);
}
void launch_update_kf(const DevBuf& d, int cur, int trial, hipStream_t s) {
    hipLaunchKernelGGL(k_update_kf, dim3((d.K + 63) / 64), dim3(64), 0, s, d, cur, trial);
}
void launch_reduce(const DevBuf& d, bool owns_pose_edges, double* red, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, s, d, d.E ? edge_blocks(d) : 0, d.L ? lm_blocks(d) : 0, owns_pose_edges ? 1 : 0, red);
}
void launch_lambda_init2(const DevBuf& d, const LmParams& lp, const double* red, bool first_iter, int iteration, hipStream_t s) {
    hipLaunchKernelGGL(k_lambda_init, dim3(1), dim3(256), 0, s, d, lp, red, first_iter ? 1 : 0, iteration);
}
void launch_decide(const DevBuf& d, const LmParams& lp, const double* red, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(256), 0, s, d, lp, red);
}
void launch_gate(const DevBuf& d, int state, double thresh, hipStream_t s) {
    if (d.E == 0) return;
    const size_t sh = (size_t)d.K * KFCAM_STRIDE * sizeof(double);
    hipLaunchKernelGGL(k_gate, dim3(edge_blocks(d)), dim3(256), sh, s, d, state, thresh, (uint8_t*)nullptr, 1);
}
void launch_depth(const DevBuf& d, int state, uint8_t* out, hipStream_t s) {
    if (d.E == 0) return;
    const size_t sh = (size_t)d.K * KFCAM_STRIDE * sizeof(double);
    hipLaunchKernelGGL(k_gate, dim3(edge_blocks(d)), dim3(256), sh, s, d, state, 0.0, out, 0);
}

}  // namespace plba

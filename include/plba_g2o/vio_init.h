// vio_init.h — the closed-form steps of MapHandler::tryVioInit that sit between its g2o graphs: gravity from three
// consecutive keyframes' positions and preintegrated deltas (src/mapHandler.cpp:4853-4900), the accelerometer bias given that gravity
// (:4903-4945) — both `JacobiSVD<MatrixXd>(A, ComputeThinU | ComputeThinV).solve(b)` of a 3 (N - 2) x 3 system there — and the
// velocities (:4955-4980).  Plain arrays, no Eigen: a caller that has Eigen keeps its own lines; this is what the harness re-enacts
// the call site with, checked against an independent restatement in the oracle (orc_vio_init, Householder QR).
//
// Indexing: N keyframes; interval m = 0 .. N - 2 is the preintegration from keyframe m to m + 1 (KeyFrame m + 1's GetIMUPreInt()).
// Rc / pc: camera-to-world rotation (row-major 3 x 3) and camera centre of every keyframe (T_kf_w); Rcb / pcb = Tbs^-1.
#pragma once
#include <cmath>
#include <vector>

namespace plba_vio {

inline void mat3_vec(const double* R, const double* v, double* o) { for (int i = 0; i < 3; ++i) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2]; }
inline void mat3_mul(const double* A, const double* B, double* C) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j]; }

// minimum-norm least squares of a (rows x 3) system as JacobiSVD<MatrixXd>(A, ComputeThinU | ComputeThinV).solve(b) computes it
// (src/mapHandler.cpp:4896, 4940): a ONE-SIDED Jacobi on the columns of A (Hestenes) — the rotated columns are U Sigma, the accumulated
// rotations V — so that small singular values keep their RELATIVE accuracy; singular values at or below Eigen's threshold
// max(rows, 3) * epsilon * sigma_max are treated as zero.  (Round 3 went through the eigen-decomposition of A^T A, which resolves
// singular values only down to sqrt(epsilon) * sigma_max: with cond(A) > 1e7 — an accelerometer-bias system observed through little
// rotation — a singular value that is rounding noise passed the threshold and was inverted.  ADVICE r03.)
inline void lstsq3(const std::vector<double>& A /* rows x 3 row-major */, const std::vector<double>& b, double* x) {
    const int rows = (int)b.size();
    std::vector<double> G(A.begin(), A.begin() + 3 * (size_t)rows);
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            double al = 0.0, be = 0.0, ga = 0.0;
            for (int i = 0; i < rows; ++i) { const double gp = G[3 * i + p], gq = G[3 * i + q]; al += gp * gp; be += gq * gq; ga += gp * gq; }
            if (ga == 0.0 || std::fabs(ga) <= 2.220446049250313e-16 * std::sqrt(al * be)) continue;
            rotated = true;
            const double zeta = (be - al) / (2.0 * ga), t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(zeta * zeta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
            for (int i = 0; i < rows; ++i) { const double gp = G[3 * i + p], gq = G[3 * i + q]; G[3 * i + p] = c * gp - sn * gq; G[3 * i + q] = sn * gp + c * gq; }
            for (int k = 0; k < 3; ++k) { const double vp = V[3 * k + p], vq = V[3 * k + q]; V[3 * k + p] = c * vp - sn * vq; V[3 * k + q] = sn * vp + c * vq; }
        }
        if (!rotated) break;
    }
    double sg2[3], gb[3], smax = 0.0;
    for (int k = 0; k < 3; ++k) {
        sg2[k] = gb[k] = 0.0;
        for (int i = 0; i < rows; ++i) { sg2[k] += G[3 * i + k] * G[3 * i + k]; gb[k] += G[3 * i + k] * b[i]; }
        smax = std::fmax(smax, std::sqrt(sg2[k]));
    }
    const double thr = (rows > 3 ? rows : 3) * 2.220446049250313e-16 * smax;
    x[0] = x[1] = x[2] = 0.0;
    for (int k = 0; k < 3; ++k) {
        if (!(std::sqrt(sg2[k]) > thr)) continue;
        const double w = gb[k] / sg2[k];      // (u_k . b) / sigma_k with u_k = g_k / sigma_k
        for (int a = 0; a < 3; ++a) x[a] += V[3 * a + k] * w;
    }
}

// src/mapHandler.cpp:4853-4900.  g0 = gpre / |gpre| * 9.810
inline void gravity(int N, const double* dt, const double* dP, const double* dV, const double* Rc, const double* pc, const double* Rcb, const double* pcb,
                    double* gpre, double* g0) {
    std::vector<double> C(9 * (size_t)(N - 2), 0.0), D(3 * (size_t)(N - 2), 0.0);
    for (int i = 0; i < N - 2; ++i) {
        const double dt12 = dt[i], dt23 = dt[i + 1];
        const double *dp12 = dP + 3 * i, *dv12 = dV + 3 * i, *dp23 = dP + 3 * (i + 1);
        const double *R1 = Rc + 9 * i, *R2 = Rc + 9 * (i + 1), *R3 = Rc + 9 * (i + 2), *p1 = pc + 3 * i, *p2 = pc + 3 * (i + 1), *p3 = pc + 3 * (i + 2);
        const double beta = 0.5 * (dt12 * dt12 * dt23 + dt12 * dt23 * dt23);
        double R1cb[9], R2cb[9], a[3], b[3], c[3], d32[3], d12[3], e[3], f[3], R32[9], R12[9];
        mat3_mul(R1, Rcb, R1cb); mat3_mul(R2, Rcb, R2cb);
        for (int t = 0; t < 9; ++t) { R32[t] = R3[t] - R2[t]; R12[t] = R1[t] - R2[t]; }
        mat3_vec(R32, pcb, d32); mat3_vec(R12, pcb, d12); mat3_vec(R1cb, dp12, a); mat3_vec(R2cb, dp23, b); mat3_vec(R1cb, dv12, c);
        for (int t = 0; t < 3; ++t) {
            e[t] = (p2[t] - p1[t]) * dt23 + (p2[t] - p3[t]) * dt12;                                                   // lambda
            f[t] = d32[t] * dt12 + d12[t] * dt23 + a[t] * dt23 - b[t] * dt12 - c[t] * dt12 * dt23;                     // gamma
            C[3 * (3 * i + t) + t] = beta;
            D[3 * i + t] = f[t] - e[t];
        }
    }
    lstsq3(C, D, gpre);
    const double n = std::sqrt(gpre[0] * gpre[0] + gpre[1] * gpre[1] + gpre[2] * gpre[2]);
    for (int t = 0; t < 3; ++t) g0[t] = gpre[t] / n * 9.810;
}

// src/mapHandler.cpp:4903-4945
inline void acc_bias(int N, const double* dt, const double* dP, const double* dV, const double* JPa, const double* JVa, const double* Rc, const double* pc,
                     const double* Rcb, const double* pcb, const double* g0, double* ba) {
    std::vector<double> A(9 * (size_t)(N - 2), 0.0), B(3 * (size_t)(N - 2), 0.0);
    for (int i = 0; i < N - 2; ++i) {
        const double dt12 = dt[i], dt23 = dt[i + 1];
        const double *dp12 = dP + 3 * i, *dv12 = dV + 3 * i, *dp23 = dP + 3 * (i + 1);
        const double *R1 = Rc + 9 * i, *R2 = Rc + 9 * (i + 1), *R3 = Rc + 9 * (i + 2), *p1 = pc + 3 * i, *p2 = pc + 3 * (i + 1), *p3 = pc + 3 * (i + 2);
        double R1cb[9], R2cb[9], a[3], b[3], c[3], d12[3], d23[3], R12[9], R23[9], F1[9], F2[9], F3[9];
        mat3_mul(R1, Rcb, R1cb); mat3_mul(R2, Rcb, R2cb);
        for (int t = 0; t < 9; ++t) { R12[t] = R1[t] - R2[t]; R23[t] = R2[t] - R3[t]; }
        mat3_vec(R12, pcb, d12); mat3_vec(R23, pcb, d23); mat3_vec(R1cb, dp12, a); mat3_vec(R2cb, dp23, b); mat3_vec(R1cb, dv12, c);
        mat3_mul(R1cb, JPa + 9 * i, F1); mat3_mul(R2cb, JPa + 9 * (i + 1), F2); mat3_mul(R1cb, JVa + 9 * i, F3);
        for (int t = 0; t < 3; ++t) {
            B[3 * i + t] = p2[t] * dt23 - p3[t] * dt12 - p1[t] * dt23 + p2[t] * dt12 + 0.5 * g0[t] * (dt12 * dt12 * dt23 + dt23 * dt23 * dt12)
                         - a[t] * dt23 + b[t] * dt12 - d12[t] * dt23 + d23[t] * dt12 + c[t] * dt12 * dt23;
            for (int u = 0; u < 3; ++u) A[3 * (3 * i + t) + u] = F1[3 * t + u] * dt23 - F2[3 * t + u] * dt12 - F3[3 * t + u] * dt12 * dt23;
        }
    }
    lstsq3(A, B, ba);
}

// src/mapHandler.cpp:4955-4980.  Rb / pb: body rotation (row-major) and position of every keyframe (imuState); V out: N x 3
inline void velocities(int N, const double* dt, const double* dP, const double* dV, const double* Rb, const double* pb, const double* g0, double* V) {
    for (int i = 0; i < N; ++i) {
        if (i != N - 1) {
            double r[3];
            mat3_vec(Rb + 9 * i, dP + 3 * i, r);
            for (int t = 0; t < 3; ++t) V[3 * i + t] = (pb[3 * (i + 1) + t] - pb[3 * i + t] - 0.5 * g0[t] * dt[i] * dt[i] - r[t]) / dt[i];
        } else {
            double r[3];
            mat3_vec(Rb + 9 * (i - 1), dV + 3 * (i - 1), r);
            for (int t = 0; t < 3; ++t) V[3 * i + t] = V[3 * (i - 1) + t] + g0[t] * dt[i - 1] + r[t];
        }
    }
}

}  // namespace plba_vio

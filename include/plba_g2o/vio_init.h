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

// minimum-norm least squares of a (rows x 3) system through the eigen-decomposition of A^T A (= the thin SVD's V and sigma^2), singular
// values below Eigen's JacobiSVD threshold (max(rows, 3) * epsilon * sigma_max) treated as zero
inline void lstsq3(const std::vector<double>& A /* rows x 3 row-major */, const std::vector<double>& b, double* x) {
    const int rows = (int)b.size();
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, r[3] = {0, 0, 0};
    for (int i = 0; i < rows; ++i) for (int a = 0; a < 3; ++a) { r[a] += A[3 * i + a] * b[i]; for (int c = 0; c < 3; ++c) M[3 * a + c] += A[3 * i + a] * A[3 * i + c]; }
    for (int sweep = 0; sweep < 50; ++sweep) {
        const double off = M[1] * M[1] + M[2] * M[2] + M[5] * M[5];
        if (off == 0.0) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            const double apq = M[3 * p + q];
            if (apq == 0.0) continue;
            const double th = (M[3 * q + q] - M[3 * p + p]) / (2.0 * apq), t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; ++k) { const double a = M[3 * k + p], bq = M[3 * k + q]; M[3 * k + p] = c * a - s * bq; M[3 * k + q] = s * a + c * bq; }
            for (int k = 0; k < 3; ++k) { const double a = M[3 * p + k], bq = M[3 * q + k]; M[3 * p + k] = c * a - s * bq; M[3 * q + k] = s * a + c * bq; }
            for (int k = 0; k < 3; ++k) { const double a = V[3 * k + p], bq = V[3 * k + q]; V[3 * k + p] = c * a - s * bq; V[3 * k + q] = s * a + c * bq; }
        }
    }
    double smax = 0.0;
    for (int k = 0; k < 3; ++k) smax = std::fmax(smax, std::sqrt(std::fmax(M[4 * k], 0.0)));
    const double thr = (rows > 3 ? rows : 3) * 2.220446049250313e-16 * smax;
    x[0] = x[1] = x[2] = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double sg = std::sqrt(std::fmax(M[4 * k], 0.0));
        if (!(sg > thr)) continue;
        const double w = (V[k] * r[0] + V[3 + k] * r[1] + V[6 + k] * r[2]) / (sg * sg);
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

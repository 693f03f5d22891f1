// plba_math.h — per-thread fp64 math of the VI-BA hot path for gfx950 device code.
//
// Written for registers, not for generality: fixed-size, fully unrolled, FMA-friendly, no
// local arrays indexed at run time.  Every function is PLBA_HD so the same source can be compiled
// by a host compiler for the CPU-side formula cross-check in tests/test_device_math_host.py
// (that check is test plumbing; the product only ever runs these functions inside HIP kernels).
//
// Reference semantics followed (file:line under /root/reference):
//   quaternion/SO3 ....... IMU/so3.cpp:32-89,199-280 (+ Eigen Quaterniond conventions)
//   NavState oplus ....... IMU/NavState.cpp:69-121
//   point edge ........... IMU/g2otypes.h:230-275, IMU/g2otypes.cpp:286-341
//   line edge ............ IMU/g2otypes.h:783-825, IMU/g2otypes.cpp:1306-1359
//   IMU PVR / bias edge .. IMU/g2otypes.cpp:27-284
//   Huber ................ g2o RobustKernelHuber
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PLBA_HD __host__ __device__ __forceinline__
#else
#define PLBA_HD inline
#endif

namespace plba {

struct V3 { double x, y, z; };
struct M3 { double a[9]; };                 // row-major
struct Q4 { double x, y, z, w; };           // Eigen coeff order

PLBA_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PLBA_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PLBA_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PLBA_HD V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
PLBA_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PLBA_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PLBA_HD double norm(V3 a) { return sqrt(dot(a, a)); }

PLBA_HD V3 mul(const M3& A, V3 v) {
    return v3(A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
              A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z);
}
PLBA_HD V3 mulT(const M3& A, V3 v) {  // A^T v
    return v3(A.a[0] * v.x + A.a[3] * v.y + A.a[6] * v.z, A.a[1] * v.x + A.a[4] * v.y + A.a[7] * v.z,
              A.a[2] * v.x + A.a[5] * v.y + A.a[8] * v.z);
}
PLBA_HD M3 mul(const M3& A, const M3& B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.a[i * 3 + j] = A.a[i * 3] * B.a[j] + A.a[i * 3 + 1] * B.a[3 + j] + A.a[i * 3 + 2] * B.a[6 + j];
    return C;
}
PLBA_HD M3 mulABt(const M3& A, const M3& B) {  // A * B^T
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.a[i * 3 + j] = A.a[i * 3] * B.a[j * 3] + A.a[i * 3 + 1] * B.a[j * 3 + 1] + A.a[i * 3 + 2] * B.a[j * 3 + 2];
    return C;
}
PLBA_HD M3 mulAtB(const M3& A, const M3& B) {  // A^T * B
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.a[i * 3 + j] = A.a[i] * B.a[j] + A.a[3 + i] * B.a[3 + j] + A.a[6 + i] * B.a[6 + j];
    return C;
}
PLBA_HD M3 transpose(const M3& A) {
    M3 T;
    T.a[0] = A.a[0]; T.a[1] = A.a[3]; T.a[2] = A.a[6];
    T.a[3] = A.a[1]; T.a[4] = A.a[4]; T.a[5] = A.a[7];
    T.a[6] = A.a[2]; T.a[7] = A.a[5]; T.a[8] = A.a[8];
    return T;
}
PLBA_HD M3 hat(V3 v) {  // IMU/so3.cpp:283-290
    M3 O;
    O.a[0] = 0; O.a[1] = -v.z; O.a[2] = v.y;
    O.a[3] = v.z; O.a[4] = 0; O.a[5] = -v.x;
    O.a[6] = -v.y; O.a[7] = v.x; O.a[8] = 0;
    return O;
}
PLBA_HD M3 eye3() { M3 I; I.a[0] = 1; I.a[1] = 0; I.a[2] = 0; I.a[3] = 0; I.a[4] = 1; I.a[5] = 0; I.a[6] = 0; I.a[7] = 0; I.a[8] = 1; return I; }

// ---- quaternion (unit), Eigen conventions ---------------------------------------------------
PLBA_HD Q4 q_normalized(Q4 q) {
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    Q4 r; r.x = q.x / n; r.y = q.y / n; r.z = q.z / n; r.w = q.w / n; return r;
}
PLBA_HD Q4 q_mul(Q4 a, Q4 b) {
    Q4 r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
PLBA_HD Q4 q_conj(Q4 a) { Q4 r; r.x = -a.x; r.y = -a.y; r.z = -a.z; r.w = a.w; return r; }
PLBA_HD M3 q_to_R(Q4 q) {  // Eigen toRotationMatrix
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 R;
    R.a[0] = 1 - (tyy + tzz); R.a[1] = txy - twz; R.a[2] = txz + twy;
    R.a[3] = txy + twz; R.a[4] = 1 - (txx + tzz); R.a[5] = tyz - twx;
    R.a[6] = txz - twy; R.a[7] = tyz + twx; R.a[8] = 1 - (txx + tyy);
    return R;
}
PLBA_HD Q4 R_to_q(const M3& m) {  // Eigen Quaterniond(Matrix3d); branch-free selects instead of run-time indexing
    Q4 q;
    double t = m.a[0] + m.a[4] + m.a[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m.a[7] - m.a[5]) * t;
        q.y = (m.a[2] - m.a[6]) * t;
        q.z = (m.a[3] - m.a[1]) * t;
    } else if (m.a[0] >= m.a[4] && m.a[0] >= m.a[8]) {  // i = 0, j = 1, k = 2
        t = sqrt(m.a[0] - m.a[4] - m.a[8] + 1.0);
        q.x = 0.5 * t; t = 0.5 / t;
        q.w = (m.a[7] - m.a[5]) * t; q.y = (m.a[3] + m.a[1]) * t; q.z = (m.a[6] + m.a[2]) * t;
    } else if (m.a[4] > m.a[0] && m.a[4] >= m.a[8]) {    // i = 1, j = 2, k = 0
        t = sqrt(m.a[4] - m.a[8] - m.a[0] + 1.0);
        q.y = 0.5 * t; t = 0.5 / t;
        q.w = (m.a[2] - m.a[6]) * t; q.z = (m.a[7] + m.a[5]) * t; q.x = (m.a[1] + m.a[3]) * t;
    } else {                                               // i = 2, j = 0, k = 1
        t = sqrt(m.a[8] - m.a[0] - m.a[4] + 1.0);
        q.z = 0.5 * t; t = 0.5 / t;
        q.w = (m.a[3] - m.a[1]) * t; q.x = (m.a[2] + m.a[6]) * t; q.y = (m.a[5] + m.a[7]) * t;
    }
    return q;
}
PLBA_HD V3 q_rot(Q4 q, V3 v) {  // Eigen _transformVector
    V3 qv = v3(q.x, q.y, q.z);
    V3 uv = cross(qv, v);
    uv = uv + uv;
    return v + q.w * uv + cross(qv, uv);
}
PLBA_HD Q4 so3_exp(V3 w) {  // IMU/so3.cpp:257-280
    double theta = norm(w), half = 0.5 * theta, imag, real = cos(half);
    if (theta < 1e-10) {
        double t2 = theta * theta, t4 = t2 * t2;
        imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4;
    } else {
        imag = sin(half) / theta;
    }
    Q4 q; q.x = imag * w.x; q.y = imag * w.y; q.z = imag * w.z; q.w = real;
    return q_normalized(q);
}
PLBA_HD V3 so3_log(Q4 q) {  // IMU/so3.cpp:206-247 (atan, not atan2; SURVEY B-Q13)
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z), w = q.w, f;
    if (n < 1e-10) f = 2. / w - 2. * (n * n) / (w * w * w);
    else f = 2 * atan(n / w) / n;
    return v3(f * q.x, f * q.y, f * q.z);
}
PLBA_HD M3 so3_Jr(V3 w) {  // IMU/so3.cpp:32-49
    double theta = norm(w);
    if (theta < 0.00001) return eye3();
    M3 K = hat((1.0 / theta) * w), KK = mul(K, K), I = eye3(), J;
    double a = (1 - cos(theta)) / theta, b = 1 - sin(theta) / theta;
#pragma unroll
    for (int i = 0; i < 9; ++i) J.a[i] = I.a[i] - a * K.a[i] + b * KK.a[i];
    return J;
}
PLBA_HD M3 so3_JrInv(V3 w) {  // IMU/so3.cpp:50-68
    double theta = norm(w);
    if (theta < 0.00001) return eye3();
    M3 K = hat((1.0 / theta) * w), KK = mul(K, K), W = hat(w), I = eye3(), J;
    double c = 1.0 - (1.0 + cos(theta)) * theta / (2.0 * sin(theta));
#pragma unroll
    for (int i = 0; i < 9; ++i) J.a[i] = I.a[i] + 0.5 * W.a[i] + c * KK.a[i];
    return J;
}

// ---- keyframe state record (24 doubles in HBM) ------------------------------------------------
// [0..2] P  [3..5] V  [6..9] q(x,y,z,w)  [10..12] bg  [13..15] ba  [16..18] dbg  [19..21] dba  [22,23] pad
constexpr int KF_STRIDE = 24;
struct KfPose { V3 P; Q4 q; };

PLBA_HD void kf_oplus_pvr(const double* s, const double* u /*9*/, double* o) {  // IMU/NavState.cpp:69-98
    Q4 q; q.x = s[6]; q.y = s[7]; q.z = s[8]; q.w = s[9];
    Q4 qc = q_normalized(q);                       // Get_R() copy-normalises
    M3 R = q_to_R(qc);
    V3 d = mul(R, v3(u[0], u[1], u[2]));
    o[0] = s[0] + d.x; o[1] = s[1] + d.y; o[2] = s[2] + d.z;
    o[3] = s[3] + u[3]; o[4] = s[4] + u[4]; o[5] = s[5] + u[5];
    Q4 dq = so3_exp(v3(u[6], u[7], u[8]));
    Q4 qn = q_normalized(q_mul(qc, dq));
    o[6] = qn.x; o[7] = qn.y; o[8] = qn.z; o[9] = qn.w;
}

// ---- Huber (g2o RobustKernelHuber::robustify) ---------------------------------------------------
PLBA_HD void huber(double e, double delta, double& rho0, double& rho1) {
    double dsqr = delta * delta;
    if (e <= dsqr) { rho0 = e; rho1 = 1.0; }
    else { double s = sqrt(e); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// ---- camera block staged per keyframe in LDS: M = Rcb * Rwb^T (9), Pwb (3) ---------------------
struct Cam { double fx, fy, cx, cy; M3 Rcb; V3 c0; /* c0 = Rcb * Pbc */ };
constexpr int KFCAM_STRIDE = 12;

PLBA_HD void kfcam_make(const Cam& cam, const double* s, double* out /*12*/) {
    Q4 q; q.x = s[6]; q.y = s[7]; q.z = s[8]; q.w = s[9];
    M3 Rwb = q_to_R(q);
    M3 M = mulABt(cam.Rcb, Rwb);
#pragma unroll
    for (int i = 0; i < 9; ++i) out[i] = M.a[i];
    out[9] = s[0]; out[10] = s[1]; out[11] = s[2];
}

// Reprojection of one world point through a staged keyframe: Pc = M (Pw - Pwb) - c0   (g2otypes.h:243-260)
PLBA_HD V3 cam_Pc(const Cam& cam, const double* kc, V3 Pw) {
    M3 M;
#pragma unroll
    for (int i = 0; i < 9; ++i) M.a[i] = kc[i];
    V3 d = Pw - v3(kc[9], kc[10], kc[11]);
    return mul(M, d) - cam.c0;
}

// Jacobian rows of one projected point.  Given Pc and the staged M, returns
//   jl[2][3] : d(u,v)/dPw            =  Jpi * M
//   jp[2][3] : d(u,v)/d(dp, body)    = -Jpi * Rcb
//   jr[2][3] : d(u,v)/d(dphi)        =  Jpi * hat(Pc + c0) * Rcb
// where Jpi = [[fx/z, 0, -fx x/z^2], [0, fy/z, -fy y/z^2]].  (signs of d proj; the edges apply theirs)
struct ProjJac { double jl[6], jp[6], jr[6]; };
PLBA_HD ProjJac proj_jac(const Cam& cam, const double* kc, V3 Pc) {
    double iz = 1.0 / Pc.z;
    double a0 = cam.fx * iz, a2 = -cam.fx * Pc.x * iz * iz;
    double b1 = cam.fy * iz, b2 = -cam.fy * Pc.y * iz * iz;
    ProjJac J;
    // Jpi * M
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        J.jl[c] = a0 * kc[c] + a2 * kc[6 + c];
        J.jl[3 + c] = b1 * kc[3 + c] + b2 * kc[6 + c];
        J.jp[c] = -(a0 * cam.Rcb.a[c] + a2 * cam.Rcb.a[6 + c]);
        J.jp[3 + c] = -(b1 * cam.Rcb.a[3 + c] + b2 * cam.Rcb.a[6 + c]);
    }
    // hat(Paux) * Rcb with Paux = Pc + c0 = M (Pw - Pwb)
    V3 pa = Pc + cam.c0;
    M3 HR = mul(hat(pa), cam.Rcb);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        J.jr[c] = a0 * HR.a[c] + a2 * HR.a[6 + c];
        J.jr[3 + c] = b1 * HR.a[3 + c] + b2 * HR.a[6 + c];
    }
    return J;
}

// ---- compact edge record written by the linearisation kernel: 16 doubles = one 128-byte line -------------
// Both residual rows a = 0,1 are described in the CAMERA frame by a 3-vector u_a (= row of the projection
// Jacobian; for a line end-point already contracted with the line normal l12) and the point P_a = M (Pw - Pwb):
//   [0..2] uA  [3..5] PA  [6..8] uB  [9..11] PB  [12] w = rho1 * inv_sigma2 (0: inactive)  [13] e0  [14] e1  [15] chi2
// From these and the per-keyframe block M = Rcb Rwb^T every consumer rebuilds what it needs:
//   landmark Jacobian row   Jl_a = sl * (M^T u_a)^T             sl = -1 point edge, +1 line edge
//   pose Jacobian row       Jp_a = g_a^T * blkdiag(Rcb, Rcb)    g_a = [ u_a ; P_a x u_a ]                 (point)
//                                                               g_a = [ -Rcb M^T u_a ; -(P_a x u_a) ]     (line, reference
//                            world-frame position block, SURVEY B-Q1;  [ -u_a ; ... ] with fix_line_position_jacobian)
// (derivation: -u^T hat(P) = (P x u)^T, and -u^T M = -(Rcb M^T u)^T Rcb.)
constexpr int EREC = 16;
// A POINT observation's record is its camera-frame point alone: both rows of the projection Jacobian (u_A, u_B) and
// P_A = P_B follow from it and the intrinsics, with exactly the expressions of point_edge_rec.  Point records therefore
// take 64 bytes (line records keep 128; both kinds are packed back to back in keyframe-major order),
//   [0..2] Pc  [3] w  [4] e0  [5] e1  [6] chi2  [7] 0
// which is what k_linearize writes and the landmark / Schur / back-substitution passes read for 5 observations in 6.
constexpr int EREC_PT_W = 3, EREC_PT_E0 = 4;
constexpr int EREC_UNIT = 8;        // record positions (ob_pos, ent_pi / ent_pj) count 64-byte units: 1 per point record, 2 per line record
PLBA_HD void point_rows_from_Pc(const Cam& cam, V3 Pc, V3& ua, V3& ub, V3& P) {
    const double iz = 1.0 / Pc.z;
    ua = v3(cam.fx * iz, 0.0, -cam.fx * Pc.x * iz * iz);
    ub = v3(0.0, cam.fy * iz, -cam.fy * Pc.y * iz * iz);
    P = Pc + cam.c0;
}

PLBA_HD void point_edge_rec(const Cam& cam, const double* kc, V3 Pw, double u, double v, double* e2, double* rec12, bool& depth_pos, bool jac, V3* Pc_out = nullptr) {
    V3 Pc = cam_Pc(cam, kc, Pw);
    if (Pc_out) *Pc_out = Pc;
    const double iz = 1.0 / Pc.z;
    e2[0] = u - (Pc.x * iz * cam.fx + cam.cx);
    e2[1] = v - (Pc.y * iz * cam.fy + cam.cy);
    depth_pos = Pc.z > 0.0;
    if (!jac) return;
    const V3 P = Pc + cam.c0;
    rec12[0] = cam.fx * iz; rec12[1] = 0.0; rec12[2] = -cam.fx * Pc.x * iz * iz;
    rec12[3] = P.x; rec12[4] = P.y; rec12[5] = P.z;
    rec12[6] = 0.0; rec12[7] = cam.fy * iz; rec12[8] = -cam.fy * Pc.y * iz * iz;
    rec12[9] = P.x; rec12[10] = P.y; rec12[11] = P.z;
}
PLBA_HD void line_edge_rec(const Cam& cam, const double* kc, V3 Ps_w, V3 Pe_w, double lx, double ly, double lz,
                           double* e2, double* rec12, bool& depth_pos, bool jac) {
    V3 Ps = cam_Pc(cam, kc, Ps_w), Pe = cam_Pc(cam, kc, Pe_w);
    const double izs = 1.0 / Ps.z, ize = 1.0 / Pe.z;
    e2[0] = lx * (Ps.x * izs * cam.fx + cam.cx) + ly * (Ps.y * izs * cam.fy + cam.cy) + lz;
    e2[1] = lx * (Pe.x * ize * cam.fx + cam.cx) + ly * (Pe.y * ize * cam.fy + cam.cy) + lz;
    depth_pos = (Ps.z > 0.0) && (Pe.z > 0.0);
    if (!jac) return;
    const V3 A = Ps + cam.c0, B = Pe + cam.c0;
    rec12[0] = lx * cam.fx * izs; rec12[1] = ly * cam.fy * izs; rec12[2] = -(lx * cam.fx * Ps.x + ly * cam.fy * Ps.y) * izs * izs;
    rec12[3] = A.x; rec12[4] = A.y; rec12[5] = A.z;
    rec12[6] = lx * cam.fx * ize; rec12[7] = ly * cam.fy * ize; rec12[8] = -(lx * cam.fx * Pe.x + ly * cam.fy * Pe.y) * ize * ize;
    rec12[9] = B.x; rec12[10] = B.y; rec12[11] = B.z;
}
// one residual row rebuilt from its record half: v = M^T u (world frame), g = pose coefficients in the blkdiag(Rcb,Rcb) basis
PLBA_HD void rec_row(bool is_pt, bool fix_q1, const double* kc, const M3& Rcb, V3 u, V3 P, V3& v, double* g6) {
    M3 M;
#pragma unroll
    for (int i = 0; i < 9; ++i) M.a[i] = kc[i];
    v = mulT(M, u);
    const V3 px = cross(P, u);
    if (is_pt) {
        g6[0] = u.x; g6[1] = u.y; g6[2] = u.z; g6[3] = px.x; g6[4] = px.y; g6[5] = px.z;
    } else {
        const V3 dp = fix_q1 ? u : mul(Rcb, v);
        g6[0] = -dp.x; g6[1] = -dp.y; g6[2] = -dp.z; g6[3] = -px.x; g6[4] = -px.y; g6[5] = -px.z;
    }
}
// Jp row (6) = g^T blkdiag(Rcb, Rcb)
PLBA_HD void basis_apply(const M3& Rcb, const double* g6, double* jp6) {
    const V3 a = mulT(Rcb, v3(g6[0], g6[1], g6[2])), b = mulT(Rcb, v3(g6[3], g6[4], g6[5]));
    jp6[0] = a.x; jp6[1] = a.y; jp6[2] = a.z; jp6[3] = b.x; jp6[4] = b.y; jp6[5] = b.z;
}

// Point edge: e = obs - proj  =>  Jl = -Jpi M, Jp(dp) = +Jpi Rcb, Jp(dphi) = -Jpi hat(Paux) Rcb   (g2otypes.cpp:286-341)
PLBA_HD void point_edge(const Cam& cam, const double* kc, V3 Pw, double u, double v, double* e2, double* Jp12, double* Jl6, bool& depth_pos, bool jac) {
    V3 Pc = cam_Pc(cam, kc, Pw);
    double iz = 1.0 / Pc.z;
    e2[0] = u - (Pc.x * iz * cam.fx + cam.cx);
    e2[1] = v - (Pc.y * iz * cam.fy + cam.cy);
    depth_pos = Pc.z > 0.0;
    if (!jac) return;
    ProjJac J = proj_jac(cam, kc, Pc);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Jl6[r * 3 + c] = -J.jl[r * 3 + c];
            Jp12[r * 6 + c] = -J.jp[r * 3 + c];
            Jp12[r * 6 + 3 + c] = -J.jr[r * 3 + c];
        }
}

// Line edge (g2otypes.h:783-825, g2otypes.cpp:1306-1359): e0 = l . (proj(sP),1), e1 = l . (proj(eP),1), e2 == 0.
//   Jl row0 = l12^T Jpi_s M,  row1 = l12^T Jpi_e M
//   Jp(dphi) rowk = l12^T Jpi_k Rcb hat(Rwb^T d_k) = l12^T Jpi_k hat(M d_k) Rcb
//   Jp(dp)   rowk = l12^T Jpi_k (-M)   [reference: world-frame, B-Q1]   or  l12^T Jpi_k (-Rcb)  [fix_q1]
PLBA_HD void line_edge(const Cam& cam, const double* kc, V3 Ps_w, V3 Pe_w, double lx, double ly, double lz, bool fix_q1,
                       double* e2, double* Jp12, double* Jl6, bool& depth_pos, bool jac) {
    V3 Ps = cam_Pc(cam, kc, Ps_w), Pe = cam_Pc(cam, kc, Pe_w);
    double izs = 1.0 / Ps.z, ize = 1.0 / Pe.z;
    e2[0] = lx * (Ps.x * izs * cam.fx + cam.cx) + ly * (Ps.y * izs * cam.fy + cam.cy) + lz;
    e2[1] = lx * (Pe.x * ize * cam.fx + cam.cx) + ly * (Pe.y * ize * cam.fy + cam.cy) + lz;
    depth_pos = (Ps.z > 0.0) && (Pe.z > 0.0);
    if (!jac) return;
    ProjJac Js = proj_jac(cam, kc, Ps), Je = proj_jac(cam, kc, Pe);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double ls = lx * Js.jl[c] + ly * Js.jl[3 + c];      // l12^T Jpi_s M
        double le = lx * Je.jl[c] + ly * Je.jl[3 + c];
        Jl6[c] = ls;
        Jl6[3 + c] = le;
        if (fix_q1) {
            Jp12[c] = lx * Js.jp[c] + ly * Js.jp[3 + c];    // -l12^T Jpi Rcb
            Jp12[6 + c] = lx * Je.jp[c] + ly * Je.jp[3 + c];
        } else {
            Jp12[c] = -ls;
            Jp12[6 + c] = -le;
        }
        Jp12[3 + c] = lx * Js.jr[c] + ly * Js.jr[3 + c];
        Jp12[6 + 3 + c] = lx * Je.jr[c] + ly * Je.jr[3 + c];
    }
}

// ---- IMU preintegration payload (142 doubles) -----------------------------------------------------
constexpr int PRE_STRIDE = 142;
// offsets: dP 0, dV 3, dR 6, JPg 15, JPa 24, JVg 33, JVa 42, JRg 51, cov 60, dt 141
PLBA_HD M3 ld_m3(const double* p) { M3 A;
#pragma unroll
    for (int i = 0; i < 9; ++i) A.a[i] = p[i];
    return A; }
PLBA_HD V3 ld_v3(const double* p) { return v3(p[0], p[1], p[2]); }

// EdgeNavStatePVR::computeError (g2otypes.cpp:27-92).  si/sj = keyframe records; bias deltas come from si.
PLBA_HD void pvr_error(const double* si, const double* sj, const double* pre, V3 gw, double* e9) {
    Q4 qi; qi.x = si[6]; qi.y = si[7]; qi.z = si[8]; qi.w = si[9];
    Q4 qj; qj.x = sj[6]; qj.y = sj[7]; qj.z = sj[8]; qj.w = sj[9];
    qi = q_normalized(qi); qj = q_normalized(qj);
    Q4 qiT = q_normalized(q_conj(qi));
    double dT = pre[141], dT2 = dT * dT;
    V3 Pi = ld_v3(si), Vi = ld_v3(si + 3), Pj = ld_v3(sj), Vj = ld_v3(sj + 3);
    V3 dbg = ld_v3(si + 16), dba = ld_v3(si + 19);
    V3 a = Pj - Pi - dT * Vi - (0.5 * dT2) * gw;
    V3 rP = q_rot(qiT, a) - (ld_v3(pre) + mul(ld_m3(pre + 15), dbg) + mul(ld_m3(pre + 24), dba));
    V3 b = Vj - Vi - dT * gw;
    V3 rV = q_rot(qiT, b) - (ld_v3(pre + 3) + mul(ld_m3(pre + 33), dbg) + mul(ld_m3(pre + 42), dba));
    Q4 dRij = q_normalized(R_to_q(ld_m3(pre + 6)));
    Q4 dRdbg = so3_exp(mul(ld_m3(pre + 51), dbg));
    Q4 A = q_normalized(q_mul(dRij, dRdbg));
    Q4 Ainv = q_normalized(q_conj(A));
    Q4 B = q_normalized(q_mul(Ainv, qiT));
    Q4 C = q_normalized(q_mul(B, qj));
    V3 rPhi = so3_log(C);
    e9[0] = rP.x; e9[1] = rP.y; e9[2] = rP.z; e9[3] = rV.x; e9[4] = rV.y; e9[5] = rV.z; e9[6] = rPhi.x; e9[7] = rPhi.y; e9[8] = rPhi.z;
}
// lda: row stride of J0 / J1, ldb: row stride of J2 (9 / 6 for separate blocks; 24 / 24 when the three sit side by side).
// Split in two so that the device can run them on different wavefronts: everything that does not depend on the
// rotation residual (pvr_jac_static: it runs next to pvr_error) and the three blocks that do (pvr_jac_rphi).
PLBA_HD void pvr_jac_static(const double* si, const double* sj, const double* pre, V3 gw, double* J0, double* J1, double* J2, int lda, int ldb, M3& RjTRi, M3& JrB) {
    Q4 qi; qi.x = si[6]; qi.y = si[7]; qi.z = si[8]; qi.w = si[9];
    Q4 qj; qj.x = sj[6]; qj.y = sj[7]; qj.z = sj[8]; qj.w = sj[9];
    M3 Ri = q_to_R(qi), Rj = q_to_R(qj), RiT = transpose(Ri);
    double dT = pre[141], dT2 = dT * dT;
    V3 Pi = ld_v3(si), Vi = ld_v3(si + 3), Pj = ld_v3(sj), Vj = ld_v3(sj + 3);
    V3 dbg = ld_v3(si + 16);
    M3 H1 = hat(mul(RiT, Pj - Pi - dT * Vi - (0.5 * dT2) * gw));
    M3 H2 = hat(mul(RiT, Vj - Vi - dT * gw));
    RjTRi = mulAtB(Rj, Ri);
    M3 RiTRj = mulAtB(Ri, Rj);
    JrB = so3_Jr(mul(ld_m3(pre + 51), dbg));
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
    for (int r = 0; r < 3; ++r) {
        J0[r * lda + r] = -1.0;
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
        for (int c = 0; c < 3; ++c) {
            J0[r * lda + 3 + c] = -RiT.a[r * 3 + c] * dT;
            J0[r * lda + 6 + c] = H1.a[r * 3 + c];
            J0[(3 + r) * lda + 3 + c] = -RiT.a[r * 3 + c];
            J0[(3 + r) * lda + 6 + c] = H2.a[r * 3 + c];
            J1[r * lda + c] = RiTRj.a[r * 3 + c];
            J1[(3 + r) * lda + 3 + c] = RiT.a[r * 3 + c];
            J2[r * ldb + c] = -pre[15 + r * 3 + c];
            J2[r * ldb + 3 + c] = -pre[24 + r * 3 + c];
            J2[(3 + r) * ldb + c] = -pre[33 + r * 3 + c];
            J2[(3 + r) * ldb + 3 + c] = -pre[42 + r * 3 + c];
        }
    }
}
PLBA_HD void pvr_jac_rphi(const double* pre, const double* e9, const M3& RjTRi, const M3& JrB, double* J0, double* J1, double* J2, int lda, int ldb) {
    V3 rPhi = v3(e9[6], e9[7], e9[8]);
    M3 JrInv = so3_JrInv(rPhi);
    M3 A33 = mul(JrInv, RjTRi);       // J0(6,6) = -JrInv * Rj^T * Ri
    M3 ET = q_to_R(q_normalized(q_conj(so3_exp(rPhi))));
    M3 B33 = mul(mul(mul(JrInv, ET), JrB), ld_m3(pre + 51));   // J2(6,0) = -JrInv * Exp(rPhi)^T * Jr * JRg
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
    for (int r = 0; r < 3; ++r) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
        for (int c = 0; c < 3; ++c) {
            J0[(6 + r) * lda + 6 + c] = -A33.a[r * 3 + c];
            J1[(6 + r) * lda + 6 + c] = JrInv.a[r * 3 + c];
            J2[(6 + r) * ldb + c] = -B33.a[r * 3 + c];
        }
    }
}
// EdgeNavStatePVR::linearizeOplus (g2otypes.cpp:94-234).  J0, J1: 9x9 row-major; J2: 9x6.  Caller zero-fills.
PLBA_HD void pvr_jacobians(const double* si, const double* sj, const double* pre, V3 gw, const double* e9, double* J0, double* J1, double* J2, int lda = 9, int ldb = 6) {
    M3 RjTRi, JrB;
    pvr_jac_static(si, sj, pre, gw, J0, J1, J2, lda, ldb, RjTRi, JrB);
    pvr_jac_rphi(pre, e9, RjTRi, JrB, J0, J1, J2, lda, ldb);
}
// EdgeNavStateBias::computeError (g2otypes.cpp:236-262)
PLBA_HD void bias_error(const double* si, const double* sj, double* e6) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        e6[c] = (sj[10 + c] + sj[16 + c]) - (si[10 + c] + si[16 + c]);
        e6[3 + c] = (sj[13 + c] + sj[19 + c]) - (si[13 + c] + si[19 + c]);
    }
}
// Prior dx for one kept PVR vertex (g2otypes.cpp:1462-1466): P - P0, V - V0, 2 vec(q0^-1 * Quaterniond(R(q)))
PLBA_HD void prior_dx_pvr(const double* s, const double* x0 /*10*/, double* dx9) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { dx9[c] = s[c] - x0[c]; dx9[3 + c] = s[3 + c] - x0[3 + c]; }
    Q4 q0; q0.x = x0[6]; q0.y = x0[7]; q0.z = x0[8]; q0.w = x0[9];
    double n2 = q0.x * q0.x + q0.y * q0.y + q0.z * q0.z + q0.w * q0.w;
    Q4 qi; qi.x = -q0.x / n2; qi.y = -q0.y / n2; qi.z = -q0.z / n2; qi.w = q0.w / n2;
    Q4 q; q.x = s[6]; q.y = s[7]; q.z = s[8]; q.w = s[9];
    Q4 qc = R_to_q(q_to_R(q));
    Q4 r = q_mul(qi, qc);
    dx9[6] = 2.0 * r.x; dx9[7] = 2.0 * r.y; dx9[8] = 2.0 * r.z;
}
PLBA_HD void prior_dx_bias(const double* s, const double* x0 /*6*/, double* dx6) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { dx6[c] = (s[10 + c] + s[16 + c]) - x0[c]; dx6[3 + c] = (s[13 + c] + s[19 + c]) - x0[3 + c]; }
}

// symmetric 3x3 inverse of (H + lambda I) by LDL^T (backward stable for the SPD blocks of the path; the cofactor
// form loses digits on poorly triangulated landmarks when lambda is small); H given as upper triangle
// [h00 h01 h02 h11 h12 h22], result in the same packing
// -------------------------------------------------------------------------------------------------
// IMU preintegration producer: one IMUPreintegrator::update step
// (IMU/IMUPreintegrator.cpp:80-139) on the 142-double payload
//   [dP3 dV3 dR9 JPg9 JPa9 JVg9 JVa9 JRg9 cov81 dt]   (IMU/IMUPreintegrator.h:187-201)
// omega / acc are bias-corrected; gcov / acov = diagonal value of IMUData::getGyrMeasCov / getAccMeasCov.
// The 9 x 9 covariance propagation cov' = A cov A^T + Bg Sg Bg^T + Ca Sa Ca^T uses the block structure
//   A = [I dt.I a13; 0 I a23; 0 0 a33],  a13 = -dR skew(acc) dt^2/2, a23 = -dR skew(acc) dt, a33 = Exp(w dt)^T
// with the reference's summation order inside every entry (540 multiply-adds instead of 1458).
// -------------------------------------------------------------------------------------------------
constexpr int PREINT_DOUBLES = 142;
PLBA_HD void st_m3(double* p, const M3& A) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
    for (int i = 0; i < 9; ++i) p[i] = A.a[i];
}
PLBA_HD void preint_reset(double* pre) {   // IMU/IMUPreintegrator.cpp:47-76
    for (int i = 0; i < PREINT_DOUBLES; ++i) pre[i] = 0.0;
    pre[6] = 1.0; pre[10] = 1.0; pre[14] = 1.0;
}
PLBA_HD void preint_update(double* pre, V3 omega, V3 acc, double dt, double gcov, double acov) {
    const double dt2 = dt * dt;
    const V3 wdt = dt * omega;
    const M3 dRk = (norm(wdt) < 1e-10) ? eye3() : q_to_R(so3_exp(wdt));     // Expmap, IMU/IMUPreintegrator.h:85-90
    const M3 Jr = so3_Jr(wdt);
    const M3 dR = ld_m3(pre + 6);
    const M3 RS = mul(dR, hat(acc));
    double* C = pre + 60;
    double a13[9], a23[9], a33[9];
    {
        const M3 dRkT = transpose(dRk);
        for (int i = 0; i < 9; ++i) { a13[i] = -0.5 * RS.a[i] * dt2; a23[i] = -RS.a[i] * dt; a33[i] = dRkT.a[i]; }
    }
    // cov <- (A cov) A^T, one row of T = A cov at a time; rows are overwritten only after every later row that needs
    // them has been formed (A is block upper triangular), rows 6-8 together
    double T[27];
    for (int r = 0; r < 6; ++r) {
        const int i = r % 3;
        const double* arow = (r < 3) ? a13 + 3 * i : a23 + 3 * i;
        for (int c = 0; c < 9; ++c) {
            double s = C[r * 9 + c];
            if (r < 3) s += dt * C[(3 + r) * 9 + c];
            s += arow[0] * C[54 + c];
            s += arow[1] * C[63 + c];
            s += arow[2] * C[72 + c];
            T[c] = s;
        }
        for (int c = 0; c < 3; ++c) {
            double s = T[c] + dt * T[3 + c];
            s += T[6] * a13[3 * c]; s += T[7] * a13[3 * c + 1]; s += T[8] * a13[3 * c + 2];
            C[r * 9 + c] = s;
            double u = T[3 + c];
            u += T[6] * a23[3 * c]; u += T[7] * a23[3 * c + 1]; u += T[8] * a23[3 * c + 2];
            C[r * 9 + 3 + c] = u;
            double v = T[6] * a33[3 * c];
            v += T[7] * a33[3 * c + 1]; v += T[8] * a33[3 * c + 2];
            C[r * 9 + 6 + c] = v;
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 9; ++c) {
            double s = a33[3 * i] * C[54 + c];
            s += a33[3 * i + 1] * C[63 + c];
            s += a33[3 * i + 2] * C[72 + c];
            T[i * 9 + c] = s;
        }
    for (int i = 0; i < 3; ++i) {
        const double* t = T + i * 9;
        for (int c = 0; c < 3; ++c) {
            double s = t[c] + dt * t[3 + c];
            s += t[6] * a13[3 * c]; s += t[7] * a13[3 * c + 1]; s += t[8] * a13[3 * c + 2];
            C[(6 + i) * 9 + c] = s;
            double u = t[3 + c];
            u += t[6] * a23[3 * c]; u += t[7] * a23[3 * c + 1]; u += t[8] * a23[3 * c + 2];
            C[(6 + i) * 9 + 3 + c] = u;
            double v = t[6] * a33[3 * c];
            v += t[7] * a33[3 * c + 1]; v += t[8] * a33[3 * c + 2];
            C[(6 + i) * 9 + 6 + c] = v;
        }
    }
    {   // + Bg Sg Bg^T (rotation block) + Ca Sa Ca^T (position / velocity blocks)
        const M3 JJ = mulABt(Jr, Jr), RR = mulABt(dR, dR);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                const double rr = RR.a[r * 3 + c];
                C[(6 + r) * 9 + 6 + c] += (dt * dt * gcov) * JJ.a[r * 3 + c];
                C[r * 9 + c] += (0.5 * dt2) * (0.5 * dt2) * acov * rr;
                C[r * 9 + 3 + c] += (0.5 * dt2) * dt * acov * rr;
                C[(3 + r) * 9 + c] += dt * (0.5 * dt2) * acov * rr;
                C[(3 + r) * 9 + 3 + c] += dt * dt * acov * rr;
            }
    }
    // Jacobians w.r.t. the biases: P first, then V, then R (cpp:104-110)
    const M3 JRg = ld_m3(pre + 51);
    const M3 RSJ = mul(RS, JRg);
    for (int i = 0; i < 9; ++i) {
        pre[24 + i] += pre[42 + i] * dt - 0.5 * dR.a[i] * dt2;       // JPa += JVa dt - dR dt^2/2
        pre[15 + i] += pre[33 + i] * dt - 0.5 * RSJ.a[i] * dt2;      // JPg += JVg dt - dR skew(a) JRg dt^2/2
    }
    for (int i = 0; i < 9; ++i) {
        pre[42 + i] += -dR.a[i] * dt;
        pre[33 + i] += -RSJ.a[i] * dt;
    }
    {
        const M3 t9 = mulAtB(dRk, JRg);
        for (int i = 0; i < 9; ++i) pre[51 + i] = t9.a[i] - Jr.a[i] * dt;
    }
    // delta measurements: P first, then V, then R (cpp:112-116)
    const V3 Ra = mul(dR, acc);
    pre[0] += pre[3] * dt + 0.5 * Ra.x * dt2; pre[1] += pre[4] * dt + 0.5 * Ra.y * dt2; pre[2] += pre[5] * dt + 0.5 * Ra.z * dt2;
    pre[3] += Ra.x * dt; pre[4] += Ra.y * dt; pre[5] += Ra.z * dt;
    {   // normalizeRotationM (IMU/IMUPreintegrator.h:163-178): through a w >= 0 unit quaternion
        Q4 q = R_to_q(mul(dR, dRk));
        if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
        st_m3(pre + 6, q_to_R(q_normalized(q)));
    }
    pre[141] += dt;
}

PLBA_HD bool sym3_inv(const double* h, double lambda, double* d /*6 upper*/) {
    const double a = h[0] + lambda, b = h[1], c = h[2], e = h[3] + lambda, f = h[4], g = h[5] + lambda;
    const double i0 = 1.0 / a;
    const double l10 = b * i0, l20 = c * i0;
    const double d1 = e - l10 * b;
    const double i1 = 1.0 / d1;
    const double l21 = (f - l20 * b) * i1;
    const double d2 = g - l20 * c - l21 * l21 * d1;
    const double i2 = 1.0 / d2;
    if (!(a != 0.0) || !(d1 != 0.0) || !(d2 != 0.0)) { d[0] = d[1] = d[2] = d[3] = d[4] = d[5] = 0.0; return false; }
    const double m10 = -l10, m21 = -l21, m20 = l10 * l21 - l20;      // L^-1 below the diagonal
    d[5] = i2;
    d[4] = m21 * i2;
    d[3] = i1 + m21 * m21 * i2;
    d[2] = m20 * i2;
    d[1] = m10 * i1 + m20 * m21 * i2;
    d[0] = i0 + m10 * m10 * i1 + m20 * m20 * i2;
    return true;
}
PLBA_HD V3 sym3_mul(const double* d, V3 v) {
    return v3(d[0] * v.x + d[1] * v.y + d[2] * v.z, d[1] * v.x + d[3] * v.y + d[4] * v.z, d[2] * v.x + d[4] * v.y + d[5] * v.z);
}

}  // namespace plba

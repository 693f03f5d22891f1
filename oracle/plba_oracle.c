/*
 * plba_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded, double-precision restatement of the reference's local-mapping
 * visual-inertial bundle adjustment hot path (HeadReaper-hc/PL-inertial-slam):
 *   IMU/so3.cpp, IMU/NavState.cpp, IMU/g2otypes.{h,cpp}, IMU/marginalization.{h,cpp},
 *   IMU/IMUPreintegrator.cpp and the call-site protocol of src/mapHandler.cpp:5741-6254,
 * plus the g2o semantics that path relies on (g2o is a third-party
 * dependency that is NOT vendored in the reference and whose version is unpinned:
 * find_package(G2O REQUIRED), reference CMakeLists.txt:14).
 *
 * PARITY UNPINNED: the reference ships no test, fixture or golden vector for this path
 * (test/test.cpp has no assertions and is not built) and cannot be compiled here (Eigen, g2o,
 * OpenCV, Boost ... absent).  The only known answer derivable from the reference's
 * own files is test/test.cpp's scenario (error = (603,0,0)), checked in tests/test_oracle_math.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It exports the same entry points as include/plba.h with the prefix orc_ instead of plba_.
 *
 * Every function cites the reference file:line it follows.  Eigen expression evaluation order is
 * followed where it is visible in the source (left-to-right matrix products, explicit
 * temporaries); Eigen's own kernels (quaternion<->matrix, normalisation, PartialPivLU inverse,
 * SelfAdjointEigenSolver) are restated from their published algorithms.
 */
#include "../include/plba.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define SMALL_EPS 1e-10 /* IMU/so3.h:35 */

/* ============================================================================================
 * 1. small dense helpers (row-major)
 * ========================================================================================== */
static void mat_mul(const double* A, const double* B, double* C, int r, int k, int c) {
    /* C(r x c) = A(r x k) * B(k x c) */
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) {
            double s = 0.0;
            for (int t = 0; t < k; ++t) s += A[i * k + t] * B[t * c + j];
            C[i * c + j] = s;
        }
}
static void mat_T(const double* A, double* At, int r, int c) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) At[j * r + i] = A[i * c + j];
}
static void m3_mul(const double* A, const double* B, double* C) { mat_mul(A, B, C, 3, 3, 3); }
static void m3_T(const double* A, double* At) { mat_T(A, At, 3, 3); }
static void m3_v(const double* A, const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
static double v3_norm(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* IMU/so3.cpp:283-290 (== skew(), IMU/se3_ops.hpp:27-38) */
static void so3_hat(const double* v, double* O) {
    O[0] = 0;     O[1] = -v[2]; O[2] = v[1];
    O[3] = v[2];  O[4] = 0;     O[5] = -v[0];
    O[6] = -v[1]; O[7] = v[0];  O[8] = 0;
}

/* ============================================================================================
 * 2. quaternion / SO3 (Sophus copy vendored by the reference, IMU/so3.cpp) on top of Eigen's
 *    Quaterniond semantics.  Storage order (x,y,z,w) like Eigen's coeffs().
 * ========================================================================================== */
static void q_normalize(double* q) { /* Eigen: coeffs() /= norm() */
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void q_mul(const double* a, const double* b, double* o) { /* Eigen quat product a*b */
    double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static void q_to_R(const double* q, double* R) { /* Eigen QuaternionBase::toRotationMatrix */
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
static void R_to_q(const double* m, double* q) { /* Eigen quaternionbase_assign_impl<Mat,3,3> */
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}
static void q_rotate(const double* q, const double* v, double* o) { /* Eigen _transformVector */
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
/* SO3 copy-ctor normalises, IMU/so3.cpp:108-113 */
static void so3_copy(const double* a, double* o) { memcpy(o, a, 32); q_normalize(o); }
/* SO3(Matrix3d), IMU/so3.cpp:115-119 */
static void so3_from_R(const double* R, double* o) { R_to_q(R, o); q_normalize(o); }
/* SO3::operator*, IMU/so3.cpp:142-149 */
static void so3_mul(const double* a, const double* b, double* o) {
    double r[4], t[4];
    so3_copy(a, r);
    q_mul(r, b, t);
    q_normalize(t);
    memcpy(o, t, 32);
}
/* SO3::inverse, IMU/so3.cpp:164-168 (ctor from quaternion normalises, :121-126) */
static void so3_inverse(const double* a, double* o) {
    o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3];
    q_normalize(o);
}
/* SO3::expAndTheta, IMU/so3.cpp:257-280 */
static void so3_exp(const double* w, double* o) {
    double theta = v3_norm(w);
    double half = 0.5 * theta;
    double imag, real = cos(half);
    if (theta < SMALL_EPS) {
        double t2 = theta * theta, t4 = t2 * t2;
        imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4;
    } else {
        imag = sin(half) / theta;
    }
    o[0] = imag * w[0]; o[1] = imag * w[1]; o[2] = imag * w[2]; o[3] = real;
    q_normalize(o);
}
/* SO3::logAndTheta, IMU/so3.cpp:206-247 — the fabs(w)<eps branch is overwritten unconditionally
 * by 2*atan(n/w)/n exactly as in the reference (SURVEY B-Q13). */
static void so3_log(const double* q, double* o) {
    double n = v3_norm(q);
    double w = q[3];
    double sw = w * w;
    double f;
    if (n < SMALL_EPS) {
        f = 2. / w - 2. * (n * n) / (w * sw);
    } else {
        f = 2 * atan(n / w) / n;
    }
    o[0] = f * q[0]; o[1] = f * q[1]; o[2] = f * q[2];
}
/* SO3::JacobianR, IMU/so3.cpp:32-49 */
static void so3_Jr(const double* w, double* J) {
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double theta = v3_norm(w);
    memcpy(J, I, 72);
    if (theta < 0.00001) return;
    double k[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    double K[9], KK[9];
    so3_hat(k, K);
    m3_mul(K, K, KK);
    double a = (1 - cos(theta)) / theta, b = (1 - sin(theta) / theta);
    for (int i = 0; i < 9; ++i) J[i] = I[i] - a * K[i] + b * KK[i];
}
/* SO3::JacobianRInv, IMU/so3.cpp:50-68 */
static void so3_JrInv(const double* w, double* J) {
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double theta = v3_norm(w);
    memcpy(J, I, 72);
    if (theta < 0.00001) return;
    double k[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    double K[9], KK[9], W[9];
    so3_hat(k, K);
    m3_mul(K, K, KK);
    so3_hat(w, W);
    double c = (1.0 - (1.0 + cos(theta)) * theta / (2.0 * sin(theta)));
    for (int i = 0; i < 9; ++i) J[i] = I[i] + 0.5 * W[i] + c * KK[i];
}

/* ============================================================================================
 * 3. NavState (IMU/NavState.h:69-80) and its increments (IMU/NavState.cpp:69-121)
 * ========================================================================================== */
typedef struct {
    double P[3], V[3], q[4], bg[3], ba[3], dbg[3], dba[3];
} nav_t;

static void nav_RotMatrix(const nav_t* s, double* R) { q_to_R(s->q, R); } /* Get_RotMatrix h:25 */

static void nav_IncSmallPVR(nav_t* s, const double* u) { /* IMU/NavState.cpp:69-98 */
    double qc[4], R[9], d[3], dR[4], qn[4];
    so3_copy(s->q, qc); /* Get_R() returns a copy */
    q_to_R(qc, R);
    m3_v(R, u, d);
    s->P[0] += d[0]; s->P[1] += d[1]; s->P[2] += d[2];
    s->V[0] += u[3]; s->V[1] += u[4]; s->V[2] += u[5];
    so3_exp(u + 6, dR);
    so3_copy(s->q, qc);
    so3_mul(qc, dR, qn);
    memcpy(s->q, qn, 32);
}
static void nav_IncSmallBias(nav_t* s, const double* u) { /* IMU/NavState.cpp:100-121 */
    for (int i = 0; i < 3; ++i) { s->dbg[i] += u[i]; s->dba[i] += u[3 + i]; }
}

/* ============================================================================================
 * 4. problem container
 * ========================================================================================== */
typedef struct {
    double dP[3], dV[3], dR[9], JPg[9], JPa[9], JVg[9], JVa[9], JRg[9], cov[81], dt;
} preint_t; /* IMU/IMUPreintegrator.h:187-201, 142 doubles */

struct plba_problem {
    plba_options opt;
    char err[256];
    /* camera / gravity */
    double fx, fy, cx, cy, Rbc[9], Pbc[3], gw[3];
    int have_cam;
    /* keyframes */
    int K;
    int32_t *vid_pvr, *vid_bias;
    nav_t *ns, *ns_bak, *ns_saved;
    uint8_t *fix_pvr, *fix_bias;
    /* landmarks */
    int Np, Nl;
    double *pt, *pt_bak, *pt_saved;
    double *ln, *ln_bak, *ln_saved;
    uint8_t *pt_fixed, *ln_fixed;
    /* observations */
    int Ep, El;
    int32_t *po_pt, *po_kf, *lo_ln, *lo_kf;
    double *po_uv, *po_w, *lo_l, *lo_w;
    uint8_t *po_level, *lo_level;
    double *po_err; /* Ep*2, cached _error */
    double *lo_err; /* El*3 */
    /* IMU edges */
    int M;
    int32_t *im_i, *im_j;
    preint_t* im_pre;
    double *im_info_pvr, *im_info_bias;
    double *im_err_pvr; /* M*9 */
    double *im_err_bias; /* M*6 */
    /* prior */
    int pr_n, pr_nv;
    int32_t *pr_vid, *pr_size, *pr_idx;
    double *pr_x0; /* packed */
    int* pr_x0_off;
    double *pr_J0, *pr_r0, *pr_err;
    /* robust */
    int rob_on[5];
    double rob_delta[5];
    /* index mapping (rebuilt per optimize) */
    int P;                /* pose-side dimension */
    int *off_pvr, *off_bias; /* per keyframe hessian offset or -1 */
    uint8_t *pt_active, *ln_active;
    int *pt_xoff, *ln_xoff;  /* offsets into the landmark part of x, or -1 */
    int Ldim;
    /* linear system */
    double *Hpp, *bp, *bpg, *Hs, *bs, *x; /* bp = local gradient, bpg = global (all-reduced) */
    double *Hll_pt, *bl_pt, *Hll_ln, *bl_ln;
    double *Hpl_pt; /* Ep * 27 (9x3) */
    double *Hpl_ln; /* El * 54 (9x6) */
    double *Dinv_pt, *Dinv_ln;
    double *bl_all; /* Ldim */
    double chi2_last, maxdiag_last;
    /* trace */
    plba_trace_row* trace;
    int trace_n, trace_cap;
    /* shard */
    int rank, world;
    plba_allreduce_fn xfn;
    void* xuser;
};
typedef struct plba_problem prob_t;

static char g_create_err[256];

#define FAIL(p, code, ...)                                   \
    do {                                                     \
        snprintf((p)->err, sizeof((p)->err), __VA_ARGS__);   \
        return (code);                                       \
    } while (0)

static void* xcalloc(size_t n, size_t s) {
    void* r = calloc(n ? n : 1, s);
    if (!r) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return r;
}
static void* xdup(const void* src, size_t bytes) {
    void* r = xcalloc(bytes ? bytes : 1, 1);
    if (src && bytes) memcpy(r, src, bytes);
    return r;
}
#define REPL(field, src, bytes) do { free(field); field = xdup(src, bytes); } while (0)

void orc_default_options(plba_options* o) {
    memset(o, 0, sizeof(*o));
    o->tau = 1e-5;                 /* g2o OptimizationAlgorithmLevenberg _tau */
    o->good_step_lower = 1. / 3.;  /* _goodStepLowerScale */
    o->good_step_upper = 2. / 3.;  /* _goodStepUpperScale */
    o->max_trials = 10;            /* _maxTrialsAfterFailure */
    o->user_lambda_init = 0.0;
    o->marg_eps = 1e-8;            /* IMU/marginalization.h:99 */
    o->device = -1;
    o->use_mfma = 1;
    o->factor_block = 32;
    o->factor_flow = 0;
    o->chain_elim = 1;             /* solver-structure options of the HIP library: ignored here */
    o->wide_steps = 0;
    o->band_solve = 1;
    o->marg_exact = 1;             /* the oracle always takes the dense eigen pseudo-inverse (cpp:351-353) */
    o->lm_fused = 1;
    o->lm_fused_min_obs = 40000;
}
const char* orc_backend_name(void) { return "cpu-oracle"; }

int orc_create(const plba_options* opt, plba_problem** out) {
    if (!out) { snprintf(g_create_err, sizeof g_create_err, "out is NULL"); return PLBA_ERR_INVALID; }
    prob_t* p = (prob_t*)xcalloc(1, sizeof(prob_t));
    if (opt) p->opt = *opt; else orc_default_options(&p->opt);
    p->world = 1;
    *out = p;
    return PLBA_OK;
}
void orc_destroy(plba_problem* p) {
    if (!p) return;
    free(p->vid_pvr); free(p->vid_bias); free(p->ns); free(p->ns_bak); free(p->ns_saved);
    free(p->fix_pvr); free(p->fix_bias);
    free(p->pt); free(p->pt_bak); free(p->pt_saved); free(p->ln); free(p->ln_bak); free(p->ln_saved);
    free(p->pt_fixed); free(p->ln_fixed);
    free(p->po_pt); free(p->po_kf); free(p->lo_ln); free(p->lo_kf);
    free(p->po_uv); free(p->po_w); free(p->lo_l); free(p->lo_w);
    free(p->po_level); free(p->lo_level); free(p->po_err); free(p->lo_err);
    free(p->im_i); free(p->im_j); free(p->im_pre); free(p->im_info_pvr); free(p->im_info_bias);
    free(p->im_err_pvr); free(p->im_err_bias);
    free(p->pr_vid); free(p->pr_size); free(p->pr_idx); free(p->pr_x0); free(p->pr_x0_off);
    free(p->pr_J0); free(p->pr_r0); free(p->pr_err);
    free(p->off_pvr); free(p->off_bias); free(p->pt_active); free(p->ln_active);
    free(p->pt_xoff); free(p->ln_xoff);
    free(p->Hpp); free(p->bp); free(p->bpg); free(p->Hs); free(p->bs); free(p->x);
    free(p->Hll_pt); free(p->bl_pt); free(p->Hll_ln); free(p->bl_ln);
    free(p->Hpl_pt); free(p->Hpl_ln); free(p->Dinv_pt); free(p->Dinv_ln); free(p->bl_all);
    free(p->trace);
    free(p);
}
const char* orc_last_error(const plba_problem* p) { return p ? p->err : g_create_err; }

static int all_finite(const double* a, size_t n) {
    for (size_t i = 0; i < n; ++i) if (!isfinite(a[i])) return 0;
    return 1;
}

int orc_set_camera(plba_problem* p, double fx, double fy, double cx, double cy, const double* Rbc, const double* Pbc) {
    if (!p || !Rbc || !Pbc) return PLBA_ERR_INVALID;
    p->fx = fx; p->fy = fy; p->cx = cx; p->cy = cy;
    memcpy(p->Rbc, Rbc, 72); memcpy(p->Pbc, Pbc, 24);
    p->have_cam = 1;
    return PLBA_OK;
}
int orc_set_gravity(plba_problem* p, const double* gw) {
    if (!p || !gw) return PLBA_ERR_INVALID;
    memcpy(p->gw, gw, 24);
    return PLBA_OK;
}
int orc_set_keyframes(plba_problem* p, int K, const int32_t* vid_pvr, const int32_t* vid_bias, const double* P3,
                      const double* V3, const double* q4, const double* bg3, const double* ba3, const double* dbg3,
                      const double* dba3, const uint8_t* fixed_pvr, const uint8_t* fixed_bias) {
    if (!p || K <= 0 || !vid_pvr || !P3 || !V3 || !q4) return PLBA_ERR_INVALID;
    for (int k = 1; k < K; ++k)
        if (vid_pvr[k] <= vid_pvr[k - 1]) FAIL(p, PLBA_ERR_INVALID, "keyframe vertex ids must be ascending");
    if (!all_finite(P3, 3 * (size_t)K) || !all_finite(q4, 4 * (size_t)K)) FAIL(p, PLBA_ERR_NUMERIC, "non-finite keyframe state");
    p->K = K;
    REPL(p->vid_pvr, vid_pvr, sizeof(int32_t) * K);
    free(p->vid_bias);
    p->vid_bias = (int32_t*)xcalloc(K, sizeof(int32_t));
    free(p->ns); p->ns = (nav_t*)xcalloc(K, sizeof(nav_t));
    free(p->ns_bak); p->ns_bak = (nav_t*)xcalloc(K, sizeof(nav_t));
    free(p->ns_saved); p->ns_saved = (nav_t*)xcalloc(K, sizeof(nav_t));
    free(p->fix_pvr); p->fix_pvr = (uint8_t*)xcalloc(K, 1);
    free(p->fix_bias); p->fix_bias = (uint8_t*)xcalloc(K, 1);
    for (int k = 0; k < K; ++k) {
        nav_t* s = &p->ns[k];
        p->vid_bias[k] = vid_bias ? vid_bias[k] : -1;
        memcpy(s->P, P3 + 3 * k, 24); memcpy(s->V, V3 + 3 * k, 24); memcpy(s->q, q4 + 4 * k, 32);
        if (bg3) memcpy(s->bg, bg3 + 3 * k, 24);
        if (ba3) memcpy(s->ba, ba3 + 3 * k, 24);
        if (dbg3) memcpy(s->dbg, dbg3 + 3 * k, 24);
        if (dba3) memcpy(s->dba, dba3 + 3 * k, 24);
        p->fix_pvr[k] = fixed_pvr ? fixed_pvr[k] : 0;
        p->fix_bias[k] = fixed_bias ? fixed_bias[k] : 0;
    }
    return PLBA_OK;
}
int orc_set_points(plba_problem* p, int Np, const double* xyz, const uint8_t* fixed) {
    if (!p || Np < 0 || (Np && !xyz)) return PLBA_ERR_INVALID;
    if (!all_finite(xyz, 3 * (size_t)Np)) FAIL(p, PLBA_ERR_NUMERIC, "non-finite point");
    p->Np = Np;
    REPL(p->pt, xyz, 24 * (size_t)Np);
    REPL(p->pt_bak, xyz, 24 * (size_t)Np);
    REPL(p->pt_saved, xyz, 24 * (size_t)Np);
    free(p->pt_fixed); p->pt_fixed = (uint8_t*)xcalloc(Np, 1);
    if (fixed) memcpy(p->pt_fixed, fixed, Np);
    return PLBA_OK;
}
int orc_set_lines(plba_problem* p, int Nl, const double* sPeP, const uint8_t* fixed) {
    if (!p || Nl < 0 || (Nl && !sPeP)) return PLBA_ERR_INVALID;
    if (!all_finite(sPeP, 6 * (size_t)Nl)) FAIL(p, PLBA_ERR_NUMERIC, "non-finite line");
    p->Nl = Nl;
    REPL(p->ln, sPeP, 48 * (size_t)Nl);
    REPL(p->ln_bak, sPeP, 48 * (size_t)Nl);
    REPL(p->ln_saved, sPeP, 48 * (size_t)Nl);
    free(p->ln_fixed); p->ln_fixed = (uint8_t*)xcalloc(Nl, 1);
    if (fixed) memcpy(p->ln_fixed, fixed, Nl);
    return PLBA_OK;
}
static int check_obs(prob_t* p, int E, const int32_t* lm, const int32_t* kf, int Nlm) {
    for (int e = 0; e < E; ++e) {
        if (lm[e] < 0 || lm[e] >= Nlm) FAIL(p, PLBA_ERR_INVALID, "observation %d: landmark index out of range", e);
        if (kf[e] < 0 || kf[e] >= p->K) FAIL(p, PLBA_ERR_INVALID, "observation %d: keyframe index out of range", e);
        if (e && lm[e] < lm[e - 1]) FAIL(p, PLBA_ERR_INVALID, "observations must be landmark-major (sorted by landmark)");
    }
    /* one observation per (landmark, keyframe) — the reference's data model (kf_obs_list) */
    for (int e = 0; e < E; ++e)
        for (int f = e + 1; f < E && lm[f] == lm[e]; ++f)
            if (kf[f] == kf[e]) FAIL(p, PLBA_ERR_INVALID, "landmark %d observed twice by keyframe %d", lm[e], kf[e]);
    return PLBA_OK;
}
int orc_set_point_obs(plba_problem* p, int Ep, const int32_t* pt, const int32_t* kf, const double* uv, const double* w) {
    if (!p || Ep < 0 || (Ep && (!pt || !kf || !uv))) return PLBA_ERR_INVALID;
    if (!p->K) FAIL(p, PLBA_ERR_STATE, "set_keyframes first");
    int rc = check_obs(p, Ep, pt, kf, p->Np);
    if (rc) return rc;
    p->Ep = Ep;
    REPL(p->po_pt, pt, 4 * (size_t)Ep); REPL(p->po_kf, kf, 4 * (size_t)Ep);
    REPL(p->po_uv, uv, 16 * (size_t)Ep);
    free(p->po_w); p->po_w = (double*)xcalloc(Ep, 8);
    /* const float& invSigma2 = 1.0/sigma: information rounded to float (mapHandler.cpp:5340, B-Q12) */
    for (int e = 0; e < Ep; ++e) p->po_w[e] = (double)(float)(w ? w[e] : 1.0);
    free(p->po_level); p->po_level = (uint8_t*)xcalloc(Ep, 1);
    free(p->po_err); p->po_err = (double*)xcalloc(Ep, 16);
    return PLBA_OK;
}
int orc_set_line_obs(plba_problem* p, int El, const int32_t* ln, const int32_t* kf, const double* l3, const double* w) {
    if (!p || El < 0 || (El && (!ln || !kf || !l3))) return PLBA_ERR_INVALID;
    if (!p->K) FAIL(p, PLBA_ERR_STATE, "set_keyframes first");
    int rc = check_obs(p, El, ln, kf, p->Nl);
    if (rc) return rc;
    p->El = El;
    REPL(p->lo_ln, ln, 4 * (size_t)El); REPL(p->lo_kf, kf, 4 * (size_t)El);
    REPL(p->lo_l, l3, 24 * (size_t)El);
    free(p->lo_w); p->lo_w = (double*)xcalloc(El, 8);
    for (int e = 0; e < El; ++e) p->lo_w[e] = (double)(float)(w ? w[e] : 1.0); /* mapHandler.cpp:5398 */
    free(p->lo_level); p->lo_level = (uint8_t*)xcalloc(El, 1);
    free(p->lo_err); p->lo_err = (double*)xcalloc(El, 24);
    return PLBA_OK;
}
int orc_set_imu_edges(plba_problem* p, int M, const int32_t* ki, const int32_t* kj, const double* pre, const double* ipvr,
                      const double* ibias) {
    if (!p || M < 0 || (M && (!ki || !kj || !pre || !ipvr || !ibias))) return PLBA_ERR_INVALID;
    for (int m = 0; m < M; ++m) {
        if (ki[m] < 0 || ki[m] >= p->K || kj[m] < 0 || kj[m] >= p->K) FAIL(p, PLBA_ERR_INVALID, "imu edge %d: keyframe index", m);
        if (p->vid_bias[ki[m]] < 0 || p->vid_bias[kj[m]] < 0) FAIL(p, PLBA_ERR_INVALID, "imu edge %d: keyframe without bias vertex", m);
    }
    p->M = M;
    REPL(p->im_i, ki, 4 * (size_t)M); REPL(p->im_j, kj, 4 * (size_t)M);
    free(p->im_pre); p->im_pre = (preint_t*)xcalloc(M, sizeof(preint_t));
    for (int m = 0; m < M; ++m) memcpy(&p->im_pre[m], pre + 142 * (size_t)m, 142 * 8);
    REPL(p->im_info_pvr, ipvr, 81 * 8 * (size_t)M); REPL(p->im_info_bias, ibias, 36 * 8 * (size_t)M);
    free(p->im_err_pvr); p->im_err_pvr = (double*)xcalloc(M, 72);
    free(p->im_err_bias); p->im_err_bias = (double*)xcalloc(M, 48);
    return PLBA_OK;
}
int orc_set_prior(plba_problem* p, int n, int nv, const int32_t* vid, const int32_t* size, const int32_t* idx, const double* x0,
                  const double* J0, const double* r0) {
    if (!p) return PLBA_ERR_INVALID;
    if (nv == 0) { p->pr_n = 0; p->pr_nv = 0; return PLBA_OK; }
    if (n <= 0 || nv < 0 || !vid || !size || !idx || !x0 || !J0 || !r0) return PLBA_ERR_INVALID;
    int tot = 0, xoff = 0;
    free(p->pr_x0_off); p->pr_x0_off = (int*)xcalloc(nv, sizeof(int));
    for (int i = 0; i < nv; ++i) {
        if (size[i] != 9 && size[i] != 6) FAIL(p, PLBA_ERR_INVALID, "Undefined size of marginalization vertex: %d", size[i]); /* g2otypes.h:1071 */
        if (idx[i] < 0 || idx[i] + size[i] > n) FAIL(p, PLBA_ERR_INVALID, "prior vertex %d: idx out of range", i);
        p->pr_x0_off[i] = xoff;
        xoff += (size[i] == 9) ? 10 : 6;
        tot += size[i];
    }
    if (tot != n) FAIL(p, PLBA_ERR_INVALID, "prior: sum of kept sizes %d != n %d", tot, n);
    p->pr_n = n; p->pr_nv = nv;
    REPL(p->pr_vid, vid, 4 * (size_t)nv); REPL(p->pr_size, size, 4 * (size_t)nv); REPL(p->pr_idx, idx, 4 * (size_t)nv);
    REPL(p->pr_x0, x0, 8 * (size_t)xoff);
    REPL(p->pr_J0, J0, 8 * (size_t)n * n); REPL(p->pr_r0, r0, 8 * (size_t)n);
    free(p->pr_err); p->pr_err = (double*)xcalloc(n, 8);
    return PLBA_OK;
}
int orc_set_robust(plba_problem* p, plba_edge_kind kind, int enabled, double delta) {
    if (!p || kind < 0 || kind > 4) return PLBA_ERR_INVALID;
    p->rob_on[kind] = enabled;
    p->rob_delta[kind] = delta;
    return PLBA_OK;
}
int orc_set_marg_eps(plba_problem* p, double eps) {      /* MarginalizationInfo::eps, IMU/marginalization.h:99 */
    if (!p || !(eps >= 0.0) || !isfinite(eps)) return PLBA_ERR_INVALID;
    p->opt.marg_eps = eps;
    return PLBA_OK;
}
int orc_set_levels(plba_problem* p, plba_edge_kind kind, const uint8_t* level) {
    if (!p || !level) return PLBA_ERR_INVALID;
    if (kind == PLBA_EDGE_POINT) memcpy(p->po_level, level, p->Ep);
    else if (kind == PLBA_EDGE_LINE) memcpy(p->lo_level, level, p->El);
    else FAIL(p, PLBA_ERR_INVALID, "levels only for point/line edges");
    return PLBA_OK;
}
int orc_get_levels(plba_problem* p, plba_edge_kind kind, uint8_t* level) {
    if (!p || !level) return PLBA_ERR_INVALID;
    if (kind == PLBA_EDGE_POINT) memcpy(level, p->po_level, p->Ep);
    else if (kind == PLBA_EDGE_LINE) memcpy(level, p->lo_level, p->El);
    else return PLBA_ERR_INVALID;
    return PLBA_OK;
}
int orc_set_shard(plba_problem* p, int rank, int world, plba_allreduce_fn fn, void* user) {
    if (!p || world < 1 || rank < 0 || rank >= world) return PLBA_ERR_INVALID;
    p->rank = rank; p->world = world; p->xfn = fn; p->xuser = user;
    return PLBA_OK;
}
int orc_set_stream(plba_problem* p, void* s) { (void)p; (void)s; return PLBA_OK; }

/* ============================================================================================
 * 5. edges: computeError / linearizeOplus restatements
 * ========================================================================================== */

/* EdgeNavStatePVRPointXYZ::computePc, IMU/g2otypes.h:243-260 */
static void cam_Pc(const prob_t* p, const nav_t* s, const double* Pw, double* Pc, double* RcbRwbT /*opt*/) {
    double Rwb[9], RwbT[9], Rcb[9], M[9], d[3], a[3], b[3];
    nav_RotMatrix(s, Rwb);
    m3_T(Rwb, RwbT);
    m3_T(p->Rbc, Rcb);
    m3_mul(Rcb, RwbT, M); /* Rcb * Rwb.transpose() evaluated first (left-to-right) */
    d[0] = Pw[0] - s->P[0]; d[1] = Pw[1] - s->P[1]; d[2] = Pw[2] - s->P[2];
    m3_v(M, d, a);
    m3_v(Rcb, p->Pbc, b);
    Pc[0] = a[0] - b[0]; Pc[1] = a[1] - b[1]; Pc[2] = a[2] - b[2];
    if (RcbRwbT) memcpy(RcbRwbT, M, 72);
}
/* cam_project, IMU/g2otypes.h:262-275 */
static void cam_project(const prob_t* p, const double* Pc, double* uv) {
    double px = Pc[0] / Pc[2], py = Pc[1] / Pc[2];
    uv[0] = px * p->fx + p->cx;
    uv[1] = py * p->fy + p->cy;
}
/* EdgeNavStatePVRPointXYZ::computeError, IMU/g2otypes.h:230-236 */
static void point_error(const prob_t* p, const nav_t* s, const double* Pw, const double* obs, double* err, int* depth_pos) {
    double Pc[3], uv[2];
    cam_Pc(p, s, Pw, Pc, NULL);
    cam_project(p, Pc, uv);
    err[0] = obs[0] - uv[0];
    err[1] = obs[1] - uv[1];
    if (depth_pos) *depth_pos = Pc[2] > 0.0; /* isDepthPositive h:238-241 */
}
/* EdgeNavStatePVRPointXYZ::linearizeOplus, IMU/g2otypes.cpp:286-341.  Ji 2x3, Jj 2x9 */
static void point_linearize(const prob_t* p, const nav_t* s, const double* Pw, double* Ji, double* Jj) {
    double Pc[3], M[9], Rcb[9];
    cam_Pc(p, s, Pw, Pc, M);
    m3_T(p->Rbc, Rcb);
    double x = Pc[0], y = Pc[1], z = Pc[2];
    double Maux[6] = {p->fx, 0, -x / z * p->fx, 0, p->fy, -y / z * p->fy};
    double Jpi[6];
    for (int i = 0; i < 6; ++i) Jpi[i] = Maux[i] / z;
    /* _jacobianOplusXi = - Jpi * Rcb * Rwb.transpose()  : (-Jpi*Rcb)*RwbT, left to right */
    double Rwb[9], RwbT[9], nJpi[6], t[6];
    nav_RotMatrix(s, Rwb);
    m3_T(Rwb, RwbT);
    for (int i = 0; i < 6; ++i) nJpi[i] = -Jpi[i];
    mat_mul(nJpi, Rcb, t, 2, 3, 3);
    mat_mul(t, RwbT, Ji, 2, 3, 3);
    /* JdPwb = - Jpi * (-Rcb) */
    double nRcb[9], JdP[6];
    for (int i = 0; i < 9; ++i) nRcb[i] = -Rcb[i];
    mat_mul(nJpi, nRcb, JdP, 2, 3, 3);
    /* Paux = Rcb*Rwb.transpose()*(Pw-Pwb); JdRwb = - Jpi * (hat(Paux) * Rcb) */
    double d[3] = {Pw[0] - s->P[0], Pw[1] - s->P[1], Pw[2] - s->P[2]}, Paux[3], H[9], HR[9], JdR[6];
    m3_v(M, d, Paux);
    so3_hat(Paux, H);
    m3_mul(H, Rcb, HR);
    mat_mul(nJpi, HR, JdR, 2, 3, 3);
    memset(Jj, 0, 18 * 8);
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) {
            Jj[r * 9 + c] = JdP[r * 3 + c];
            Jj[r * 9 + 6 + c] = JdR[r * 3 + c];
        }
}
/* EdgeNavStateLine::computeError, IMU/g2otypes.h:783-793 (3-dim, e2 == 0) */
static void line_error(const prob_t* p, const nav_t* s, const double* L, const double* obs, double* err, int* depth_pos) {
    double Ps[3], Pe[3], us[2], ue[2];
    cam_Pc(p, s, L, Ps, NULL);
    cam_Pc(p, s, L + 3, Pe, NULL);
    cam_project(p, Ps, us);
    cam_project(p, Pe, ue);
    err[0] = obs[0] * us[0] + obs[1] * us[1] + obs[2];
    err[1] = obs[0] * ue[0] + obs[1] * ue[1] + obs[2];
    err[2] = 0;
    if (depth_pos) *depth_pos = (Ps[2] > 0.0 && Pe[2] > 0.0); /* h:795-798 */
}
/* EdgeNavStateLine::linearizeOplus, IMU/g2otypes.cpp:1306-1359.  Ji 3x6, Jj 3x9 */
static void line_linearize(const prob_t* p, const nav_t* s, const double* L, const double* obs, double* Ji, double* Jj) {
    double Ps[3], Pe[3], M[9], Rcb[9], Rwb[9], RwbT[9];
    cam_Pc(p, s, L, Ps, M);
    cam_Pc(p, s, L + 3, Pe, NULL);
    m3_T(p->Rbc, Rcb);
    nav_RotMatrix(s, Rwb);
    m3_T(Rwb, RwbT);
    double de_p[2] = {obs[0], obs[1]};
    double dps[6] = {p->fx / Ps[2], 0, -p->fx * Ps[0] / (Ps[2] * Ps[2]), 0, p->fy / Ps[2], -p->fy * Ps[1] / (Ps[2] * Ps[2])};
    double dpe[6] = {p->fx / Pe[2], 0, -p->fx * Pe[0] / (Pe[2] * Pe[2]), 0, p->fy / Pe[2], -p->fy * Pe[1] / (Pe[2] * Pe[2])};
    double rs[3], re[3]; /* de_p * dps_Ps (1x3) */
    mat_mul(de_p, dps, rs, 1, 2, 3);
    mat_mul(de_p, dpe, re, 1, 2, 3);
    double a[3], b[3];
    mat_mul(rs, M, a, 1, 3, 3); /* * dPs_l block = Rcb*RwbT */
    mat_mul(re, M, b, 1, 3, 3);
    memset(Ji, 0, 18 * 8);
    for (int c = 0; c < 3; ++c) { Ji[0 * 6 + c] = a[c]; Ji[1 * 6 + 3 + c] = b[c]; }
    /* de0_Pwb = de_p*dps_Ps*(-Rcb*RwbT): world-frame dp (SURVEY B-Q1), reproduced unless option set */
    double nRcb[9], nM[9], p0[3], p1[3];
    for (int i = 0; i < 9; ++i) nRcb[i] = -Rcb[i];
    m3_mul(nRcb, RwbT, nM);
    mat_mul(rs, nM, p0, 1, 3, 3);
    mat_mul(re, nM, p1, 1, 3, 3);
    if (p->opt.fix_line_position_jacobian) { /* consistent with P += R*dp : d e/d dp = -(de_p*dps)*Rcb */
        mat_mul(rs, nRcb, p0, 1, 3, 3);
        mat_mul(re, nRcb, p1, 1, 3, 3);
    }
    double ds[3] = {L[0] - s->P[0], L[1] - s->P[1], L[2] - s->P[2]};
    double de[3] = {L[3] - s->P[0], L[4] - s->P[1], L[5] - s->P[2]};
    double vs[3], ve[3], Ss[9], Se[9], RS[9], RE[9], f0[3], f1[3];
    m3_v(RwbT, ds, vs);
    m3_v(RwbT, de, ve);
    so3_hat(vs, Ss);
    so3_hat(ve, Se);
    m3_mul(Rcb, Ss, RS);
    m3_mul(Rcb, Se, RE);
    mat_mul(rs, RS, f0, 1, 3, 3);
    mat_mul(re, RE, f1, 1, 3, 3);
    memset(Jj, 0, 27 * 8);
    for (int c = 0; c < 3; ++c) {
        Jj[0 * 9 + c] = p0[c]; Jj[1 * 9 + c] = p1[c];
        Jj[0 * 9 + 6 + c] = f0[c]; Jj[1 * 9 + 6 + c] = f1[c];
    }
}
/* EdgeNavStatePVR::computeError, IMU/g2otypes.cpp:27-92.  si,sj PVR states; sb = bias vertex of i */
static void pvr_error(const prob_t* p, const nav_t* si, const nav_t* sj, const nav_t* sb, const preint_t* M, double* err) {
    double Ri[4], Rj[4], dRij[4], RiT[4];
    so3_copy(si->q, Ri); /* Get_R() */
    so3_copy(sj->q, Rj);
    double dT = M->dt, dT2 = dT * dT;
    so3_from_R(M->dR, dRij);
    so3_inverse(Ri, RiT);
    double a[3], ra[3], t1[3], t2[3];
    for (int i = 0; i < 3; ++i) a[i] = sj->P[i] - si->P[i] - si->V[i] * dT - 0.5 * p->gw[i] * dT2;
    q_rotate(RiT, a, ra);
    m3_v(M->JPg, sb->dbg, t1);
    m3_v(M->JPa, sb->dba, t2);
    for (int i = 0; i < 3; ++i) err[i] = ra[i] - (M->dP[i] + t1[i] + t2[i]);
    for (int i = 0; i < 3; ++i) a[i] = sj->V[i] - si->V[i] - p->gw[i] * dT;
    q_rotate(RiT, a, ra);
    m3_v(M->JVg, sb->dbg, t1);
    m3_v(M->JVa, sb->dba, t2);
    for (int i = 0; i < 3; ++i) err[3 + i] = ra[i] - (M->dV[i] + t1[i] + t2[i]);
    double w[3], dRdbg[4], A[4], Ainv[4], B[4], C[4];
    m3_v(M->JRg, sb->dbg, w);
    so3_exp(w, dRdbg);
    so3_mul(dRij, dRdbg, A);
    so3_inverse(A, Ainv);
    so3_mul(Ainv, RiT, B);
    so3_mul(B, Rj, C);
    so3_log(C, err + 6);
}
/* EdgeNavStatePVR::linearizeOplus, IMU/g2otypes.cpp:94-234.  Uses the cached error's rotation part
 * (cpp:127).  J0 9x9 (PVR i), J1 9x9 (PVR j), J2 9x6 (bias i). */
static void pvr_linearize(const prob_t* p, const nav_t* si, const nav_t* sj, const nav_t* sb, const preint_t* M, const double* err,
                          double* J0, double* J1, double* J2) {
    double Ri[9], Rj[9], RiT[9], RjT[9];
    nav_RotMatrix(si, Ri);
    nav_RotMatrix(sj, Rj);
    m3_T(Ri, RiT);
    m3_T(Rj, RjT);
    double dT = M->dt, dT2 = dT * dT;
    const double* rPhi = err + 6;
    double JrInv[9];
    so3_JrInv(rPhi, JrInv);
    memset(J0, 0, 81 * 8); memset(J1, 0, 81 * 8); memset(J2, 0, 54 * 8);
    double a[3], v[3], H[9];
    /* 4.1 */
    for (int i = 0; i < 3; ++i) J0[i * 9 + i] = -1.0;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J0[r * 9 + 3 + c] = -RiT[r * 3 + c] * dT;
    for (int i = 0; i < 3; ++i) a[i] = sj->P[i] - si->P[i] - si->V[i] * dT - 0.5 * p->gw[i] * dT2;
    m3_v(RiT, a, v);
    so3_hat(v, H);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J0[r * 9 + 6 + c] = H[r * 3 + c];
    /* 4.2 */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J0[(3 + r) * 9 + 3 + c] = -RiT[r * 3 + c];
    for (int i = 0; i < 3; ++i) a[i] = sj->V[i] - si->V[i] - p->gw[i] * dT;
    m3_v(RiT, a, v);
    so3_hat(v, H);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J0[(3 + r) * 9 + 6 + c] = H[r * 3 + c];
    /* 4.3  - JrInv * RjT * Ri   ((-JrInv*RjT)*Ri left to right) */
    double nJ[9], t[9], t2[9];
    for (int i = 0; i < 9; ++i) nJ[i] = -JrInv[i];
    m3_mul(nJ, RjT, t);
    m3_mul(t, Ri, t2);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J0[(6 + r) * 9 + 6 + c] = t2[r * 3 + c];
    /* 5 */
    m3_mul(RiT, Rj, t);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        J1[r * 9 + c] = t[r * 3 + c];
        J1[(3 + r) * 9 + 3 + c] = RiT[r * 3 + c];
        J1[(6 + r) * 9 + 6 + c] = JrInv[r * 3 + c];
    }
    /* 6 */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        J2[r * 6 + c] = -M->JPg[r * 3 + c];
        J2[r * 6 + 3 + c] = -M->JPa[r * 3 + c];
        J2[(3 + r) * 6 + c] = -M->JVg[r * 3 + c];
        J2[(3 + r) * 6 + 3 + c] = -M->JVa[r * 3 + c];
    }
    /* ExprPhiijTrans = exp(rPhi).inverse().matrix(); JrBiasGCorr = Jr(J_rPhi_dbg*dBgi) */
    double e4[4], ei[4], ET[9], w[3], JrB[9];
    so3_exp(rPhi, e4);
    so3_inverse(e4, ei);
    q_to_R(ei, ET);
    m3_v(M->JRg, sb->dbg, w);
    so3_Jr(w, JrB);
    /* - JrInv * ExprPhiijTrans * JrBiasGCorr * J_rPhi_dbg */
    double u1[9], u2[9], u3[9];
    m3_mul(nJ, ET, u1);
    m3_mul(u1, JrB, u2);
    m3_mul(u2, M->JRg, u3);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J2[(6 + r) * 6 + c] = u3[r * 3 + c];
}
/* EdgeNavStateBias::computeError, IMU/g2otypes.cpp:236-262 */
static void bias_error(const nav_t* si, const nav_t* sj, double* err) {
    for (int i = 0; i < 3; ++i) {
        err[i] = (sj->bg[i] + sj->dbg[i]) - (si->bg[i] + si->dbg[i]);
        err[3 + i] = (sj->ba[i] + sj->dba[i]) - (si->ba[i] + si->dba[i]);
    }
}

static int find_kf_by_vid(const prob_t* p, int vid, int* is_bias) {
    for (int k = 0; k < p->K; ++k) {
        if (p->vid_pvr[k] == vid) { *is_bias = 0; return k; }
        if (p->vid_bias[k] == vid) { *is_bias = 1; return k; }
    }
    return -1;
}
/* EdgeMarginalization::computeError, IMU/g2otypes.cpp:1423-1475 */
static int prior_error(prob_t* p, double* err) {
    int n = p->pr_n;
    double* dx = (double*)xcalloc(n, 8);
    for (int i = 0; i < p->pr_nv; ++i) {
        int size = p->pr_size[i], idx = p->pr_idx[i], isb;
        int k = find_kf_by_vid(p, p->pr_vid[i], &isb);
        if (k < 0) { free(dx); FAIL(p, PLBA_ERR_INVALID, "prior vertex id %d not in the window", p->pr_vid[i]); }
        const double* x0 = p->pr_x0 + p->pr_x0_off[i];
        const nav_t* s = &p->ns[k];
        if (size != 9) {
            for (int c = 0; c < 3; ++c) {
                dx[idx + c] = (s->bg[c] + s->dbg[c]) - x0[c];
                dx[idx + 3 + c] = (s->ba[c] + s->dba[c]) - x0[3 + c];
            }
        } else {
            for (int c = 0; c < 3; ++c) {
                dx[idx + c] = s->P[c] - x0[c];
                dx[idx + 3 + c] = s->V[c] - x0[3 + c];
            }
            /* 2 * (Quaterniond(x0(9),x0(6),x0(7),x0(8)).inverse() * Quaterniond(RotMatrix)).vec()  (no w-sign fix, B-Q5) */
            double q0[4] = {x0[6], x0[7], x0[8], x0[9]}, qi[4], R[9], qc[4], qr[4];
            double n2 = q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3];
            qi[0] = -q0[0] / n2; qi[1] = -q0[1] / n2; qi[2] = -q0[2] / n2; qi[3] = q0[3] / n2; /* Eigen inverse() */
            nav_RotMatrix(s, R);
            R_to_q(R, qc);
            q_mul(qi, qc, qr);
            dx[idx + 6] = 2.0 * qr[0]; dx[idx + 7] = 2.0 * qr[1]; dx[idx + 8] = 2.0 * qr[2];
        }
    }
    for (int r = 0; r < n; ++r) {
        double s = 0.0;
        for (int c = 0; c < n; ++c) s += p->pr_J0[(size_t)c * n + r] * dx[c]; /* col-major */
        err[r] = p->pr_r0[r] + s;
    }
    free(dx);
    return PLBA_OK;
}

/* g2o RobustKernelHuber::robustify */
static void huber(double e, double delta, double* rho) {
    double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else {
        double sqrte = sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e;
    }
}
static double quad_form(const double* e, const double* Om, int n) { /* e^T Omega e */
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += Om[i * n + j] * e[j];
        s += e[i] * t;
    }
    return s;
}

/* ============================================================================================
 * 6. g2o: active set, computeActiveErrors, buildSystem, solve (Schur), LM  
 * ========================================================================================== */
static int pose_edges_owned(const prob_t* p) { return p->rank == 0; } /* IMU+prior by rank 0 only */

static void build_index(prob_t* p) { /* initializeOptimization(0) + buildStructure, App. A.1 */
    int K = p->K;
    free(p->off_pvr); free(p->off_bias); free(p->pt_active); free(p->ln_active); free(p->pt_xoff); free(p->ln_xoff);
    p->off_pvr = (int*)xcalloc(K, sizeof(int));
    p->off_bias = (int*)xcalloc(K, sizeof(int));
    p->pt_active = (uint8_t*)xcalloc(p->Np, 1);
    p->ln_active = (uint8_t*)xcalloc(p->Nl, 1);
    p->pt_xoff = (int*)xcalloc(p->Np, sizeof(int));
    p->ln_xoff = (int*)xcalloc(p->Nl, sizeof(int));
    uint8_t* act_pvr = (uint8_t*)xcalloc(K, 1);
    uint8_t* act_bias = (uint8_t*)xcalloc(K, 1);
    for (int e = 0; e < p->Ep; ++e) if (p->po_level[e] == 0) { p->pt_active[p->po_pt[e]] = 1; act_pvr[p->po_kf[e]] = 1; }
    for (int e = 0; e < p->El; ++e) if (p->lo_level[e] == 0) { p->ln_active[p->lo_ln[e]] = 1; act_pvr[p->lo_kf[e]] = 1; }
    /* in a sharded run every rank must use the same pose layout: activity of pose-side vertices is
       decided from the replicated pose-side edges plus "any observation on any rank"; the host shards
       so that this is the same set (every keyframe keeps at least its IMU edges).  Without IMU edges
       (configs 1-2) a sharded keyframe with no local observation still gets an index: its rows are
       filled by the all-reduce. */
    if (p->world > 1) for (int k = 0; k < K; ++k) act_pvr[k] = 1;
    for (int m = 0; m < p->M; ++m) {
        act_pvr[p->im_i[m]] = 1; act_pvr[p->im_j[m]] = 1; act_bias[p->im_i[m]] = 1; act_bias[p->im_j[m]] = 1;
    }
    for (int i = 0; i < p->pr_nv; ++i) {
        int isb, k = find_kf_by_vid(p, p->pr_vid[i], &isb);
        if (k >= 0) { if (isb) act_bias[k] = 1; else act_pvr[k] = 1; }
    }
    int off = 0;
    for (int k = 0; k < K; ++k) { /* non-marginalized vertices by ascending id: PVR 2k, Bias 2k+1 interleaved */
        int pv = (act_pvr[k] && !p->fix_pvr[k]);
        int bv = (p->vid_bias[k] >= 0 && act_bias[k] && !p->fix_bias[k]);
        if (pv && bv && p->vid_bias[k] < p->vid_pvr[k]) { /* bias id below pvr id: bias first */
            p->off_bias[k] = off; off += 6; p->off_pvr[k] = off; off += 9;
        } else {
            if (pv) { p->off_pvr[k] = off; off += 9; } else p->off_pvr[k] = -1;
            if (bv) { p->off_bias[k] = off; off += 6; } else p->off_bias[k] = -1;
        }
    }
    p->P = off;
    int lo = 0;
    for (int i = 0; i < p->Np; ++i) { if (p->pt_active[i] && !p->pt_fixed[i]) { p->pt_xoff[i] = lo; lo += 3; } else p->pt_xoff[i] = -1; }
    for (int i = 0; i < p->Nl; ++i) { if (p->ln_active[i] && !p->ln_fixed[i]) { p->ln_xoff[i] = lo; lo += 6; } else p->ln_xoff[i] = -1; }
    p->Ldim = lo;
    free(act_pvr); free(act_bias);
    size_t PP = (size_t)p->P * p->P;
    free(p->Hpp); p->Hpp = (double*)xcalloc(PP, 8);
    free(p->Hs); p->Hs = (double*)xcalloc(PP, 8);
    free(p->bp); p->bp = (double*)xcalloc(p->P, 8);
    free(p->bpg); p->bpg = (double*)xcalloc(p->P, 8);
    free(p->bs); p->bs = (double*)xcalloc(p->P, 8);
    free(p->x); p->x = (double*)xcalloc((size_t)p->P + lo, 8);
    free(p->bl_all); p->bl_all = (double*)xcalloc(lo, 8);
    free(p->Hll_pt); p->Hll_pt = (double*)xcalloc((size_t)p->Np * 9, 8);
    free(p->bl_pt); p->bl_pt = (double*)xcalloc((size_t)p->Np * 3, 8);
    free(p->Hll_ln); p->Hll_ln = (double*)xcalloc((size_t)p->Nl * 36, 8);
    free(p->bl_ln); p->bl_ln = (double*)xcalloc((size_t)p->Nl * 6, 8);
    free(p->Hpl_pt); p->Hpl_pt = (double*)xcalloc((size_t)p->Ep * 27, 8);
    free(p->Hpl_ln); p->Hpl_ln = (double*)xcalloc((size_t)p->El * 54, 8);
    free(p->Dinv_pt); p->Dinv_pt = (double*)xcalloc((size_t)p->Np * 9, 8);
    free(p->Dinv_ln); p->Dinv_ln = (double*)xcalloc((size_t)p->Nl * 36, 8);
}

/* SparseOptimizer::computeActiveErrors + activeRobustChi2 (edge insertion order of the call site:
 * IMU PVR/bias pairs, points, lines, prior; mapHandler.cpp:5842-6034).  Returns local chi2. */
static int compute_active_errors(prob_t* p, double* chi_out) {
    double chi = 0.0, rho[3];
    if (pose_edges_owned(p)) {
        for (int m = 0; m < p->M; ++m) {
            const nav_t *si = &p->ns[p->im_i[m]], *sj = &p->ns[p->im_j[m]];
            double* e = p->im_err_pvr + 9 * m;
            pvr_error(p, si, sj, si, &p->im_pre[m], e);
            double c = quad_form(e, p->im_info_pvr + 81 * m, 9);
            if (p->rob_on[PLBA_EDGE_IMU_PVR]) { huber(c, p->rob_delta[PLBA_EDGE_IMU_PVR], rho); chi += rho[0]; } else chi += c;
            double* eb = p->im_err_bias + 6 * m;
            bias_error(si, sj, eb);
            c = quad_form(eb, p->im_info_bias + 36 * m, 6);
            if (p->rob_on[PLBA_EDGE_IMU_BIAS]) { huber(c, p->rob_delta[PLBA_EDGE_IMU_BIAS], rho); chi += rho[0]; } else chi += c;
        }
    }
#ifdef PLBA_ORACLE_FAST
#pragma omp parallel for schedule(static) reduction(+ : chi)
#endif
    for (int e = 0; e < p->Ep; ++e) {
        if (p->po_level[e]) continue;
        double rh[3];
        double* er = p->po_err + 2 * e;
        point_error(p, &p->ns[p->po_kf[e]], p->pt + 3 * p->po_pt[e], p->po_uv + 2 * e, er, NULL);
        double c = p->po_w[e] * (er[0] * er[0] + er[1] * er[1]);
        if (p->rob_on[PLBA_EDGE_POINT]) { huber(c, p->rob_delta[PLBA_EDGE_POINT], rh); chi += rh[0]; } else chi += c;
    }
#ifdef PLBA_ORACLE_FAST
#pragma omp parallel for schedule(static) reduction(+ : chi)
#endif
    for (int e = 0; e < p->El; ++e) {
        if (p->lo_level[e]) continue;
        double rh[3];
        double* er = p->lo_err + 3 * e;
        line_error(p, &p->ns[p->lo_kf[e]], p->ln + 6 * p->lo_ln[e], p->lo_l + 3 * e, er, NULL);
        double c = p->lo_w[e] * (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]);
        if (p->rob_on[PLBA_EDGE_LINE]) { huber(c, p->rob_delta[PLBA_EDGE_LINE], rh); chi += rh[0]; } else chi += c;
    }
    if (pose_edges_owned(p) && p->pr_nv) {
        int rc = prior_error(p, p->pr_err);
        if (rc) return rc;
        double c = 0.0;
        for (int i = 0; i < p->pr_n; ++i) c += p->pr_err[i] * p->pr_err[i]; /* Omega = I, no kernel (mapHandler.cpp:6029) */
        chi += c;
    }
    *chi_out = chi;
    return PLBA_OK;
}

/* H(oi.., oj..) += A^T W B  with A (d x ni), B (d x nj), W (d x d); b(oi..) += A^T wr */
static void add_block(double* H, int P, int oi, int ni, int oj, int nj, const double* A, const double* B, const double* W, int d) {
    /* AtO = A^T * W  (ni x d) */
    double AtO[9 * 9 * 4];
    double* use = AtO;
    double* heap = NULL;
    if ((size_t)ni * d > sizeof(AtO) / 8) { heap = (double*)xcalloc((size_t)ni * d, 8); use = heap; }
    for (int i = 0; i < ni; ++i)
        for (int k = 0; k < d; ++k) {
            double s = 0.0;
            for (int t = 0; t < d; ++t) s += A[t * ni + i] * W[t * d + k];
            use[i * d + k] = s;
        }
    for (int i = 0; i < ni; ++i)
        for (int j = 0; j < nj; ++j) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += use[i * d + k] * B[k * nj + j];
            H[(size_t)(oi + i) * P + oj + j] += s;
        }
    free(heap);
}
static void add_grad(double* b, int oi, int ni, const double* A, const double* wr, int d) {
    for (int i = 0; i < ni; ++i) {
        double s = 0.0;
        for (int t = 0; t < d; ++t) s += A[t * ni + i] * wr[t];
        b[oi + i] += s;
    }
}

/* constructQuadraticForm of one point / line edge (App. A.4); Hpp / bp: the pose-side accumulators (the problem's own, or a thread's
 * private copy in the PLBA_ORACLE_FAST build) */
static void bs_point(prob_t* p, int e, double* Hpp, double* bp) {
    if (p->po_level[e]) return;
    int P = p->P;
    double rho[3];
    int l = p->po_pt[e], k = p->po_kf[e];
    double Ji[6], Jj[18];
    point_linearize(p, &p->ns[k], p->pt + 3 * l, Ji, Jj);
    const double* er = p->po_err + 2 * e;
    double w = p->po_w[e];
    if (p->rob_on[PLBA_EDGE_POINT]) { huber(w * (er[0] * er[0] + er[1] * er[1]), p->rob_delta[PLBA_EDGE_POINT], rho); w *= rho[1]; }
    double Om[4] = {w, 0, 0, w}, wr[2] = {-w * er[0], -w * er[1]};
    int lact = (p->pt_xoff[l] >= 0), op = p->off_pvr[k];
    if (lact) { add_grad(p->bl_pt + 3 * l, 0, 3, Ji, wr, 2); add_block(p->Hll_pt + 9 * l, 3, 0, 3, 0, 3, Ji, Ji, Om, 2); }
    if (op >= 0) { add_grad(bp, op, 9, Jj, wr, 2); add_block(Hpp, P, op, 9, op, 9, Jj, Jj, Om, 2); }
    if (lact && op >= 0) add_block(p->Hpl_pt + 27 * (size_t)e, 3, 0, 9, 0, 3, Jj, Ji, Om, 2); /* Hpl = Jj^T Om Ji (9x3) */
}
static void bs_line(prob_t* p, int e, double* Hpp, double* bp) {
    if (p->lo_level[e]) return;
    int P = p->P;
    double rho[3];
    int l = p->lo_ln[e], k = p->lo_kf[e];
    double Ji[18], Jj[27];
    line_linearize(p, &p->ns[k], p->ln + 6 * l, p->lo_l + 3 * e, Ji, Jj);
    const double* er = p->lo_err + 3 * e;
    double w = p->lo_w[e];
    if (p->rob_on[PLBA_EDGE_LINE]) { huber(w * (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]), p->rob_delta[PLBA_EDGE_LINE], rho); w *= rho[1]; }
    double Om[9] = {w, 0, 0, 0, w, 0, 0, 0, w}, wr[3] = {-w * er[0], -w * er[1], -w * er[2]};
    int lact = (p->ln_xoff[l] >= 0), op = p->off_pvr[k];
    if (lact) { add_grad(p->bl_ln + 6 * l, 0, 6, Ji, wr, 3); add_block(p->Hll_ln + 36 * l, 6, 0, 6, 0, 6, Ji, Ji, Om, 3); }
    if (op >= 0) { add_grad(bp, op, 9, Jj, wr, 3); add_block(Hpp, P, op, 9, op, 9, Jj, Jj, Om, 3); }
    if (lact && op >= 0) add_block(p->Hpl_ln + 54 * (size_t)e, 6, 0, 9, 0, 6, Jj, Ji, Om, 3);
}
#ifdef PLBA_ORACLE_FAST
/* The PLBA_ORACLE_FAST build (oracle/Makefile: -O3 -march=native -ffp-contract=fast -fopenmp; bench.py's stronger CPU leg, never the
 * parity checker): the same per-edge arithmetic, spread over the host cores by LANDMARK (a landmark's edges are contiguous and its
 * blocks private to it), pose-side sums into thread-private copies that are added up afterwards; the reduced system factored inside
 * its envelope (a sparse Cholesky, as g2o's LinearSolverEigen is).  Summation orders differ from the faithful build by rounding. */
#include <omp.h>
static void lm_edge_starts(const int32_t* lm_of, int E, int N, int* start) { /* start[l] .. start[l + 1]: the edges of landmark l */
    int e = 0;
    for (int l = 0; l < N; ++l) { start[l] = e; while (e < E && lm_of[e] == l) ++e; }
    start[N] = e;
}
/* Landmarks in order of their first keyframe, dealt to the threads in contiguous runs: a thread then touches the pose blocks of a
 * narrow range of keyframes, and its private copy of the pose-side sums covers the rows [lo, hi] only (full copies, 8 x 4.3 MB at
 * configs[2], cost more to clear and add up than the edges cost to evaluate: measured). */
typedef struct { int *ordp, *ordl, *ps, *ls; } fast_order;
static void fast_order_make(prob_t* p, fast_order* o) {
    o->ps = (int*)xcalloc(p->Np + 2, sizeof(int)); o->ls = (int*)xcalloc(p->Nl + 2, sizeof(int));
    lm_edge_starts(p->po_pt, p->Ep, p->Np, o->ps); lm_edge_starts(p->lo_ln, p->El, p->Nl, o->ls);
    o->ordp = (int*)xcalloc(p->Np + 1, sizeof(int)); o->ordl = (int*)xcalloc(p->Nl + 1, sizeof(int));
    int* cnt = (int*)xcalloc(p->K + 2, sizeof(int));
    for (int kind = 0; kind < 2; ++kind) {
        const int N = kind ? p->Nl : p->Np; const int* st = kind ? o->ls : o->ps; const int32_t* kf = kind ? p->lo_kf : p->po_kf; int* ord = kind ? o->ordl : o->ordp;
        memset(cnt, 0, (size_t)(p->K + 2) * sizeof(int));
        for (int l = 0; l < N; ++l) { int k0 = p->K; for (int e = st[l]; e < st[l + 1]; ++e) if (kf[e] < k0) k0 = kf[e]; cnt[(k0 < p->K ? k0 : p->K) + 1]++; }
        for (int k = 0; k <= p->K; ++k) cnt[k + 1] += cnt[k];
        for (int l = 0; l < N; ++l) { int k0 = p->K; for (int e = st[l]; e < st[l + 1]; ++e) if (kf[e] < k0) k0 = kf[e]; ord[cnt[k0 < p->K ? k0 : p->K]++] = l; }
    }
    free(cnt);
}
static void fast_order_free(fast_order* o) { free(o->ordp); free(o->ordl); free(o->ps); free(o->ls); }
/* fn_pt / fn_ln: the per-landmark work; H receives the pose x pose terms (leading dimension P), v a pose-side vector */
typedef void (*fast_lm_fn)(prob_t* p, int l, int e0, int e1, double lambda, double* H, double* v);
static void fast_over_landmarks(prob_t* p, double lambda, fast_lm_fn fn_pt, fast_lm_fn fn_ln, double* Hdst, double* vdst) {
    const int P = p->P, nt = omp_get_max_threads();
    fast_order o;
    fast_order_make(p, &o);
    double** bufs = (double**)xcalloc(nt + 1, sizeof(double*)); double** vecs = (double**)xcalloc(nt + 1, sizeof(double*));
    int* lo = (int*)xcalloc(nt + 1, sizeof(int)); int* hi = (int*)xcalloc(nt + 1, sizeof(int));
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const int p0 = (int)((long)p->Np * t / T), p1 = (int)((long)p->Np * (t + 1) / T), l0 = (int)((long)p->Nl * t / T), l1 = (int)((long)p->Nl * (t + 1) / T);
        int rlo = P, rhi = -1;
        for (int q = p0; q < p1; ++q) for (int e = o.ps[o.ordp[q]]; e < o.ps[o.ordp[q] + 1]; ++e) { const int oa = p->off_pvr[p->po_kf[e]]; if (oa >= 0) { if (oa < rlo) rlo = oa; if (oa + 8 > rhi) rhi = oa + 8; } }
        for (int q = l0; q < l1; ++q) for (int e = o.ls[o.ordl[q]]; e < o.ls[o.ordl[q] + 1]; ++e) { const int oa = p->off_pvr[p->lo_kf[e]]; if (oa >= 0) { if (oa < rlo) rlo = oa; if (oa + 8 > rhi) rhi = oa + 8; } }
        lo[t] = rlo; hi[t] = rhi;
        double* buf = (rhi >= rlo) ? (double*)xcalloc((size_t)(rhi - rlo + 1) * P + 1, 8) : NULL;
        double* vec = (double*)xcalloc(P + 1, 8);
        bufs[t] = buf; vecs[t] = vec;
        double* H = buf ? buf - (size_t)rlo * P : NULL;      /* rows [rlo, rhi] of a P x P matrix */
        for (int q = p0; q < p1; ++q) { const int l = o.ordp[q]; fn_pt(p, l, o.ps[l], o.ps[l + 1], lambda, H, vec); }
        for (int q = l0; q < l1; ++q) { const int l = o.ordl[q]; fn_ln(p, l, o.ls[l], o.ls[l + 1], lambda, H, vec); }
#pragma omp barrier
#pragma omp for schedule(static)
        for (int r = 0; r < P; ++r) {
            double* d = Hdst + (size_t)r * P;
            for (int u = 0; u < T; ++u) {
                if (!bufs[u] || r < lo[u] || r > hi[u]) continue;
                const double* src = bufs[u] + (size_t)(r - lo[u]) * P;
                for (int c = 0; c < P; ++c) d[c] += src[c];
            }
            double sv = 0.0;
            for (int u = 0; u < T; ++u) sv += vecs[u][r];
            vdst[r] += sv;
        }
        free(buf); free(vec);
    }
    free(bufs); free(vecs); free(lo); free(hi);
    fast_order_free(&o);
}
static void fast_bs_pt(prob_t* p, int l, int e0, int e1, double lambda, double* H, double* v) { (void)l; (void)lambda; for (int e = e0; e < e1; ++e) bs_point(p, e, H, v); }
static void fast_bs_ln(prob_t* p, int l, int e0, int e1, double lambda, double* H, double* v) { (void)l; (void)lambda; for (int e = e0; e < e1; ++e) bs_line(p, e, H, v); }
static void fast_build_edges(prob_t* p) { fast_over_landmarks(p, 0.0, fast_bs_pt, fast_bs_ln, p->Hpp, p->bp); }
#endif

/* BlockSolver::buildSystem: linearizeOplus + constructQuadraticForm per active edge (App. A.4).
 * Hpp holds the upper block-triangle only (as g2o), mirrored before the dense factorisation. */
static void build_system(prob_t* p) {
    int P = p->P;
    memset(p->Hpp, 0, (size_t)P * P * 8);
    memset(p->bp, 0, (size_t)P * 8);
    memset(p->Hll_pt, 0, (size_t)p->Np * 72); memset(p->bl_pt, 0, (size_t)p->Np * 24);
    memset(p->Hll_ln, 0, (size_t)p->Nl * 288); memset(p->bl_ln, 0, (size_t)p->Nl * 48);
    memset(p->Hpl_pt, 0, (size_t)p->Ep * 27 * 8); memset(p->Hpl_ln, 0, (size_t)p->El * 54 * 8);
    double rho[3];
    if (pose_edges_owned(p)) {
        for (int m = 0; m < p->M; ++m) {
            int ki = p->im_i[m], kj = p->im_j[m];
            const nav_t *si = &p->ns[ki], *sj = &p->ns[kj];
            /* --- PVR multi-edge: vertices (PVR_i, PVR_j, Bias_i) --- */
            double J0[81], J1[81], J2[54];
            const double* e = p->im_err_pvr + 9 * m;
            pvr_linearize(p, si, sj, si, &p->im_pre[m], e, J0, J1, J2);
            double Om[81], wr[9];
            memcpy(Om, p->im_info_pvr + 81 * m, 81 * 8);
            double w = 1.0;
            if (p->rob_on[PLBA_EDGE_IMU_PVR]) { huber(quad_form(e, Om, 9), p->rob_delta[PLBA_EDGE_IMU_PVR], rho); w = rho[1]; }
            for (int i = 0; i < 9; ++i) { double s = 0; for (int j = 0; j < 9; ++j) s += Om[i * 9 + j] * e[j]; wr[i] = -s * w; }
            if (w != 1.0) for (int i = 0; i < 81; ++i) Om[i] *= w;
            int off[3] = {p->off_pvr[ki], p->off_pvr[kj], p->off_bias[ki]};
            int dim[3] = {9, 9, 6};
            const double* Js[3] = {J0, J1, J2};
            for (int a = 0; a < 3; ++a) {
                if (off[a] < 0) continue;
                add_grad(p->bp, off[a], dim[a], Js[a], wr, 9);
                for (int b = a; b < 3; ++b) {
                    if (off[b] < 0) continue;
                    if (off[a] <= off[b]) add_block(p->Hpp, P, off[a], dim[a], off[b], dim[b], Js[a], Js[b], Om, 9);
                    else add_block(p->Hpp, P, off[b], dim[b], off[a], dim[a], Js[b], Js[a], Om, 9);
                }
            }
            /* --- bias binary edge (Bias_i, Bias_j): J = -I, +I (g2otypes.cpp:264-284) --- */
            double Ji[36] = {0}, Jj[36] = {0};
            for (int i = 0; i < 6; ++i) { Ji[i * 6 + i] = -1.0; Jj[i * 6 + i] = 1.0; }
            const double* eb = p->im_err_bias + 6 * m;
            double Ob[36], wb[6];
            memcpy(Ob, p->im_info_bias + 36 * m, 36 * 8);
            w = 1.0;
            if (p->rob_on[PLBA_EDGE_IMU_BIAS]) { huber(quad_form(eb, Ob, 6), p->rob_delta[PLBA_EDGE_IMU_BIAS], rho); w = rho[1]; }
            for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += Ob[i * 6 + j] * eb[j]; wb[i] = -s * w; }
            if (w != 1.0) for (int i = 0; i < 36; ++i) Ob[i] *= w;
            int oi = p->off_bias[ki], oj = p->off_bias[kj];
            if (oi >= 0) { add_grad(p->bp, oi, 6, Ji, wb, 6); add_block(p->Hpp, P, oi, 6, oi, 6, Ji, Ji, Ob, 6); }
            if (oj >= 0) { add_grad(p->bp, oj, 6, Jj, wb, 6); add_block(p->Hpp, P, oj, 6, oj, 6, Jj, Jj, Ob, 6); }
            if (oi >= 0 && oj >= 0) {
                if (oi <= oj) add_block(p->Hpp, P, oi, 6, oj, 6, Ji, Jj, Ob, 6);
                else add_block(p->Hpp, P, oj, 6, oi, 6, Jj, Ji, Ob, 6);
            }
        }
    }
#ifdef PLBA_ORACLE_FAST
    fast_build_edges(p);
#else
    for (int e = 0; e < p->Ep; ++e) bs_point(p, e, p->Hpp, p->bp);
    for (int e = 0; e < p->El; ++e) bs_line(p, e, p->Hpp, p->bp);
#endif
    if (pose_edges_owned(p) && p->pr_nv) { /* EdgeMarginalization::linearizeOplus g2otypes.cpp:1477-1497; multi-edge: all vertex pairs */
        int n = p->pr_n;
        int* off = (int*)xcalloc(p->pr_nv, sizeof(int));
        for (int i = 0; i < p->pr_nv; ++i) {
            int isb, k = find_kf_by_vid(p, p->pr_vid[i], &isb);
            off[i] = (k < 0) ? -1 : (isb ? p->off_bias[k] : p->off_pvr[k]);
        }
        for (int a = 0; a < p->pr_nv; ++a) {
            if (off[a] < 0) continue;
            int sa = p->pr_size[a], ia = p->pr_idx[a];
            for (int c = 0; c < sa; ++c) {
                double s = 0.0;
                for (int r = 0; r < n; ++r) s += p->pr_J0[(size_t)(ia + c) * n + r] * (-p->pr_err[r]);
                p->bp[off[a] + c] += s;
            }
            for (int b = a; b < p->pr_nv; ++b) {
                if (off[b] < 0) continue;
                int sb = p->pr_size[b], ib = p->pr_idx[b];
                int up = off[a] <= off[b];
                for (int c = 0; c < sa; ++c)
                    for (int d = 0; d < sb; ++d) {
                        double s = 0.0;
                        const double* ca = p->pr_J0 + (size_t)(ia + c) * n;
                        const double* cb = p->pr_J0 + (size_t)(ib + d) * n;
                        for (int r = 0; r < n; ++r) s += ca[r] * cb[r];
                        if (up) p->Hpp[(size_t)(off[a] + c) * P + off[b] + d] += s;
                        else p->Hpp[(size_t)(off[b] + d) * P + off[a] + c] += s;
                    }
            }
        }
        free(off);
    }
    /* the landmark part of b (g2o's _b beyond sizePoses) */
    for (int i = 0; i < p->Np; ++i) if (p->pt_xoff[i] >= 0) memcpy(p->bl_all + p->pt_xoff[i], p->bl_pt + 3 * i, 24);
    for (int i = 0; i < p->Nl; ++i) if (p->ln_xoff[i] >= 0) memcpy(p->bl_all + p->ln_xoff[i], p->bl_ln + 6 * i, 48);
}

static int exchange(prob_t* p, double* buf, size_t n, int op) {
    if (p->world <= 1 || !p->xfn) return PLBA_OK;
    return p->xfn(p->xuser, buf, n, op, NULL) ? PLBA_ERR_EXCHANGE : PLBA_OK;
}

/* computeLambdaInit: tau * max |H_jj| over all active non-fixed vertices, poses and landmarks (A.3) */
static int max_diag(prob_t* p, double* out) {
    double m = 0.0;
    if (p->world > 1) { /* pose diagonals are sums over ranks */
        double* d = (double*)xcalloc(p->P + 1, 8);
        for (int i = 0; i < p->P; ++i) d[i] = p->Hpp[(size_t)i * p->P + i];
        int rc = exchange(p, d, p->P, 0);
        if (rc) { free(d); return rc; }
        for (int i = 0; i < p->P; ++i) if (fabs(d[i]) > m) m = fabs(d[i]);
        free(d);
    } else {
        for (int i = 0; i < p->P; ++i) { double v = fabs(p->Hpp[(size_t)i * p->P + i]); if (v > m) m = v; }
    }
    for (int i = 0; i < p->Np; ++i) if (p->pt_xoff[i] >= 0) for (int c = 0; c < 3; ++c) { double v = fabs(p->Hll_pt[9 * i + c * 4]); if (v > m) m = v; }
    for (int i = 0; i < p->Nl; ++i) if (p->ln_xoff[i] >= 0) for (int c = 0; c < 6; ++c) { double v = fabs(p->Hll_ln[36 * i + c * 7]); if (v > m) m = v; }
    if (p->world > 1) { int rc = exchange(p, &m, 1, 1); if (rc) return rc; }
    *out = m;
    return PLBA_OK;
}

/* Eigen PartialPivLU-based inverse for the dynamic-size landmark blocks of BlockSolverX (A.5) */
static int lu_inverse(const double* A, double* Ainv, int n) {
    double a[36], inv[36];
    int piv[6];
    memcpy(a, A, (size_t)n * n * 8);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int r = k; double best = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i) if (fabs(a[i * n + k]) > best) { best = fabs(a[i * n + k]); r = i; }
        if (best == 0.0) return 0;
        if (r != k) { for (int j = 0; j < n; ++j) { double t = a[k * n + j]; a[k * n + j] = a[r * n + j]; a[r * n + j] = t; } int t = piv[k]; piv[k] = piv[r]; piv[r] = t; }
        for (int i = k + 1; i < n; ++i) {
            a[i * n + k] /= a[k * n + k];
            for (int j = k + 1; j < n; ++j) a[i * n + j] -= a[i * n + k] * a[k * n + j];
        }
    }
    for (int c = 0; c < n; ++c) { /* solve A x = e_c */
        double y[6];
        for (int i = 0; i < n; ++i) {
            double s = (piv[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s -= a[i * n + j] * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= a[i * n + j] * inv[j * n + c];
            inv[i * n + c] = s / a[i * n + i];
        }
    }
    memcpy(Ainv, inv, (size_t)n * n * 8);
    return 1;
}

/* dense LL^T (lower), in place; returns 0 if a pivot <= 0 (LinearSolverEigen failure, A.6) */
static int chol_factor(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            const double *ri = A + (size_t)i * n, *rj = A + (size_t)j * n;
            for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    return 1;
}
static void chol_solve(const double* L, int n, const double* b, double* x) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * x[k];
        x[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * x[k];
        x[i] = s / L[(size_t)i * n + i];
    }
}

/* BlockSolverX::solve, one landmark (App. A.5): D = (Hll + lambda I)^-1, bschur and Hschur terms of its edges [e0, e1); Hs / coeff: the
 * problem's own arrays, or a thread's private copies in the PLBA_ORACLE_FAST build */
static void schur_point(prob_t* p, int l, int e0, int e1, double lambda, double* Hs, double* coeff) {
    if (p->pt_xoff[l] < 0) return;
    int P = p->P;
    double D[9], Dinv[9], db[3];
    memcpy(D, p->Hll_pt + 9 * l, 72);
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    if (!lu_inverse(D, Dinv, 3)) memset(Dinv, 0, 72);
    memcpy(p->Dinv_pt + 9 * l, Dinv, 72);
    m3_v(Dinv, p->bl_pt + 3 * l, db);
    for (int a = e0; a < e1; ++a) {
        if (p->po_level[a]) continue;
        int oa = p->off_pvr[p->po_kf[a]];
        if (oa < 0) continue;
        const double* Ba = p->Hpl_pt + 27 * (size_t)a;
        double BD[27];
        mat_mul(Ba, Dinv, BD, 9, 3, 3);
        for (int r = 0; r < 9; ++r) coeff[oa + r] += Ba[r * 3] * db[0] + Ba[r * 3 + 1] * db[1] + Ba[r * 3 + 2] * db[2];
        for (int b = e0; b < e1; ++b) {
            if (p->po_level[b]) continue;
            int ob = p->off_pvr[p->po_kf[b]];
            if (ob < 0 || ob < oa) continue; /* upper block triangle: pairs (i1, i2 >= i1) */
            if (ob == oa && b != a) continue;
            const double* Bb = p->Hpl_pt + 27 * (size_t)b;
            for (int r = 0; r < 9; ++r)
                for (int c = 0; c < 9; ++c)
                    Hs[(size_t)(oa + r) * P + ob + c] -= BD[r * 3] * Bb[c * 3] + BD[r * 3 + 1] * Bb[c * 3 + 1] + BD[r * 3 + 2] * Bb[c * 3 + 2];
        }
    }
}
static void schur_line(prob_t* p, int l, int e0, int e1, double lambda, double* Hs, double* coeff) {
    if (p->ln_xoff[l] < 0) return;
    int P = p->P;
    double D[36], Dinv[36], db[6];
    memcpy(D, p->Hll_ln + 36 * l, 288);
    for (int i = 0; i < 6; ++i) D[i * 7] += lambda;
    if (!lu_inverse(D, Dinv, 6)) memset(Dinv, 0, 288);
    memcpy(p->Dinv_ln + 36 * l, Dinv, 288);
    mat_mul(Dinv, p->bl_ln + 6 * l, db, 6, 6, 1);
    for (int a = e0; a < e1; ++a) {
        if (p->lo_level[a]) continue;
        int oa = p->off_pvr[p->lo_kf[a]];
        if (oa < 0) continue;
        const double* Ba = p->Hpl_ln + 54 * (size_t)a;
        double BD[54];
        mat_mul(Ba, Dinv, BD, 9, 6, 6);
        for (int r = 0; r < 9; ++r) { double s = 0; for (int t = 0; t < 6; ++t) s += Ba[r * 6 + t] * db[t]; coeff[oa + r] += s; }
        for (int b = e0; b < e1; ++b) {
            if (p->lo_level[b]) continue;
            int ob = p->off_pvr[p->lo_kf[b]];
            if (ob < 0 || ob < oa) continue;
            if (ob == oa && b != a) continue;
            const double* Bb = p->Hpl_ln + 54 * (size_t)b;
            for (int r = 0; r < 9; ++r)
                for (int c = 0; c < 9; ++c) {
                    double s = 0;
                    for (int t = 0; t < 6; ++t) s += BD[r * 6 + t] * Bb[c * 6 + t];
                    Hs[(size_t)(oa + r) * P + ob + c] -= s;
                }
        }
    }
}
#ifdef PLBA_ORACLE_FAST
static void fast_schur(prob_t* p, double lambda, double* coeff) { fast_over_landmarks(p, lambda, schur_point, schur_line, p->Hs, coeff); }
/* LL^T inside the envelope of the matrix (row i starts at its first non-zero column): the fill of a Cholesky factor stays inside it,
 * so this is an exact sparse Cholesky for the banded reduced system — what g2o's LinearSolverEigen (simplicial LL^T) does. */
static int chol_factor_env(double* A, int n, int* first) {
    for (int i = 0; i < n; ++i) { int f = 0; while (f < i && A[(size_t)i * n + f] == 0.0) ++f; first[i] = f; }
    for (int i = 0; i < n; ++i) {
        double* ri = A + (size_t)i * n;
        for (int j = first[i]; j < i; ++j) {
            const double* rj = A + (size_t)j * n;
            double s = ri[j];
            const int k0 = first[i] > first[j] ? first[i] : first[j];
            for (int k = k0; k < j; ++k) s -= ri[k] * rj[k];
            ri[j] = s / rj[j];
        }
        double d = ri[i];
        for (int k = first[i]; k < i; ++k) d -= ri[k] * ri[k];
        if (!(d > 0.0)) return 0;
        ri[i] = sqrt(d);
    }
    return 1;
}
static void chol_solve_env(const double* L, int n, const int* first, const double* b, double* x) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = first[i]; k < i; ++k) s -= L[(size_t)i * n + k] * x[k];
        x[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {      /* column sweep: x_k -= L(i, k) x_i for k in the envelope of row i */
        x[i] /= L[(size_t)i * n + i];
        const double xi = x[i];
        for (int k = first[i]; k < i; ++k) x[k] -= L[(size_t)i * n + k] * xi;
    }
}
#endif

/* setLambda + BlockSolverX::solve with Schur complement (App. A.5) + restoreDiagonal.
 * do_solve = 0 stops after forming Hschur/bschur (diagnostics). */
static int schur_solve(prob_t* p, double lambda, int do_solve, int* ok_out) {
    int P = p->P;
    size_t PP = (size_t)P * P;
    memcpy(p->Hs, p->Hpp, PP * 8); /* Hschur = Hpp */
    if (p->world <= 1 || p->rank == 0) for (int i = 0; i < P; ++i) p->Hs[(size_t)i * P + i] += lambda;
    double* coeff = (double*)xcalloc(P, 8);
#ifdef PLBA_ORACLE_FAST
    fast_schur(p, lambda, coeff);
#else
    {
        int e = 0;
        for (int l = 0; l < p->Np; ++l) { int e0 = e; while (e < p->Ep && p->po_pt[e] == l) ++e; schur_point(p, l, e0, e, lambda, p->Hs, coeff); }
        e = 0;
        for (int l = 0; l < p->Nl; ++l) { int e0 = e; while (e < p->El && p->lo_ln[e] == l) ++e; schur_line(p, l, e0, e, lambda, p->Hs, coeff); }
    }
#endif
    for (int i = 0; i < P; ++i) p->bs[i] = p->bp[i] - coeff[i];
    free(coeff);
    /* multi-GPU analogue: all-reduce [Hschur | bschur | bp] */
    if (p->world > 1) {
        double* buf = (double*)xcalloc(PP + 2 * (size_t)P, 8);
        memcpy(buf, p->Hs, PP * 8); memcpy(buf + PP, p->bs, P * 8); memcpy(buf + PP + P, p->bp, P * 8);
        int rc = exchange(p, buf, PP + 2 * (size_t)P, 0);
        if (rc) { free(buf); return rc; }
        memcpy(p->Hs, buf, PP * 8); memcpy(p->bs, buf + PP, P * 8); memcpy(p->bpg, buf + PP + P, P * 8);
        free(buf);
    } else {
        memcpy(p->bpg, p->bp, (size_t)P * 8);
    }
    /* mirror the upper block-triangle to a full symmetric matrix */
    for (int i = 0; i < P; ++i) for (int j = i + 1; j < P; ++j) p->Hs[(size_t)j * P + i] = p->Hs[(size_t)i * P + j];
    if (!do_solve) { if (ok_out) *ok_out = 1; return PLBA_OK; }
    /* LinearSolverEigen: exact Cholesky of Hschur */
    double* L = (double*)xdup(p->Hs, PP * 8);
#ifdef PLBA_ORACLE_FAST
    int* first = (int*)xcalloc(P + 1, sizeof(int));
    int ok = (P == 0) ? 1 : chol_factor_env(L, P, first);
    memset(p->x, 0, ((size_t)P + p->Ldim) * 8);
    if (ok) chol_solve_env(L, P, first, p->bs, p->x);
    free(first);
    if (ok) {
#else
    int ok = (P == 0) ? 1 : chol_factor(L, P);
    memset(p->x, 0, ((size_t)P + p->Ldim) * 8);
    if (ok) {
        chol_solve(L, P, p->bs, p->x);
#endif
        for (int i = 0; i < P; ++i) if (!isfinite(p->x[i])) ok = 0;
    }
    free(L);
    if (!ok) { memset(p->x, 0, ((size_t)P + p->Ldim) * 8); *ok_out = 0; return PLBA_OK; }
    /* landmarks: xl = Dinv * (bl - Hpl^T xp) */
    double* xl = p->x + P;
    int e = 0;
    for (int l = 0; l < p->Np; ++l) {
        int e0 = e;
        while (e < p->Ep && p->po_pt[e] == l) ++e;
        if (p->pt_xoff[l] < 0) continue;
        double c[3] = {p->bl_pt[3 * l], p->bl_pt[3 * l + 1], p->bl_pt[3 * l + 2]};
        for (int a = e0; a < e; ++a) {
            if (p->po_level[a]) continue;
            int oa = p->off_pvr[p->po_kf[a]];
            if (oa < 0) continue;
            const double* Ba = p->Hpl_pt + 27 * (size_t)a;
            for (int r = 0; r < 9; ++r) for (int t = 0; t < 3; ++t) c[t] -= Ba[r * 3 + t] * p->x[oa + r];
        }
        m3_v(p->Dinv_pt + 9 * l, c, xl + p->pt_xoff[l]);
    }
    e = 0;
    for (int l = 0; l < p->Nl; ++l) {
        int e0 = e;
        while (e < p->El && p->lo_ln[e] == l) ++e;
        if (p->ln_xoff[l] < 0) continue;
        double c[6];
        memcpy(c, p->bl_ln + 6 * l, 48);
        for (int a = e0; a < e; ++a) {
            if (p->lo_level[a]) continue;
            int oa = p->off_pvr[p->lo_kf[a]];
            if (oa < 0) continue;
            const double* Ba = p->Hpl_ln + 54 * (size_t)a;
            for (int r = 0; r < 9; ++r) for (int t = 0; t < 6; ++t) c[t] -= Ba[r * 6 + t] * p->x[oa + r];
        }
        mat_mul(p->Dinv_ln + 36 * l, c, xl + p->ln_xoff[l], 6, 6, 1);
    }
    *ok_out = 1;
    return PLBA_OK;
}

static void state_push(prob_t* p) {
    memcpy(p->ns_bak, p->ns, sizeof(nav_t) * p->K);
    memcpy(p->pt_bak, p->pt, 24 * (size_t)p->Np);
    memcpy(p->ln_bak, p->ln, 48 * (size_t)p->Nl);
}
static void state_pop(prob_t* p) {
    memcpy(p->ns, p->ns_bak, sizeof(nav_t) * p->K);
    memcpy(p->pt, p->pt_bak, 24 * (size_t)p->Np);
    memcpy(p->ln, p->ln_bak, 48 * (size_t)p->Nl);
}
/* SparseOptimizer::update: oplus on every active non-fixed vertex (A.3) */
static void state_update(prob_t* p) {
    for (int k = 0; k < p->K; ++k) {
        if (p->off_pvr[k] >= 0) nav_IncSmallPVR(&p->ns[k], p->x + p->off_pvr[k]);
        if (p->off_bias[k] >= 0) nav_IncSmallBias(&p->ns[k], p->x + p->off_bias[k]);
    }
    const double* xl = p->x + p->P;
    for (int i = 0; i < p->Np; ++i) if (p->pt_xoff[i] >= 0) for (int c = 0; c < 3; ++c) p->pt[3 * i + c] += xl[p->pt_xoff[i] + c];
    for (int i = 0; i < p->Nl; ++i) if (p->ln_xoff[i] >= 0) for (int c = 0; c < 6; ++c) p->ln[6 * i + c] += xl[p->ln_xoff[i] + c];
}

static void trace_add(prob_t* p, const plba_trace_row* r) {
    if (p->trace_n == p->trace_cap) {
        p->trace_cap = p->trace_cap ? 2 * p->trace_cap : 64;
        p->trace = (plba_trace_row*)realloc(p->trace, sizeof(plba_trace_row) * p->trace_cap);
    }
    p->trace[p->trace_n++] = *r;
}
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static int global_chi(prob_t* p, double* chi) { return (p->world > 1) ? exchange(p, chi, 1, 0) : PLBA_OK; }

/* SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve (App. A.2, A.3) */
static double g_t[6];
#define TPH(i, stmt) do { double t__ = now_ms(); stmt; g_t[i] += now_ms() - t__; } while (0)
int orc_optimize(plba_problem* p, int max_iters, const volatile uint8_t* abort_flag, plba_stats* out) {
    if (!p) return PLBA_ERR_INVALID;
    if (!p->have_cam || !p->K) FAIL(p, PLBA_ERR_STATE, "camera and keyframes must be set before optimize");
    double t0 = now_ms();
    plba_stats st;
    memset(&st, 0, sizeof st);
    p->trace_n = 0;
    TPH(0, build_index(p));
    double lambda = 0.0, ni = 2.0;
    int rc, ok = 1;
    for (int it = 0; it < max_iters && !(abort_flag && *abort_flag) && ok; ++it) {
        double currentChi;
        TPH(1, rc = compute_active_errors(p, &currentChi)); if (rc) return rc;
        if ((rc = global_chi(p, &currentChi))) return rc;
        double tempChi = currentChi;
        if (it == 0) st.chi2_initial = currentChi;
        TPH(2, build_system(p));
        if (it == 0) {
            if (p->opt.user_lambda_init > 0) lambda = p->opt.user_lambda_init;
            else { double md; if ((rc = max_diag(p, &md))) return rc; lambda = p->opt.tau * md; }
            ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
            state_push(p);
            int ok2 = 1;
            TPH(3, rc = schur_solve(p, lambda, 1, &ok2)); if (rc) return rc;
            if (!ok2) st.solver_failures++;
            TPH(4, state_update(p));
            TPH(1, rc = compute_active_errors(p, &tempChi)); if (rc) return rc;
            if ((rc = global_chi(p, &tempChi))) return rc;
            if (!ok2) tempChi = DBL_MAX;
            /* computeScale: sum x_j (lambda x_j + b_j) over poses and landmarks */
            double scale = 0.0, lscale = 0.0;
            for (int j = 0; j < p->P; ++j) scale += p->x[j] * (lambda * p->x[j] + p->bpg[j]);
            for (int j = 0; j < p->Ldim; ++j) lscale += p->x[p->P + j] * (lambda * p->x[p->P + j] + p->bl_all[j]);
            if ((rc = global_chi(p, &lscale))) return rc;
            scale += lscale;
            scale += 1e-3;
            rho = (currentChi - tempChi) / scale;
            plba_trace_row tr = {it, qmax, 0, ok2, lambda, currentChi, tempChi, scale, rho};
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = alpha < p->opt.good_step_upper ? alpha : p->opt.good_step_upper;
                double sf = p->opt.good_step_lower > alpha ? p->opt.good_step_lower : alpha;
                lambda *= sf;
                ni = 2;
                currentChi = tempChi;
                tr.accepted = 1;
            } else {
                lambda *= ni;
                ni *= 2;
                state_pop(p);
                if (!isfinite(lambda)) { trace_add(p, &tr); st.trials++; break; }
            }
            trace_add(p, &tr);
            st.trials++;
            qmax++;
        } while (rho < 0 && qmax < p->opt.max_trials && !(abort_flag && *abort_flag));
        st.iterations++;
        st.chi2_final = currentChi;
        if (qmax == p->opt.max_trials || rho == 0 || !isfinite(lambda)) { ok = 0; st.stop_reason = 1; }
    }
    if (abort_flag && *abort_flag && st.stop_reason == 0 && st.iterations < max_iters) st.stop_reason = 2;
    if (max_iters == 0 || st.iterations == 0) {
        double c;
        if ((rc = compute_active_errors(p, &c))) return rc;
        if ((rc = global_chi(p, &c))) return rc;
        st.chi2_initial = st.chi2_final = c;
    }
    st.lambda_final = lambda;
    st.ms_total = now_ms() - t0;
    if (getenv("PLBA_ORC_TIMING")) fprintf(stderr, "[oracle] index %.1f errors %.1f build %.1f schur+solve %.1f update %.1f ms (cumulative)\n", g_t[0], g_t[1], g_t[2], g_t[3], g_t[4]);
    if (out) *out = st;
    return PLBA_OK;
}

int orc_recompute_errors(plba_problem* p) {
    if (!p) return PLBA_ERR_INVALID;
    if (!p->off_pvr) build_index(p);
    double c;
    int rc = compute_active_errors(p, &c);
    p->chi2_last = c;
    return rc;
}
int orc_get_edge_chi2(plba_problem* p, plba_edge_kind kind, double* chi2, uint8_t* dpos) {
    if (!p) return PLBA_ERR_INVALID;
    if (kind == PLBA_EDGE_POINT) {
        for (int e = 0; e < p->Ep; ++e) {
            const double* er = p->po_err + 2 * e;
            if (chi2) chi2[e] = p->po_w[e] * (er[0] * er[0] + er[1] * er[1]);
            if (dpos) { double t[2]; int d; point_error(p, &p->ns[p->po_kf[e]], p->pt + 3 * p->po_pt[e], p->po_uv + 2 * e, t, &d); dpos[e] = (uint8_t)d; }
        }
    } else if (kind == PLBA_EDGE_LINE) {
        for (int e = 0; e < p->El; ++e) {
            const double* er = p->lo_err + 3 * e;
            if (chi2) chi2[e] = p->lo_w[e] * (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]);
            if (dpos) { double t[3]; int d; line_error(p, &p->ns[p->lo_kf[e]], p->ln + 6 * p->lo_ln[e], p->lo_l + 3 * e, t, &d); dpos[e] = (uint8_t)d; }
        }
    } else if (kind == PLBA_EDGE_IMU_PVR) {
        for (int m = 0; m < p->M; ++m) { if (chi2) chi2[m] = quad_form(p->im_err_pvr + 9 * m, p->im_info_pvr + 81 * m, 9); if (dpos) dpos[m] = 1; }
    } else if (kind == PLBA_EDGE_IMU_BIAS) {
        for (int m = 0; m < p->M; ++m) { if (chi2) chi2[m] = quad_form(p->im_err_bias + 6 * m, p->im_info_bias + 36 * m, 6); if (dpos) dpos[m] = 1; }
    } else if (kind == PLBA_EDGE_PRIOR) {
        double c = 0; for (int i = 0; i < p->pr_n; ++i) c += p->pr_err[i] * p->pr_err[i];
        if (chi2) chi2[0] = c;
        if (dpos) dpos[0] = 1;
    } else return PLBA_ERR_INVALID;
    return PLBA_OK;
}
/* mapHandler.cpp:6047-6066: chi2()>thresh || !isDepthPositive() => setLevel(1); setRobustKernel(0) on all */
int orc_gate_outliers(plba_problem* p, double thresh, int* np_out, int* nl_out) {
    if (!p) return PLBA_ERR_INVALID;
    int np = 0, nl = 0;
    for (int e = 0; e < p->Ep; ++e) {
        const double* er = p->po_err + 2 * e;
        double c = p->po_w[e] * (er[0] * er[0] + er[1] * er[1]), t[2]; int d;
        point_error(p, &p->ns[p->po_kf[e]], p->pt + 3 * p->po_pt[e], p->po_uv + 2 * e, t, &d);
        if (c > thresh || !d) { if (!p->po_level[e]) np++; p->po_level[e] = 1; }
    }
    for (int e = 0; e < p->El; ++e) {
        const double* er = p->lo_err + 3 * e;
        double c = p->lo_w[e] * (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]), t[3]; int d;
        line_error(p, &p->ns[p->lo_kf[e]], p->ln + 6 * p->lo_ln[e], p->lo_l + 3 * e, t, &d);
        if (c > thresh || !d) { if (!p->lo_level[e]) nl++; p->lo_level[e] = 1; }
    }
    p->rob_on[PLBA_EDGE_POINT] = 0;
    p->rob_on[PLBA_EDGE_LINE] = 0;
    if (np_out) *np_out = np;
    if (nl_out) *nl_out = nl;
    return np + nl;
}
/* mapHandler.cpp:5541-5556 (points), :5611-5620 (lines): level-1 edges get computeError() first, then
 * chi2() > thresh || !isDepthPositive() marks the observation for removal */
int orc_cull_observations(plba_problem* p, double thresh, uint8_t* bad_point, uint8_t* bad_line, int* np_out, int* nl_out) {
    if (!p) return PLBA_ERR_INVALID;
    int np = 0, nl = 0;
    for (int e = 0; e < p->Ep; ++e) {
        double t[2]; int d;
        point_error(p, &p->ns[p->po_kf[e]], p->pt + 3 * p->po_pt[e], p->po_uv + 2 * e, t, &d);
        if (p->po_level[e]) { p->po_err[2 * e] = t[0]; p->po_err[2 * e + 1] = t[1]; }      /* e->computeError() */
        const double* er = p->po_err + 2 * e;
        const double c = p->po_w[e] * (er[0] * er[0] + er[1] * er[1]);
        const int bad = (c > thresh || !d);
        if (bad_point) bad_point[e] = (uint8_t)bad;
        np += bad;
    }
    for (int e = 0; e < p->El; ++e) {
        double t[3]; int d;
        line_error(p, &p->ns[p->lo_kf[e]], p->ln + 6 * p->lo_ln[e], p->lo_l + 3 * e, t, &d);
        if (p->lo_level[e]) { p->lo_err[3 * e] = t[0]; p->lo_err[3 * e + 1] = t[1]; p->lo_err[3 * e + 2] = t[2]; }
        const double* er = p->lo_err + 3 * e;
        const double c = p->lo_w[e] * (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]);
        const int bad = (c > thresh || !d);
        if (bad_line) bad_line[e] = (uint8_t)bad;
        nl += bad;
    }
    if (np_out) *np_out = np;
    if (nl_out) *nl_out = nl;
    return np + nl;
}
int orc_get_trace(plba_problem* p, plba_trace_row* rows, int cap, int* n) {
    if (!p) return PLBA_ERR_INVALID;
    int c = p->trace_n < cap ? p->trace_n : cap;
    if (rows && c > 0) memcpy(rows, p->trace, sizeof(plba_trace_row) * c);
    if (n) *n = p->trace_n;
    return PLBA_OK;
}
int orc_get_keyframes(plba_problem* p, double* P3, double* V3, double* q4, double* dbg3, double* dba3) {
    if (!p) return PLBA_ERR_INVALID;
    for (int k = 0; k < p->K; ++k) {
        if (P3) memcpy(P3 + 3 * k, p->ns[k].P, 24);
        if (V3) memcpy(V3 + 3 * k, p->ns[k].V, 24);
        if (q4) memcpy(q4 + 4 * k, p->ns[k].q, 32);
        if (dbg3) memcpy(dbg3 + 3 * k, p->ns[k].dbg, 24);
        if (dba3) memcpy(dba3 + 3 * k, p->ns[k].dba, 24);
    }
    return PLBA_OK;
}
int orc_get_points(plba_problem* p, double* xyz) { if (!p || !xyz) return PLBA_ERR_INVALID; memcpy(xyz, p->pt, 24 * (size_t)p->Np); return PLBA_OK; }
int orc_get_lines(plba_problem* p, double* l) { if (!p || !l) return PLBA_ERR_INVALID; memcpy(l, p->ln, 48 * (size_t)p->Nl); return PLBA_OK; }
int orc_save_state(plba_problem* p) {
    if (!p) return PLBA_ERR_INVALID;
    memcpy(p->ns_saved, p->ns, sizeof(nav_t) * p->K);
    memcpy(p->pt_saved, p->pt, 24 * (size_t)p->Np);
    memcpy(p->ln_saved, p->ln, 48 * (size_t)p->Nl);
    return PLBA_OK;
}
int orc_restore_state(plba_problem* p) {
    if (!p) return PLBA_ERR_INVALID;
    memcpy(p->ns, p->ns_saved, sizeof(nav_t) * p->K);
    memcpy(p->pt, p->pt_saved, 24 * (size_t)p->Np);
    memcpy(p->ln, p->ln_saved, 48 * (size_t)p->Nl);
    return PLBA_OK;
}

int orc_debug_build(plba_problem* p, double lambda, int do_solve) {
    if (!p) return PLBA_ERR_INVALID;
    build_index(p);
    int rc, ok;
    double chi;
    if ((rc = compute_active_errors(p, &chi))) return rc;
    if ((rc = global_chi(p, &chi))) return rc;
    p->chi2_last = chi;
    build_system(p);
    if ((rc = max_diag(p, &p->maxdiag_last))) return rc;
    return schur_solve(p, lambda, do_solve, &ok);
}
int orc_debug_get(plba_problem* p, const char* what, double* out, size_t cap, size_t* n) {
    if (!p || !what) return PLBA_ERR_INVALID;
    const double* src = NULL;
    size_t cnt = 0;
    double tmp;
    size_t P = p->P;
    if (!strcmp(what, "Hschur")) { src = p->Hs; cnt = P * P; }
    else if (!strcmp(what, "bschur")) { src = p->bs; cnt = P; }
    else if (!strcmp(what, "bp")) { src = p->bpg; cnt = P; }
    else if (!strcmp(what, "x")) { src = p->x; cnt = P + p->Ldim; }
    else if (!strcmp(what, "hll_pt")) { src = p->Hll_pt; cnt = (size_t)p->Np * 9; }
    else if (!strcmp(what, "bl_pt")) { src = p->bl_pt; cnt = (size_t)p->Np * 3; }
    else if (!strcmp(what, "hll_ln")) { src = p->Hll_ln; cnt = (size_t)p->Nl * 36; }
    else if (!strcmp(what, "bl_ln")) { src = p->bl_ln; cnt = (size_t)p->Nl * 6; }
    else if (!strcmp(what, "err_pvr")) { src = p->im_err_pvr; cnt = (size_t)p->M * 9; }
    else if (!strcmp(what, "err_bias")) { src = p->im_err_bias; cnt = (size_t)p->M * 6; }
    else if (!strcmp(what, "err_prior")) { src = p->pr_err; cnt = p->pr_nv ? p->pr_n : 0; }
    else if (!strcmp(what, "err_pt")) { src = p->po_err; cnt = (size_t)p->Ep * 2; }
    else if (!strcmp(what, "err_ln")) { src = p->lo_err; cnt = (size_t)p->El * 3; }
    else if (!strcmp(what, "pose_dim")) { tmp = (double)p->P; src = &tmp; cnt = 1; }
    else if (!strcmp(what, "chi2")) { tmp = p->chi2_last; src = &tmp; cnt = 1; }
    else if (!strcmp(what, "maxdiag")) { tmp = p->maxdiag_last; src = &tmp; cnt = 1; }
    else FAIL(p, PLBA_ERR_INVALID, "debug_get: unknown buffer '%s'", what);
    if (n) *n = cnt;
    if (out) { size_t c = cnt < cap ? cnt : cap; if (c) memcpy(out, src, c * 8); }
    return PLBA_OK;
}

int orc_debug_dense_solve(plba_problem* p, int n, const double* A, const double* b, double* x, int* ok) {
    if (!p || n <= 0 || !A || !b || !x) return PLBA_ERR_INVALID;
    double* L = (double*)xdup(A, (size_t)n * n * 8);
    int good = chol_factor(L, n);
    if (good) chol_solve(L, n, b, x);
    free(L);
    if (ok) *ok = good;
    return PLBA_OK;
}
int orc_dense_solve(plba_problem* p, int n, const double* A, const double* b, double* x, int* ok) { return orc_debug_dense_solve(p, n, A, b, x, ok); }

/* ============================================================================================
 * 7. marginalization (IMU/marginalization.cpp:38-147, 291-384; call site mapHandler.cpp:6075-6199)
 * ========================================================================================== */

/* cyclic Jacobi eigen-decomposition of a symmetric matrix: A = V diag(w) V^T, w ascending
 * (stands in for Eigen::SelfAdjointEigenSolver, IMU/marginalization.cpp:352,364). V columns = vectors */
static void sym_eig(const double* Ain, int n, double* w, double* V) {
    double* A = (double*)xdup(Ain, (size_t)n * n * 8);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) { diag += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
        if (off <= 1e-32 * (diag + off) || off == 0.0) break;
        for (int pI = 0; pI < n - 1; ++pI)
            for (int q = pI + 1; q < n; ++q) {
                double apq = A[(size_t)pI * n + q];
                if (apq == 0.0) continue;
                double app = A[(size_t)pI * n + pI], aqq = A[(size_t)q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) { /* columns p,q */
                    double akp = A[(size_t)k * n + pI], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + pI] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) { /* rows p,q */
                    double apk = A[(size_t)pI * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)pI * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V[(size_t)k * n + pI], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + pI] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
    /* sort ascending like SelfAdjointEigenSolver */
    for (int i = 0; i < n - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j) if (w[j] < w[m]) m = j;
        if (m != i) {
            double t = w[i]; w[i] = w[m]; w[m] = t;
            for (int k = 0; k < n; ++k) { double u = V[(size_t)k * n + i]; V[(size_t)k * n + i] = V[(size_t)k * n + m]; V[(size_t)k * n + m] = u; }
        }
    }
    free(A);
}

typedef struct { /* one ResidualBlockInfo (IMU/marginalization.h:29-59) with the B-Q3 fix: own (r,J) at the final estimate */
    int nv;
    int vid[64];      /* global parameter id: vertex id for KF vertices, (1<<28)+pt, (1<<29)+ln for landmarks */
    int size[64];
    int drop[64];
    double x0[64][10];
    int nx0[64];
    int dim;
    double* r;        /* dim */
    double* J[64];    /* dim x size[i], row-major */
} factor_t;
#define PID_PT(i) ((1 << 28) + (i))
#define PID_LN(i) ((1 << 29) + (i))

static void est_pvr(const nav_t* s, double* d) { /* GetEstData PVR part: P,V,quat(x,y,z,w) of Quaterniond(Rwb) h:127-168 */
    double R[9], q[4];
    memcpy(d, s->P, 24); memcpy(d + 3, s->V, 24);
    nav_RotMatrix(s, R);
    R_to_q(R, q);
    memcpy(d + 6, q, 32);
}
static void est_bias(const nav_t* s, double* d) {
    for (int c = 0; c < 3; ++c) { d[c] = s->bg[c] + s->dbg[c]; d[3 + c] = s->ba[c] + s->dba[c]; }
}
static void factor_free(factor_t* f) { free(f->r); for (int i = 0; i < f->nv; ++i) free(f->J[i]); }

int orc_marginalize(plba_problem* p, int first_kf, int max_edges, plba_prior* out) {
    if (!p || !out || first_kf < 0 || first_kf >= p->K) return PLBA_ERR_INVALID;
    memset(out, 0, sizeof *out);
    int NUM = max_edges;
    int cap = 2 + 2 * (NUM + 2) + 1;
    factor_t* F = (factor_t*)xcalloc(cap, sizeof(factor_t));
    int nf = 0;
    int vid0 = p->vid_pvr[first_kf], vidb0 = p->vid_bias[first_kf];
    /* --- IMU PVR edge [0] and bias edge [0] (mapHandler.cpp:6078-6107): first created IMU edge --- */
    if (p->M > 0) {
        int m = 0, ki = p->im_i[m], kj = p->im_j[m];
        const nav_t *si = &p->ns[ki], *sj = &p->ns[kj];
        factor_t* f = &F[nf++];
        f->nv = 3; f->dim = 9;
        f->vid[0] = p->vid_pvr[ki]; f->vid[1] = p->vid_pvr[kj]; f->vid[2] = p->vid_bias[ki];
        f->size[0] = 9; f->size[1] = 9; f->size[2] = 6;
        f->drop[0] = 1; /* drop_set = {0} */
        est_pvr(si, f->x0[0]); f->nx0[0] = 10; est_pvr(sj, f->x0[1]); f->nx0[1] = 10; est_bias(si, f->x0[2]); f->nx0[2] = 6;
        f->r = (double*)xcalloc(9, 8);
        f->J[0] = (double*)xcalloc(81, 8); f->J[1] = (double*)xcalloc(81, 8); f->J[2] = (double*)xcalloc(54, 8);
        pvr_error(p, si, sj, si, &p->im_pre[m], f->r);
        pvr_linearize(p, si, sj, si, &p->im_pre[m], f->r, f->J[0], f->J[1], f->J[2]);
        f = &F[nf++];
        f->nv = 2; f->dim = 6;
        f->vid[0] = p->vid_bias[ki]; f->vid[1] = p->vid_bias[kj];
        f->size[0] = 6; f->size[1] = 6; f->drop[0] = 1;
        est_bias(si, f->x0[0]); f->nx0[0] = 6; est_bias(sj, f->x0[1]); f->nx0[1] = 6;
        f->r = (double*)xcalloc(6, 8);
        f->J[0] = (double*)xcalloc(36, 8); f->J[1] = (double*)xcalloc(36, 8);
        bias_error(si, sj, f->r);
        for (int i = 0; i < 6; ++i) { f->J[0][i * 6 + i] = -1.0; f->J[1][i * 6 + i] = 1.0; }
    }
    /* --- point edges whose landmark was first observed in the first KF (:6109-6136), <= NUM+1 edges --- */
    {
        int num = 0, e = 0;
        for (int l = 0; l < p->Np && num <= NUM; ++l) {
            int e0 = e;
            while (e < p->Ep && p->po_pt[e] == l) ++e;
            if (e0 == e || p->po_kf[e0] != first_kf) continue;
            for (int a = e0; a < e; ++a) {
                int k = p->po_kf[a];
                factor_t* f = &F[nf++];
                f->nv = 2; f->dim = 2;
                f->vid[0] = PID_PT(l); f->vid[1] = p->vid_pvr[k];
                f->size[0] = 3; f->size[1] = 9;
                f->drop[0] = 1; f->drop[1] = (p->vid_pvr[k] == vid0);
                memcpy(f->x0[0], p->pt + 3 * l, 24); f->nx0[0] = 3; est_pvr(&p->ns[k], f->x0[1]); f->nx0[1] = 10;
                f->r = (double*)xcalloc(2, 8);
                f->J[0] = (double*)xcalloc(6, 8); f->J[1] = (double*)xcalloc(18, 8);
                point_error(p, &p->ns[k], p->pt + 3 * l, p->po_uv + 2 * a, f->r, NULL);
                point_linearize(p, &p->ns[k], p->pt + 3 * l, f->J[0], f->J[1]);
                num++;
                if (num > NUM) break; /* `num>NUM` after increment admits NUM+1 (B-Q10) */
            }
        }
    }
    /* --- line edges (:6138-6165) --- */
    {
        int num = 0, e = 0;
        for (int l = 0; l < p->Nl && num <= NUM; ++l) {
            int e0 = e;
            while (e < p->El && p->lo_ln[e] == l) ++e;
            if (e0 == e || p->lo_kf[e0] != first_kf) continue;
            for (int a = e0; a < e; ++a) {
                int k = p->lo_kf[a];
                factor_t* f = &F[nf++];
                f->nv = 2; f->dim = 3;
                f->vid[0] = PID_LN(l); f->vid[1] = p->vid_pvr[k];
                f->size[0] = 6; f->size[1] = 9;
                f->drop[0] = 1; f->drop[1] = (p->vid_pvr[k] == vid0);
                memcpy(f->x0[0], p->ln + 6 * l, 48); f->nx0[0] = 6; est_pvr(&p->ns[k], f->x0[1]); f->nx0[1] = 10;
                f->r = (double*)xcalloc(3, 8);
                f->J[0] = (double*)xcalloc(18, 8); f->J[1] = (double*)xcalloc(27, 8);
                line_error(p, &p->ns[k], p->ln + 6 * l, p->lo_l + 3 * a, f->r, NULL);
                line_linearize(p, &p->ns[k], p->ln + 6 * l, p->lo_l + 3 * a, f->J[0], f->J[1]);
                num++;
                if (num > NUM) break;
            }
        }
    }
    /* --- old prior edge (:6167-6188) --- */
    if (p->pr_nv) {
        if (p->pr_nv > 64) { for (int i = 0; i < nf; ++i) factor_free(&F[i]); free(F); FAIL(p, PLBA_ERR_INVALID, "oracle: prior with more than 64 vertices"); }
        factor_t* f = &F[nf++];
        int n = p->pr_n;
        f->nv = p->pr_nv; f->dim = n;
        f->r = (double*)xcalloc(n, 8);
        int rc = prior_error(p, f->r);
        if (rc) { for (int i = 0; i < nf; ++i) factor_free(&F[i]); free(F); return rc; }
        for (int i = 0; i < p->pr_nv; ++i) {
            int isb = 0, k = find_kf_by_vid(p, p->pr_vid[i], &isb);
            f->vid[i] = p->pr_vid[i]; f->size[i] = p->pr_size[i];
            f->drop[i] = (p->pr_vid[i] == vid0 || p->pr_vid[i] == vidb0);
            if (isb) { est_bias(&p->ns[k], f->x0[i]); f->nx0[i] = 6; } else { est_pvr(&p->ns[k], f->x0[i]); f->nx0[i] = 10; }
            int s = f->size[i], ix = p->pr_idx[i];
            f->J[i] = (double*)xcalloc((size_t)n * s, 8);
            for (int r = 0; r < n; ++r) for (int c = 0; c < s; ++c) f->J[i][(size_t)r * s + c] = p->pr_J0[(size_t)(ix + c) * n + r];
        }
    }
    /* --- parameter ordering: dropped first then kept, each by ascending id (B-Q6) --- */
    int np = 0, pcap = 4 * cap + 64;
    int* pid = (int*)xcalloc(pcap, sizeof(int));
    int* psz = (int*)xcalloc(pcap, sizeof(int));
    int* pdrop = (int*)xcalloc(pcap, sizeof(int));
    double(*px0)[10] = (double(*)[10])xcalloc(pcap, sizeof(double[10]));
    int* pnx0 = (int*)xcalloc(pcap, sizeof(int));
    for (int i = 0; i < nf; ++i)
        for (int v = 0; v < F[i].nv; ++v) {
            int j;
            for (j = 0; j < np; ++j) if (pid[j] == F[i].vid[v]) break;
            if (j == np) { pid[np] = F[i].vid[v]; psz[np] = F[i].size[v]; pdrop[np] = 0; np++; }
            else if (psz[j] != F[i].size[v]) { FAIL(p, PLBA_ERR_INVALID, "Wrong! The same param block with different param size"); }
            if (F[i].drop[v]) pdrop[j] = 1;
            memcpy(px0[j], F[i].x0[v], 80); pnx0[j] = F[i].nx0[v]; /* later factors overwrite (marginalization.cpp:118-119) */
        }
    /* sort by (kept, id) */
    int* order = (int*)xcalloc(np, sizeof(int));
    for (int i = 0; i < np; ++i) order[i] = i;
    for (int i = 0; i < np - 1; ++i)
        for (int j = i + 1; j < np; ++j) {
            int a = order[i], b = order[j];
            int ka = !pdrop[a], kb = !pdrop[b];
            if (ka > kb || (ka == kb && pid[a] > pid[b])) { order[i] = b; order[j] = a; }
        }
    int* pidx = (int*)xcalloc(np, sizeof(int));
    int pos = 0, m = 0;
    for (int i = 0; i < np; ++i) { int a = order[i]; if (pdrop[a]) { pidx[a] = pos; pos += psz[a]; } }
    m = pos;
    int nv_keep = 0;
    for (int i = 0; i < np; ++i) { int a = order[i]; if (!pdrop[a]) { pidx[a] = pos; pos += psz[a]; nv_keep++; } }
    int n = pos - m;
    /* --- A = sum J^T J, b = sum J^T r (ThreadsConstructA, marginalization.cpp:8-36; raw J, no Omega, B-Q4) --- */
    double* A = (double*)xcalloc((size_t)pos * pos, 8);
    double* b = (double*)xcalloc(pos, 8);
    for (int fi = 0; fi < nf; ++fi) {
        factor_t* f = &F[fi];
        for (int i = 0; i < f->nv; ++i) {
            int ai; for (ai = 0; ai < np; ++ai) if (pid[ai] == f->vid[i]) break;
            int ii = pidx[ai], si = f->size[i];
            for (int j = i; j < f->nv; ++j) {
                int aj; for (aj = 0; aj < np; ++aj) if (pid[aj] == f->vid[j]) break;
                int ij = pidx[aj], sj = f->size[j];
                for (int c = 0; c < si; ++c)
                    for (int d = 0; d < sj; ++d) {
                        double s = 0.0;
                        for (int r = 0; r < f->dim; ++r) s += f->J[i][(size_t)r * si + c] * f->J[j][(size_t)r * sj + d];
                        A[(size_t)(ii + c) * pos + ij + d] += s;
                        if (i != j) A[(size_t)(ij + d) * pos + ii + c] = A[(size_t)(ii + c) * pos + ij + d];
                    }
            }
            for (int c = 0; c < si; ++c) {
                double s = 0.0;
                for (int r = 0; r < f->dim; ++r) s += f->J[i][(size_t)r * si + c] * f->r[r];
                b[ii + c] += s;
            }
        }
    }
    /* --- Amm pseudo-inverse, Schur (marginalization.cpp:351-362) --- */
    double eps = p->opt.marg_eps;
    double* Amm = (double*)xcalloc((size_t)m * m + 1, 8);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Amm[(size_t)i * m + j] = 0.5 * (A[(size_t)i * pos + j] + A[(size_t)j * pos + i]);
    double* wv = (double*)xcalloc(m + 1, 8);
    double* Vm = (double*)xcalloc((size_t)m * m + 1, 8);
    if (m) sym_eig(Amm, m, wv, Vm);
    double* Ainv = (double*)xcalloc((size_t)m * m + 1, 8);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int k = 0; k < m; ++k) { double iv = (wv[k] > eps) ? 1.0 / wv[k] : 0.0; s += Vm[(size_t)i * m + k] * iv * Vm[(size_t)j * m + k]; }
            Ainv[(size_t)i * m + j] = s;
        }
    /* T = Arm * Amm_inv  (n x m) */
    double* T = (double*)xcalloc((size_t)n * m + 1, 8);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int k = 0; k < m; ++k) s += A[(size_t)(m + i) * pos + k] * Ainv[(size_t)k * m + j];
            T[(size_t)i * m + j] = s;
        }
    double* Ar = (double*)xcalloc((size_t)n * n + 1, 8);
    double* br = (double*)xcalloc(n + 1, 8);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < m; ++k) s += T[(size_t)i * m + k] * A[(size_t)k * pos + m + j];
            Ar[(size_t)i * n + j] = A[(size_t)(m + i) * pos + m + j] - s;
        }
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += T[(size_t)i * m + k] * b[k];
        br[i] = b[m + i] - s;
    }
    /* --- eigen-sqrt into (J0, r0) (marginalization.cpp:364-372) --- */
    double* w2 = (double*)xcalloc(n + 1, 8);
    double* V2 = (double*)xcalloc((size_t)n * n + 1, 8);
    if (n) sym_eig(Ar, n, w2, V2);
    out->n = n; out->m = m; out->nv = nv_keep;
    out->vid = (int32_t*)xcalloc(nv_keep, 4); out->size = (int32_t*)xcalloc(nv_keep, 4); out->idx = (int32_t*)xcalloc(nv_keep, 4);
    out->J0 = (double*)xcalloc((size_t)n * n + 1, 8); out->r0 = (double*)xcalloc(n + 1, 8);
    out->Ar = Ar; out->br = br;
    int nx = 0, kv = 0;
    for (int i = 0; i < np; ++i) { int a = order[i]; if (!pdrop[a]) nx += (psz[a] == 9) ? 10 : 6; }
    out->x0 = (double*)xcalloc(nx + 1, 8);
    nx = 0;
    for (int i = 0; i < np; ++i) {
        int a = order[i];
        if (pdrop[a]) continue;
        if (psz[a] != 9 && psz[a] != 6) { FAIL(p, PLBA_ERR_INVALID, "kept parameter %d is a landmark: only PVR/bias vertices may be kept", pid[a]); }
        out->vid[kv] = pid[a]; out->size[kv] = psz[a]; out->idx[kv] = pidx[a] - m;
        int c = (psz[a] == 9) ? 10 : 6;
        memcpy(out->x0 + nx, px0[a], 8 * (size_t)c);
        nx += c; kv++;
    }
    for (int r = 0; r < n; ++r) { /* J0 = diag(sqrt(S)) V^T (colmajor); r0 = diag(sqrt(S_inv)) V^T b' */
        double S = (w2[r] > eps) ? w2[r] : 0.0, Si = (w2[r] > eps) ? 1.0 / w2[r] : 0.0;
        double ss = sqrt(S), sis = sqrt(Si), acc = 0.0;
        for (int c = 0; c < n; ++c) { out->J0[(size_t)c * n + r] = ss * V2[(size_t)c * n + r]; acc += V2[(size_t)c * n + r] * br[c]; }
        out->r0[r] = sis * acc;
    }
    for (int i = 0; i < nf; ++i) factor_free(&F[i]);
    free(F); free(pid); free(psz); free(pdrop); free(px0); free(pnx0); free(order); free(pidx);
    free(A); free(b); free(Amm); free(wv); free(Vm); free(Ainv); free(T); free(w2); free(V2);
    return PLBA_OK;
}
/* oracle counterpart of plba_marginalize_factors: only the reference's own selection is restated, so the general
 * entry accepts exactly the factor set orc_marginalize would pick for the dropped keyframe and defers to it. */
int orc_marginalize_factors(plba_problem* p, int n_imu, const int32_t* imu_edges, int n_pt, const int32_t* point_edges,
                            int n_ln, const int32_t* line_edges, int use_prior, int n_drop, const int32_t* drop_vid, plba_prior* out) {
    (void)imu_edges; (void)point_edges; (void)line_edges; (void)use_prior;
    if (!p || !out || n_drop < 1) return PLBA_ERR_INVALID;
    int first_kf = -1;
    for (int k = 0; k < p->K; ++k) if (p->vid_pvr[k] == drop_vid[0]) first_kf = k;
    if (first_kf < 0) FAIL(p, PLBA_ERR_INVALID, "marginalize_factors: dropped vertex %d is not a PVR vertex of the window", drop_vid[0]);
    int num_pt = n_pt > 0 ? n_pt - 1 : 0, num_ln = n_ln > 0 ? n_ln - 1 : 0;
    (void)n_imu; (void)num_ln;
    return orc_marginalize(p, first_kf, num_pt > num_ln ? num_pt : num_ln, out);
}
void orc_prior_free(plba_prior* pr) {
    if (!pr) return;
    free(pr->vid); free(pr->size); free(pr->idx); free(pr->x0); free(pr->J0); free(pr->r0); free(pr->Ar); free(pr->br);
    memset(pr, 0, sizeof *pr);
}

/* ============================================================================================
 * 8. stand-alone evaluators exported for the unit tests (finite differences, known answers)
 * ========================================================================================== */
static void fill_cam(prob_t* p, const double* cam /*fx fy cx cy Rbc9 Pbc3*/) {
    memset(p, 0, sizeof *p);
    p->fx = cam[0]; p->fy = cam[1]; p->cx = cam[2]; p->cy = cam[3];
    memcpy(p->Rbc, cam + 4, 72); memcpy(p->Pbc, cam + 13, 24);
}
static void fill_nav(nav_t* s, const double* v /*P3 V3 q4 bg3 ba3 dbg3 dba3 = 22*/) {
    memcpy(s->P, v, 24); memcpy(s->V, v + 3, 24); memcpy(s->q, v + 6, 32);
    memcpy(s->bg, v + 10, 24); memcpy(s->ba, v + 13, 24); memcpy(s->dbg, v + 16, 24); memcpy(s->dba, v + 19, 24);
}
static void dump_nav(const nav_t* s, double* v) {
    memcpy(v, s->P, 24); memcpy(v + 3, s->V, 24); memcpy(v + 6, s->q, 32);
    memcpy(v + 10, s->bg, 24); memcpy(v + 13, s->ba, 24); memcpy(v + 16, s->dbg, 24); memcpy(v + 19, s->dba, 24);
}
void orc_eval_point_edge(const double* cam, const double* nav22, const double* Pw, const double* obs, double* err2, double* Ji6, double* Jj18, int* dpos) {
    prob_t p; nav_t s;
    fill_cam(&p, cam); fill_nav(&s, nav22);
    point_error(&p, &s, Pw, obs, err2, dpos);
    if (Ji6 && Jj18) point_linearize(&p, &s, Pw, Ji6, Jj18);
}
void orc_eval_line_edge(const double* cam, const double* nav22, const double* L6, const double* obs3, int fix_q1, double* err3, double* Ji18, double* Jj27, int* dpos) {
    prob_t p; nav_t s;
    fill_cam(&p, cam); fill_nav(&s, nav22);
    p.opt.fix_line_position_jacobian = fix_q1;
    line_error(&p, &s, L6, obs3, err3, dpos);
    if (Ji18 && Jj27) line_linearize(&p, &s, L6, obs3, Ji18, Jj27);
}
/* EdgeNavStateLinePoint::computeError (IMU/g2otypes.h:929-939): the edge of the reference's test/test.cpp */
void orc_eval_linepoint_edge(const double* cam, const double* nav22, const double* Pw, const double* obs3, double* err3) {
    prob_t p; nav_t s; double Pc[3], uv[2];
    fill_cam(&p, cam); fill_nav(&s, nav22);
    cam_Pc(&p, &s, Pw, Pc, NULL);
    cam_project(&p, Pc, uv);
    err3[0] = obs3[0] * uv[0] + obs3[1] * uv[1] + obs3[2];
    err3[1] = 0; err3[2] = 0;
}
void orc_eval_pvr_edge(const double* gw, const double* navi22, const double* navj22, const double* navb22, const double* pre142, double* err9,
                       double* J0_81, double* J1_81, double* J2_54) {
    prob_t p; nav_t si, sj, sb; preint_t M;
    memset(&p, 0, sizeof p);
    memcpy(p.gw, gw, 24);
    fill_nav(&si, navi22); fill_nav(&sj, navj22); fill_nav(&sb, navb22);
    memcpy(&M, pre142, 142 * 8);
    pvr_error(&p, &si, &sj, &sb, &M, err9);
    if (J0_81) pvr_linearize(&p, &si, &sj, &sb, &M, err9, J0_81, J1_81, J2_54);
}
void orc_nav_oplus_pvr(const double* nav22, const double* u9, double* out22) { nav_t s; fill_nav(&s, nav22); nav_IncSmallPVR(&s, u9); dump_nav(&s, out22); }
void orc_nav_oplus_bias(const double* nav22, const double* u6, double* out22) { nav_t s; fill_nav(&s, nav22); nav_IncSmallBias(&s, u6); dump_nav(&s, out22); }
void orc_so3_exp(const double* w, double* q) { so3_exp(w, q); }
void orc_so3_log(const double* q, double* w) { so3_log(q, w); }
void orc_so3_jr(const double* w, double* J) { so3_Jr(w, J); }
void orc_so3_jrinv(const double* w, double* J) { so3_JrInv(w, J); }
void orc_quat_to_R(const double* q, double* R) { q_to_R(q, R); }
void orc_R_to_quat(const double* R, double* q) { R_to_q(R, q); }
void orc_huber(double e, double delta, double* rho3) { huber(e, delta, rho3); }
void orc_sym_eig(const double* A, int n, double* w, double* V) { sym_eig(A, n, w, V); }

/* IMUPreintegrator::update (IMU/IMUPreintegrator.cpp:80-139), state = the 142-double payload.
 * gyr_cov / acc_cov = IMUData::_gyrMeasCov / _accMeasCov diagonal value (imudata.cpp:27-28). */
void orc_preint_update(double* pre142, const double* omega, const double* acc, double dt, double gyr_cov, double acc_cov) {
    preint_t M;
    memcpy(&M, pre142, 142 * 8);
    double dt2 = dt * dt;
    double w[3] = {omega[0] * dt, omega[1] * dt, omega[2] * dt};
    double dR[9], Jr[9];
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (v3_norm(w) < 1e-10) memcpy(dR, I3, 72); /* Expmap, IMUPreintegrator.h:85-90 */
    else { double q[4]; so3_exp(w, q); q_to_R(q, dR); }
    so3_Jr(w, Jr);
    double A[81] = {0}, Bg[27] = {0}, Ca[27] = {0}, S[9], dRT[9], RS[9];
    for (int i = 0; i < 9; ++i) A[i * 9 + i] = 1.0;
    m3_T(dR, dRT);
    so3_hat(acc, S);
    m3_mul(M.dR, S, RS);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        A[(6 + r) * 9 + 6 + c] = dRT[r * 3 + c];
        A[(3 + r) * 9 + 6 + c] = -RS[r * 3 + c] * dt;
        A[(0 + r) * 9 + 6 + c] = -0.5 * RS[r * 3 + c] * dt2;
        A[(0 + r) * 9 + 3 + c] = I3[r * 3 + c] * dt;
        Bg[(6 + r) * 3 + c] = Jr[r * 3 + c] * dt;
        Ca[(3 + r) * 3 + c] = M.dR[r * 3 + c] * dt;
        Ca[(0 + r) * 3 + c] = 0.5 * M.dR[r * 3 + c] * dt2;
    }
    double AC[81], AT[81], ACA[81], BgT[27], CaT[27], BB[81], CC[81];
    mat_mul(A, M.cov, AC, 9, 9, 9);
    mat_T(A, AT, 9, 9);
    mat_mul(AC, AT, ACA, 9, 9, 9);
    mat_T(Bg, BgT, 9, 3);
    mat_T(Ca, CaT, 9, 3);
    double Bgs[27], Cas[27];
    for (int i = 0; i < 27; ++i) { Bgs[i] = Bg[i] * gyr_cov; Cas[i] = Ca[i] * acc_cov; }
    mat_mul(Bgs, BgT, BB, 9, 3, 9);
    mat_mul(Cas, CaT, CC, 9, 3, 9);
    for (int i = 0; i < 81; ++i) M.cov[i] = ACA[i] + BB[i] + CC[i];
    /* jacobians wrt bias: P first, then V, then R */
    double RSJ[9];
    m3_mul(RS, M.JRg, RSJ);
    for (int i = 0; i < 9; ++i) {
        M.JPa[i] += M.JVa[i] * dt - 0.5 * M.dR[i] * dt2;
        M.JPg[i] += M.JVg[i] * dt - 0.5 * RSJ[i] * dt2;
    }
    for (int i = 0; i < 9; ++i) {
        M.JVa[i] += -M.dR[i] * dt;
        M.JVg[i] += -RSJ[i] * dt;
    }
    double t9[9];
    m3_mul(dRT, M.JRg, t9);
    for (int i = 0; i < 9; ++i) M.JRg[i] = t9[i] - Jr[i] * dt;
    /* delta measurements */
    double Ra[3];
    m3_v(M.dR, acc, Ra);
    for (int i = 0; i < 3; ++i) M.dP[i] += M.dV[i] * dt + 0.5 * Ra[i] * dt2;
    for (int i = 0; i < 3; ++i) M.dV[i] += Ra[i] * dt;
    double RdR[9], q[4];
    m3_mul(M.dR, dR, RdR);
    R_to_q(RdR, q); /* normalizeRotationM, IMUPreintegrator.h:166-180 */
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    q_normalize(q);
    q_to_R(q, M.dR);
    M.dt += dt;
    memcpy(pre142, &M, 142 * 8);
}

/* KeyFrame::ComputeIMUPreIntSinceLastFrame (src/keyFrame.cpp:139-172) batched over M keyframe intervals; same
 * contract as plba_preintegrate (include/plba.h).  p is unused. */
int orc_preintegrate(void* p, int M, const int32_t* sample_start, const long double* t, const double* gyr3, const double* acc3,
                     const long double* t_prev, const long double* t_curr, const double* bg3, const double* ba3,
                     double gyr_meas_cov, double acc_meas_cov, double* out142) {
    (void)p;
    for (int m = 0; m < M; ++m) {
        double* pre = out142 + (size_t)m * 142;
        memset(pre, 0, 142 * 8);                 /* IMUPreintegrator::reset, IMU/IMUPreintegrator.cpp:47-76 */
        pre[6] = pre[10] = pre[14] = 1.0;
        const int end = sample_start[m + 1];
        int i = sample_start[m];
        const double* bg = bg3 + 3 * m; const double* ba = ba3 + 3 * m;
        double w[3], a[3], dt;
#define ORC_STEP(ii, DT) do { for (int q = 0; q < 3; ++q) { w[q] = gyr3[3 * (size_t)(ii) + q] - bg[q]; a[q] = acc3[3 * (size_t)(ii) + q] - ba[q]; } \
        dt = (double)(DT); orc_preint_update(pre, w, a, dt, gyr_meas_cov, acc_meas_cov); } while (0)
        while (i < end && t[i] < t_prev[m]) ++i;             /* keyFrame.cpp:147-149 */
        if (i >= end) continue;
        ORC_STEP(i, t[i] - t_prev[m]); ++i;                  /* :150-154 */
        while (i < end && t[i] <= t_curr[m]) { ORC_STEP(i, t[i] - t[i - 1]); ++i; }   /* :155-161 */
        if (i < end) ORC_STEP(i, t_curr[m] - t[i]);          /* :162-167, dt as coded */
#undef ORC_STEP
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * API-surface types of IMU/types_six_dof_expmap.{h,cpp} and IMU/se3quat.h (named by the task, no call
 * site in the local-BA path).  Plain restatements used only to check the host-side facade classes.
 * SE3 = unit quaternion (x,y,z,w; w >= 0) + translation.
 * ------------------------------------------------------------------------------------------------ */
static void se3_normalize(double* q) { /* SE3Quat::normalizeRotation, IMU/se3quat.h:283-288 */
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    q_normalize(q);
}
/* SE3Quat::exp, IMU/se3quat.h:223-257: update = (omega, upsilon) */
void orc_se3_exp(const double* u6, double* q, double* t) {
    const double* om = u6; const double* up = u6 + 3;
    double theta = v3_norm(om), W[9], W2[9], R[9], V[9];
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    so3_hat(om, W);
    m3_mul(W, W, W2);
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = I3[i] + W[i] + W2[i]; V[i] = R[i]; }
    } else {
        double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; ++i) { R[i] = I3[i] + a * W[i] + b * W2[i]; V[i] = I3[i] + b * W[i] + c * W2[i]; }
    }
    R_to_q(R, q);
    m3_v(V, up, t);
    se3_normalize(q);
}
/* SE3Quat::operator*, IMU/se3quat.h:101-107 */
void orc_se3_mul(const double* qa, const double* ta, const double* qb, const double* tb, double* qo, double* to) {
    double r[3];
    q_rotate(qa, tb, r);
    for (int i = 0; i < 3; ++i) to[i] = ta[i] + r[i];
    q_mul(qa, qb, qo);
    se3_normalize(qo);
}
/* SE3Quat::map, IMU/se3quat.h:217-220 */
void orc_se3_map(const double* q, const double* t, const double* x, double* out) {
    q_rotate(q, x, out);
    for (int i = 0; i < 3; ++i) out[i] += t[i];
}
/* SE3Quat::log, IMU/se3quat.h:178-215 */
void orc_se3_log(const double* q, const double* t, double* out6) {
    double R[9], om[3], W[9], W2[9], Vi[9], dR[3];
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    q_to_R(q, R);
    double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    dR[0] = R[7] - R[5]; dR[1] = R[2] - R[6]; dR[2] = R[3] - R[1];
    if (d > 0.99999) {
        for (int i = 0; i < 3; ++i) om[i] = 0.5 * dR[i];
        so3_hat(om, W); m3_mul(W, W, W2);
        for (int i = 0; i < 9; ++i) Vi[i] = I3[i] - 0.5 * W[i] + (1. / 12.) * W2[i];
    } else {
        double theta = acos(d);
        for (int i = 0; i < 3; ++i) om[i] = theta / (2 * sqrt(1 - d * d)) * dR[i];
        so3_hat(om, W); m3_mul(W, W, W2);
        double k = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
        for (int i = 0; i < 9; ++i) Vi[i] = I3[i] - 0.5 * W[i] + k * W2[i];
    }
    out6[0] = om[0]; out6[1] = om[1]; out6[2] = om[2];
    m3_v(Vi, t, out6 + 3);
}
/* VertexSE3Expmap::oplusImpl, IMU/types_six_dof_expmap.h:73-76: estimate <- exp(update) * estimate */
void orc_se3_oplus(const double* q, const double* t, const double* u6, double* qo, double* to) {
    double qe[4], te[3];
    orc_se3_exp(u6, qe, te);
    orc_se3_mul(qe, te, q, t, qo, to);
}
/* The four reprojection edges.  kind: 0 EdgeSE3ProjectXYZ (cpp:103-145), 1 EdgeStereoSE3ProjectXYZ (:150-235),
 * 2 EdgeSE3ProjectXYZOnlyPose (:258-289), 3 EdgeStereoSE3ProjectXYZOnlyPose (:299-358).  cam = fx fy cx cy bf.
 * err: 2 or 3; Jpoint: D x 3 row-major (kinds 0/1; untouched otherwise); Jpose: D x 6 row-major; depth_pos: map(X).z > 0. */
void orc_eval_se3_edge(int kind, const double* cam, const double* q, const double* t, const double* X, const double* obs,
                       double* err, double* Jpoint, double* Jpose, int* depth_pos) {
    const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], bf = cam[4];
    double p[3], R[9];
    orc_se3_map(q, t, X, p);
    q_to_R(q, R);
    const double x = p[0], y = p[1], z = p[2];
    const int stereo = (kind == 1 || kind == 3), D = stereo ? 3 : 2;
    if (depth_pos) *depth_pos = z > 0.0;
    if (!stereo) {
        err[0] = obs[0] - (x / z * fx + cx);        /* project2d then scale, cpp:139-145 / 283-289 */
        err[1] = obs[1] - (y / z * fy + cy);
    } else {
        const float invz = (float)(1.0f / z);        /* cpp:151 / 300: invz is a float */
        double r0 = x * invz * fx + cx, r1 = y * invz * fy + cy, r2;
        if (kind == 1) { const float bff = (float)bf; r2 = r0 - bff * invz; }   /* float * float, cam_project(xyz, const float& bf) */
        else r2 = r0 - bf * invz;                                              /* double member * float */
        err[0] = obs[0] - r0; err[1] = obs[1] - r1; err[2] = obs[2] - r2;
    }
    if (kind == 0) {
        const double z_2 = z * z;
        const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (-1. / z * tmp[r * 3 + k]) * R[k * 3 + c];
            Jpoint[r * 3 + c] = s;
        }
        double* j = Jpose;
        j[0] = x * y / z_2 * fx; j[1] = -(1 + (x * x / z_2)) * fx; j[2] = y / z * fx; j[3] = -1. / z * fx; j[4] = 0; j[5] = x / z_2 * fx;
        j[6] = (1 + y * y / z_2) * fy; j[7] = -x * y / z_2 * fy; j[8] = -x / z * fy; j[9] = 0; j[10] = -1. / z * fy; j[11] = y / z_2 * fy;
    } else if (kind == 1) {
        const double z_2 = z * z;
        for (int c = 0; c < 3; ++c) {
            Jpoint[c] = -fx * R[c] / z + fx * x * R[6 + c] / z_2;
            Jpoint[3 + c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z_2;
            Jpoint[6 + c] = Jpoint[c] - bf * R[6 + c] / z_2;
        }
        double* j = Jpose;
        j[0] = x * y / z_2 * fx; j[1] = -(1 + (x * x / z_2)) * fx; j[2] = y / z * fx; j[3] = -1. / z * fx; j[4] = 0; j[5] = x / z_2 * fx;
        j[6] = (1 + y * y / z_2) * fy; j[7] = -x * y / z_2 * fy; j[8] = -x / z * fy; j[9] = 0; j[10] = -1. / z * fy; j[11] = y / z_2 * fy;
        j[12] = j[0] - bf * y / z_2; j[13] = j[1] + bf * x / z_2; j[14] = j[2]; j[15] = j[3]; j[16] = 0; j[17] = j[5] - bf / z_2;
    } else {
        const double invz = 1.0 / z, invz_2 = invz * invz;
        double* j = Jpose;
        j[0] = x * y * invz_2 * fx; j[1] = -(1 + (x * x * invz_2)) * fx; j[2] = y * invz * fx; j[3] = -invz * fx; j[4] = 0; j[5] = x * invz_2 * fx;
        j[6] = (1 + y * y * invz_2) * fy; j[7] = -x * y * invz_2 * fy; j[8] = -x * invz * fy; j[9] = 0; j[10] = -invz * fy; j[11] = y * invz_2 * fy;
        if (kind == 3) { j[12] = j[0] - bf * y * invz_2; j[13] = j[1] + bf * x * invz_2; j[14] = j[2]; j[15] = j[3]; j[16] = 0; j[17] = j[5] - bf * invz_2; }
    }
    (void)D;
}

/* EdgeGyrBias (IMU/g2otypes.cpp:1263-1287): err = Log((dRbij Exp(J bg))^T Rwbi^T Rwbj); J = -JlInv(Log(dRbij^T Rwbi^T Rwbj)) J_dR_bg
 * (the Jacobian is evaluated without the bias term, as coded). All matrices row-major. */
void orc_eval_gyrbias_edge(const double* dRbij, const double* JdRbg, const double* Rwbi, const double* Rwbj, const double* bg,
                           double* err3, double* J9) {
    double w[3], q[4], dRbg[9], A[9], At[9], RiT[9], B[9], E[9];
    m3_v(JdRbg, bg, w);
    so3_exp(w, q); q_to_R(q, dRbg);
    m3_mul(dRbij, dRbg, A); m3_T(A, At); m3_T(Rwbi, RiT);
    m3_mul(At, RiT, B); m3_mul(B, Rwbj, E);
    R_to_q(E, q); q_normalize(q); so3_log(q, err3);
    if (J9) {
        double l[3], nl[3], Jl[9], Jx[9];
        m3_T(dRbij, At); m3_mul(At, RiT, B); m3_mul(B, Rwbj, E);
        R_to_q(E, q); q_normalize(q); so3_log(q, l);
        nl[0] = -l[0]; nl[1] = -l[1]; nl[2] = -l[2];
        so3_JrInv(nl, Jl);                      /* JacobianLInv(w) = JacobianRInv(-w), IMU/so3.h */
        m3_mul(Jl, JdRbg, Jx);
        for (int i = 0; i < 9; ++i) J9[i] = -Jx[i];
    }
}

/* =================================================================================================================
 * SURVEY §8f row 4 — the OTHER g2o users of src/mapHandler.cpp, restated for the checker:
 *   MapHandler::IMUInitEstBg (src/mapHandler.cpp:4989-5036): one VertexGyrBias, M EdgeGyrBias, Levenberg, optimize(1)
 *   the pose-graph optimisers (src/mapHandler.cpp:4068-4297, 4299-4528): g2o VertexSE3 / EdgeSE3, Levenberg with
 *   userLambdaInit 1e-10.
 * g2o is third-party and un-vendored: the slam3d parametrisation is restated from its published form
 * (update X <- X * fromVectorMQT(u); error toVectorMQT(Z^-1 Xi^-1 Xj)).  The EdgeSE3 Jacobians are taken here by
 * CENTRAL DIFFERENCES of the error through the vertex update, so that this checker does not share the analytic derivation
 * of include/plba_g2o/types_slam3d.h.  The LM loop is SURVEY App. A.3 on dense normal equations.
 * ================================================================================================================= */
typedef struct {
    int N;
    void* ctx;
    double (*errors)(void* ctx);                          /* computeActiveErrors + activeRobustChi2 */
    void (*build)(void* ctx, double* H, double* b);       /* buildSystem: H = sum J^T W J, b = -sum J^T W e */
    void (*push)(void* ctx);
    void (*pop)(void* ctx);
    void (*update)(void* ctx, const double* x);
} graph_ops;

/* stats: chi2 before, chi2 after, iterations done, trials */
static void lm_dense(graph_ops* g, int iters, double user_lambda, int gauss_newton, double* stats) {
    const int N = g->N;
    double* H = (double*)xcalloc((size_t)N * N, 8); double* A = (double*)xcalloc((size_t)N * N, 8);
    double* b = (double*)xcalloc(N, 8); double* x = (double*)xcalloc(N, 8);
    double lambda = 0.0, ni = 2.0;
    int done = 0, trials = 0, ok = 1;
    stats[0] = stats[1] = g->errors(g->ctx);
    for (int it = 0; it < iters && ok; ++it) {
        double currentChi = g->errors(g->ctx);
        g->build(g->ctx, H, b);
        if (gauss_newton) {
            memcpy(A, H, (size_t)N * N * 8);
            if (chol_factor(A, N)) { chol_solve(A, N, b, x); g->update(g->ctx, x); } else ok = 0;
            ++trials; ++done;
            stats[1] = g->errors(g->ctx);
            continue;
        }
        if (it == 0) {
            if (user_lambda > 0) lambda = user_lambda;
            else { double md = 0.0; for (int i = 0; i < N; ++i) if (fabs(H[(size_t)i * N + i]) > md) md = fabs(H[(size_t)i * N + i]); lambda = 1e-5 * md; }
            ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
            g->push(g->ctx);
            memcpy(A, H, (size_t)N * N * 8);
            for (int i = 0; i < N; ++i) A[(size_t)i * N + i] += lambda;
            const int sok = chol_factor(A, N);
            if (sok) chol_solve(A, N, b, x); else memset(x, 0, (size_t)N * 8);
            g->update(g->ctx, x);
            double tempChi = g->errors(g->ctx);
            if (!sok) tempChi = DBL_MAX;
            double scale = 1e-3;
            for (int i = 0; i < N; ++i) scale += x[i] * (lambda * x[i] + b[i]);
            rho = (currentChi - tempChi) / scale;
            ++trials;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                if (alpha > 2. / 3.) alpha = 2. / 3.;
                lambda *= (alpha > 1. / 3.) ? alpha : 1. / 3.;
                ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                g->pop(g->ctx);
                if (!isfinite(lambda)) break;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10);
        ++done;
        stats[1] = currentChi;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) ok = 0;
    }
    stats[2] = done; stats[3] = trials;
    free(H); free(A); free(b); free(x);
}

/* ---- IMUInitEstBg ------------------------------------------------------------------------------------------------ */
typedef struct { int M; const double *dR, *J, *Ri, *Rj, *info; double bg[3], saved[3]; } gyr_ctx;
static double gyr_errors(void* c) {
    gyr_ctx* g = (gyr_ctx*)c;
    double chi = 0.0;
    for (int m = 0; m < g->M; ++m) {
        double e[3];
        orc_eval_gyrbias_edge(g->dR + 9 * m, g->J + 9 * m, g->Ri + 9 * m, g->Rj + 9 * m, g->bg, e, NULL);
        chi += quad_form(e, g->info + 9 * m, 3);
    }
    return chi;
}
static void gyr_build(void* c, double* H, double* b) {
    gyr_ctx* g = (gyr_ctx*)c;
    memset(H, 0, 72); memset(b, 0, 24);
    for (int m = 0; m < g->M; ++m) {
        double e[3], J[9], OJ[9], Oe[3];
        orc_eval_gyrbias_edge(g->dR + 9 * m, g->J + 9 * m, g->Ri + 9 * m, g->Rj + 9 * m, g->bg, e, J);
        const double* Om = g->info + 9 * m;
        mat_mul(Om, J, OJ, 3, 3, 3); m3_v(Om, e, Oe);
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc) { double s = 0; for (int k = 0; k < 3; ++k) s += J[k * 3 + r] * OJ[k * 3 + cc]; H[r * 3 + cc] += s; }
            double s = 0; for (int k = 0; k < 3; ++k) s += J[k * 3 + r] * Oe[k];
            b[r] -= s;
        }
    }
}
static void gyr_push(void* c) { gyr_ctx* g = (gyr_ctx*)c; memcpy(g->saved, g->bg, 24); }
static void gyr_pop(void* c) { gyr_ctx* g = (gyr_ctx*)c; memcpy(g->bg, g->saved, 24); }
static void gyr_update(void* c, const double* x) { gyr_ctx* g = (gyr_ctx*)c; for (int i = 0; i < 3; ++i) g->bg[i] += x[i]; }
/* bg3: in = start (the reference starts at zero), out = estimate; stats4: chi2 before / after, iterations, trials */
void orc_gyrbias_estimate(int M, const double* dR9, const double* JRg9, const double* Rwbi9, const double* Rwbj9, const double* info9,
                          int iters, int gauss_newton, double* bg3, double* stats4) {
    gyr_ctx c; c.M = M; c.dR = dR9; c.J = JRg9; c.Ri = Rwbi9; c.Rj = Rwbj9; c.info = info9;
    memcpy(c.bg, bg3, 24);
    graph_ops g = {3, &c, gyr_errors, gyr_build, gyr_push, gyr_pop, gyr_update};
    lm_dense(&g, iters, 0.0, gauss_newton, stats4);
    memcpy(bg3, c.bg, 24);
}

/* ---- pose graph: g2o VertexSE3 / EdgeSE3 ------------------------------------------------------------------------------ */
typedef struct { double R[9], t[3]; } iso_t;
static void iso_mul(const iso_t* a, const iso_t* b, iso_t* o) { iso_t r; m3_mul(a->R, b->R, r.R); m3_v(a->R, b->t, r.t); for (int i = 0; i < 3; ++i) r.t[i] += a->t[i]; *o = r; }
static void iso_inv(const iso_t* a, iso_t* o) { iso_t r; m3_T(a->R, r.R); m3_v(r.R, a->t, r.t); for (int i = 0; i < 3; ++i) r.t[i] = -r.t[i]; *o = r; }
static void iso_from_mqt(const double* u, iso_t* o) { /* internal::fromVectorMQT */
    memcpy(o->t, u, 24);
    const double w2 = 1.0 - (u[3] * u[3] + u[4] * u[4] + u[5] * u[5]);
    if (w2 < 0) { memset(o->R, 0, 72); o->R[0] = o->R[4] = o->R[8] = 1.0; }
    else { double q[4] = {u[3], u[4], u[5], sqrt(w2)}; q_to_R(q, o->R); }
}
static void iso_to_mqt(const iso_t* a, double* e) { /* internal::toVectorMQT: q normalised, w >= 0 */
    double q[4];
    R_to_q(a->R, q); q_normalize(q);
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    memcpy(e, a->t, 24); e[3] = q[0]; e[4] = q[1]; e[5] = q[2];
}
typedef struct {
    int nv, ne;
    iso_t *X, *saved;
    const int32_t *fixed, *ei, *ej;
    int* hidx;
    iso_t *Z, *Zi;
    const double* info;
} pgo_ctx;
static void pgo_edge_error(const pgo_ctx* g, int k, const iso_t* Xi, const iso_t* Xj, double* e) {
    iso_t a, b;
    iso_inv(Xi, &a); iso_mul(&g->Zi[k], &a, &b); iso_mul(&b, Xj, &a);
    iso_to_mqt(&a, e);
}
static double pgo_errors(void* c) {
    pgo_ctx* g = (pgo_ctx*)c;
    double chi = 0.0;
    for (int k = 0; k < g->ne; ++k) { double e[6]; pgo_edge_error(g, k, &g->X[g->ei[k]], &g->X[g->ej[k]], e); chi += quad_form(e, g->info + 36 * k, 6); }
    return chi;
}
static void pgo_numjac(const pgo_ctx* g, int k, int which, double* J /*6x6 row-major*/) {
    const double h = 1e-6;
    for (int c = 0; c < 6; ++c) {
        double u[6] = {0, 0, 0, 0, 0, 0}, ep[6], em[6];
        iso_t d, Xp, Xm;
        const iso_t* Xi = &g->X[g->ei[k]]; const iso_t* Xj = &g->X[g->ej[k]];
        u[c] = h; iso_from_mqt(u, &d); iso_mul(which ? Xj : Xi, &d, &Xp);
        u[c] = -h; iso_from_mqt(u, &d); iso_mul(which ? Xj : Xi, &d, &Xm);
        pgo_edge_error(g, k, which ? Xi : &Xp, which ? &Xp : Xj, ep);
        pgo_edge_error(g, k, which ? Xi : &Xm, which ? &Xm : Xj, em);
        for (int r = 0; r < 6; ++r) J[r * 6 + c] = (ep[r] - em[r]) / (2 * h);
    }
}
static void pgo_build(void* c, double* H, double* b) {
    pgo_ctx* g = (pgo_ctx*)c;
    int N = 0;
    for (int v = 0; v < g->nv; ++v) if (g->hidx[v] >= 0) N += 6;
    memset(H, 0, (size_t)N * N * 8); memset(b, 0, (size_t)N * 8);
    for (int k = 0; k < g->ne; ++k) {
        double e[6], J[2][36], Oe[6];
        const int vv[2] = {g->ei[k], g->ej[k]};
        pgo_edge_error(g, k, &g->X[vv[0]], &g->X[vv[1]], e);
        pgo_numjac(g, k, 0, J[0]); pgo_numjac(g, k, 1, J[1]);
        const double* Om = g->info + 36 * k;
        for (int r = 0; r < 6; ++r) { double s = 0; for (int q = 0; q < 6; ++q) s += Om[r * 6 + q] * e[q]; Oe[r] = s; }
        for (int a = 0; a < 2; ++a) {
            const int oa = g->hidx[vv[a]];
            if (oa < 0) continue;
            for (int cc = 0; cc < 6; ++cc) { double s = 0; for (int r = 0; r < 6; ++r) s += J[a][r * 6 + cc] * Oe[r]; b[oa + cc] -= s; }
            for (int bb = 0; bb < 2; ++bb) {
                const int ob = g->hidx[vv[bb]];
                if (ob < 0) continue;
                for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) {
                    double s = 0;
                    for (int p = 0; p < 6; ++p) for (int q = 0; q < 6; ++q) s += J[a][p * 6 + r] * Om[p * 6 + q] * J[bb][q * 6 + cc];
                    H[(size_t)(oa + r) * N + ob + cc] += s;
                }
            }
        }
    }
}
static void pgo_push(void* c) { pgo_ctx* g = (pgo_ctx*)c; memcpy(g->saved, g->X, sizeof(iso_t) * (size_t)g->nv); }
static void pgo_pop(void* c) { pgo_ctx* g = (pgo_ctx*)c; memcpy(g->X, g->saved, sizeof(iso_t) * (size_t)g->nv); }
static void pgo_update(void* c, const double* x) {
    pgo_ctx* g = (pgo_ctx*)c;
    for (int v = 0; v < g->nv; ++v) if (g->hidx[v] >= 0) { iso_t d; iso_from_mqt(x + g->hidx[v], &d); iso_mul(&g->X[v], &d, &g->X[v]); }
}
/* pose12: nv x (R row-major 9, t 3), in = initial estimates, out = optimised; vertices are addressed by their INDEX in these arrays
 * and must be given in ascending vertex-id order (Hessian order, App. A.1); meas12 likewise per edge; info36 row-major.
 * initial_guess: breadth-first propagation from the fixed vertices through the edges in insertion order (the facade's reading of
 * g2o's computeInitialGuess).  stats4: chi2 before (after the initial guess) / after, iterations, trials. */
void orc_pgo(int nv, const int32_t* fixed, double* pose12, int ne, const int32_t* ei, const int32_t* ej, const double* meas12,
             const double* info36, int iters, double user_lambda, int gauss_newton, int initial_guess, double* stats4) {
    pgo_ctx g;
    g.nv = nv; g.ne = ne; g.fixed = fixed; g.ei = ei; g.ej = ej; g.info = info36;
    g.X = (iso_t*)xcalloc(nv, sizeof(iso_t)); g.saved = (iso_t*)xcalloc(nv, sizeof(iso_t));
    g.Z = (iso_t*)xcalloc(ne > 0 ? ne : 1, sizeof(iso_t)); g.Zi = (iso_t*)xcalloc(ne > 0 ? ne : 1, sizeof(iso_t));
    g.hidx = (int*)xcalloc(nv, sizeof(int));
    int N = 0;
    for (int v = 0; v < nv; ++v) { memcpy(g.X[v].R, pose12 + 12 * v, 72); memcpy(g.X[v].t, pose12 + 12 * v + 9, 24); if (fixed[v]) g.hidx[v] = -1; else { g.hidx[v] = N; N += 6; } }
    for (int k = 0; k < ne; ++k) { memcpy(g.Z[k].R, meas12 + 12 * k, 72); memcpy(g.Z[k].t, meas12 + 12 * k + 9, 24); iso_inv(&g.Z[k], &g.Zi[k]); }
    if (initial_guess) {
        char* done = (char*)xcalloc(nv, 1); int* front = (int*)xcalloc(nv, sizeof(int)); int* next = (int*)xcalloc(nv, sizeof(int));
        int nf = 0;
        for (int k = 0; k < ne; ++k) { const int vv[2] = {ei[k], ej[k]}; for (int a = 0; a < 2; ++a) if (fixed[vv[a]] && !done[vv[a]]) { done[vv[a]] = 1; front[nf++] = vv[a]; } }
        while (nf > 0) {
            int nn = 0;
            for (int f = 0; f < nf; ++f) for (int k = 0; k < ne; ++k) {
                const int from = front[f];
                const int to = ei[k] == from ? ej[k] : ej[k] == from ? ei[k] : -1;
                if (to < 0 || done[to] || fixed[to]) continue;
                if (ei[k] == from) iso_mul(&g.X[from], &g.Z[k], &g.X[to]); else iso_mul(&g.X[from], &g.Zi[k], &g.X[to]);
                done[to] = 1; next[nn++] = to;
            }
            memcpy(front, next, sizeof(int) * (size_t)nn); nf = nn;
        }
        free(done); free(front); free(next);
    }
    graph_ops ops = {N, &g, pgo_errors, pgo_build, pgo_push, pgo_pop, pgo_update};
    if (N > 0) lm_dense(&ops, iters, user_lambda, gauss_newton, stats4);
    else { stats4[0] = stats4[1] = pgo_errors(&g); stats4[2] = stats4[3] = 0; }
    for (int v = 0; v < nv; ++v) { memcpy(pose12 + 12 * v, g.X[v].R, 72); memcpy(pose12 + 12 * v + 9, g.X[v].t, 24); }
    free(g.X); free(g.saved); free(g.Z); free(g.Zi); free(g.hidx);
}
/* the EdgeSE3 pieces on their own, for Jacobian checks of include/plba_g2o/types_slam3d.h: error of one edge and the vertex update */
void orc_se3_edge_error(const double* Xi12, const double* Xj12, const double* Z12, double* e6) {
    pgo_ctx g; iso_t Xi, Xj, Z, Zi;
    memcpy(Xi.R, Xi12, 72); memcpy(Xi.t, Xi12 + 9, 24); memcpy(Xj.R, Xj12, 72); memcpy(Xj.t, Xj12 + 9, 24); memcpy(Z.R, Z12, 72); memcpy(Z.t, Z12 + 9, 24);
    iso_inv(&Z, &Zi); g.Zi = &Zi;
    pgo_edge_error(&g, 0, &Xi, &Xj, e6);
}
void orc_se3_vertex_oplus(const double* X12, const double* u6, double* out12) {
    iso_t X, d; memcpy(X.R, X12, 72); memcpy(X.t, X12 + 9, 24);
    iso_from_mqt(u6, &d); iso_mul(&X, &d, &X);
    memcpy(out12, X.R, 72); memcpy(out12 + 9, X.t, 24);
}

/* =================================================================================================================
 * SURVEY §8f row 2 — the pre-VIO-init visual-only local BA: MapHandler::levMarquardtOptimizationLBA
 * (src/mapHandler.cpp:1441-2098), restated.  Hand-rolled LM on X = [6 per local keyframe | 3 per point | 6 per line]:
 * scalar residual = NORM of the reprojection error, Jacobian row e^T J / max(homogTh, |e|), Cauchy weights
 * (stvo-pl/src/auxiliar.cpp:556-559), multiplicative damping H_ii += lambda H_ii (:1662-1663, :1887-1888), SimplicialLDLT of
 * the full system, pose update T <- T * inverse(expmap(dx)) (:1672-1676), lambda schedule as coded (:1895-1900: divided by
 * lambda_k when the error grew, MULTIPLIED when it shrank).  SE(3) helpers: stvo-pl/src/auxiliar.cpp:113-173.
 * Decisions on the defects of SURVEY App. B-Q9:
 *   :1650  err /= (Npt_obs + Nls_obs) with both counters 0: reproduced (IEEE: +inf), it makes the first comparison of :1894 a "success";
 *   :1787-1788  P and Q of a line read from the same, 3-strided offset -> NOT reproduced (P = X[.. + 6 l], Q = X[.. + 6 l + 3]);
 *   :1790  the line pass inside the loop takes keyframe poses from the MAP, not from X -> reproduced unless use_iterate_poses.
 * variant = 1 is levMarquardtOptimizationGBA (:2210-2812), the same text over all keyframes with three differences that matter:
 * `int Hmax` (:2468), err divided by the zero counters in EVERY pass (:2744; every step is taken, lambda only grows) and
 * machine epsilon as both thresholds (:2746, :2776; the caller passes them as min_error / min_error_change).
 * ================================================================================================================= */
typedef struct { double lambda_lm, lambda_k; int max_iters; double homog_th, min_error, min_error_change; int use_iterate_poses; int variant; } lba_opt_t;
typedef struct { int iterations; int updates; double err_first, err_last, lambda; int solver_failed; int reserved; } lba_stats_t;

static void se3_inverse(const double* T, double* Ti) { /* inverse_se3, auxiliar.cpp:113-122 */
    memset(Ti, 0, 128); Ti[15] = 1.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ti[i * 4 + j] = T[j * 4 + i];
    for (int i = 0; i < 3; ++i) Ti[i * 4 + 3] = -(Ti[i * 4] * T[3] + Ti[i * 4 + 1] * T[7] + Ti[i * 4 + 2] * T[11]);
}
static void se3_mul(const double* A, const double* B, double* C) { double t[16]; mat_mul(A, B, t, 4, 4, 4); memcpy(C, t, 128); }
static void se3_expmap(const double* x, double* T) { /* expmap_se3, auxiliar.cpp:124-141: x = (t, w) */
    const double* w = x + 3;
    double th = v3_norm(w), R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {x[0], x[1], x[2]};
    if (!(th < 0.000001)) {
        double s[9], wn[3] = {w[0] / th, w[1] / th, w[2] / th}, ss[9], V[9], tv[3];
        so3_hat(wn, s); m3_mul(s, s, ss);
        const double sn = sin(th), cs = cos(th);
        for (int i = 0; i < 9; ++i) {
            const double I = (i % 4 == 0) ? 1.0 : 0.0;
            R[i] = I + s[i] * sn + ss[i] * (1.0 - cs);
            V[i] = I + s[i] * (1.0 - cs) / th + ss[i] * (th - sn) / th;
        }
        m3_v(V, t, tv); memcpy(t, tv, 24);
    }
    memset(T, 0, 128); T[15] = 1.0;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j]; T[i * 4 + 3] = t[i]; }
}
static void se3_logmap(const double* T, double* x) { /* logmap_se3, auxiliar.cpp:143-173 */
    double R[9], Vt[3] = {T[3], T[7], T[11]}, w[3] = {0, 0, 0}, V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j];
    double cosine = (R[0] + R[4] + R[8] - 1.0) / 2.0;
    if (cosine > 1.0) cosine = 1.0; else if (cosine < -1.0) cosine = -1.0;
    double sine = sqrt(1.0 - cosine * cosine);
    if (sine > 1.0) sine = 1.0;
    const double theta = acos(cosine);
    if (theta > 0.000001) {
        const double f = theta / (2.0 * sine);
        w[0] = f * (R[7] - R[5]); w[1] = f * (R[2] - R[6]); w[2] = f * (R[3] - R[1]);     /* skewcoords(theta (R - R^T) / (2 sine)) */
        double s[9], ss[9], wn[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
        so3_hat(wn, s); m3_mul(s, s, ss);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s[i] * (1.0 - cosine) / theta + ss[i] * (theta - sine) / theta;
    }
    double Vi[9];
    lu_inverse(V, Vi, 3);
    m3_v(Vi, Vt, x);
    x[3] = w[0]; x[4] = w[1]; x[5] = w[2];
}
/* pose / landmark Jacobian pieces shared by the point and the line-end-point terms (:1490-1512): g = point in the camera
 * frame, (a, b) = (fx dx, fy dy) resp. (fx lx, fy ly); Jp6 unscaled, Jl3 = (.,.,.) unscaled (before * R^T and the norm) */
static void lba_jac_pieces(const double* g, double a, double b, double homog_th, double* Jp6, double* Jl3) {
    double gz2 = g[2] * g[2];
    gz2 = 1.0 / (homog_th > gz2 ? homog_th : gz2);
    Jp6[0] = gz2 * a * g[2];
    Jp6[1] = gz2 * b * g[2];
    Jp6[2] = -gz2 * (a * g[0] + b * g[1]);
    Jp6[3] = -gz2 * (a * g[0] * g[1] + b * g[1] * g[1] + b * g[2] * g[2]);
    Jp6[4] = gz2 * (a * g[0] * g[0] + a * g[2] * g[2] + b * g[0] * g[1]);
    Jp6[5] = gz2 * (b * g[0] * g[2] - a * g[1] * g[2]);
    Jl3[0] = Jp6[0]; Jl3[1] = Jp6[1]; Jl3[2] = Jp6[2];
}
static void rowvec_R(const double* v3, const double* Tiw, double* o) { /* v^T * Tiw.block(0,0,3,3) */
    for (int c = 0; c < 3; ++c) o[c] = v3[0] * Tiw[0 * 4 + c] + v3[1] * Tiw[1 * 4 + c] + v3[2] * Tiw[2 * 4 + c];
}
/* one observation: residual norm, weight, pose Jacobian (6), landmark Jacobian (3 or 6); Tiw = inverse(T_kf_w) */
static void lba_point_obs(const double* cam4, double homog_th, const double* Tiw, const double* Xw, const double* uv, double* rn, double* w, double* Jp, double* Jl) {
    double g[3], e[2], Jl3[3];
    for (int i = 0; i < 3; ++i) g[i] = Tiw[i * 4] * Xw[0] + Tiw[i * 4 + 1] * Xw[1] + Tiw[i * 4 + 2] * Xw[2] + Tiw[i * 4 + 3];
    e[0] = uv[0] - (cam4[2] + cam4[0] * g[0] / g[2]); e[1] = uv[1] - (cam4[3] + cam4[1] * g[1] / g[2]);
    const double n = sqrt(e[0] * e[0] + e[1] * e[1]), dn = homog_th > n ? homog_th : n;
    lba_jac_pieces(g, cam4[0] * e[0], cam4[1] * e[1], homog_th, Jp, Jl3);
    for (int i = 0; i < 6; ++i) Jp[i] /= dn;
    rowvec_R(Jl3, Tiw, Jl);
    for (int i = 0; i < 3; ++i) Jl[i] /= dn;
    *rn = n; *w = 1.0 / (1.0 + n * n);
}
static void lba_line_obs(const double* cam4, double homog_th, const double* Tiw, const double* PQ, const double* l3, double* rn, double* w, double* Jp, double* Jl) {
    double gp[3], gq[3], e[2], JP[6], JQ[6], a3[3], b3[3];
    for (int i = 0; i < 3; ++i) {
        gp[i] = Tiw[i * 4] * PQ[0] + Tiw[i * 4 + 1] * PQ[1] + Tiw[i * 4 + 2] * PQ[2] + Tiw[i * 4 + 3];
        gq[i] = Tiw[i * 4] * PQ[3] + Tiw[i * 4 + 1] * PQ[4] + Tiw[i * 4 + 2] * PQ[5] + Tiw[i * 4 + 3];
    }
    e[0] = l3[0] * (cam4[2] + cam4[0] * gp[0] / gp[2]) + l3[1] * (cam4[3] + cam4[1] * gp[1] / gp[2]) + l3[2];
    e[1] = l3[0] * (cam4[2] + cam4[0] * gq[0] / gq[2]) + l3[1] * (cam4[3] + cam4[1] * gq[1] / gq[2]) + l3[2];
    const double n = sqrt(e[0] * e[0] + e[1] * e[1]), dn = homog_th > n ? homog_th : n;
    /* NB the reference forms BOTH end points' pieces with (fx lx, fy ly) = (fx e0, fy e1)  (:1580-1583, reused at :1612) */
    lba_jac_pieces(gp, cam4[0] * e[0], cam4[1] * e[1], homog_th, JP, a3);
    lba_jac_pieces(gq, cam4[0] * e[0], cam4[1] * e[1], homog_th, JQ, b3);
    rowvec_R(a3, Tiw, Jl); rowvec_R(b3, Tiw, Jl + 3);
    for (int i = 0; i < 3; ++i) { Jl[i] *= e[0] / dn; Jl[3 + i] *= e[1] / dn; }
    for (int i = 0; i < 6; ++i) Jp[i] = (JP[i] * e[0] + JQ[i] * e[1]) / dn;
    *rn = n; *w = 1.0 / (1.0 + n * n);
}
/* dense LDL^T without pivoting (Eigen SimplicialLDLT on an SPD-by-construction matrix), solves A x = b in place of x */
static int ldlt_solve(double* A, int n, const double* b, double* x) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
        if (d == 0.0 || !isfinite(d)) return 0;
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * x[k]; x[i] = s; }
    for (int i = 0; i < n; ++i) x[i] /= A[(size_t)i * n + i];
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k]; x[i] = s; }
    return 1;
}
/* kf_loc[k]: local index of keyframe k (0 .. Nkf-1, ascending with k) or -1 = not optimised; T_kf_w16: K x 4x4 row-major map poses;
 * xyz3 / pq6 in = map estimates, out = optimised; T_out16: K x 16.  Returns 0, or -1 when there is no observation. */
void orc_lba_default_options(void* o_) {
    lba_opt_t* o = (lba_opt_t*)o_;
    o->lambda_lm = 1e-5; o->lambda_k = 10.0; o->max_iters = 15; o->homog_th = 1e-7; o->min_error = 1e-7; o->min_error_change = 1e-7; o->use_iterate_poses = 0; o->variant = 0;
}
/* same argument list as plba_lba_visual (include/plba.h); the problem handle is not used */
int orc_lba_visual(void* problem, const void* opt_, int K, const double* T_kf_w16, const int32_t* kf_loc, int Np, double* xyz3, int Nl, double* pq6,
                   int Ep, const int32_t* po_pt, const int32_t* po_kf, const double* uv2, int El, const int32_t* lo_ln, const int32_t* lo_kf, const double* l3,
                   double fx, double fy, double cx, double cy, double* T_out16, uint8_t* pt_moved, uint8_t* ln_moved, void* stats_) {
    (void)problem;
    const lba_opt_t* o = (const lba_opt_t*)opt_;
    lba_stats_t* st = (lba_stats_t*)stats_;
    memset(st, 0, sizeof *st);
    if (Ep + El == 0) return -1;
    int Nkf = 0;
    for (int k = 0; k < K; ++k) if (kf_loc[k] >= 0) ++Nkf;
    const int N = 6 * Nkf + 3 * Np + 6 * Nl;
    const double cam4[4] = {fx, fy, cx, cy};
    double* X = (double*)xcalloc(N, 8); double* DX = (double*)xcalloc(N, 8); double* g = (double*)xcalloc(N, 8);
    double* H = (double*)xcalloc((size_t)N * N, 8);
    for (int k = 0; k < K; ++k) if (kf_loc[k] >= 0) se3_logmap(T_kf_w16 + 16 * k, X + 6 * kf_loc[k]);      /* x_kf_w */
    memcpy(X + 6 * Nkf, xyz3, (size_t)3 * Np * 8); memcpy(X + 6 * Nkf + 3 * Np, pq6, (size_t)6 * Nl * 8);
    double err = 0.0, err_prev = 999999999.9, lambda = o->lambda_lm;
    int iters, updates = 0;
    for (iters = 0; iters < o->max_iters; ++iters) {
        memset(H, 0, (size_t)N * N * 8); memset(g, 0, (size_t)N * 8); memset(DX, 0, (size_t)N * 8);
        err = 0.0;
        for (int e = 0; e < Ep; ++e) {
            const int k = po_kf[e], loc = kf_loc[k], l = po_pt[e];
            double T[16], Tiw[16], rn, w, Jp[6], Jl[3];
            if (iters > 0 && loc >= 0) se3_expmap(X + 6 * loc, T); else memcpy(T, T_kf_w16 + 16 * k, 128);      /* :1726-1730 */
            se3_inverse(T, Tiw);
            lba_point_obs(cam4, o->homog_th, Tiw, X + 6 * Nkf + 3 * l, uv2 + 2 * e, &rn, &w, Jp, Jl);
            const int idx = 6 * loc, jdx = 6 * Nkf + 3 * l;
            err += rn * rn * w;
            for (int a = 0; a < 3; ++a) { g[jdx + a] += Jl[a] * rn * w; for (int b = 0; b < 3; ++b) H[(size_t)(jdx + a) * N + jdx + b] += Jl[a] * Jl[b] * w; }
            if (loc >= 0) {
                for (int a = 0; a < 6; ++a) {
                    g[idx + a] += Jp[a] * rn * w;
                    for (int b = 0; b < 6; ++b) H[(size_t)(idx + a) * N + idx + b] += Jp[a] * Jp[b] * w;
                    for (int b = 0; b < 3; ++b) { H[(size_t)(jdx + b) * N + idx + a] += Jl[b] * Jp[a] * w; H[(size_t)(idx + a) * N + jdx + b] += Jl[b] * Jp[a] * w; }
                }
            }
        }
        for (int e = 0; e < El; ++e) {
            const int k = lo_kf[e], loc = kf_loc[k], l = lo_ln[e];
            double T[16], Tiw[16], rn, w, Jp[6], Jl[6];
            if (iters > 0 && loc >= 0 && o->use_iterate_poses) se3_expmap(X + 6 * loc, T); else memcpy(T, T_kf_w16 + 16 * k, 128);   /* :1790: the map pose */
            se3_inverse(T, Tiw);
            lba_line_obs(cam4, o->homog_th, Tiw, X + 6 * Nkf + 3 * Np + 6 * l, l3 + 3 * e, &rn, &w, Jp, Jl);
            const int idx = 6 * loc, jdx = 6 * Nkf + 3 * Np + 6 * l;
            err += rn * rn * w;
            for (int a = 0; a < 6; ++a) { g[jdx + a] += Jl[a] * rn * w; for (int b = 0; b < 6; ++b) H[(size_t)(jdx + a) * N + jdx + b] += Jl[a] * Jl[b] * w; }
            if (loc >= 0) {
                for (int a = 0; a < 6; ++a) {
                    g[idx + a] += Jp[a] * rn * w;
                    for (int b = 0; b < 6; ++b) H[(size_t)(idx + a) * N + idx + b] += Jp[a] * Jp[b] * w;
                    for (int b = 0; b < 6; ++b) { H[(size_t)(jdx + b) * N + idx + a] += Jl[b] * Jp[a] * w; H[(size_t)(idx + a) * N + jdx + b] += Jl[b] * Jp[a] * w; }
                }
            }
        }
        if (iters == 0) {
            st->err_first = err / (double)(Ep + El);                       /* reported only */
            err /= 0.0;                                                    /* :1650 as coded: both counters are still 0 -> +inf (NaN for a zero sum), IEEE 754 */
            double Hmax = 0.0;
            for (int i = 0; i < N; ++i) { const double d = H[(size_t)i * N + i]; if (d > Hmax || d < -Hmax) Hmax = fabs(d); }
            if (o->variant == 1) Hmax = (double)(int)Hmax;                 /* GBA: `int Hmax` (:2468) truncates */
            lambda *= Hmax;                                                /* :1653-1659 */
        } else {
            if (o->variant == 1) err /= 0.0;                               /* GBA :2744: the zero counters again -> every pass is a "success" */
            else err /= (double)(Np + Nl);                                 /* :1882, as coded: the LANDMARK counts */
            if (fabs(err - err_prev) < o->min_error_change || err < o->min_error) break;      /* :1884-1885 */
        }
        for (int i = 0; i < N; ++i) H[(size_t)i * N + i] += lambda * H[(size_t)i * N + i];
        if (!ldlt_solve(H, N, g, DX)) { st->solver_failed = 1; break; }
        int do_update = 1;
        if (iters > 0) {
            if (err > err_prev) { lambda /= o->lambda_k; do_update = 0; }   /* :1894-1897 */
            else lambda *= o->lambda_k;
        }
        if (do_update) {
            ++updates;
            for (int i = 0; i < Nkf; ++i) {                                /* :1670-1675, :1901-1906 */
                double Tp[16], Td[16], Tdi[16], Tc[16];
                se3_expmap(X + 6 * i, Tp); se3_expmap(DX + 6 * i, Td); se3_inverse(Td, Tdi); se3_mul(Tp, Tdi, Tc);
                se3_logmap(Tc, X + 6 * i);
            }
            for (int i = 6 * Nkf; i < N; ++i) X[i] += DX[i];
        }
        if (iters > 0) { double nn = 0.0; for (int i = 0; i < N; ++i) nn += DX[i] * DX[i]; if (sqrt(nn) < o->min_error_change) { err_prev = err; ++iters; break; } }
        err_prev = err;
    }
    for (int k = 0; k < K; ++k) { if (kf_loc[k] >= 0) se3_expmap(X + 6 * kf_loc[k], T_out16 + 16 * k); else memcpy(T_out16 + 16 * k, T_kf_w16 + 16 * k, 128); }
    for (int i = 0; i < Np; ++i) {          /* :1944-1970: moved more than 1 cm -> the reference clears `inlier` */
        double n2 = 0.0;
        for (int c = 0; c < 3; ++c) { const double dl = X[6 * Nkf + 3 * i + c] - xyz3[3 * i + c]; n2 += dl * dl; }
        if (pt_moved) pt_moved[i] = sqrt(n2) > 0.01;
    }
    for (int i = 0; i < Nl; ++i) {
        double n2 = 0.0;
        for (int c = 0; c < 6; ++c) { const double dl = X[6 * Nkf + 3 * Np + 6 * i + c] - pq6[6 * i + c]; n2 += dl * dl; }
        if (ln_moved) ln_moved[i] = sqrt(n2) > 0.01;
    }
    memcpy(xyz3, X + 6 * Nkf, (size_t)3 * Np * 8); memcpy(pq6, X + 6 * Nkf + 3 * Np, (size_t)6 * Nl * 8);
    st->iterations = iters; st->updates = updates; st->err_last = err; st->lambda = lambda;
    free(X); free(DX); free(g); free(H);
    return 0;
}


/* ---- MapHandler::tryVioInit, the closed-form steps (src/mapHandler.cpp:4853-4980): gravity, accelerometer bias, velocities -------------
 * Independent of include/plba_g2o/vio_init.h: the two least-squares problems go through Householder QR here (eigen-decomposition of
 * the normal equations there).  Layout as vio_init.h: N keyframes, interval m = keyframe m -> m + 1. */
static void ls3_qr(double* A /* rows x 4: [A | b], overwritten */, int rows, double* x) {
    for (int k = 0; k < 3; ++k) {
        double nrm = 0.0;
        for (int i = k; i < rows; ++i) nrm += A[4 * i + k] * A[4 * i + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = A[4 * k + k] > 0 ? -nrm : nrm;
        double* v = (double*)xcalloc(rows, 8);
        for (int i = k; i < rows; ++i) v[i] = A[4 * i + k];
        v[k] -= alpha;
        double vv = 0.0;
        for (int i = k; i < rows; ++i) vv += v[i] * v[i];
        if (vv > 0.0)
            for (int c = k; c < 4; ++c) {
                double d = 0.0;
                for (int i = k; i < rows; ++i) d += v[i] * A[4 * i + c];
                d = 2.0 * d / vv;
                for (int i = k; i < rows; ++i) A[4 * i + c] -= d * v[i];
            }
        free(v);
    }
    for (int k = 2; k >= 0; --k) { double sacc = A[4 * k + 3]; for (int c = k + 1; c < 3; ++c) sacc -= A[4 * k + c] * x[c]; x[k] = sacc / A[4 * k + k]; }
}
void orc_vio_init(int N, const double* dt, const double* dP, const double* dV, const double* JPa, const double* JVa, const double* Rc, const double* pc,
                  const double* Rb, const double* pb, const double* Rcb, const double* pcb, double* gpre, double* g0, double* ba, double* V) {
    const int rows = 3 * (N - 2);
    double* S = (double*)xcalloc((size_t)rows * 4 + 4, 8);
    for (int pass = 0; pass < 2; ++pass) {
        memset(S, 0, (size_t)rows * 32);
        for (int i = 0; i < N - 2; ++i) {
            const double dt12 = dt[i], dt23 = dt[i + 1];
            const double *R1 = Rc + 9 * i, *R2 = Rc + 9 * (i + 1), *R3 = Rc + 9 * (i + 2), *p1 = pc + 3 * i, *p2 = pc + 3 * (i + 1), *p3 = pc + 3 * (i + 2);
            double R1cb[9], R2cb[9], a[3], b[3], c[3], t1[3], t2[3], D12[9], D23[9];
            m3_mul(R1, Rcb, R1cb); m3_mul(R2, Rcb, R2cb);
            m3_v(R1cb, dP + 3 * i, a); m3_v(R2cb, dP + 3 * (i + 1), b); m3_v(R1cb, dV + 3 * i, c);
            for (int t = 0; t < 9; ++t) { D12[t] = R1[t] - R2[t]; D23[t] = R2[t] - R3[t]; }
            m3_v(D12, pcb, t1); m3_v(D23, pcb, t2);
            if (pass == 0) {      /* C y = D: C = beta I, D = gamma - lambda (:4853-4893) */
                const double beta = 0.5 * (dt12 * dt12 * dt23 + dt12 * dt23 * dt23);
                for (int t = 0; t < 3; ++t) {
                    const double lam = (p2[t] - p1[t]) * dt23 + (p2[t] - p3[t]) * dt12;
                    const double gam = -t2[t] * dt12 + t1[t] * dt23 + a[t] * dt23 - b[t] * dt12 - c[t] * dt12 * dt23;      /* (Rc3 - Rc2) pcb = -D23 pcb */
                    S[4 * (3 * i + t) + t] = beta; S[4 * (3 * i + t) + 3] = gam - lam;
                }
            } else {              /* A y = B (:4903-4940) */
                double F1[9], F2[9], F3[9];
                m3_mul(R1cb, JPa + 9 * i, F1); m3_mul(R2cb, JPa + 9 * (i + 1), F2); m3_mul(R1cb, JVa + 9 * i, F3);
                for (int t = 0; t < 3; ++t) {
                    S[4 * (3 * i + t) + 3] = p2[t] * dt23 - p3[t] * dt12 - p1[t] * dt23 + p2[t] * dt12 + 0.5 * g0[t] * (dt12 * dt12 * dt23 + dt23 * dt23 * dt12)
                                           - a[t] * dt23 + b[t] * dt12 - t1[t] * dt23 + t2[t] * dt12 + c[t] * dt12 * dt23;
                    for (int u = 0; u < 3; ++u) S[4 * (3 * i + t) + u] = F1[3 * t + u] * dt23 - F2[3 * t + u] * dt12 - F3[3 * t + u] * dt12 * dt23;
                }
            }
        }
        if (pass == 0) {
            ls3_qr(S, rows, gpre);
            const double n = v3_norm(gpre);
            for (int t = 0; t < 3; ++t) g0[t] = gpre[t] / n * 9.810;
        } else ls3_qr(S, rows, ba);
    }
    free(S);
    for (int i = 0; i < N; ++i) {      /* :4955-4980 */
        double r[3];
        if (i != N - 1) { m3_v(Rb + 9 * i, dP + 3 * i, r); for (int t = 0; t < 3; ++t) V[3 * i + t] = (pb[3 * (i + 1) + t] - pb[3 * i + t] - 0.5 * g0[t] * dt[i] * dt[i] - r[t]) / dt[i]; }
        else { m3_v(Rb + 9 * (i - 1), dV + 3 * (i - 1), r); for (int t = 0; t < 3; ++t) V[3 * i + t] = V[3 * (i - 1) + t] + g0[t] * dt[i - 1] + r[t]; }
    }
}

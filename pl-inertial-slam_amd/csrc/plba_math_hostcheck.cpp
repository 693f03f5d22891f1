// Host-compiled wrapper around plba_math.h used ONLY by tests/test_device_math_host.py to compare the
// device formulas against the oracle on the CPU before they ever run on a GPU.  Not linked into
// libplba_hip.so and never used by the product path.
#include <string.h>
#include "plba_math.h"
using namespace plba;

static Cam mk_cam(const double* c) {
    Cam cam;
    cam.fx = c[0]; cam.fy = c[1]; cam.cx = c[2]; cam.cy = c[3];
    M3 Rbc = ld_m3(c + 4);
    cam.Rcb = transpose(Rbc);
    cam.c0 = mul(cam.Rcb, v3(c[13], c[14], c[15]));
    return cam;
}
extern "C" {
void hc_point_edge(const double* camv, const double* nav22, const double* Pw, const double* obs, double* e2, double* Jp12, double* Jl6, int* dpos) {
    Cam cam = mk_cam(camv);
    double s[24] = {0}, kc[12];
    memcpy(s, nav22, 22 * 8);
    kfcam_make(cam, s, kc);
    bool d;
    point_edge(cam, kc, v3(Pw[0], Pw[1], Pw[2]), obs[0], obs[1], e2, Jp12, Jl6, d, true);
    *dpos = d;
}
void hc_line_edge(const double* camv, const double* nav22, const double* L, const double* l3, int fix_q1, double* e2, double* Jp12, double* Jl6, int* dpos) {
    Cam cam = mk_cam(camv);
    double s[24] = {0}, kc[12];
    memcpy(s, nav22, 22 * 8);
    kfcam_make(cam, s, kc);
    bool d;
    line_edge(cam, kc, v3(L[0], L[1], L[2]), v3(L[3], L[4], L[5]), l3[0], l3[1], l3[2], fix_q1 != 0, e2, Jp12, Jl6, d, true);
    *dpos = d;
}
void hc_pvr_edge(const double* gw, const double* navi22, const double* navj22, const double* pre142, double* e9, double* J0, double* J1, double* J2) {
    double si[24] = {0}, sj[24] = {0};
    memcpy(si, navi22, 22 * 8); memcpy(sj, navj22, 22 * 8);
    V3 g = v3(gw[0], gw[1], gw[2]);
    pvr_error(si, sj, pre142, g, e9);
    memset(J0, 0, 81 * 8); memset(J1, 0, 81 * 8); memset(J2, 0, 54 * 8);
    pvr_jacobians(si, sj, pre142, g, e9, J0, J1, J2);
}
void hc_oplus_pvr(const double* nav22, const double* u9, double* out22) {
    double s[24] = {0}, o[24];
    memcpy(s, nav22, 22 * 8); memcpy(o, s, sizeof o);
    kf_oplus_pvr(s, u9, o);
    memcpy(out22, o, 22 * 8);
}
void hc_prior_dx_pvr(const double* nav22, const double* x0, double* dx9) {
    double s[24] = {0};
    memcpy(s, nav22, 22 * 8);
    prior_dx_pvr(s, x0, dx9);
}
// compact record path (what the kernels actually use): record -> rows -> Jp (2x6), Jl (2x3)
void hc_rec_edge(const double* camv, const double* nav22, const double* L, const double* obs, int is_pt, int fix_q1, double* e2, double* Jp12, double* Jl6) {
    Cam cam = mk_cam(camv);
    double s[24] = {0}, kc[12], rec[12];
    memcpy(s, nav22, 22 * 8);
    kfcam_make(cam, s, kc);
    bool d;
    if (is_pt) point_edge_rec(cam, kc, v3(L[0], L[1], L[2]), obs[0], obs[1], e2, rec, d, true);
    else line_edge_rec(cam, kc, v3(L[0], L[1], L[2]), v3(L[3], L[4], L[5]), obs[0], obs[1], obs[2], e2, rec, d, true);
    V3 va, vb; double ga[6], gb[6];
    rec_row(is_pt != 0, fix_q1 != 0, kc, cam.Rcb, v3(rec[0], rec[1], rec[2]), v3(rec[3], rec[4], rec[5]), va, ga);
    rec_row(is_pt != 0, fix_q1 != 0, kc, cam.Rcb, v3(rec[6], rec[7], rec[8]), v3(rec[9], rec[10], rec[11]), vb, gb);
    basis_apply(cam.Rcb, ga, Jp12);
    basis_apply(cam.Rcb, gb, Jp12 + 6);
    const double sl = is_pt ? -1.0 : 1.0;
    Jl6[0] = sl * va.x; Jl6[1] = sl * va.y; Jl6[2] = sl * va.z; Jl6[3] = sl * vb.x; Jl6[4] = sl * vb.y; Jl6[5] = sl * vb.z;
}
void hc_so3_exp(const double* w, double* q) { const Q4 r = so3_exp(v3(w[0], w[1], w[2])); q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
void hc_so3_log(const double* q, double* w) { Q4 a; a.x = q[0]; a.y = q[1]; a.z = q[2]; a.w = q[3]; const V3 r = so3_log(a); w[0] = r.x; w[1] = r.y; w[2] = r.z; }
void hc_so3_jr(const double* w, double* J) { const M3 r = so3_Jr(v3(w[0], w[1], w[2])); memcpy(J, r.a, 72); }
void hc_so3_jrinv(const double* w, double* J) { const M3 r = so3_JrInv(v3(w[0], w[1], w[2])); memcpy(J, r.a, 72); }
void hc_bias_error(const double* navi22, const double* navj22, double* e6) {
    double si[24] = {0}, sj[24] = {0};
    memcpy(si, navi22, 22 * 8); memcpy(sj, navj22, 22 * 8);
    bias_error(si, sj, e6);
}
void hc_sym3_inv(const double* h6, double lambda, double* d6) { sym3_inv(h6, lambda, d6); }
void hc_huber(double e, double delta, double* r) { huber(e, delta, r[0], r[1]); }
void hc_preint_update(double* pre142, const double* w, const double* a, double dt, double gcov, double acov) {
    preint_update(pre142, ld_v3(w), ld_v3(a), dt, gcov, acov);
}
}

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
void hc_sym3_inv(const double* h6, double lambda, double* d6) { sym3_inv(h6, lambda, d6); }
void hc_huber(double e, double delta, double* r) { huber(e, delta, r[0], r[1]); }
}

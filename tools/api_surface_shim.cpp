// C entry points over the host-evaluated ("API surface only") vertex / edge classes of the g2o facade, so
// that tests/test_api_surface.py can compare them with the oracle's restatements and with finite differences.
// Built by the test with plain g++ against libplba_hip.so; not part of the product library.
#include <sstream>

#include "plba_g2o/types_six_dof_expmap.h"
#include "plba_g2o/g2otypes.h"
#include "plba_g2o/types_slam3d.h"
#include "plba_g2o/vio_init.h"

using namespace g2o;

static SE3Quat make_se3(const double* q, const double* t) {
    plba::Q4 qq; qq.x = q[0]; qq.y = q[1]; qq.z = q[2]; qq.w = q[3];
    return SE3Quat::fromRaw(qq, plba::v3(t[0], t[1], t[2]));
}
static void put_se3(const SE3Quat& T, double* q, double* t) {
    q[0] = T.rawRotation().x; q[1] = T.rawRotation().y; q[2] = T.rawRotation().z; q[3] = T.rawRotation().w;
    t[0] = T.rawTranslation().x; t[1] = T.rawTranslation().y; t[2] = T.rawTranslation().z;
}

template <class S> static void cp(const S& s, double* d) { if (d) for (size_t i = 0; i < s.size(); ++i) d[i] = s[i]; }

extern "C" {

void shim_se3_exp(const double* u6, double* q, double* t) { Vector6d u; for (int i = 0; i < 6; ++i) u[i] = u6[i]; put_se3(SE3Quat::exp(u), q, t); }
void shim_se3_log(const double* q, const double* t, double* out6) { const Vector6d l = make_se3(q, t).log(); for (int i = 0; i < 6; ++i) out6[i] = l[i]; }
void shim_se3_oplus(const double* q, const double* t, const double* u6, double* qo, double* to) {
    VertexSE3Expmap v;
    v.setEstimate(make_se3(q, t));
    v.oplusImpl(u6);
    put_se3(v.estimate(), qo, to);
}
void shim_se3_inverse_mul(const double* q, const double* t, double* qo, double* to) { const SE3Quat T = make_se3(q, t); put_se3(T.inverse() * T, qo, to); }
void shim_se3_vertex_io(const double* q, const double* t, double* qo, double* to) {
    VertexSE3Expmap a, b;
    a.setEstimate(make_se3(q, t));
    std::stringstream ss;
    ss.precision(17);
    a.write(ss);
    b.read(ss);
    put_se3(b.estimate(), qo, to);
}
void shim_eval_se3_edge(int kind, const double* cam, const double* q, const double* t, const double* X, const double* obs,
                        double* err, double* Jpoint, double* Jpose, int* depth_pos, double* chi2) {
    VertexSE3Expmap pose; pose.setEstimate(make_se3(q, t)); pose.setId(1);
    VertexSBAPointXYZ pt; pt.setEstimate(Vector3d(X[0], X[1], X[2])); pt.setId(0);
    auto copy = [](const auto& s, double* d) { if (d) for (size_t i = 0; i < s.size(); ++i) d[i] = s[i]; };
    Matrix2d I2; I2.setIdentity();
    Matrix3d I3; I3.setIdentity();
    if (kind == 0) {
        EdgeSE3ProjectXYZ e; e.fx = cam[0]; e.fy = cam[1]; e.cx = cam[2]; e.cy = cam[3];
        e.setVertex(0, &pt); e.setVertex(1, &pose); e.setMeasurement(Vector2d(obs[0], obs[1])); e.setInformation(I2 * 2.0);
        e.computeError(); e.linearizeOplus();
        copy(e.error(), err); copy(e.jacobianOplusXi(), Jpoint); copy(e.jacobianOplusXj(), Jpose); *depth_pos = e.isDepthPositive(); *chi2 = e.chi2();
    } else if (kind == 1) {
        EdgeStereoSE3ProjectXYZ e; e.fx = cam[0]; e.fy = cam[1]; e.cx = cam[2]; e.cy = cam[3]; e.bf = cam[4];
        e.setVertex(0, &pt); e.setVertex(1, &pose); e.setMeasurement(Vector3d(obs[0], obs[1], obs[2])); e.setInformation(I3 * 2.0);
        e.computeError(); e.linearizeOplus();
        copy(e.error(), err); copy(e.jacobianOplusXi(), Jpoint); copy(e.jacobianOplusXj(), Jpose); *depth_pos = e.isDepthPositive(); *chi2 = e.chi2();
    } else if (kind == 2) {
        EdgeSE3ProjectXYZOnlyPose e; e.fx = cam[0]; e.fy = cam[1]; e.cx = cam[2]; e.cy = cam[3]; e.Xw = Vector3d(X[0], X[1], X[2]);
        e.setVertex(0, &pose); e.setMeasurement(Vector2d(obs[0], obs[1])); e.setInformation(I2 * 2.0);
        e.computeError(); e.linearizeOplus();
        copy(e.error(), err); copy(e.jacobianOplusXi(), Jpose); *depth_pos = e.isDepthPositive(); *chi2 = e.chi2();
    } else {
        EdgeStereoSE3ProjectXYZOnlyPose e; e.fx = cam[0]; e.fy = cam[1]; e.cx = cam[2]; e.cy = cam[3]; e.bf = cam[4]; e.Xw = Vector3d(X[0], X[1], X[2]);
        e.setVertex(0, &pose); e.setMeasurement(Vector3d(obs[0], obs[1], obs[2])); e.setInformation(I3 * 2.0);
        e.computeError(); e.linearizeOplus();
        copy(e.error(), err); copy(e.jacobianOplusXi(), Jpose); *depth_pos = e.isDepthPositive(); *chi2 = e.chi2();
    }
}


static NavState make_nav(const double* nav22) {
    NavState ns;
    double* s = ns.raw();
    for (int i = 0; i < 22; ++i) s[i] = nav22[i];
    return ns;
}
static void set_cam(const double* camv, double& fx, double& fy, double& cx, double& cy, Matrix3d& Rbc, Vector3d& Pbc) {
    fx = camv[0]; fy = camv[1]; cx = camv[2]; cy = camv[3];
    for (int i = 0; i < 3; ++i) { Pbc[i] = camv[13 + i]; for (int j = 0; j < 3; ++j) Rbc(i, j) = camv[4 + i * 3 + j]; }
}
// camv = fx fy cx cy Rbc(9, row-major) Pbc(3)  (oracle.cam_vec); nav22 = P V q(xyzw) bg ba dbg dba
void shim_eval_pvr_point_onlypose(const double* camv, const double* nav22, const double* Pw, const double* obs, double* err2, double* J18, int* dpos) {
    double fx, fy, cx, cy; Matrix3d Rbc; Vector3d Pbc;
    set_cam(camv, fx, fy, cx, cy, Rbc, Pbc);
    VertexNavStatePVR v; v.setEstimate(make_nav(nav22));
    EdgeNavStatePVRPointXYZOnlyPose e;
    e.SetParams(fx, fy, cx, cy, Rbc, Pbc, Vector3d(Pw[0], Pw[1], Pw[2]));
    e.setVertex(0, &v); e.setMeasurement(Vector2d(obs[0], obs[1]));
    e.computeError(); e.linearizeOplus();
    for (int i = 0; i < 2; ++i) err2[i] = e.error()[i];
    for (int i = 0; i < 18; ++i) J18[i] = e.jacobianOplusXi()[i];
    *dpos = e.isDepthPositive();
}
void shim_eval_linepoint(const double* camv, const double* nav22, const double* Pw, const double* obs3, double* err3, double* Ji9, double* Jj27, int* dpos) {
    double fx, fy, cx, cy; Matrix3d Rbc; Vector3d Pbc;
    set_cam(camv, fx, fy, cx, cy, Rbc, Pbc);
    VertexNavStatePVR v; v.setEstimate(make_nav(nav22));
    VertexLinePoint lp; lp.setEstimate(Vector3d(Pw[0], Pw[1], Pw[2]));
    EdgeNavStateLinePoint e;
    e.SetParams(fx, fy, cx, cy, Rbc, Pbc);
    e.setVertex(0, &lp); e.setVertex(1, &v); e.setMeasurement(Vector3d(obs3[0], obs3[1], obs3[2]));
    e.computeError(); e.linearizeOplus();
    for (int i = 0; i < 3; ++i) err3[i] = e.error()[i];
    for (int i = 0; i < 9; ++i) Ji9[i] = e.jacobianOplusXi()[i];
    for (int i = 0; i < 27; ++i) Jj27[i] = e.jacobianOplusXj()[i];
    *dpos = e.isDepthPositive();
}
void shim_eval_gyrbias(const double* dRbij, const double* JdRbg, const double* Rwbi, const double* Rwbj, const double* bg, double* err3, double* J9) {
    VertexGyrBias v; v.setEstimate(Vector3d(bg[0], bg[1], bg[2]));
    EdgeGyrBias e;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { e.dRbij(i, j) = dRbij[i * 3 + j]; e.J_dR_bg(i, j) = JdRbg[i * 3 + j]; e.Rwbi(i, j) = Rwbi[i * 3 + j]; e.Rwbj(i, j) = Rwbj[i * 3 + j]; }
    e.setVertex(0, &v);
    e.computeError(); e.linearizeOplus();
    for (int i = 0; i < 3; ++i) err3[i] = e.error()[i];
    for (int i = 0; i < 9; ++i) J9[i] = e.jacobianOplusXi()[i];
}


static void put_nav(const NavState& ns, double* out22) { for (int i = 0; i < 22; ++i) out22[i] = ns.raw()[i]; }

void shim_navstate_oplus(const double* nav22, const double* u15, double* out22) {
    VertexNavState v; v.setEstimate(make_nav(nav22)); v.oplusImpl(u15); put_nav(v.estimate(), out22);
}
void shim_gravity_oplus(const double* g3, const double* u2, double* out3) {
    VertexGravityW v; v.setEstimate(Vector3d(g3[0], g3[1], g3[2])); v.oplusImpl(u2);
    for (int i = 0; i < 3; ++i) out3[i] = v.estimate()[i];
}
void shim_navstate_edge(const double* gw, const double* navi, const double* navj, const double* pre142, int with_gw_vertex,
                        double* err15, double* Ji225, double* Jj225, double* Jg30) {
    VertexNavState vi, vj; vi.setEstimate(make_nav(navi)); vj.setEstimate(make_nav(navj));
    IMUPreintegrator M; M.setPayload(pre142);
    if (!with_gw_vertex) {
        EdgeNavState e; e.SetParams(Vector3d(gw[0], gw[1], gw[2]));
        e.setVertex(0, &vi); e.setVertex(1, &vj); e.setMeasurement(M);
        e.computeError(); e.linearizeOplus();
        cp(e.error(), err15); cp(e.jacobianOplusXi(), Ji225); cp(e.jacobianOplusXj(), Jj225);
    } else {
        VertexGravityW vg; vg.setEstimate(Vector3d(gw[0], gw[1], gw[2]));
        EdgeNavStateGw e;
        e.setVertex(0, &vi); e.setVertex(1, &vj); e.setVertex(2, &vg); e.setMeasurement(M);
        e.computeError(); e.linearizeOplus();
        cp(e.error(), err15); cp(e.jacobianOplus(0), Ji225); cp(e.jacobianOplus(1), Jj225); cp(e.jacobianOplus(2), Jg30);
    }
}
void shim_prior_edge(const double* prior22, const double* est22, double* err15, double* J225) {
    VertexNavState v; v.setEstimate(make_nav(est22));
    EdgeNavStatePrior e; e.setVertex(0, &v); e.setMeasurement(make_nav(prior22));
    e.computeError(); e.linearizeOplus();
    cp(e.error(), err15); cp(e.jacobianOplusXi(), J225);
}
void shim_prior_pvrbias_edge(const double* prior22, const double* pvr22, const double* bias22, double* err15, double* Ji135, double* Jj90) {
    VertexNavStatePVR a; a.setEstimate(make_nav(pvr22));
    VertexNavStateBias b; b.setEstimate(make_nav(bias22));
    EdgeNavStatePriorPVRBias e; e.setVertex(0, &a); e.setVertex(1, &b); e.setMeasurement(make_nav(prior22));
    e.computeError(); e.linearizeOplus();
    cp(e.error(), err15); cp(e.jacobianOplusXi(), Ji135); cp(e.jacobianOplusXj(), Jj90);
}
void shim_pvr_oplus(const double* nav22, const double* u9, double* out22) { VertexNavStatePVR v; v.setEstimate(make_nav(nav22)); v.oplusImpl(u9); put_nav(v.estimate(), out22); }
void shim_bias_oplus(const double* nav22, const double* u6, double* out22) { VertexNavStateBias v; v.setEstimate(make_nav(nav22)); v.oplusImpl(u6); put_nav(v.estimate(), out22); }
void shim_navstate_point(const double* camv, const double* nav22, const double* Pw, const double* obs, int only_pose, double* err2, double* Ji6, double* Jj30, int* dpos) {
    double fx, fy, cx, cy; Matrix3d Rbc; Vector3d Pbc;
    set_cam(camv, fx, fy, cx, cy, Rbc, Pbc);
    VertexNavState v; v.setEstimate(make_nav(nav22));
    if (only_pose) {
        EdgeNavStatePointXYZOnlyPose e; e.SetParams(fx, fy, cx, cy, Rbc, Pbc, Vector3d(Pw[0], Pw[1], Pw[2]));
        e.setVertex(0, &v); e.setMeasurement(Vector2d(obs[0], obs[1]));
        e.computeError(); e.linearizeOplus();
        cp(e.error(), err2); cp(e.jacobianOplusXi(), Jj30); *dpos = e.isDepthPositive();
    } else {
        VertexLMPointXYZ pt; pt.setEstimate(Vector3d(Pw[0], Pw[1], Pw[2]));
        EdgeNavStatePointXYZ e; e.SetParams(fx, fy, cx, cy, Rbc, Pbc);
        e.setVertex(0, &pt); e.setVertex(1, &v); e.setMeasurement(Vector2d(obs[0], obs[1]));
        e.computeError(); e.linearizeOplus();
        cp(e.error(), err2); cp(e.jacobianOplusXi(), Ji6); cp(e.jacobianOplusXj(), Jj30); *dpos = e.isDepthPositive();
    }
}

// g2o slam3d EdgeSE3 between two VertexSE3 (include/plba_g2o/types_slam3d.h); poses as (R row-major 9, t 3)
void shim_eval_edge_se3(const double* Xi12, const double* Xj12, const double* Z12, double* err6, double* Ji36, double* Jj36) {
    auto iso = [](const double* p) {
        Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
        Matrix3d R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = p[i * 3 + j];
        T.linear() = R; T.translation() = Vector3d(p[9], p[10], p[11]);
        return T;
    };
    VertexSE3 a, b; a.setEstimate(iso(Xi12)); b.setEstimate(iso(Xj12)); a.setId(0); b.setId(1);
    EdgeSE3 e; e.setVertex(0, &a); e.setVertex(1, &b); e.setMeasurement(iso(Z12));
    e.computeError(); e.linearizeOplus();
    for (int i = 0; i < 6; ++i) err6[i] = e.error()[i];
    for (int i = 0; i < 36; ++i) { Ji36[i] = e.jacobianOplusXi()[i]; Jj36[i] = e.jacobianOplusXj()[i]; }
}

// plba_vio::lstsq3 (include/plba_g2o/vio_init.h): JacobiSVD(A).solve(b) of src/mapHandler.cpp:4896, 4940
void shim_lstsq3(int rows, const double* A, const double* b, double* x3) {
    plba_vio::lstsq3(std::vector<double>(A, A + 3 * (size_t)rows), std::vector<double>(b, b + rows), x3);
}

}  // extern "C"

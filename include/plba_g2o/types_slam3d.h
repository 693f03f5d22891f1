// types_slam3d.h — drop-in for the two g2o slam3d types the reference's pose-graph optimisers use
// (src/mapHandler.cpp:4068-4297 loopClosureOptimizationEssGraphG2O, :4299-4528 ...CovGraphG2O; headers pulled by
// include/mapHandler.h:35-36): g2o::VertexSE3 and g2o::EdgeSE3.  SURVEY §8f row 4: host-evaluated, the graph runs
// SparseOptimizer's host Levenberg loop (g2o_compat.h).
//
// g2o is third-party and un-vendored; what is restated here is its published slam3d parametrisation:
//   vertex estimate     Isometry3 X; update  X <- X * fromVectorMQT(u),  u = (t, q_xyz), q_w = sqrt(1 - |q_xyz|^2)
//                       (re-orthogonalised every 1000 updates: VertexSE3::orthogonalizeAfter)
//   edge error          e = toVectorMQT(Z^-1 * Xi^-1 * Xj) = (translation, xyz of the unit quaternion with w >= 0)
//   edge Jacobians      analytic first derivatives of e with respect to the two updates (computeEdgeSE3Gradient computes the
//                       same quantities; derivation next to the code)
#pragma once
#include "plba_g2o/g2o_compat.h"
#include "plba_g2o/se3quat.h"

namespace g2o {

namespace slam3d_detail {
struct Rt { plba::M3 R; plba::V3 t; };
inline Rt from_iso(const Eigen::Isometry3d& T) {
    Rt r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.R.a[i * 3 + j] = T.linear()(i, j);
    r.t = plba::v3(T.translation()(0), T.translation()(1), T.translation()(2));
    return r;
}
inline Eigen::Isometry3d to_iso(const Rt& a) {
    Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
    Matrix3d R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = a.R.a[i * 3 + j];
    T.linear() = R;
    T.translation() = Vector3d(a.t.x, a.t.y, a.t.z);
    return T;
}
inline Rt mul(const Rt& a, const Rt& b) { Rt r; r.R = plba::mul(a.R, b.R); r.t = plba::mul(a.R, b.t) + a.t; return r; }
inline Rt inv(const Rt& a) { Rt r; r.R = plba::transpose(a.R); const plba::V3 x = plba::mul(r.R, a.t); r.t = plba::v3(-x.x, -x.y, -x.z); return r; }
// internal::fromVectorMQT: translation + compact quaternion (w = sqrt(1 - |v|^2), identity rotation if |v| > 1)
inline Rt from_mqt(const double* u) {
    Rt r;
    r.t = plba::v3(u[0], u[1], u[2]);
    const double w2 = 1.0 - (u[3] * u[3] + u[4] * u[4] + u[5] * u[5]);
    if (w2 < 0) r.R = plba::eye3();
    else { plba::Q4 q; q.x = u[3]; q.y = u[4]; q.z = u[5]; q.w = std::sqrt(w2); r.R = plba::q_to_R(q); }
    return r;
}
// internal::toVectorMQT: translation + xyz of the normalised quaternion with w >= 0
inline plba::Q4 unit_q(const plba::M3& R) {
    plba::Q4 q = plba::q_normalized(plba::R_to_q(R));
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    return q;
}
}  // namespace slam3d_detail

class VertexSE3 : public BaseVertex<6, Eigen::Isometry3d> {
public:
    static const int orthogonalizeAfter = 1000;
    VertexSE3() { setToOriginImpl(); }
    void setToOriginImpl() override { _estimate = Eigen::Isometry3d::Identity(); }
    void oplusImpl(const double* u) override {
        using namespace slam3d_detail;
        Rt X = mul(from_iso(_estimate), from_mqt(u));
        if (++_numOplusCalls > orthogonalizeAfter) {            // approximateNearestOrthogonalMatrix: R -= 0.5 R (R^T R - I)
            _numOplusCalls = 0;
            plba::M3 E = plba::mulAtB(X.R, X.R);
            E.a[0] -= 1; E.a[4] -= 1; E.a[8] -= 1;
            const plba::M3 RE = plba::mul(X.R, E);
            for (int i = 0; i < 9; ++i) X.R.a[i] -= 0.5 * RE.a[i];
        }
        _estimate = to_iso(X);
    }
    int estimateDimension() const override { return 7; }
    SE3Quat estimateAsSE3Quat() const { return SE3Quat(_estimate.linear(), _estimate.translation()); }      // internal::toSE3Quat
    void setEstimateFromSE3Quat(const SE3Quat& s) { _estimate = s; }
private:
    int _numOplusCalls = 0;
};

class EdgeSE3 : public BaseBinaryEdge<6, Eigen::Isometry3d, VertexSE3, VertexSE3> {
public:
    EdgeSE3() { allocJacobians({6, 6}); _measurement = Eigen::Isometry3d::Identity(); _inverseMeasurement = _measurement; for (int i = 0; i < 6; ++i) _info[(size_t)i * 6 + i] = 1.0; }
    void setMeasurement(const Eigen::Isometry3d& m) { _measurement = m; _inverseMeasurement = m.inverse(); }
    void computeError() override {
        using namespace slam3d_detail;
        const Rt E = mul(mul(from_iso(_inverseMeasurement), inv(X(0))), X(1));
        const plba::Q4 q = unit_q(E.R);
        _error[0] = E.t.x; _error[1] = E.t.y; _error[2] = E.t.z; _error[3] = q.x; _error[4] = q.y; _error[5] = q.z;
    }
    // E = Z^-1 Xi^-1 Xj = (R_E, t_E), q_E = (w, v) its unit quaternion (w >= 0), B = Xi^-1 Xj.  First order in the updates:
    //   Xj <- Xj D(u):  t' = t_E + R_E u_t,  q' = q_E (1, u_r)            =>  de_t/du_t = R_E,      de_q/du_r = w I + [v]x
    //   Xi <- Xi D(u):  E' = Z^-1 D^-1 B:  t' = R_Z^T (t_B - u_t - 2 u_r x t_B) - R_Z^T t_Z,  q' = (1, -R_Z^T u_r) q_E
    //                                                   =>  de_t/du_t = -R_Z^T,  de_t/du_r = 2 R_Z^T [t_B]x,  de_q/du_r = -(w I - [v]x) R_Z^T
    void linearizeOplus() override {
        using namespace slam3d_detail;
        const Rt Zi = from_iso(_inverseMeasurement), B = mul(inv(X(0)), X(1)), E = mul(Zi, B);
        const plba::Q4 q = unit_q(E.R);
        const plba::M3 RZt = Zi.R, V = plba::hat(plba::v3(q.x, q.y, q.z)), TB = plba::hat(B.t), RZtTB = plba::mul(RZt, TB);
        plba::M3 Qm, Qp;                                          // w I - [v]x,  w I + [v]x
        for (int i = 0; i < 9; ++i) { const double d = (i % 4 == 0) ? q.w : 0.0; Qm.a[i] = d - V.a[i]; Qp.a[i] = d + V.a[i]; }
        const plba::M3 QmRZt = plba::mul(Qm, RZt);
        std::fill(_jac[0].begin(), _jac[0].end(), 0.0); std::fill(_jac[1].begin(), _jac[1].end(), 0.0);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            J(0, 6, r, c) = -RZt.a[r * 3 + c];
            J(0, 6, r, 3 + c) = 2.0 * RZtTB.a[r * 3 + c];
            J(0, 6, 3 + r, 3 + c) = -QmRZt.a[r * 3 + c];
            J(1, 6, r, c) = E.R.a[r * 3 + c];
            J(1, 6, 3 + r, 3 + c) = Qp.a[r * 3 + c];
        }
    }
    double chi2() const override { return chi2FromError(); }
    // EdgeSE3::initialEstimatePossible / initialEstimate: to = from * Z (from = vertex 0), from = to * Z^-1 otherwise
    double initialEstimatePossible(const OptimizableGraph::Vertex*, const OptimizableGraph::Vertex*) override { return 1.0; }
    void initialEstimate(const OptimizableGraph::Vertex* from, OptimizableGraph::Vertex* to) override {
        VertexSE3* a = static_cast<VertexSE3*>(_vertices[0]); VertexSE3* b = static_cast<VertexSE3*>(_vertices[1]);
        if (from == a && to == b) b->setEstimate(a->estimate() * _measurement);
        else if (from == b && to == a) a->setEstimate(b->estimate() * _inverseMeasurement);
    }
private:
    slam3d_detail::Rt X(int k) const { return slam3d_detail::from_iso(static_cast<const VertexSE3*>(_vertices[k])->estimate()); }
    Eigen::Isometry3d _inverseMeasurement;
};

}  // namespace g2o

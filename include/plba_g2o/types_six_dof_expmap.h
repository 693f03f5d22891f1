// types_six_dof_expmap.h — drop-in for the reference header IMU/types_six_dof_expmap.h (:59-202, .cpp:40-364):
// the SE(3)-pose vertex and the four mono / stereo reprojection edges of the ORB-SLAM lineage.  They are named by the
// task but have NO call site in the local-BA path, so they are API surface only: residuals and Jacobians
// are evaluated on the host with the reference's formulas (including the float `invz` of the stereo projections,
// .cpp:150-157, 299-306) and a graph containing them is not accepted by the HIP-backed SparseOptimizer::optimize.
#pragma once
#include "plba_g2o/g2o_compat.h"
#include "plba_g2o/se3quat.h"
#include "plba_g2o/types_sba.h"

namespace g2o {

// SE3 vertex: estimate = world -> camera, update = exp(delta) * estimate (h:59-77)
class VertexSE3Expmap : public BaseVertex<6, SE3Quat> {
public:
    void setToOriginImpl() override { _estimate = SE3Quat(); }
    void oplusImpl(const double* u) override { Vector6d d; for (int i = 0; i < 6; ++i) d[i] = u[i]; setEstimate(SE3Quat::exp(d) * estimate()); }
    int estimateDimension() const override { return 6; }
    bool read(std::istream& is) override {          // cpp:48-57: the file holds camera -> world
        Vector7d est; for (int i = 0; i < 7; ++i) is >> est[i];
        SE3Quat cam2world; cam2world.fromVector(est);
        setEstimate(cam2world.inverse());
        return true;
    }
    bool write(std::ostream& os) const override {    // cpp:59-64
        const SE3Quat cam2world(estimate().inverse());
        for (int i = 0; i < 7; ++i) os << cam2world[i] << " ";
        return os.good();
    }
};

namespace plba_detail {
// d(pixel)/d(pose update) of the mono projection, rows 0-1 of every edge below (cpp:124-136, 207-219, 262-274, 333-345)
inline void pose_jac_rows(double* Jrow0, double* Jrow1, double x, double y, double invz, double invz_2, double fx, double fy) {
    Jrow0[0] = x * y * invz_2 * fx; Jrow0[1] = -(1 + (x * x * invz_2)) * fx; Jrow0[2] = y * invz * fx;
    Jrow0[3] = -invz * fx; Jrow0[4] = 0; Jrow0[5] = x * invz_2 * fx;
    Jrow1[0] = (1 + y * y * invz_2) * fy; Jrow1[1] = -x * y * invz_2 * fy; Jrow1[2] = -x * invz * fy;
    Jrow1[3] = 0; Jrow1[4] = -invz * fy; Jrow1[5] = y * invz_2 * fy;
}
template <typename Edge> inline void read_edge(Edge& e, std::istream& is, int nmeas, int D) {
    auto m = e.measurement();
    for (int i = 0; i < nmeas; ++i) is >> m[i];
    e.setMeasurement(m);
    std::vector<double> info((size_t)D * D, 0.0);
    for (int i = 0; i < D; ++i) for (int j = i; j < D; ++j) { is >> info[(size_t)i * D + j]; info[(size_t)j * D + i] = info[(size_t)i * D + j]; }
    e.setInformationRowMajor(info);
}
template <typename Edge> inline bool write_edge(const Edge& e, std::ostream& os, int nmeas, int D) {
    for (int i = 0; i < nmeas; ++i) os << e.measurement()[i] << " ";
    for (int i = 0; i < D; ++i) for (int j = i; j < D; ++j) os << " " << e.informationRowMajor()[(size_t)i * D + j];
    return os.good();
}
}  // namespace plba_detail

// mono point-to-pose edge (h:80-109)
class EdgeSE3ProjectXYZ : public BaseBinaryEdge<2, Vector2d, VertexSBAPointXYZ, VertexSE3Expmap> {
public:
    EdgeSE3ProjectXYZ() { allocJacobians({3, 6}); }
    void computeError() override {
        const plba::V3 pc = transformed();
        const Vector2d z = cam_project(Vector3d(pc.x, pc.y, pc.z));
        _error[0] = _measurement[0] - z[0]; _error[1] = _measurement[1] - z[1];
    }
    bool isDepthPositive() { return transformed().z > 0.0; }
    void linearizeOplus() override {                 // cpp:103-137
        const SE3Quat& T = static_cast<const VertexSE3Expmap*>(_vertices[1])->estimate();
        const plba::V3 p = transformed();
        const double x = p.x, y = p.y, z = p.z, z_2 = z * z;
        const plba::M3 R = plba::q_to_R(T.rawRotation());
        const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += (-1. / z * tmp[r * 3 + k]) * R.a[k * 3 + c];
                J(0, 3, r, c) = s;
            }
        double* j = _jac[1].data();
        j[0] = x * y / z_2 * fx; j[1] = -(1 + (x * x / z_2)) * fx; j[2] = y / z * fx; j[3] = -1. / z * fx; j[4] = 0; j[5] = x / z_2 * fx;
        j[6] = (1 + y * y / z_2) * fy; j[7] = -x * y / z_2 * fy; j[8] = -x / z * fy; j[9] = 0; j[10] = -1. / z * fy; j[11] = y / z_2 * fy;
    }
    Vector2d cam_project(const Vector3d& p) const { return Vector2d(p[0] / p[2] * fx + cx, p[1] / p[2] * fy + cy); }   // cpp:139-145
    double chi2() const override { return chi2FromError(); }
    bool read(std::istream& is) override { plba_detail::read_edge(*this, is, 2, 2); return true; }
    bool write(std::ostream& os) const override { return plba_detail::write_edge(*this, os, 2, 2); }
    double fx = 0, fy = 0, cx = 0, cy = 0;
private:
    plba::V3 transformed() const {
        const SE3Quat& T = static_cast<const VertexSE3Expmap*>(_vertices[1])->estimate();
        const Vector3d& X = static_cast<const VertexSBAPointXYZ*>(_vertices[0])->estimate();
        return T.mapRaw(plba::v3(X[0], X[1], X[2]));
    }
};

// stereo point-to-pose edge, measurement (u_l, v, u_r) (h:112-141)
class EdgeStereoSE3ProjectXYZ : public BaseBinaryEdge<3, Vector3d, VertexSBAPointXYZ, VertexSE3Expmap> {
public:
    EdgeStereoSE3ProjectXYZ() { allocJacobians({3, 6}); }
    void computeError() override {
        const plba::V3 pc = transformed();
        const Vector3d z = cam_project(Vector3d(pc.x, pc.y, pc.z), (float)bf);
        for (int i = 0; i < 3; ++i) _error[i] = _measurement[i] - z[i];
    }
    bool isDepthPositive() { return transformed().z > 0.0; }
    void linearizeOplus() override {                 // cpp:182-235
        const SE3Quat& T = static_cast<const VertexSE3Expmap*>(_vertices[1])->estimate();
        const plba::V3 p = transformed();
        const plba::M3 R = plba::q_to_R(T.rawRotation());
        const double x = p.x, y = p.y, z = p.z, z_2 = z * z;
        for (int c = 0; c < 3; ++c) {
            J(0, 3, 0, c) = -fx * R.a[c] / z + fx * x * R.a[6 + c] / z_2;
            J(0, 3, 1, c) = -fy * R.a[3 + c] / z + fy * y * R.a[6 + c] / z_2;
            J(0, 3, 2, c) = J(0, 3, 0, c) - bf * R.a[6 + c] / z_2;
        }
        double* j = _jac[1].data();
        j[0] = x * y / z_2 * fx; j[1] = -(1 + (x * x / z_2)) * fx; j[2] = y / z * fx; j[3] = -1. / z * fx; j[4] = 0; j[5] = x / z_2 * fx;
        j[6] = (1 + y * y / z_2) * fy; j[7] = -x * y / z_2 * fy; j[8] = -x / z * fy; j[9] = 0; j[10] = -1. / z * fy; j[11] = y / z_2 * fy;
        j[12] = j[0] - bf * y / z_2; j[13] = j[1] + bf * x / z_2; j[14] = j[2]; j[15] = j[3]; j[16] = 0; j[17] = j[5] - bf / z_2;
    }
    // cpp:150-157: 1/z is rounded to float, and so is the baseline term bf * invz (both operands are float there)
    Vector3d cam_project(const Vector3d& p, const float& bf_) const {
        const float invz = (float)(1.0f / p[2]);
        Vector3d r;
        r[0] = p[0] * invz * fx + cx;
        r[1] = p[1] * invz * fy + cy;
        r[2] = r[0] - bf_ * invz;
        return r;
    }
    double chi2() const override { return chi2FromError(); }
    bool read(std::istream& is) override { plba_detail::read_edge(*this, is, 3, 3); return true; }
    bool write(std::ostream& os) const override { return plba_detail::write_edge(*this, os, 3, 3); }
    double fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
private:
    plba::V3 transformed() const {
        const SE3Quat& T = static_cast<const VertexSE3Expmap*>(_vertices[1])->estimate();
        const Vector3d& X = static_cast<const VertexSBAPointXYZ*>(_vertices[0])->estimate();
        return T.mapRaw(plba::v3(X[0], X[1], X[2]));
    }
};

// mono pose-only edge, fixed world point Xw (h:143-171)
class EdgeSE3ProjectXYZOnlyPose : public BaseUnaryEdge<2, Vector2d, VertexSE3Expmap> {
public:
    EdgeSE3ProjectXYZOnlyPose() { allocJacobians({6}); }
    void computeError() override {
        const plba::V3 pc = transformed();
        const Vector2d z = cam_project(Vector3d(pc.x, pc.y, pc.z));
        _error[0] = _measurement[0] - z[0]; _error[1] = _measurement[1] - z[1];
    }
    bool isDepthPositive() { return transformed().z > 0.0; }
    void linearizeOplus() override {                 // cpp:258-281
        const plba::V3 p = transformed();
        const double invz = 1.0 / p.z, invz_2 = invz * invz;
        plba_detail::pose_jac_rows(_jac[0].data(), _jac[0].data() + 6, p.x, p.y, invz, invz_2, fx, fy);
    }
    Vector2d cam_project(const Vector3d& p) const { return Vector2d(p[0] / p[2] * fx + cx, p[1] / p[2] * fy + cy); }   // cpp:283-289
    double chi2() const override { return chi2FromError(); }
    bool read(std::istream& is) override { plba_detail::read_edge(*this, is, 2, 2); return true; }
    bool write(std::ostream& os) const override { return plba_detail::write_edge(*this, os, 2, 2); }
    Vector3d Xw;
    double fx = 0, fy = 0, cx = 0, cy = 0;
private:
    plba::V3 transformed() const { return static_cast<const VertexSE3Expmap*>(_vertices[0])->estimate().mapRaw(plba::v3(Xw[0], Xw[1], Xw[2])); }
};

// stereo pose-only edge (h:174-202)
class EdgeStereoSE3ProjectXYZOnlyPose : public BaseUnaryEdge<3, Vector3d, VertexSE3Expmap> {
public:
    EdgeStereoSE3ProjectXYZOnlyPose() { allocJacobians({6}); }
    void computeError() override {
        const plba::V3 pc = transformed();
        const Vector3d z = cam_project(Vector3d(pc.x, pc.y, pc.z));
        for (int i = 0; i < 3; ++i) _error[i] = _measurement[i] - z[i];
    }
    bool isDepthPositive() { return transformed().z > 0.0; }
    void linearizeOplus() override {                 // cpp:328-358
        const plba::V3 p = transformed();
        const double x = p.x, y = p.y, invz = 1.0 / p.z, invz_2 = invz * invz;
        double* j = _jac[0].data();
        plba_detail::pose_jac_rows(j, j + 6, x, y, invz, invz_2, fx, fy);
        j[12] = j[0] - bf * y * invz_2; j[13] = j[1] + bf * x * invz_2; j[14] = j[2]; j[15] = j[3]; j[16] = 0; j[17] = j[5] - bf * invz_2;
    }
    // cpp:299-306: 1/z is rounded to float; here bf stays double
    Vector3d cam_project(const Vector3d& p) const {
        const float invz = (float)(1.0f / p[2]);
        Vector3d r;
        r[0] = p[0] * invz * fx + cx;
        r[1] = p[1] * invz * fy + cy;
        r[2] = r[0] - bf * invz;
        return r;
    }
    double chi2() const override { return chi2FromError(); }
    bool read(std::istream& is) override { plba_detail::read_edge(*this, is, 3, 3); return true; }
    bool write(std::ostream& os) const override { return plba_detail::write_edge(*this, os, 3, 3); }
    Vector3d Xw;
    double fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
private:
    plba::V3 transformed() const { return static_cast<const VertexSE3Expmap*>(_vertices[0])->estimate().mapRaw(plba::v3(Xw[0], Xw[1], Xw[2])); }
};

}  // namespace g2o

// se3quat.h — drop-in for the reference header IMU/se3quat.h (g2o::SE3Quat, :43-305): rigid transform stored as a
// unit quaternion (w >= 0) plus a translation.  API surface only ("named by north_star but not reached
// by the localBA call sites"): plain host arithmetic, no device path.  Conventions follow the reference: exp() takes
// (omega, upsilon) with the small-angle branch R = I + W + W^2, V = R below 1e-5 (:237-243); log() the 0.99999 branch
// (:181-200); map(x) = r * x + t with Eigen's quaternion rotation.
#pragma once
#include "plba_g2o/g2o_compat.h"

namespace g2o {

class SE3Quat {
public:
    SE3Quat() { q_.x = q_.y = q_.z = 0.0; q_.w = 1.0; t_ = plba::v3(0, 0, 0); }
    SE3Quat(const Matrix3d& R, const Vector3d& t) {            // :57-59
        plba::M3 m;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.a[i * 3 + j] = R(i, j);
        q_ = plba::R_to_q(m); t_ = plba::v3(t[0], t[1], t[2]);
        normalizeRotation();
    }
    SE3Quat(const Quaterniond& q, const Vector3d& t) {         // :61-63
        q_.x = q.x(); q_.y = q.y(); q_.z = q.z(); q_.w = q.w(); t_ = plba::v3(t[0], t[1], t[2]);
        normalizeRotation();
    }
    static SE3Quat fromRaw(plba::Q4 q, plba::V3 t) { SE3Quat s; s.q_ = q; s.t_ = t; s.normalizeRotation(); return s; }

    Vector3d translation() const { return Vector3d(t_.x, t_.y, t_.z); }
    void setTranslation(const Vector3d& t) { t_ = plba::v3(t[0], t[1], t[2]); }
    Quaterniond rotation() const { return Quaterniond(q_.w, q_.x, q_.y, q_.z); }
    void setRotation(const Quaterniond& q) { q_.x = q.x(); q_.y = q.y(); q_.z = q.z(); q_.w = q.w(); }
    const plba::Q4& rawRotation() const { return q_; }
    const plba::V3& rawTranslation() const { return t_; }
    Matrix3d rotationMatrix() const { const plba::M3 m = plba::q_to_R(q_); Matrix3d R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = m.a[i * 3 + j]; return R; }

    SE3Quat operator*(const SE3Quat& o) const {                // :101-107
        SE3Quat r(*this);
        r.t_ = r.t_ + plba::q_rot(q_, o.t_);
        r.q_ = plba::q_mul(q_, o.q_);
        r.normalizeRotation();
        return r;
    }
    SE3Quat& operator*=(const SE3Quat& o) { *this = *this * o; return *this; }
    Vector3d operator*(const Vector3d& v) const { return map(v); }
    SE3Quat inverse() const {                                  // :121-126 (no re-normalisation there either)
        SE3Quat r;
        r.q_ = plba::q_conj(q_);
        r.t_ = plba::q_rot(r.q_, plba::v3(-t_.x, -t_.y, -t_.z));
        return r;
    }
    double operator[](int i) const { return i < 3 ? (i == 0 ? t_.x : i == 1 ? t_.y : t_.z) : (i == 3 ? q_.x : i == 4 ? q_.y : i == 5 ? q_.z : q_.w); }
    Vector7d toVector() const { Vector7d v; for (int i = 0; i < 7; ++i) v[i] = (*this)[i]; return v; }
    void fromVector(const Vector7d& v) { t_ = plba::v3(v[0], v[1], v[2]); q_.x = v[3]; q_.y = v[4]; q_.z = v[5]; q_.w = v[6]; }   // :150-153 (as given)
    Vector6d toMinimalVector() const { Vector6d v; for (int i = 0; i < 6; ++i) v[i] = (*this)[i]; return v; }
    void fromMinimalVector(const Vector6d& v) {                // :166-174
        const double w = 1. - v[3] * v[3] - v[4] * v[4] - v[5] * v[5];
        if (w > 0) { q_.w = std::sqrt(w); q_.x = v[3]; q_.y = v[4]; q_.z = v[5]; }
        else { q_.w = 0; q_.x = -v[3]; q_.y = -v[4]; q_.z = -v[5]; }
        t_ = plba::v3(v[0], v[1], v[2]);
    }
    Vector3d map(const Vector3d& p) const { const plba::V3 r = mapRaw(plba::v3(p[0], p[1], p[2])); return Vector3d(r.x, r.y, r.z); }   // :217-220
    plba::V3 mapRaw(plba::V3 p) const { return plba::q_rot(q_, p) + t_; }

    static void expRaw(const double* u6, plba::Q4& q, plba::V3& t) {   // :223-257
        const plba::V3 omega = plba::v3(u6[0], u6[1], u6[2]), upsilon = plba::v3(u6[3], u6[4], u6[5]);
        const double theta = plba::norm(omega);
        const plba::M3 W = plba::hat(omega), W2 = plba::mul(W, W), I = plba::eye3();
        plba::M3 R, V;
        if (theta < 0.00001) {
            for (int i = 0; i < 9; ++i) R.a[i] = I.a[i] + W.a[i] + W2.a[i];
            V = R;
        } else {
            const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta), c = (theta - std::sin(theta)) / std::pow(theta, 3);
            for (int i = 0; i < 9; ++i) { R.a[i] = I.a[i] + a * W.a[i] + b * W2.a[i]; V.a[i] = I.a[i] + b * W.a[i] + c * W2.a[i]; }
        }
        q = plba::R_to_q(R);
        t = plba::mul(V, upsilon);
    }
    static SE3Quat exp(const Vector6d& update) {
        double u[6]; for (int i = 0; i < 6; ++i) u[i] = update[i];
        plba::Q4 q; plba::V3 t;
        expRaw(u, q, t);
        return fromRaw(q, t);                                  // SE3Quat(Quaterniond(R), V * upsilon) normalises
    }
    Vector6d log() const {                                     // :178-215
        const plba::M3 R = plba::q_to_R(q_);
        const double d = 0.5 * (R.a[0] + R.a[4] + R.a[8] - 1);
        const plba::V3 dR = plba::v3(R.a[7] - R.a[5], R.a[2] - R.a[6], R.a[3] - R.a[1]);
        plba::V3 omega;
        plba::M3 Vinv;
        const plba::M3 I = plba::eye3();
        if (d > 0.99999) {
            omega = 0.5 * dR;
            const plba::M3 W = plba::hat(omega), W2 = plba::mul(W, W);
            for (int i = 0; i < 9; ++i) Vinv.a[i] = I.a[i] - 0.5 * W.a[i] + (1. / 12.) * W2.a[i];
        } else {
            const double theta = std::acos(d);
            omega = (theta / (2 * std::sqrt(1 - d * d))) * dR;
            const plba::M3 W = plba::hat(omega), W2 = plba::mul(W, W);
            const double k = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
            for (int i = 0; i < 9; ++i) Vinv.a[i] = I.a[i] - 0.5 * W.a[i] + k * W2.a[i];
        }
        const plba::V3 ups = plba::mul(Vinv, t_);
        Vector6d r;
        r[0] = omega.x; r[1] = omega.y; r[2] = omega.z; r[3] = ups.x; r[4] = ups.y; r[5] = ups.z;
        return r;
    }
    // g2o's slam3d types take rigid transforms as Eigen::Isometry3d: `v_se3->setEstimate(g2o::SE3Quat::exp(x))` and
    // `e_se3->setMeasurement(g2o::SE3Quat::exp(x))` (src/mapHandler.cpp:4122,4152,4173) rely on this conversion
    operator Eigen::Isometry3d() const {
        Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
        const plba::M3 m = plba::q_to_R(q_);
        Matrix3d R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = m.a[i * 3 + j];
        T.linear() = R;
        T.translation() = Vector3d(t_.x, t_.y, t_.z);
        return T;
    }
    void normalizeRotation() {                                 // :283-288
        if (q_.w < 0) { q_.x = -q_.x; q_.y = -q_.y; q_.z = -q_.z; q_.w = -q_.w; }
        q_ = plba::q_normalized(q_);
    }
private:
    plba::Q4 q_;
    plba::V3 t_;
};

}  // namespace g2o

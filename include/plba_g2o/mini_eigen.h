// mini_eigen.h — the handful of Eigen types the g2o-compatible facade touches, for builds where Eigen3 is not
// installed (this image).  With Eigen present (`__has_include(<Eigen/Core>)`, the reference's build environment)
// g2o_compat.h includes the real headers instead and this file is not used.  Column-major like Eigen, so
// `.data()` layouts agree.  Only storage, element access and the few operators the facade/harness need.
#pragma once
#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <vector>

namespace Eigen {

constexpr int Dynamic = -1;

template <typename S, int R, int C>
class Matrix {
    static constexpr bool kDyn = (R == Dynamic || C == Dynamic);
    std::vector<S> v_;
    int r_ = (R == Dynamic ? 0 : R), c_ = (C == Dynamic ? 0 : C);

public:
    typedef S Scalar;
    Matrix() { if (!kDyn) v_.assign((size_t)R * C, S(0)); }
    Matrix(int r, int c) : r_(r), c_(c) { v_.assign((size_t)r * c, S(0)); }
    explicit Matrix(int n) : r_(C == 1 ? n : 1), c_(C == 1 ? 1 : n) { if (kDyn) v_.assign((size_t)n, S(0)); else v_.assign((size_t)R * C, S(0)); }
    Matrix(S a, S b) { static_assert(!kDyn && R * C == 2, "2-vector ctor"); v_ = {a, b}; }
    Matrix(S a, S b, S c) { static_assert(!kDyn && R * C == 3, "3-vector ctor"); v_ = {a, b, c}; }
    Matrix(S a, S b, S c, S d) { static_assert(!kDyn && R * C == 4, "4-vector ctor"); v_ = {a, b, c, d}; }
    int rows() const { return r_; }
    int cols() const { return c_; }
    int size() const { return r_ * c_; }
    void resize(int r, int c) { r_ = r; c_ = c; v_.assign((size_t)r * c, S(0)); }
    void resize(int n) { if (C == 1) resize(n, 1); else resize(1, n); }
    S* data() { return v_.data(); }
    const S* data() const { return v_.data(); }
    S& operator()(int i, int j) { return v_[(size_t)j * r_ + i]; }
    const S& operator()(int i, int j) const { return v_[(size_t)j * r_ + i]; }
    S& operator()(int i) { return v_[i]; }
    const S& operator()(int i) const { return v_[i]; }
    S& operator[](int i) { return v_[i]; }
    const S& operator[](int i) const { return v_[i]; }
    void setZero() { for (auto& x : v_) x = S(0); }
    void setIdentity() { setZero(); for (int i = 0; i < r_ && i < c_; ++i) (*this)(i, i) = S(1); }
    void fill(S s) { for (auto& x : v_) x = s; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Zero(int r, int c) { return Matrix(r, c); }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Identity(int r, int c) { Matrix m(r, c); m.setIdentity(); return m; }
    Matrix operator*(S s) const { Matrix m(*this); for (auto& x : m.v_) x *= s; return m; }
    Matrix operator/(S s) const { Matrix m(*this); for (auto& x : m.v_) x /= s; return m; }
    Matrix operator+(const Matrix& o) const { Matrix m(*this); for (size_t i = 0; i < v_.size(); ++i) m.v_[i] += o.v_[i]; return m; }
    Matrix operator-(const Matrix& o) const { Matrix m(*this); for (size_t i = 0; i < v_.size(); ++i) m.v_[i] -= o.v_[i]; return m; }
    Matrix<S, C, R> transpose() const { Matrix<S, C, R> t(c_, r_); for (int i = 0; i < r_; ++i) for (int j = 0; j < c_; ++j) t(j, i) = (*this)(i, j); return t; }
    S norm() const { S s = 0; for (auto x : v_) s += x * x; return std::sqrt(s); }
};
template <typename S, int R, int K, int C>
Matrix<S, R, C> operator*(const Matrix<S, R, K>& a, const Matrix<S, K, C>& b) {
    Matrix<S, R, C> m(a.rows(), b.cols());
    for (int i = 0; i < a.rows(); ++i) for (int j = 0; j < b.cols(); ++j) { S s = 0; for (int k = 0; k < a.cols(); ++k) s += a(i, k) * b(k, j); m(i, j) = s; }
    return m;
}

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, Dynamic, 1> VectorXd;

class Quaterniond {
    double c_[4] = {0, 0, 0, 1};   // x y z w
public:
    Quaterniond() {}
    Quaterniond(double w, double x, double y, double z) { c_[0] = x; c_[1] = y; c_[2] = z; c_[3] = w; }
    double x() const { return c_[0]; } double y() const { return c_[1]; } double z() const { return c_[2]; } double w() const { return c_[3]; }
    double& x() { return c_[0]; } double& y() { return c_[1]; } double& z() { return c_[2]; } double& w() { return c_[3]; }
    const double* coeffs() const { return c_; }
};

// rigid transform (Eigen::Isometry3d): rotation matrix + translation, the estimate type of g2o's VertexSE3
class Isometry3d {
    Matrix3d R_;
    Vector3d t_;
public:
    Isometry3d() { R_.setIdentity(); }
    static Isometry3d Identity() { return Isometry3d(); }
    Matrix3d& linear() { return R_; }
    const Matrix3d& linear() const { return R_; }
    Matrix3d rotation() const { return R_; }
    Vector3d& translation() { return t_; }
    const Vector3d& translation() const { return t_; }
    Isometry3d inverse() const {
        Isometry3d r;
        r.R_ = R_.transpose();
        const Vector3d rt = r.R_ * t_;
        for (int i = 0; i < 3; ++i) r.t_[i] = -rt[i];
        return r;
    }
    Isometry3d operator*(const Isometry3d& o) const { Isometry3d r; r.R_ = R_ * o.R_; r.t_ = R_ * o.t_ + t_; return r; }
    Vector3d operator*(const Vector3d& v) const { return R_ * v + t_; }
    Matrix4d matrix() const { Matrix4d m; m.setIdentity(); for (int i = 0; i < 3; ++i) { m(i, 3) = t_[i]; for (int j = 0; j < 3; ++j) m(i, j) = R_(i, j); } return m; }
};

}  // namespace Eigen

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

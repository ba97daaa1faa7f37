// mini_eigen.hpp — FUNCTIONAL STAND-IN for the part of Eigen 3.3.7 that the reference's hot path uses.
// TEST INFRASTRUCTURE ONLY (oracle/_ref): the reference's own map_eval.cpp / voxel_calculator.cpp are compiled, unmodified,
// from /root/reference against these headers, because Eigen itself is absent from this image.  Everything in this file is
// the BUILDER's arithmetic, not Eigen's; where a result depends on the operation order the order of Eigen 3.3.7 (x86-64,
// SSE2, the reference's README pins 3.3.7) is followed and said so at the function:
//   * squaredNorm / dot of a fixed 3-vector: (x*x + y*y) + z*z               (linear-vectorised redux, packet of 2 + tail)
//   * Matrix3d::determinant: cofactor expansion along row 0                   (bruteforce_det3_helper)
//   * SelfAdjointEigenSolver<Matrix3d>::compute: scale, closed-form 3x3 Householder tridiagonalisation, implicit symmetric
//     QR steps with Wilkinson shift, ascending sort                            (the iterative solver, not computeDirect)
//   * LLT: unblocked lower Cholesky, early return at a non-positive pivot     (llt_inplace<Lower>::unblocked)
//   * dynamic-size reductions (rowwise().mean()): sequential, first to last
// Dense products use the plain triple loop (Eigen's GEBP kernel sums in a different order: rounding-level difference).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <iomanip>
#include <limits>
#include <ostream>
#include <sstream>
#include <type_traits>
#include <utility>
#include <vector>

namespace Eigen {

constexpr int Dynamic = -1;
typedef std::ptrdiff_t Index;

namespace internal {
template <typename T, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)>
struct DenseStorage;
template <typename T, int R, int C>
struct DenseStorage<T, R, C, false> {
    T d[R * C];
    DenseStorage() {
        for (int i = 0; i < R * C; ++i) d[i] = T(0);  // Eigen leaves this uninitialised; zero is a defined instance of it
    }
    int rows() const { return R; }
    int cols() const { return C; }
    void resize(int r, int c) {
        assert(r == R && c == C);
        (void) r;
        (void) c;
    }
    void conservative_resize(int r, int c) { resize(r, c); }
    T *data() { return d; }
    const T *data() const { return d; }
};
template <typename T, int R, int C>
struct DenseStorage<T, R, C, true> {
    std::vector<T> d;
    int r_ = (R == Dynamic ? 0 : R), c_ = (C == Dynamic ? 0 : C);
    int rows() const { return r_; }
    int cols() const { return c_; }
    void resize(int r, int c) {
        r_ = r;
        c_ = c;
        d.assign((size_t) r * (size_t) c, T(0));
    }
    void conservative_resize(int r, int c) {  // keeps the top-left block (column-major)
        std::vector<T> n((size_t) r * (size_t) c, T(0));
        for (int j = 0; j < std::min(c, c_); ++j)
            for (int i = 0; i < std::min(r, r_); ++i) n[(size_t) j * r + i] = d[(size_t) j * r_ + i];
        d.swap(n);
        r_ = r;
        c_ = c;
    }
    T *data() { return d.data(); }
    const T *data() const { return d.data(); }
};
}  // namespace internal

template <typename T, int R, int C>
class Matrix;
template <typename M>
class LLT;

// L of an LLT (TriangularView<const MatrixType, Lower>) and its transpose: products skip the structural zeros.
template <typename T, int N, bool Upper>
struct TriangularMatrix {
    Matrix<T, N, N> m;  // full storage; only the triangle is meaningful
    T operator()(int i, int j) const { return (Upper ? (i <= j) : (i >= j)) ? m(i, j) : T(0); }
    TriangularMatrix<T, N, !Upper> transpose() const {
        TriangularMatrix<T, N, !Upper> t;
        t.m = m.transpose();
        return t;
    }
    operator Matrix<T, N, N>() const {
        Matrix<T, N, N> o;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) o(i, j) = (*this)(i, j);
        return o;
    }
};

template <typename T, int N>
struct DiagonalWrapper {
    Matrix<T, N, 1> d;
};

template <typename T, int R, int C>
class Matrix {
    internal::DenseStorage<T, R, C> s_;

  public:
    typedef T Scalar;
    enum { RowsAtCompileTime = R, ColsAtCompileTime = C, IsDynamic = (R == Dynamic || C == Dynamic) };

    Matrix() {}
    // MatrixXd(rows, cols)  |  Vector2(x, y)
    template <typename A, typename B, typename = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
    Matrix(const A &a, const B &b) {
        if constexpr (IsDynamic) {
            s_.resize((int) a, (int) b);
        } else {
            static_assert(R * C == 2 || IsDynamic, "two-coefficient constructor on a non-2-vector");
            s_.d[0] = (T) a;
            s_.d[1] = (T) b;
        }
    }
    Matrix(T x, T y, T z) {
        static_assert(R * C == 3, "three-coefficient constructor on a non-3-vector");
        s_.d[0] = x;
        s_.d[1] = y;
        s_.d[2] = z;
    }
    Matrix(T x, T y, T z, T w) {
        static_assert(R * C == 4, "four-coefficient constructor on a non-4-vector");
        s_.d[0] = x;
        s_.d[1] = y;
        s_.d[2] = z;
        s_.d[3] = w;
    }
    template <int R2, int C2, typename = std::enable_if_t<(R2 != R || C2 != C)>>
    Matrix(const Matrix<T, R2, C2> &o) {
        s_.resize(o.rows(), o.cols());
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) (*this)(i, j) = o(i, j);
    }
    int rows() const { return s_.rows(); }
    int cols() const { return s_.cols(); }
    Index size() const { return (Index) rows() * cols(); }
    T *data() { return s_.data(); }
    const T *data() const { return s_.data(); }
    void resize(Index r, Index c) { s_.resize((int) r, (int) c); }
    void conservativeResize(Index r, Index c) { s_.conservative_resize((int) r, (int) c); }
    Matrix &setZero() {
        for (Index i = 0; i < size(); ++i) data()[i] = T(0);
        return *this;
    }
    Matrix &setIdentity() {
        setZero();
        for (int i = 0; i < std::min(rows(), cols()); ++i) (*this)(i, i) = T(1);
        return *this;
    }

    // column-major, as Eigen's default
    T &operator()(Index i, Index j) { return data()[(size_t) j * rows() + i]; }
    const T &operator()(Index i, Index j) const { return data()[(size_t) j * rows() + i]; }
    T &operator()(Index i) { return data()[i]; }
    const T &operator()(Index i) const { return data()[i]; }
    T &operator[](Index i) { return data()[i]; }
    const T &operator[](Index i) const { return data()[i]; }
    T &x() { return data()[0]; }
    T &y() { return data()[1]; }
    T &z() { return data()[2]; }
    const T &x() const { return data()[0]; }
    const T &y() const { return data()[1]; }
    const T &z() const { return data()[2]; }
    T coeff(Index i, Index j) const { return (*this)(i, j); }

    static Matrix Zero() { return Matrix(); }
    static Matrix Ones() {
        Matrix m;
        for (Index i = 0; i < m.size(); ++i) m.data()[i] = T(1);
        return m;
    }
    static Matrix Identity() {
        Matrix m;
        m.setIdentity();
        return m;
    }

    const Matrix &matrix() const { return *this; }

    template <typename U>
    Matrix<U, R, C> cast() const {
        Matrix<U, R, C> o;
        if constexpr (IsDynamic) o.resize(rows(), cols());
        for (Index i = 0; i < size(); ++i) o.data()[i] = (U) data()[i];
        return o;
    }

    Matrix<T, C, R> transpose() const {
        Matrix<T, C, R> o;
        if constexpr (IsDynamic) o.resize(cols(), rows());
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) o(j, i) = (*this)(i, j);
        return o;
    }

    // ---- coefficient-wise arithmetic -------------------------------------------------------------------------
    Matrix operator+(const Matrix &b) const {
        Matrix o(*this);
        for (Index i = 0; i < size(); ++i) o.data()[i] = data()[i] + b.data()[i];
        return o;
    }
    Matrix operator-(const Matrix &b) const {
        Matrix o(*this);
        for (Index i = 0; i < size(); ++i) o.data()[i] = data()[i] - b.data()[i];
        return o;
    }
    Matrix operator-() const {
        Matrix o(*this);
        for (Index i = 0; i < size(); ++i) o.data()[i] = -data()[i];
        return o;
    }
    Matrix operator*(Scalar k) const {
        Matrix o(*this);
        for (Index i = 0; i < size(); ++i) o.data()[i] = data()[i] * k;
        return o;
    }
    friend Matrix operator*(Scalar k, const Matrix &m) {
        Matrix o(m);
        for (Index i = 0; i < m.size(); ++i) o.data()[i] = k * m.data()[i];
        return o;
    }
    Matrix operator/(Scalar k) const {
        Matrix o(*this);
        for (Index i = 0; i < size(); ++i) o.data()[i] = data()[i] / k;
        return o;
    }
    Matrix &operator+=(const Matrix &b) {
        for (Index i = 0; i < size(); ++i) data()[i] += b.data()[i];
        return *this;
    }
    Matrix &operator-=(const Matrix &b) {
        for (Index i = 0; i < size(); ++i) data()[i] -= b.data()[i];
        return *this;
    }
    Matrix &operator*=(Scalar k) {
        for (Index i = 0; i < size(); ++i) data()[i] *= k;
        return *this;
    }
    Matrix &operator/=(Scalar k) {
        for (Index i = 0; i < size(); ++i) data()[i] /= k;
        return *this;
    }
    bool operator==(const Matrix &b) const {
        if (rows() != b.rows() || cols() != b.cols()) return false;
        for (Index i = 0; i < size(); ++i)
            if (!(data()[i] == b.data()[i])) return false;
        return true;
    }
    bool operator!=(const Matrix &b) const { return !(*this == b); }

    // ---- reductions --------------------------------------------------------------------------------------------
    // fixed 3-vector: (x*x + y*y) + z*z — Eigen 3.3.7's linear-vectorised redux on SSE2 (one packet of two, then the tail);
    // any other size: sequential.
    Scalar dot(const Matrix &b) const {
        Scalar r = data()[0] * b.data()[0];
        for (Index i = 1; i < size(); ++i) r = r + data()[i] * b.data()[i];
        return r;
    }
    Scalar squaredNorm() const { return dot(*this); }
    Scalar norm() const { return std::sqrt(squaredNorm()); }
    void normalize() {
        const Scalar n2 = squaredNorm();
        if (n2 > Scalar(0)) *this /= std::sqrt(n2);
    }
    Scalar sum() const {
        Scalar r = size() ? data()[0] : Scalar(0);
        for (Index i = 1; i < size(); ++i) r = r + data()[i];
        return r;
    }
    Scalar trace() const {
        Scalar r = (*this)(0, 0);
        for (int i = 1; i < std::min(rows(), cols()); ++i) r = r + (*this)(i, i);
        return r;
    }
    Scalar maxCoeff() const {
        Scalar r = data()[0];
        for (Index i = 1; i < size(); ++i) r = std::max(r, data()[i]);
        return r;
    }
    Matrix cwiseAbs() const {
        Matrix o(*this);
        for (Index i = 0; i < size(); ++i) o.data()[i] = std::abs(data()[i]);
        return o;
    }
    Matrix cwiseMax(Scalar k) const {
        Matrix o(*this);
        for (Index i = 0; i < size(); ++i) o.data()[i] = std::max(data()[i], k);  // (std::max(a, b): a < b ? b : a)
        return o;
    }
    DiagonalWrapper<T, R> asDiagonal() const {
        static_assert(C == 1, "asDiagonal on a non-vector");
        return DiagonalWrapper<T, R>{*this};
    }

    // Matrix3d::determinant — Eigen's determinant_impl<Derived, 3>: cofactor expansion along row 0 with
    //   helper(a,b,c) = m(0,a) * (m(1,b)*m(2,c) - m(1,c)*m(2,b)),  det = helper(0,1,2) - helper(1,0,2) + helper(2,0,1).
    Scalar determinant() const {
        static_assert(R == 3 && C == 3, "determinant: only 3x3 is provided");
        const Matrix &m = *this;
        auto h = [&](int a, int b, int c) { return m(0, a) * (m(1, b) * m(2, c) - m(1, c) * m(2, b)); };
        return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
    }

    LLT<Matrix> llt() const;

    // ---- blocks (only what map_eval.cpp touches) -------------------------------------------------------------
    struct ColProxy {
        Matrix &m;
        int j;
        template <int R2>
        ColProxy &operator=(const Matrix<T, R2, 1> &v) {
            for (int i = 0; i < m.rows(); ++i) m(i, j) = v(i);
            return *this;
        }
    };
    ColProxy col(Index j) { return ColProxy{*this, (int) j}; }

    template <int BR, int BC>
    struct BlockProxy {
        Matrix &m;
        int i0, j0;
        template <int R2, int C2>
        BlockProxy &operator=(const Matrix<T, R2, C2> &v) {  // (sizes must agree; Eigen asserts)
            assert(v.rows() == BR && v.cols() == BC);
            for (int j = 0; j < BC && j < v.cols(); ++j)
                for (int i = 0; i < BR && i < v.rows(); ++i) m(i0 + i, j0 + j) = v(i, j);
            return *this;
        }
    };
    template <int BR, int BC>
    BlockProxy<BR, BC> block(Index i, Index j) {
        return BlockProxy<BR, BC>{*this, (int) i, (int) j};
    }

    // rowwise().mean(): Eigen = rowwise().sum() / Scalar(cols()); each row's sum is a sequential (non-vectorised, strided) redux.
    struct RowwiseOp {
        const Matrix &m;
        Matrix<T, R, 1> mean() const {
            Matrix<T, R, 1> o;
            if constexpr (R == Dynamic) o.resize(m.rows(), 1);
            for (int i = 0; i < m.rows(); ++i) {
                Scalar sacc = m(i, 0);
                for (int j = 1; j < m.cols(); ++j) sacc = sacc + m(i, j);
                o(i) = sacc / Scalar(m.cols());
            }
            return o;
        }
    };
    RowwiseOp rowwise() const { return RowwiseOp{*this}; }
    struct ColwiseOp {
        const Matrix &m;
        template <int R2>
        Matrix operator-(const Matrix<T, R2, 1> &v) const {
            Matrix o(m);
            for (int j = 0; j < m.cols(); ++j)
                for (int i = 0; i < m.rows(); ++i) o(i, j) = m(i, j) - v(i);
            return o;
        }
    };
    ColwiseOp colwise() const { return ColwiseOp{*this}; }
};

// ---- products ----------------------------------------------------------------------------------------------------
template <typename T, int R, int K, int C>
Matrix<T, R, C> operator*(const Matrix<T, R, K> &a, const Matrix<T, K, C> &b) {
    Matrix<T, R, C> o;
    if constexpr (R == Dynamic || C == Dynamic) o.resize(a.rows(), b.cols());
    const int kk = a.cols();
    for (int j = 0; j < o.cols(); ++j)
        for (int i = 0; i < o.rows(); ++i) {
            T acc = a(i, 0) * b(0, j);
            for (int k = 1; k < kk; ++k) acc = acc + a(i, k) * b(k, j);
            o(i, j) = acc;
        }
    return o;
}
// a dynamic product assigned to a fixed matrix (Matrix3d cov = (centered * centered.transpose()) / k): the converting
// constructor above takes care of it.

template <typename T, int R, int N>
Matrix<T, R, N> operator*(const Matrix<T, R, N> &a, const DiagonalWrapper<T, N> &d) {  // scales the columns
    Matrix<T, R, N> o(a);
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < a.rows(); ++i) o(i, j) = a(i, j) * d.d(j);
    return o;
}

template <typename T, int N, bool Upper>
Matrix<T, N, N> operator*(const TriangularMatrix<T, N, Upper> &t, const Matrix<T, N, N> &b) {
    Matrix<T, N, N> o;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            T acc = T(0);
            bool first = true;
            for (int k = (Upper ? i : 0); k <= (Upper ? N - 1 : i); ++k) {
                const T p = t.m(i, k) * b(k, j);
                acc = first ? p : acc + p;
                first = false;
            }
            o(i, j) = acc;
        }
    return o;
}
template <typename T, int N, bool Upper>
Matrix<T, N, N> operator*(const Matrix<T, N, N> &a, const TriangularMatrix<T, N, Upper> &t) {
    Matrix<T, N, N> o;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            T acc = T(0);
            bool first = true;
            for (int k = (Upper ? 0 : j); k <= (Upper ? j : N - 1); ++k) {
                const T p = a(i, k) * t.m(k, j);
                acc = first ? p : acc + p;
                first = false;
            }
            o(i, j) = acc;
        }
    return o;
}

// ---- LLT (Eigen: llt_inplace<Scalar, Lower>::unblocked; returns at the first non-positive pivot, leaving the rest) ----
template <typename M>
class LLT {
    M m_;
    bool ok_ = true;

  public:
    typedef typename M::Scalar T;
    explicit LLT(const M &a) : m_(a) {
        const int n = m_.rows();
        for (int k = 0; k < n; ++k) {
            T x = m_(k, k);
            if (k > 0) {
                T s2 = m_(k, 0) * m_(k, 0);
                for (int j = 1; j < k; ++j) s2 = s2 + m_(k, j) * m_(k, j);
                x -= s2;
            }
            if (x <= T(0)) {  // Eigen returns here and leaves the rest of the matrix as it is (a NaN pivot passes on)
                ok_ = false;
                return;
            }
            m_(k, k) = x = std::sqrt(x);
            for (int i = k + 1; i < n; ++i) {
                if (k > 0) {
                    T acc = m_(i, 0) * m_(k, 0);
                    for (int j = 1; j < k; ++j) acc = acc + m_(i, j) * m_(k, j);
                    m_(i, k) -= acc;
                }
                m_(i, k) /= x;
            }
        }
    }
    bool ok() const { return ok_; }
    TriangularMatrix<T, M::RowsAtCompileTime, false> matrixL() const {
        TriangularMatrix<T, M::RowsAtCompileTime, false> t;
        t.m = m_;
        return t;
    }
};
template <typename T, int R, int C>
LLT<Matrix<T, R, C>> Matrix<T, R, C>::llt() const {
    return LLT<Matrix<T, R, C>>(*this);
}

// ---- SelfAdjointEigenSolver<Matrix3d> (iterative path of Eigen 3.3.7; reads the lower triangle) ----------------------
template <typename M>
class SelfAdjointEigenSolver {
    typedef typename M::Scalar T;
    enum { N = M::RowsAtCompileTime };
    Matrix<T, N, 1> eval_;
    M evec_;

    static void make_givens(T p, T q, T &c, T &s) {
        if (q == T(0)) {
            c = p < T(0) ? T(-1) : T(1);
            s = T(0);
        } else if (p == T(0)) {
            c = T(0);
            s = q < T(0) ? T(1) : T(-1);
        } else if (std::abs(p) > std::abs(q)) {
            T t = q / p, u = std::sqrt(T(1) + t * t);
            if (p < T(0)) u = -u;
            c = T(1) / u;
            s = -t * c;
        } else {
            T t = p / q, u = std::sqrt(T(1) + t * t);
            if (q < T(0)) u = -u;
            s = -T(1) / u;
            c = -t * s;
        }
    }
    static void qr_step(T *diag, T *sub, int start, int end, M &q) {
        T td = (diag[end - 1] - diag[end]) * T(0.5);
        T e = sub[end - 1];
        T mu = diag[end];
        if (td == T(0)) {
            mu -= std::abs(e);
        } else if (e != T(0)) {
            const T e2 = e * e;
            const T h = std::hypot(td, e);
            if (e2 == T(0))
                mu -= e / ((td + (td > T(0) ? h : -h)) / e);
            else
                mu -= e2 / (td + (td > T(0) ? h : -h));
        }
        T x = diag[start] - mu;
        T z = sub[start];
        for (int k = start; k < end && z != T(0); ++k) {
            T c, s;
            make_givens(x, z, c, s);
            const T sdk = s * diag[k] + c * sub[k];
            const T dkp1 = s * sub[k] + c * diag[k + 1];
            diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
            diag[k + 1] = s * sdk + c * dkp1;
            sub[k] = c * sdk - s * dkp1;
            if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
            x = sub[k];
            if (k < end - 1) {
                z = -s * sub[k + 1];
                sub[k + 1] = c * sub[k + 1];
            }
            for (int i = 0; i < N; ++i) {  // Q = Q * G
                const T xi = q(i, k), yi = q(i, k + 1);
                q(i, k) = c * xi - s * yi;
                q(i, k + 1) = s * xi + c * yi;
            }
        }
    }

  public:
    explicit SelfAdjointEigenSolver(const M &a) {
        static_assert(N == 3, "SelfAdjointEigenSolver: only 3x3 is provided");
        M mat;
        for (int j = 0; j < 3; ++j)
            for (int i = j; i < 3; ++i) mat(i, j) = a(i, j);  // lower triangle
        T scale = T(0);
        for (int j = 0; j < 3; ++j)
            for (int i = j; i < 3; ++i) scale = std::max(scale, std::abs(mat(i, j)));
        if (scale == T(0)) scale = T(1);
        for (int j = 0; j < 3; ++j)
            for (int i = j; i < 3; ++i) mat(i, j) /= scale;
        T diag[3], sub[2];
        // closed-form 3x3 tridiagonalisation
        diag[0] = mat(0, 0);
        const T v1norm2 = mat(2, 0) * mat(2, 0);
        if (v1norm2 <= std::numeric_limits<T>::min()) {
            diag[1] = mat(1, 1);
            diag[2] = mat(2, 2);
            sub[0] = mat(1, 0);
            sub[1] = mat(2, 1);
            evec_.setIdentity();
        } else {
            const T beta = std::sqrt(mat(1, 0) * mat(1, 0) + v1norm2);
            const T inv = T(1) / beta;
            const T m01 = mat(1, 0) * inv, m02 = mat(2, 0) * inv;
            const T q = T(2) * m01 * mat(2, 1) + m02 * (mat(2, 2) - mat(1, 1));
            diag[1] = mat(1, 1) + m02 * q;
            diag[2] = mat(2, 2) - m02 * q;
            sub[0] = beta;
            sub[1] = mat(2, 1) - m01 * q;
            evec_.setZero();
            evec_(0, 0) = T(1);
            evec_(1, 1) = m01;
            evec_(1, 2) = m02;
            evec_(2, 1) = m02;
            evec_(2, 2) = -m01;
        }
        int end = 2, start = 0, iter = 0;
        const T tiny = std::numeric_limits<T>::min();
        const T prec = T(2) * std::numeric_limits<T>::epsilon();
        while (end > 0) {
            for (int i = start; i < end; ++i)
                if (std::abs(sub[i]) <= (std::abs(diag[i]) + std::abs(diag[i + 1])) * prec || std::abs(sub[i]) <= tiny)
                    sub[i] = T(0);
            while (end > 0 && sub[end - 1] == T(0)) end--;
            if (end <= 0) break;
            if (++iter > 30 * 3) break;
            start = end - 1;
            while (start > 0 && sub[start - 1] != T(0)) start--;
            qr_step(diag, sub, start, end, evec_);
        }
        for (int i = 0; i < 2; ++i) {  // ascending selection sort, columns follow
            int k = i;
            for (int j = i + 1; j < 3; ++j)
                if (diag[j] < diag[k]) k = j;
            if (k != i) {
                std::swap(diag[i], diag[k]);
                for (int r = 0; r < 3; ++r) std::swap(evec_(r, i), evec_(r, k));
            }
        }
        for (int i = 0; i < 3; ++i) eval_(i) = diag[i] * scale;
    }
    const Matrix<T, N, 1> &eigenvalues() const { return eval_; }
    const M &eigenvectors() const { return evec_; }
};

// ---- printing: Eigen's default IOFormat (stream precision, " " between coefficients, "\n" between rows, every
// coefficient padded to the widest one) ---------------------------------------------------------------------------
template <typename T, int R, int C>
std::ostream &operator<<(std::ostream &os, const Matrix<T, R, C> &m) {
    if (m.size() == 0) return os;
    std::streamsize width = 0;
    for (int j = 0; j < m.cols(); ++j)
        for (int i = 0; i < m.rows(); ++i) {
            std::stringstream ss;
            ss.copyfmt(os);
            ss << m(i, j);
            width = std::max<std::streamsize>(width, (std::streamsize) ss.str().length());
        }
    for (int i = 0; i < m.rows(); ++i) {
        if (i) os << "\n";
        for (int j = 0; j < m.cols(); ++j) {
            if (j) os << " ";
            if (width) os.width(width);
            os << m(i, j);
        }
    }
    return os;
}

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<int, 2, 1> Vector2i;
typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, Dynamic, 1> VectorXd;

// Only named in a declaration of the reference (voxel_calculator.hpp: updateVoxelMap(cloud, pose)), never used.
struct Isometry3d {
    Matrix4d m = Matrix4d::Identity();
};

}  // namespace Eigen

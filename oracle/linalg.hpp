// ORACLE — TEST INFRASTRUCTURE ONLY (see wbm.hpp header).
// Small dense column-major helpers + a restatement of Eigen 3.3's pivoted LDLT
// (Eigen/src/Cholesky/LDLT.h: ldlt_inplace<Lower>::unblocked, LDLT::_solve_impl), which the reference
// uses as `Chol = Eigen::LDLT<DMat<T>>` (HSDDPSolver/common/HSDDP_CPPTypes.h:64) in
// SinglePhase::backward_sweep (SinglePhase.cpp:366-375).  Eigen is a third-party dependency that is
// not under /root/reference; the algorithm is restated from its published source.
#pragma once
#include <cmath>
#include <vector>
#include <algorithm>
#include <limits>

namespace orc {

// C(r x c) = alpha * op(A) * op(B) + beta * C ; all column-major with explicit dims
inline void gemm(bool ta, bool tb, int r, int c, int k, const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                 double alpha = 1.0, double beta = 0.0) {
    for (int j = 0; j < c; j++)
        for (int i = 0; i < r; i++) {
            double s = 0;
            for (int t = 0; t < k; t++) {
                double a = ta ? A[t + lda * i] : A[i + lda * t];
                double b = tb ? B[j + ldb * t] : B[t + ldb * j];
                s += a * b;
            }
            C[i + ldc * j] = alpha * s + (beta == 0.0 ? 0.0 : beta * C[i + ldc * j]);
        }
}
inline void gemv(bool ta, int r, int c, const double* A, int lda, const double* x, double* y, double alpha = 1.0, double beta = 0.0) {
    // y = alpha*op(A)*x + beta*y ; A is r x c (before op)
    int ny = ta ? c : r, nx = ta ? r : c;
    for (int i = 0; i < ny; i++) {
        double s = 0;
        for (int t = 0; t < nx; t++) s += (ta ? A[t + lda * i] : A[i + lda * t]) * x[t];
        y[i] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[i]);
    }
}
inline double dotv(int n, const double* a, const double* b) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }

// Eigen-3.3-style LDLT with diagonal pivoting on the lower triangle, in place on a column-major n x n.
struct LDLT {
    int n = 0;
    std::vector<double> m;     // L (unit lower) and D on the diagonal
    std::vector<int> tr;       // transpositions
    int sign = 0;              // 0 ZeroSign, 1 PositiveSemiDef, -1 NegativeSemiDef, 2 Indefinite
    bool ok = true;
    void compute(const double* A, int n_) {
        n = n_; m.assign(A, A + n * n); tr.assign(n, 0); sign = 0; ok = true;
        auto M = [&](int i, int j) -> double& { return m[i + n * j]; };
        bool found_zero_pivot = false;
        std::vector<double> temp(n);
        for (int k = 0; k < n; k++) {
            int big = k; double best = std::fabs(M(k, k));
            for (int i = k + 1; i < n; i++) if (std::fabs(M(i, i)) > best) { best = std::fabs(M(i, i)); big = i; }
            tr[k] = big;
            if (k != big) {
                int s = n - big - 1;
                for (int j = 0; j < k; j++) std::swap(M(k, j), M(big, j));          // row(k).head(k) <-> row(big).head(k)
                for (int i = 0; i < s; i++) std::swap(M(big + 1 + i, k), M(big + 1 + i, big));  // col(k).tail(s) <-> col(big).tail(s)
                std::swap(M(k, k), M(big, big));
                for (int i = k + 1; i < big; i++) std::swap(M(i, k), M(big, i));
            }
            int rs = n - k - 1;
            if (k > 0) {
                for (int j = 0; j < k; j++) temp[j] = M(j, j) * M(k, j);
                double s = 0; for (int j = 0; j < k; j++) s += M(k, j) * temp[j];
                M(k, k) -= s;
                for (int i = 0; i < rs; i++) { double t = 0; for (int j = 0; j < k; j++) t += M(k + 1 + i, j) * temp[j]; M(k + 1 + i, k) -= t; }
            }
            double akk = M(k, k);
            bool valid = std::fabs(akk) > 0.0;
            if (k == 0 && !valid) { sign = 0; for (int j = 0; j < n; j++) tr[j] = j; return; }
            if (rs > 0 && valid) for (int i = 0; i < rs; i++) M(k + 1 + i, k) /= akk;
            else if (rs > 0) { for (int i = 0; i < rs; i++) if (M(k + 1 + i, k) != 0.0) ok = false; }
            if (found_zero_pivot && valid) ok = false; else if (!valid) found_zero_pivot = true;
            if (sign == 1) { if (akk < 0) sign = 2; }
            else if (sign == -1) { if (akk > 0) sign = 2; }
            else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
        }
    }
    bool isPositive() const { return sign == 1 || sign == 0; }
    // X (n x nrhs, column-major) <- A^-1 B
    void solve(const double* B, int nrhs, double* X) const {
        const double tol = 1.0 / std::numeric_limits<double>::max();
        for (int c = 0; c < nrhs; c++) {
            double* x = X + n * c; const double* b = B + n * c;
            for (int i = 0; i < n; i++) x[i] = b[i];
            for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
            for (int i = 0; i < n; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= m[i + n * j] * x[j]; x[i] = s; }
            for (int i = 0; i < n; i++) { double d = m[i + n * i]; x[i] = (std::fabs(d) > tol) ? x[i] / d : 0.0; }
            for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < n; j++) s -= m[j + n * i] * x[j]; x[i] = s; }
            for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
        }
    }
};

}  // namespace orc

// oracle/_ref: the reference's OWN numerical libraries, compiled from where they lie under
// /root/reference/third_party/eigen (vendored Eigen 3.2.x: its BLAS sources blas/double.cpp,
// blas/xerbla.cpp and its header-only decompositions) - TEST INFRASTRUCTURE, like everything else
// in oracle/.  No reference source is copied and no stand-in for a missing header or library is
// written: the solver core itself (src/epsilon/**) needs glog and protoc-generated headers the
// image lacks and stays unbuildable; what CAN be built is the arithmetic the
// core delegates to, called here exactly the way the core calls it:
//
//   ref_dgemv        y = op(A) x through dgemv_   as linear/dense_matrix_impl.cc:55-67
//   ref_dgemm        C = op(A) op(B) through dgemm_ as linear/linear_map_multiply.cc:14-37
//   ref_ldlt_inverse Eigen::LDLT + solve(Identity)  as linear/dense_matrix_impl.cc:21-30
//   ref_gram_svd     SelfAdjointEigenSolver(Y^T Y + 1e-15 I), sqrt, U = Y V diag(1/d) [1/0 -> 0]
//                                                     as prox/ortho_invariant.cc:36-50
//   ref_llt_solve    Eigen::LLT solve, the check of vector/block_cholesky_test.cc:95-103
//   ref_lasso_sweeps the unrolled lasso sweep (oracle/lasso_sweep.c) with its three mat-vecs through
//                    dgemv_: what bench.py times as the reference's CPU path
//
// (The reference links the system's -lblas, Makefile:46; Eigen's own BLAS is the one implementation
// of that interface its tree carries.)  tests/test_oracle_ref.py pins the numpy oracle's dense
// kernels against these; bench.py may use ref_dgemv as the CPU baseline's mat-vec.
#include <Eigen/Dense>

#include <cmath>

extern "C" {
// declarations as in the reference's linear/lapack.h
void dgemv_(char* transa, int* m, int* n, double* alpha, double* A, int* lda, double* x, int* incx,
            double* beta, double* y, int* incy);
void dgemm_(char* transa, char* transb, int* m, int* n, int* k, double* alpha, double* A, int* lda,
            double* B, int* ldb, double* beta, double* C, int* ldc);

void ref_dgemv(char trans, int m, int n, const double* A, const double* x, double* y) {
  double alpha = 1, beta = 0;
  int incx = 1, incy = 1;
  dgemv_(&trans, &m, &n, &alpha, const_cast<double*>(A), &m, const_cast<double*>(x), &incx, &beta, y,
         &incy);
}

// A is m x k or (transa == 'T') k x m, B is k x n or n x k; C is m x n, all column-major
void ref_dgemm(char transa, char transb, int m, int n, int k, const double* A, const double* B,
               double* C) {
  int lda = transa == 'N' ? m : k;
  int ldb = transb == 'N' ? k : n;
  double alpha = 1, beta = 0;
  dgemm_(&transa, &transb, &m, &n, &k, &alpha, const_cast<double*>(A), &lda, const_cast<double*>(B),
         &ldb, &beta, C, &m);
}

// k sweeps of the compiled lasso (no stopping test) with the three mat-vecs of a sweep going
// through the reference tree's dgemv_ the way the reference's solver reaches them: forward
// substitution t = A v ('N'), the cached explicit inverse w = Minv (b - t) (a dense map applied
// with 'N', dense_matrix_impl.cc:55-67), back substitution g = A^T w ('T')
// (vector/block_cholesky.cc:86-117, 135-137); the elementwise steps between them are those of
// oracle/lasso_sweep.c (prox_admm.cc:135-147, scaled_zone.cc:90-101).  bench.py's cpu_baseline
// times THIS: one thread, the BLAS of the reference's own tree.  Scratch: t, w (m), g (n).
void ref_lasso_sweeps(int m, int n, const double* A, const double* Minv, const double* b, double lam,
                      double* x0, double* x1, double* u, double* y0, double* y1, int k, double* t,
                      double* w, double* g) {
  for (int it = 0; it < k; ++it) {
    for (int j = 0; j < n; ++j) u[j] = (u[j] - y0[j]) - y1[j];
    for (int j = 0; j < n; ++j) u[j] += y0[j];
    ref_dgemv('N', m, n, A, u, t);
    for (int i = 0; i < m; ++i) t[i] = b[i] - t[i];
    ref_dgemv('N', m, m, Minv, t, w);
    ref_dgemv('T', m, n, A, w, g);
    for (int j = 0; j < n; ++j) {
      x0[j] = u[j] + 2.0 * g[j];
      y0[j] = x0[j];
      u[j] -= y0[j];
    }
    for (int j = 0; j < n; ++j) {
      u[j] += y1[j];
      const double v = -u[j];
      double x;
      if (std::fabs(v) <= 0) x = v;
      else if (v > lam) x = v - lam;
      else if (v < -lam) x = v + lam;
      else x = 0;
      x1[j] = x;
      y1[j] = -x;
      u[j] -= y1[j];
    }
  }
}

// returns 0 on Eigen::Success
int ref_ldlt_inverse(int n, const double* A, double* out) {
  Eigen::Map<const Eigen::MatrixXd> Am(A, n, n);
  Eigen::LDLT<Eigen::MatrixXd> ldlt;
  ldlt.compute(Am);
  if (ldlt.info() != Eigen::Success) return 1;
  Eigen::Map<Eigen::MatrixXd>(out, n, n) = ldlt.solve(Eigen::MatrixXd::Identity(n, n));
  return 0;
}

int ref_llt_solve(int n, const double* A, const double* b, double* x) {
  Eigen::Map<const Eigen::MatrixXd> Am(A, n, n);
  Eigen::LLT<Eigen::MatrixXd> llt;
  llt.compute(Am);
  if (llt.info() != Eigen::Success) return 1;
  Eigen::Map<Eigen::VectorXd>(x, n) = llt.solve(Eigen::Map<const Eigen::VectorXd>(b, n));
  return 0;
}

// Y is m x n; d (n), V (n x n), U (m x n)
int ref_gram_svd(int m, int n, const double* Y, double* d_out, double* V_out, double* U_out) {
  Eigen::Map<const Eigen::MatrixXd> Ym(Y, m, n);
  Eigen::MatrixXd EPS = Eigen::VectorXd::Constant(n, 1e-15).asDiagonal();
  Eigen::SelfAdjointEigenSolver<Eigen::MatrixXd> solver(Ym.transpose() * Ym + EPS);
  if (solver.info() != Eigen::Success) return 1;
  Eigen::VectorXd d = solver.eigenvalues();
  Eigen::MatrixXd V = solver.eigenvectors();
  d = d.cwiseMax(0).cwiseSqrt();
  Eigen::VectorXd dinv(d.rows());
  for (int i = 0; i < d.rows(); i++) dinv(i) = d(i) != 0 ? 1 / d(i) : 0;
  Eigen::Map<Eigen::VectorXd>(d_out, n) = d;
  Eigen::Map<Eigen::MatrixXd>(V_out, n, n) = V;
  Eigen::Map<Eigen::MatrixXd>(U_out, m, n) = Ym * V * dinv.asDiagonal();
  return 0;
}
}

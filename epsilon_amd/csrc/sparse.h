// Sparse linear maps (reference src/epsilon/linear/sparse_matrix_impl.{h,cc}).
//
// The reference holds an Eigen::SparseMatrix<double> (CSC) and leans on Eigen for every
// operation.  Here:
//   * the fp64 CSC master copy lives on the host and carries the SETUP-time structure algebra
//     (sparse x sparse, sparse + sparse, Kronecker expansion): irregular, done once per Init,
//     the same split the diagonal type uses;
//   * everything a sweep touches runs on the device: Apply is a CSR SpMV over a compute-dtype
//     copy in HBM (the CSC arrays of A double as the CSR arrays of A^T, so both directions are
//     resident after one upload each), and products / sums with dense maps are device kernels
//     (kernels_sparse.hip) that never densify the sparse operand.
#pragma once

#include <memory>
#include <vector>

#include "linear_map.h"

namespace eps {

// Column-compressed host matrix, row indices sorted within each column, no duplicates.
struct HostCsc {
  int64_t m = 0, n = 0;
  std::vector<int32_t> colptr;  // n + 1
  std::vector<int32_t> rowidx;  // nnz
  std::vector<double> val;      // nnz
  int64_t nnz() const { return static_cast<int64_t>(val.size()); }
};

HostCsc CscTranspose(const HostCsc& A);
// diag(d) * A (rows scaled) / A * diag(d) (columns scaled); d of size 1 = a scalar
HostCsc CscScaleRows(const HostCsc& A, const std::vector<double>& d);
HostCsc CscScaleCols(const HostCsc& A, const std::vector<double>& d);
HostCsc CscAdd(const HostCsc& A, const HostCsc& B);
HostCsc CscMultiply(const HostCsc& A, const HostCsc& B);
HostCsc CscKron(const HostCsc& A, const HostCsc& B);
HostCsc CscDiagonal(const std::vector<double>& d);
HostCsc CscFromDense(const std::vector<double>& colmajor, int64_t m, int64_t n);
// reference vector/vector_util.cc:182-199
bool CscIsDiagonal(const HostCsc& A);
bool CscIsScalar(const HostCsc& A, double* alpha);
// Wire blob (int32 col_ptr[n+1] | int32 row_index[nnz] | float64 values[nnz]) -> CSC
// (reference vector/vector_util.cc:261-281).
HostCsc CscFromBlob(const pb::Constant& c, const void* bytes, size_t len);

// Row-compressed device copy (int32 structure, compute-dtype values).
struct DeviceCsr {
  int64_t rows = 0, cols = 0, nnz = 0;
  std::shared_ptr<Buffer> ptr, idx;
  DVec val;
  const int32_t* rowptr() const { return static_cast<const int32_t*>(ptr->p); }
  const int32_t* colidx() const { return static_cast<const int32_t*>(idx->p); }
};

// What every handle on one sparse matrix shares: the host CSC arrays as they arrived, their
// transpose (built on first use, once), and the device CSR forms of both.
struct SparseCore {
  std::shared_ptr<const HostCsc> A;            // as constructed
  mutable std::shared_ptr<const HostCsc> At;   // CscTranspose(*A), lazily
  mutable std::shared_ptr<DeviceCsr> csr_a;    // CSR of A   (= the arrays of At)
  mutable std::shared_ptr<DeviceCsr> csr_at;   // CSR of A^T (= the arrays of A)
  const HostCsc& Transposed() const;
};

// A sparse map = (core, transposed?, scalar factor): transposing and multiplying by a scalar are
// O(1) and share everything - as the dense maps do (the reference's Eigen expressions copy the
// 7.5e6-entry data matrix of its lasso_sparse benchmark for each of them; the first version here
// did too: five 60-80 ms sparse products with a diagonal and four 40 ms transposes in one setup).
// The device CSR arrays carry the UNSCALED values: callers fold scale() into their alpha.
class SparseMatrixImpl final : public LinearMapImpl {  // linear/sparse_matrix_impl.h:12-38
 public:
  SparseMatrixImpl(HostCsc A, DType dt);
  SparseMatrixImpl(std::shared_ptr<const SparseCore> core, bool transposed, double scale, DType dt);
  int64_t m() const override { return transposed_ ? core_->A->n : core_->A->m; }
  int64_t n() const override { return transposed_ ? core_->A->m : core_->A->n; }
  std::string DebugString() const override;
  std::shared_ptr<const LinearMapImpl> Transpose() const override;
  std::shared_ptr<const LinearMapImpl> Inverse() const override;
  bool Equals(const LinearMapImpl& other) const override;
  void Apply(double alpha, const DVec& x, double beta, const DVec& y) const override;
  std::vector<double> AsDenseHost() const override;

  // host CSC of the map itself (transposition and factor applied; materialised once when needed)
  const HostCsc& csc() const;
  // the same without the scalar factor (structure and unscaled values; never copies)
  const HostCsc& csc_unscaled() const { return transposed_ ? core_->Transposed() : *core_->A; }
  DType dtype() const { return dt_; }
  double scale() const { return scale_; }
  std::shared_ptr<const LinearMapImpl> Scaled(double alpha) const;  // alpha * this, O(1)
  // CSR of this map / of its transpose, UNSCALED values, uploaded on first use.
  const DeviceCsr& csr() const;
  const DeviceCsr& csr_of_transpose() const;

 private:
  std::shared_ptr<const SparseCore> core_;
  bool transposed_ = false;
  double scale_ = 1.0;
  DType dt_;
  mutable std::shared_ptr<const HostCsc> scaled_;  // csc() when scale_ != 1
};

// Host CSC form of any map (the reference's AsSparse(): kronecker_product_impl.cc:24-43).
HostCsc AsSparseHost(const LinearMapImpl& A);

namespace k {
// y = alpha * S x + beta * y, S in CSR (kernels_sparse.hip; replaces Eigen's A_*x,
// sparse_matrix_impl.h:25).  beta == 0 never reads y.
void SpmvCsr(const DeviceCsr& S, double alpha, const DVec& x, double beta, const DVec& y);
// C (S.rows x N, ld = S.rows) = alpha * S * B, B is S.cols x N column-major (ldb)
void SpmmCsrDense(const DeviceCsr& S, double alpha, const DVec& B, int64_t ldb, int64_t N,
                  const DVec& C);
// C (M x St.rows, ld = M) = alpha * A * S where St is the CSR of S^T (i.e. S in CSC) and A is
// M x S.rows column-major (lda)
void DenseSpmmCsc(const DeviceCsr& St, double alpha, const DVec& A, int64_t lda, int64_t M,
                  const DVec& C);
// W (St.cols x St.rows column-major, ld = St.cols) += alpha * S, St = CSR of S^T
void ScatterAddCsc(const DeviceCsr& St, double alpha, const DVec& W);
}  // namespace k

}  // namespace eps

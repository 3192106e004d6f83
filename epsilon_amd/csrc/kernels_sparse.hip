// Sparse kernels for gfx950 (reference linear/sparse_matrix_impl.h:25 and
// the Sparse rows of linear_map_{multiply,add}.cc, which are Eigen expressions there).
//
// SpMV is HBM-bound on the (value, column-index) stream: (s + 4) bytes per non-zero plus the
// gathers of x.  CSR with a power-of-two group of lanes per row, chosen from the average row
// length so that a 64-lane wave always covers whole rows: short rows (selection matrices, one
// non-zero per row) get one lane each and read the stream fully coalesced; long rows get a
// whole wave and reduce with DPP shuffles.  Sums run in a fixed order (no atomics), so two runs
// give the same bits.  Matrices whose rows average >= 1024 entries get a workgroup per row.
//
// The dense x sparse products never densify the sparse operand: output columns are sparse
// combinations of dense columns (coalesced), output rows of S*B are gathered per column of B.
#include <hip/hip_runtime.h>

#include "sparse.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;

template <class T, int LANES>
__global__ __launch_bounds__(kBlock) void SpmvCsrKernel(int64_t rows,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colidx,
                                                         const T* __restrict__ val,
                                                         const T* __restrict__ x, T alpha, T beta,
                                                         T* __restrict__ y) {
  constexpr int kRowsPerBlock = kBlock / LANES;
  const int lane = threadIdx.x % LANES;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock + threadIdx.x / LANES;
  T acc = 0;
  if (row < rows) {
    const int32_t e = rowptr[row + 1];
    for (int32_t p = rowptr[row] + lane; p < e; p += LANES) acc += val[p] * x[colidx[p]];
  }
#pragma unroll
  for (int off = LANES / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, LANES);
  if (row < rows && lane == 0) y[row] = beta == T(0) ? alpha * acc : alpha * acc + beta * y[row];
}

// one workgroup per row, for matrices whose rows are thousands of entries long
template <class T>
__global__ __launch_bounds__(kBlock) void SpmvCsrBlockKernel(const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ colidx,
                                                              const T* __restrict__ val,
                                                              const T* __restrict__ x, T alpha,
                                                              T beta, T* __restrict__ y) {
  __shared__ T part[kBlock / 64];
  const int64_t row = blockIdx.x;
  const int32_t e = rowptr[row + 1];
  T acc = 0;
  for (int32_t p = rowptr[row] + threadIdx.x; p < e; p += kBlock) acc += val[p] * x[colidx[p]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T s = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) s += part[w];
    y[row] = beta == T(0) ? alpha * s : alpha * s + beta * y[row];
  }
}

template <class T, int LANES>
void LaunchSpmv(const DeviceCsr& S, T alpha, const T* x, T beta, T* y) {
  const int64_t rows_per_block = kBlock / LANES;
  const int64_t grid = (S.rows + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((SpmvCsrKernel<T, LANES>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                     Runtime::Get().stream(), S.rows, S.rowptr(), S.colidx(), S.val.as<T>(), x,
                     alpha, beta, y);
}

template <class T>
void Spmv(const DeviceCsr& S, double alpha, const DVec& x, double beta, const DVec& y) {
  const T a = static_cast<T>(alpha), b = static_cast<T>(beta);
  const T* xp = x.as<T>();
  T* yp = y.as<T>();
  const int64_t avg = S.rows ? (S.nnz + S.rows - 1) / S.rows : 0;
  if (avg >= 1024 && S.rows <= (int64_t(1) << 20)) {
    hipLaunchKernelGGL((SpmvCsrBlockKernel<T>), dim3(static_cast<unsigned>(S.rows)), dim3(kBlock),
                       0, Runtime::Get().stream(), S.rowptr(), S.colidx(), S.val.as<T>(), xp, a, b,
                       yp);
    return;
  }
  if (avg <= 1) LaunchSpmv<T, 1>(S, a, xp, b, yp);
  else if (avg <= 2) LaunchSpmv<T, 2>(S, a, xp, b, yp);
  else if (avg <= 4) LaunchSpmv<T, 4>(S, a, xp, b, yp);
  else if (avg <= 8) LaunchSpmv<T, 8>(S, a, xp, b, yp);
  else if (avg <= 16) LaunchSpmv<T, 16>(S, a, xp, b, yp);
  else if (avg <= 32) LaunchSpmv<T, 32>(S, a, xp, b, yp);
  else LaunchSpmv<T, 64>(S, a, xp, b, yp);
}

// C[i, j] = alpha * sum_p S[i, p] B[p, j]: thread per output row, blockIdx.y = column of B
template <class T>
__global__ __launch_bounds__(kBlock) void SpmmCsrDenseKernel(int64_t rows,
                                                              const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ colidx,
                                                              const T* __restrict__ val,
                                                              const T* __restrict__ B, int64_t ldb,
                                                              T alpha, T* __restrict__ C) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= rows) return;
  const T* b = B + static_cast<int64_t>(blockIdx.y) * ldb;
  T acc = 0;
  for (int32_t p = rowptr[i], e = rowptr[i + 1]; p < e; ++p) acc += val[p] * b[colidx[p]];
  C[i + static_cast<int64_t>(blockIdx.y) * rows] = alpha * acc;
}

// C[:, j] = alpha * sum_{p in column j of S} S.val[p] * A[:, S.row[p]]; coalesced over rows of A
template <class T>
__global__ __launch_bounds__(kBlock) void DenseSpmmCscKernel(int64_t M,
                                                              const int32_t* __restrict__ colptr,
                                                              const int32_t* __restrict__ rowidx,
                                                              const T* __restrict__ val,
                                                              const T* __restrict__ A, int64_t lda,
                                                              T alpha, T* __restrict__ C) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= M) return;
  const int64_t j = blockIdx.y;
  T acc = 0;
  for (int32_t p = colptr[j], e = colptr[j + 1]; p < e; ++p)
    acc += val[p] * A[i + static_cast<int64_t>(rowidx[p]) * lda];
  C[i + j * M] = alpha * acc;
}

// W[row[p], j] += alpha * val[p]; one lane per stored entry of column j (entries are unique)
template <class T>
__global__ __launch_bounds__(kBlock) void ScatterAddCscKernel(int64_t cols, int64_t ld,
                                                               const int32_t* __restrict__ colptr,
                                                               const int32_t* __restrict__ rowidx,
                                                               const T* __restrict__ val, T alpha,
                                                               T* __restrict__ W) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + threadIdx.x / 64;
  if (j >= cols) return;
  for (int32_t p = colptr[j] + (threadIdx.x & 63), e = colptr[j + 1]; p < e; p += 64)
    W[rowidx[p] + j * ld] += alpha * val[p];
}

void CheckVals(const DeviceCsr& S, const DVec& v) {
  EPS_CHECK_MSG(S.val.dt == v.dt, "sparse kernel: dtype mismatch");
}

}  // namespace

#define EPS_DISPATCH(dt, ...) \
  do {                        \
    if ((dt) == F32) {        \
      using T = float;        \
      __VA_ARGS__;            \
    } else {                  \
      using T = double;       \
      __VA_ARGS__;            \
    }                         \
  } while (0)

void SpmvCsr(const DeviceCsr& S, double alpha, const DVec& x, double beta, const DVec& y) {
  EPS_CHECK(x.n == S.cols && y.n == S.rows && x.dt == y.dt);
  CheckVals(S, x);
  if (S.rows == 0) return;
  ProfScope prof("spmv_csr", S.rows, S.nnz);
  EPS_DISPATCH(x.dt, Spmv<T>(S, alpha, x, beta, y));
  EPS_HIP(hipGetLastError());
}

void SpmmCsrDense(const DeviceCsr& S, double alpha, const DVec& B, int64_t ldb, int64_t N,
                  const DVec& C) {
  EPS_CHECK(ldb >= S.cols && B.n >= (N - 1) * ldb + S.cols && C.n == S.rows * N && B.dt == C.dt);
  CheckVals(S, B);
  if (S.rows == 0 || N == 0) return;
  EPS_CHECK_MSG(N <= 65535, "SpmmCsrDense: too many columns");
  ProfScope prof("spmm_csr_dense", S.rows, N);
  dim3 grid(static_cast<unsigned>((S.rows + kBlock - 1) / kBlock), static_cast<unsigned>(N));
  EPS_DISPATCH(B.dt, hipLaunchKernelGGL((SpmmCsrDenseKernel<T>), grid, dim3(kBlock), 0,
                                        Runtime::Get().stream(), S.rows, S.rowptr(), S.colidx(),
                                        S.val.as<T>(), B.as<T>(), ldb, static_cast<T>(alpha),
                                        C.as<T>()));
  EPS_HIP(hipGetLastError());
}

void DenseSpmmCsc(const DeviceCsr& St, double alpha, const DVec& A, int64_t lda, int64_t M,
                  const DVec& C) {
  // St is the CSR of S^T: St.rows = columns of S, St.cols = rows of S
  EPS_CHECK(lda >= M && A.n >= (St.cols - 1) * lda + M && C.n == M * St.rows && A.dt == C.dt);
  CheckVals(St, A);
  if (M == 0 || St.rows == 0) return;
  EPS_CHECK_MSG(St.rows <= 65535, "DenseSpmmCsc: too many columns");
  ProfScope prof("dense_spmm_csc", M, St.rows);
  dim3 grid(static_cast<unsigned>((M + kBlock - 1) / kBlock), static_cast<unsigned>(St.rows));
  EPS_DISPATCH(A.dt, hipLaunchKernelGGL((DenseSpmmCscKernel<T>), grid, dim3(kBlock), 0,
                                        Runtime::Get().stream(), M, St.rowptr(), St.colidx(),
                                        St.val.as<T>(), A.as<T>(), lda, static_cast<T>(alpha),
                                        C.as<T>()));
  EPS_HIP(hipGetLastError());
}

void ScatterAddCsc(const DeviceCsr& St, double alpha, const DVec& W) {
  EPS_CHECK(W.n == St.rows * St.cols);
  CheckVals(St, W);
  if (St.nnz == 0) return;
  const int64_t cols_per_block = kBlock / 64;
  dim3 grid(static_cast<unsigned>((St.rows + cols_per_block - 1) / cols_per_block));
  EPS_DISPATCH(W.dt, hipLaunchKernelGGL((ScatterAddCscKernel<T>), grid, dim3(kBlock), 0,
                                        Runtime::Get().stream(), St.rows, St.cols, St.rowptr(),
                                        St.colidx(), St.val.as<T>(), static_cast<T>(alpha),
                                        W.as<T>()));
  EPS_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace eps

#include "sparse.h"

#include <thread>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <sstream>

#include "kernels.h"

namespace eps {

// ---- host CSC structure algebra (setup time only) ---------------------------------------------

namespace {
// host-side timing of the setup algebra on stderr (EPSILON_HIP_INIT_TRACE=2)
struct HostTrace {
  const char* name;
  int64_t a, b;
  std::chrono::steady_clock::time_point t0;
  bool on;
  HostTrace(const char* n, int64_t a_ = 0, int64_t b_ = 0) : name(n), a(a_), b(b_) {
    static const bool enabled = [] {
      const char* e = std::getenv("EPSILON_HIP_INIT_TRACE");
      return e && e[0] == '2';
    }();
    on = enabled;
    if (on) t0 = std::chrono::steady_clock::now();
  }
  ~HostTrace() {
    if (!on) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms >= 1.0) std::fprintf(stderr, "[host] %-22s %8.2f ms  (%lld, %lld)\n", name, ms,
                                static_cast<long long>(a), static_cast<long long>(b));
  }
};
}  // namespace

namespace {
// f(t) for t in [0, T) on T host threads (T == 1: inline)
template <class F> void OnHostThreads(int T, F f) {
  if (T <= 1) {
    f(0);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(static_cast<size_t>(T - 1));
  for (int t = 1; t < T; ++t) th.emplace_back(f, t);
  f(0);
  for (auto& x : th) x.join();
}
}  // namespace

// A stable counting sort by row.  Large matrices (the 7.5e6-entry data matrix of the reference's
// lasso_sparse problem: 46 ms on one thread) are split into chunks of columns, one per host
// thread: every chunk counts its entries per row, a prefix over (row, chunk) gives each chunk its
// own range inside every row of the result, and the chunks scatter independently - the entries
// of a row end up in ascending column order whatever the number of threads.
HostCsc CscTranspose(const HostCsc& A) {
  HostTrace trace_("CscTranspose", static_cast<int64_t>(A.rowidx.size()), static_cast<int64_t>(A.n));
  HostCsc T;
  T.m = A.n;
  T.n = A.m;
  T.colptr.assign(T.n + 1, 0);
  T.rowidx.resize(A.nnz());
  T.val.resize(A.nnz());
  const int64_t nnz = A.nnz();
  int nt = 1;
  if (nnz >= (int64_t(1) << 20)) {
    nt = HostThreadCount();
    // the per-chunk counters are nt * rows integers: keep them small against the entries
    while (nt > 1 && static_cast<int64_t>(nt) * A.m > nnz / 4) nt /= 2;
  }
  std::vector<int64_t> cbeg(static_cast<size_t>(nt) + 1);  // chunk t: columns [cbeg[t], cbeg[t + 1])
  for (int t = 0; t <= nt; ++t) {  // equal shares of the entries, not of the columns
    const int64_t target = nnz * t / nt;
    cbeg[static_cast<size_t>(t)] =
        t == nt ? A.n : std::lower_bound(A.colptr.begin(), A.colptr.end(), static_cast<int32_t>(target)) - A.colptr.begin();
    if (cbeg[static_cast<size_t>(t)] > A.n) cbeg[static_cast<size_t>(t)] = A.n;
  }
  cbeg[0] = 0;
  std::vector<int32_t> cnt(static_cast<size_t>(nt) * static_cast<size_t>(A.m), 0);  // [chunk][row]
  OnHostThreads(nt, [&](int t) {
    int32_t* c = cnt.data() + static_cast<size_t>(t) * static_cast<size_t>(A.m);
    for (int32_t p = A.colptr[cbeg[t]]; p < A.colptr[cbeg[t + 1]]; ++p) c[A.rowidx[p]]++;
  });
  // exclusive prefix in (row, chunk) order: cnt[t][r] becomes the first position of chunk t in row r
  int32_t run = 0;
  for (int64_t r = 0; r < A.m; ++r) {
    T.colptr[r] = run;
    for (int t = 0; t < nt; ++t) {
      int32_t& c = cnt[static_cast<size_t>(t) * static_cast<size_t>(A.m) + static_cast<size_t>(r)];
      const int32_t k = c;
      c = run;
      run += k;
    }
  }
  T.colptr[T.n] = run;
  OnHostThreads(nt, [&](int t) {
    int32_t* next = cnt.data() + static_cast<size_t>(t) * static_cast<size_t>(A.m);
    for (int64_t j = cbeg[t]; j < cbeg[t + 1]; ++j)  // columns in order => rows of T sorted
      for (int32_t p = A.colptr[j]; p < A.colptr[j + 1]; ++p) {
        const int32_t q = next[A.rowidx[p]]++;
        T.rowidx[q] = static_cast<int32_t>(j);
        T.val[q] = A.val[p];
      }
  });
  return T;
}

HostCsc CscScaleRows(const HostCsc& A, const std::vector<double>& d) {
  HostTrace trace_("CscScaleRows", static_cast<int64_t>(A.rowidx.size()), static_cast<int64_t>(d.size()));
  HostCsc S = A;
  if (d.size() == 1) {
    for (double& v : S.val) v *= d[0];
  } else {
    EPS_CHECK(static_cast<int64_t>(d.size()) == A.m);
    for (size_t q = 0; q < S.val.size(); ++q) S.val[q] *= d[static_cast<size_t>(S.rowidx[q])];
  }
  return S;
}

HostCsc CscScaleCols(const HostCsc& A, const std::vector<double>& d) {
  HostTrace trace_("CscScaleCols", static_cast<int64_t>(A.rowidx.size()), static_cast<int64_t>(d.size()));
  HostCsc S = A;
  if (d.size() == 1) {
    for (double& v : S.val) v *= d[0];
  } else {
    EPS_CHECK(static_cast<int64_t>(d.size()) == A.n);
    for (int64_t j = 0; j < A.n; ++j)
      for (int32_t q = S.colptr[j]; q < S.colptr[j + 1]; ++q) S.val[q] *= d[static_cast<size_t>(j)];
  }
  return S;
}

HostCsc CscAdd(const HostCsc& A, const HostCsc& B) {
  HostTrace trace_("CscAdd", static_cast<int64_t>(A.rowidx.size()), static_cast<int64_t>(B.rowidx.size()));
  EPS_CHECK(A.m == B.m && A.n == B.n);
  HostCsc C;
  C.m = A.m;
  C.n = A.n;
  C.colptr.assign(C.n + 1, 0);
  C.rowidx.reserve(A.nnz() + B.nnz());
  C.val.reserve(A.nnz() + B.nnz());
  for (int64_t j = 0; j < C.n; ++j) {
    int32_t p = A.colptr[j], pe = A.colptr[j + 1], q = B.colptr[j], qe = B.colptr[j + 1];
    while (p < pe || q < qe) {
      if (q >= qe || (p < pe && A.rowidx[p] < B.rowidx[q])) {
        C.rowidx.push_back(A.rowidx[p]);
        C.val.push_back(A.val[p++]);
      } else if (p >= pe || B.rowidx[q] < A.rowidx[p]) {
        C.rowidx.push_back(B.rowidx[q]);
        C.val.push_back(B.val[q++]);
      } else {
        C.rowidx.push_back(A.rowidx[p]);
        C.val.push_back(A.val[p++] + B.val[q++]);
      }
    }
    EPS_CHECK_MSG(C.rowidx.size() < (size_t(1) << 31), "sparse sum exceeds int32 indexing");
    C.colptr[j + 1] = static_cast<int32_t>(C.rowidx.size());
  }
  return C;
}

HostCsc CscMultiply(const HostCsc& A, const HostCsc& B) {
  HostTrace trace_("CscMultiply", static_cast<int64_t>(A.rowidx.size()), static_cast<int64_t>(B.rowidx.size()));
  // Gustavson, one column of C at a time: C(:,j) = sum_k A(:,k) B(k,j)
  EPS_CHECK(A.n == B.m);
  HostCsc C;
  C.m = A.m;
  C.n = B.n;
  C.colptr.assign(C.n + 1, 0);
  std::vector<double> acc(A.m, 0.0);
  std::vector<int32_t> mark(A.m, -1), rows;
  for (int64_t j = 0; j < B.n; ++j) {
    rows.clear();
    for (int32_t q = B.colptr[j]; q < B.colptr[j + 1]; ++q) {
      const int32_t k = B.rowidx[q];
      const double b = B.val[q];
      for (int32_t p = A.colptr[k]; p < A.colptr[k + 1]; ++p) {
        const int32_t i = A.rowidx[p];
        if (mark[i] != j) {
          mark[i] = static_cast<int32_t>(j);
          acc[i] = 0.0;
          rows.push_back(i);
        }
        acc[i] += A.val[p] * b;
      }
    }
    std::sort(rows.begin(), rows.end());
    for (int32_t i : rows) {
      C.rowidx.push_back(i);
      C.val.push_back(acc[i]);
    }
    EPS_CHECK_MSG(C.rowidx.size() < (size_t(1) << 31), "sparse product exceeds int32 indexing");
    C.colptr[j + 1] = static_cast<int32_t>(C.rowidx.size());
  }
  return C;
}

HostCsc CscKron(const HostCsc& A, const HostCsc& B) {
  HostCsc C;
  C.m = A.m * B.m;
  C.n = A.n * B.n;
  EPS_CHECK_MSG(C.m < (int64_t(1) << 31) && C.n < (int64_t(1) << 31) &&
                    A.nnz() * B.nnz() < (int64_t(1) << 31),
                "Kronecker product too large for a sparse expansion (" << C.m << " x " << C.n
                                                                       << ")");
  C.colptr.assign(C.n + 1, 0);
  C.rowidx.reserve(A.nnz() * B.nnz());
  C.val.reserve(A.nnz() * B.nnz());
  for (int64_t ja = 0; ja < A.n; ++ja) {
    for (int64_t jb = 0; jb < B.n; ++jb) {
      for (int32_t p = A.colptr[ja]; p < A.colptr[ja + 1]; ++p) {
        for (int32_t q = B.colptr[jb]; q < B.colptr[jb + 1]; ++q) {
          C.rowidx.push_back(static_cast<int32_t>(A.rowidx[p] * B.m + B.rowidx[q]));
          C.val.push_back(A.val[p] * B.val[q]);
        }
      }
      C.colptr[ja * B.n + jb + 1] = static_cast<int32_t>(C.rowidx.size());
    }
  }
  return C;
}

HostCsc CscDiagonal(const std::vector<double>& d) {
  HostCsc C;
  C.m = C.n = static_cast<int64_t>(d.size());
  C.colptr.resize(C.n + 1);
  C.rowidx.resize(C.n);
  C.val = d;
  for (int64_t j = 0; j <= C.n; ++j) C.colptr[j] = static_cast<int32_t>(j);
  for (int64_t j = 0; j < C.n; ++j) C.rowidx[j] = static_cast<int32_t>(j);
  return C;
}

HostCsc CscFromDense(const std::vector<double>& D, int64_t m, int64_t n) {
  HostTrace trace_("CscFromDense", static_cast<int64_t>(m), static_cast<int64_t>(n));
  HostCsc C;
  C.m = m;
  C.n = n;
  C.colptr.assign(n + 1, 0);
  for (int64_t j = 0; j < n; ++j) {
    for (int64_t i = 0; i < m; ++i) {
      const double v = D[i + j * m];
      if (v != 0.0) {
        C.rowidx.push_back(static_cast<int32_t>(i));
        C.val.push_back(v);
      }
    }
    EPS_CHECK(C.rowidx.size() < (size_t(1) << 31));
    C.colptr[j + 1] = static_cast<int32_t>(C.rowidx.size());
  }
  return C;
}

bool CscIsDiagonal(const HostCsc& A) {
  for (int64_t j = 0; j < A.n; ++j)
    for (int32_t p = A.colptr[j]; p < A.colptr[j + 1]; ++p)
      if (A.rowidx[p] != j) return false;
  return true;
}

bool CscIsScalar(const HostCsc& A, double* alpha) {
  if (!CscIsDiagonal(A)) return false;
  const int64_t n = std::min(A.m, A.n);
  if (n == 0) return false;
  auto diag = [&](int64_t j) {
    return A.colptr[j + 1] > A.colptr[j] ? A.val[A.colptr[j]] : 0.0;
  };
  const double a0 = diag(0);
  for (int64_t j = 1; j < n; ++j)
    if (diag(j) != a0) return false;
  *alpha = a0;
  return true;
}

HostCsc CscFromBlob(const pb::Constant& c, const void* bytes, size_t len) {
  HostTrace trace_("CscFromBlob", static_cast<int64_t>(static_cast<int64_t>(len)), static_cast<int64_t>(0));
  EPS_CHECK_MSG(c.constant_type == pb::Constant::SPARSE_MATRIX, "constant is not a sparse matrix");
  const int64_t m = c.m, n = c.n, nnz = c.nnz;
  EPS_CHECK_MSG(len == static_cast<size_t>(nnz) * sizeof(double) +
                           static_cast<size_t>(n + nnz + 1) * sizeof(int32_t),
                "sparse blob '" << c.data_location << "' has " << len << " bytes for " << m
                                << " x " << n << " nnz=" << nnz);
  HostCsc A;
  A.m = m;
  A.n = n;
  A.colptr.resize(n + 1);
  A.rowidx.resize(nnz);
  A.val.resize(nnz);
  const char* p = static_cast<const char*>(bytes);
  std::memcpy(A.colptr.data(), p, (n + 1) * sizeof(int32_t));
  std::memcpy(A.rowidx.data(), p + (n + 1) * sizeof(int32_t), nnz * sizeof(int32_t));
  std::memcpy(A.val.data(), p + (n + 1 + nnz) * sizeof(int32_t), nnz * sizeof(double));
  EPS_CHECK_MSG(A.colptr[0] == 0 && A.colptr[n] == nnz, "sparse blob: bad column pointers");
  // validation on the host threads (23 ms on one for 7.5e6 entries); every column pointer is
  // checked against [0, nnz] before anything is indexed with it
  for (int64_t j = 0; j < n; ++j)
    EPS_CHECK_MSG(A.colptr[j] <= A.colptr[j + 1] && A.colptr[j] >= 0 && A.colptr[j + 1] <= nnz,
                  "sparse blob: column pointers not monotone");
  const int nt = nnz >= (int64_t(1) << 20) ? HostThreadCount() : 1;
  std::vector<int> flags(static_cast<size_t>(nt), 0);  // bit 0: unsorted / duplicate, bit 1: index out of range
  OnHostThreads(nt, [&](int t) {
    int f = 0;
    for (int64_t j = n * t / nt; j < n * (t + 1) / nt; ++j)
      for (int32_t q = A.colptr[j]; q < A.colptr[j + 1]; ++q) {
        if (A.rowidx[q] < 0 || A.rowidx[q] >= m) f |= 2;
        if (q > A.colptr[j] && A.rowidx[q] <= A.rowidx[q - 1]) f |= 1;
      }
    flags[static_cast<size_t>(t)] = f;
  });
  bool sorted = true;
  for (int f : flags) {
    EPS_CHECK_MSG((f & 2) == 0, "sparse blob: row index out of range");
    if (f & 1) sorted = false;
  }
  if (!sorted) {
    // unsorted / duplicated indices within a column: sort and combine (Eigen's mapped matrix
    // would use them as they are; the value of A x is the same)
    HostCsc S;
    S.m = m;
    S.n = n;
    S.colptr.assign(n + 1, 0);
    std::vector<int32_t> order;
    for (int64_t j = 0; j < n; ++j) {
      order.resize(A.colptr[j + 1] - A.colptr[j]);
      std::iota(order.begin(), order.end(), A.colptr[j]);
      std::stable_sort(order.begin(), order.end(),
                       [&](int32_t a, int32_t b) { return A.rowidx[a] < A.rowidx[b]; });
      for (int32_t q : order) {
        if (static_cast<int32_t>(S.rowidx.size()) > S.colptr[j] && S.rowidx.back() == A.rowidx[q]) {
          S.val.back() += A.val[q];
        } else {
          S.rowidx.push_back(A.rowidx[q]);
          S.val.push_back(A.val[q]);
        }
      }
      S.colptr[j + 1] = static_cast<int32_t>(S.rowidx.size());
    }
    return S;
  }
  return A;
}

// ---- SparseMatrixImpl -----------------------------------------------------------------------------

SparseMatrixImpl::SparseMatrixImpl(HostCsc A, DType dt) : LinearMapImpl(SPARSE_MATRIX), dt_(dt) {
  EPS_CHECK(static_cast<int64_t>(A.colptr.size()) == A.n + 1);
  auto core = std::make_shared<SparseCore>();
  core->A = std::make_shared<const HostCsc>(std::move(A));
  core_ = core;
}

SparseMatrixImpl::SparseMatrixImpl(std::shared_ptr<const SparseCore> core, bool transposed, double scale,
                                   DType dt)
    : LinearMapImpl(SPARSE_MATRIX), core_(std::move(core)), transposed_(transposed), scale_(scale), dt_(dt) {
  EPS_CHECK(core_ != nullptr && core_->A != nullptr);
}

const HostCsc& SparseCore::Transposed() const {
  if (!At) At = std::make_shared<const HostCsc>(CscTranspose(*A));
  return *At;
}

const HostCsc& SparseMatrixImpl::csc() const {
  const HostCsc& base = transposed_ ? core_->Transposed() : *core_->A;
  if (scale_ == 1.0) return base;
  if (!scaled_) scaled_ = std::make_shared<const HostCsc>(CscScaleRows(base, std::vector<double>(1, scale_)));
  return *scaled_;
}

std::string SparseMatrixImpl::DebugString() const {
  std::ostringstream os;
  os << "sparse matrix " << m() << " x " << n() << " nnz=" << core_->A->nnz();
  return os.str();
}

namespace {
std::shared_ptr<Buffer> UploadI32(const std::vector<int32_t>& v) {
  Runtime& rt = Runtime::Get();
  auto buf = rt.Alloc(std::max<size_t>(v.size(), 1) * sizeof(int32_t));
  if (!v.empty()) {
    EPS_HIP(hipMemcpyAsync(buf->p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                           rt.stream()));
    EPS_HIP(hipStreamSynchronize(rt.stream()));
  }
  return buf;
}

// the CSC arrays of a matrix are the CSR arrays of its transpose
std::shared_ptr<DeviceCsr> UploadAsCsrOfTranspose(const HostCsc& A, DType dt) {
  HostTrace trace_("UploadCsr", A.nnz(), A.n);
  auto d = std::make_shared<DeviceCsr>();
  d->rows = A.n;
  d->cols = A.m;
  d->nnz = A.nnz();
  d->ptr = UploadI32(A.colptr);
  d->idx = UploadI32(A.rowidx);
  d->val = DVec::FromHost(A.val.data(), A.nnz(), dt);
  return d;
}

const DeviceCsr& CoreCsrOfA(const SparseCore& c, DType dt) {
  if (!c.csr_a) c.csr_a = UploadAsCsrOfTranspose(c.Transposed(), dt);
  return *c.csr_a;
}
const DeviceCsr& CoreCsrOfAt(const SparseCore& c, DType dt) {
  if (!c.csr_at) c.csr_at = UploadAsCsrOfTranspose(*c.A, dt);
  return *c.csr_at;
}
}  // namespace

const DeviceCsr& SparseMatrixImpl::csr() const {
  return transposed_ ? CoreCsrOfAt(*core_, dt_) : CoreCsrOfA(*core_, dt_);
}

const DeviceCsr& SparseMatrixImpl::csr_of_transpose() const {
  return transposed_ ? CoreCsrOfA(*core_, dt_) : CoreCsrOfAt(*core_, dt_);
}

std::shared_ptr<const LinearMapImpl> SparseMatrixImpl::Transpose() const {
  return std::make_shared<SparseMatrixImpl>(core_, !transposed_, scale_, dt_);
}

std::shared_ptr<const LinearMapImpl> SparseMatrixImpl::Scaled(double alpha) const {
  return std::make_shared<SparseMatrixImpl>(core_, transposed_, scale_ * alpha, dt_);
}

std::shared_ptr<const LinearMapImpl> SparseMatrixImpl::Inverse() const {
  // reference sparse_matrix_impl.cc:60-78: a multiple of the identity inverts as a scalar map,
  // anything else is densified and inverted as a dense matrix.
  EPS_CHECK_MSG(m() == n(), "inverting non-square sparse matrix");
  double alpha;
  if (CscIsScalar(csc(), &alpha)) return ScalarMatrixImpl(n(), alpha).Inverse();
  return ToDense(*this, dt_)->Inverse();
}

bool SparseMatrixImpl::Equals(const LinearMapImpl&) const {
  // reference sparse_matrix_impl.cc:80-89: "Sparse matrix equality not implemented" - always
  // false, which decides e.g. that a sum of Kronecker products with sparse factors is not
  // merged (linear_map_add.cc:213-226).  Kept, so result types match.
  return false;
}

void SparseMatrixImpl::Apply(double alpha, const DVec& x, double beta, const DVec& y) const {
  EPS_CHECK_MSG(x.n == n() && y.n == m(),
                "sparse map " << m() << " x " << n() << " applied to " << x.n << " -> " << y.n);
  k::SpmvCsr(csr(), alpha * scale_, x, beta, y);
}

std::vector<double> SparseMatrixImpl::AsDenseHost() const {
  const HostCsc& A = csc();
  std::vector<double> D(static_cast<size_t>(A.m * A.n), 0.0);
  for (int64_t j = 0; j < A.n; ++j)
    for (int32_t p = A.colptr[j]; p < A.colptr[j + 1]; ++p) D[A.rowidx[p] + j * A.m] = A.val[p];
  return D;
}

HostCsc AsSparseHost(const LinearMapImpl& A) {
  switch (A.type()) {
    case SPARSE_MATRIX: return static_cast<const SparseMatrixImpl&>(A).csc();
    case SCALAR_MATRIX: {
      const auto& S = static_cast<const ScalarMatrixImpl&>(A);
      return CscDiagonal(std::vector<double>(S.n(), S.alpha()));
    }
    case DIAGONAL_MATRIX: return CscDiagonal(static_cast<const DiagonalMatrixImpl&>(A).diagonal());
    case DENSE_MATRIX: return CscFromDense(A.AsDenseHost(), A.m(), A.n());
    case KRONECKER_PRODUCT: {
      const auto& K = static_cast<const KroneckerProductImpl&>(A);
      return CscKron(AsSparseHost(K.A().impl()), AsSparseHost(K.B().impl()));
    }
    default: EPS_FATAL("AsSparseHost: unsupported type " << ImplTypeName(A.type()));
  }
}

}  // namespace eps

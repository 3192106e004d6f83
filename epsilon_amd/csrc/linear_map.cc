#include "linear_map.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>

#include "comm.h"
#include "kernels.h"
#include "sparse.h"

namespace eps {

const char* ImplTypeName(ImplType t) {
  switch (t) {
    case DENSE_MATRIX: return "DENSE_MATRIX";
    case SPARSE_MATRIX: return "SPARSE_MATRIX";
    case DIAGONAL_MATRIX: return "DIAGONAL_MATRIX";
    case SCALAR_MATRIX: return "SCALAR_MATRIX";
    case KRONECKER_PRODUCT: return "KRONECKER_PRODUCT";
    default: return "?";
  }
}

// ---- content ids / OpCache ---------------------------------------------------------------------------

uint64_t HashCombine(uint64_t h, uint64_t v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  return h;
}

uint64_t HashBytes(const void* p, size_t n, uint64_t seed) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  uint64_t h = 1469598103934665603ull ^ seed;
  for (size_t i = 0; i < n; ++i) {
    h ^= b[i];
    h *= 1099511628211ull;
  }
  return h ? h : 1;
}

namespace {
thread_local OpCache* g_op_cache = nullptr;
}
OpCache* CurrentOpCache() { return g_op_cache; }
OpCacheScope::OpCacheScope(OpCache* c) : saved(g_op_cache) { g_op_cache = c; }
OpCacheScope::~OpCacheScope() { g_op_cache = saved; }

// ---- DataMap --------------------------------------------------------------------------------------

void DataMap::InsertOwned(const std::string& key, Blob b) {
  if (b.kind == 0 && b.len > 0) {
    EPS_CHECK_MSG(b.ptr != nullptr, "host blob '" << key << "' has a null pointer");
    const char* src = static_cast<const char*>(b.ptr);
    // A large blob (the 376 MB data matrix of the MNIST-shape problem) is copied by the host
    // threads in stripes into huge pages: one thread's memcpy into freshly mapped 4 KB pages runs
    // at ~6 GB/s (64 ms), the page faults and the copy both scale with the threads (4 ms).
    b.owned = AllocHostBuffer(b.len);
    char* dst = b.owned.get();
    ParallelHostCopy(dst, src, b.len);
    b.ptr = dst;
  }
  Insert(key, b);
}

const Blob& DataMap::Get(const std::string& key) const {
  auto it = blobs_.find(key);
  EPS_CHECK_MSG(it != blobs_.end(), "data location '" << key << "' not in data map");
  return it->second;
}

const pb::Constant& DataMap::Resolve(const pb::Constant& c) const {
  if (c.parameter_id.empty()) return c;
  auto it = params_.find(c.parameter_id);
  EPS_CHECK_MSG(it != params_.end(), "parameter '" << c.parameter_id << "' has no value");
  return it->second;
}

DVec DataMap::DenseDevice(const pb::Constant& c_in) {
  const pb::Constant& c = Resolve(c_in);
  // reference vector/vector_util.cc:247-259
  EPS_CHECK_MSG(c.constant_type == pb::Constant::DENSE_MATRIX, "constant is not a dense matrix");
  const int64_t count = static_cast<int64_t>(c.m) * c.n;
  auto up = uploaded_.find(c.data_location);
  if (up != uploaded_.end()) {
    EPS_CHECK(up->second.n == count);
    return up->second;
  }
  const Blob& b = Get(c.data_location);
  DVec v;
  if (b.kind == 0) {
    EPS_CHECK_MSG(b.len == static_cast<size_t>(count) * sizeof(double),
                  "dense blob '" << c.data_location << "' has " << b.len << " bytes, expected "
                                 << count * sizeof(double));
    static const bool trace = [] {
      const char* e = std::getenv("EPSILON_HIP_INIT_TRACE");
      return e && std::atoi(e) >= 2;
    }();
    auto now = [] {
      return std::chrono::duration<double, std::milli>(
                 std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t0 = trace ? now() : 0;
    v = DVec::FromHost(static_cast<const double*>(b.ptr), count, dtype_);
    const double t1 = trace ? now() : 0;
    // (the owned host copy of a solver handle stays until the handle goes: unmapping it here cost
    // 33 ms of Init for the 376 MB MNIST matrix, on this thread or - through the address-space
    // lock - on a helper thread alike)
    if (trace && count >= (1 << 20))
      std::fprintf(stderr, "[host] dense '%s' %lld elements: to device %.2f ms\n",
                   c.data_location.c_str(), static_cast<long long>(count), t1 - t0);
  } else {
    const DType bdt = b.kind == 1 ? F32 : F64;
    EPS_CHECK_MSG(b.len == static_cast<size_t>(count),
                  "device blob '" << c.data_location << "' has " << b.len
                                  << " elements, expected " << count);
    DVec src = DVec::Borrow(const_cast<void*>(b.ptr), count, bdt);
    if (bdt == dtype_) {
      v = src;  // zero-copy: the matrix stays where the caller put it
    } else {
      v = DVec::Empty(count, dtype_);
      if (bdt == F32) k::ConvertFromF32(v, src.as<float>());
      else k::ConvertFromF64(v, src.as<double>());
    }
  }
  uploaded_[c.data_location] = v;
  return v;
}

uint64_t DataMap::DenseId(const pb::Constant& c_in) const {
  const pb::Constant& c = Resolve(c_in);
  if (c.data_location.empty()) return 0;
  const Blob& b = Get(c.data_location);
  uint64_t h = HashBytes(c.data_location.data(), c.data_location.size(), 0x5eed);
  // owned host copies are identified by (key, generation): their address means nothing and
  // re-binding the key must miss; borrowed memory by its address as well
  if (b.kind != 0) h = HashCombine(h, reinterpret_cast<uintptr_t>(b.ptr));
  auto g = generation_.find(c.data_location);
  h = HashCombine(h, g == generation_.end() ? 0 : g->second);
  h = HashCombine(h, b.len);
  h = HashCombine(h, static_cast<uint64_t>(c.m) << 32 | static_cast<uint32_t>(c.n));
  return h ? h : 1;
}

std::vector<double> DataMap::DenseHost(const pb::Constant& c_in) {
  const pb::Constant& c = Resolve(c_in);
  EPS_CHECK_MSG(c.constant_type == pb::Constant::DENSE_MATRIX, "constant is not a dense matrix");
  const int64_t count = static_cast<int64_t>(c.m) * c.n;
  const Blob& b = Get(c.data_location);
  if (b.kind == 0 && b.ptr != nullptr) {
    EPS_CHECK(b.len == static_cast<size_t>(count) * sizeof(double));
    std::vector<double> out(count);
    std::memcpy(out.data(), b.ptr, b.len);
    return out;
  }
  return DenseDevice(c).ToHost();
}

// ---- LinearMap wrapper ------------------------------------------------------------------------------

LinearMap::LinearMap() : impl_(std::make_shared<ScalarMatrixImpl>(0, 0.0)) {}

LinearMap& LinearMap::operator+=(const LinearMap& rhs) {
  *this = *this + rhs;
  return *this;
}
LinearMap& LinearMap::operator*=(const LinearMap& rhs) {
  *this = *this * rhs;
  return *this;
}

LinearMap LinearMap::Identity(int64_t n) { return Scalar(1.0, n); }
LinearMap LinearMap::Scalar(double alpha, int64_t n) {
  return LinearMap(std::make_shared<ScalarMatrixImpl>(n, alpha));
}
LinearMap LinearMap::Diagonal(std::vector<double> d, DType dt) {
  return LinearMap(std::make_shared<DiagonalMatrixImpl>(std::move(d), dt));
}
LinearMap LinearMap::Dense(DVec data, int64_t rows, int64_t cols, uint64_t id) {
  EPS_CHECK(data.n == rows * cols);
  return LinearMap(std::make_shared<DenseMatrixImpl>(std::move(data), rows, cols, false, 1.0, id));
}
LinearMap LinearMap::Kronecker(LinearMap A, LinearMap B) {
  return LinearMap(std::make_shared<KroneckerProductImpl>(std::move(A), std::move(B)));
}

bool operator==(const LinearMap& lhs, const LinearMap& rhs) { return lhs.impl().Equals(rhs.impl()); }

LinearMap operator*(double alpha, const LinearMap& A) {  // linear_map.cc:33-35
  return LinearMap::Scalar(alpha, A.impl().m()) * A;
}

DType MapDType(const LinearMapImpl& A, DType fallback) {
  switch (A.type()) {
    case DENSE_MATRIX: return static_cast<const DenseMatrixImpl&>(A).dtype();
    case DIAGONAL_MATRIX: return static_cast<const DiagonalMatrixImpl&>(A).dtype();
    case SPARSE_MATRIX: return static_cast<const SparseMatrixImpl&>(A).dtype();
    case KRONECKER_PRODUCT: {
      const auto& K = static_cast<const KroneckerProductImpl&>(A);
      return MapDType(K.A().impl(), MapDType(K.B().impl(), fallback));
    }
    default: return fallback;
  }
}

// ---- Scalar ---------------------------------------------------------------------------------------

std::string ScalarMatrixImpl::DebugString() const {
  std::ostringstream os;
  os << "scalar matrix: n=" << n_ << " alpha=" << alpha_;
  return os.str();
}
std::shared_ptr<const LinearMapImpl> ScalarMatrixImpl::Transpose() const {
  return std::make_shared<ScalarMatrixImpl>(n_, alpha_);
}
std::shared_ptr<const LinearMapImpl> ScalarMatrixImpl::Inverse() const {
  return std::make_shared<ScalarMatrixImpl>(n_, 1 / alpha_);  // scalar_matrix_impl.h:30-32
}
bool ScalarMatrixImpl::Equals(const LinearMapImpl& o) const {
  if (o.type() != SCALAR_MATRIX || o.m() != m() || o.n() != n()) return false;
  return static_cast<const ScalarMatrixImpl&>(o).alpha() == alpha_;
}
void ScalarMatrixImpl::Apply(double alpha, const DVec& x, double beta, const DVec& y) const {
  EPS_CHECK_MSG(x.n == n_ && y.n == n_, "scalar map of size " << n_ << " applied to " << x.n);
  k::Axpby(y, alpha * alpha_, x, beta);
}
std::vector<double> ScalarMatrixImpl::AsDenseHost() const {
  std::vector<double> D(n_ * n_, 0.0);
  for (int64_t i = 0; i < n_; ++i) D[i + i * n_] = alpha_;
  return D;
}

// ---- Diagonal -------------------------------------------------------------------------------------

DiagonalMatrixImpl::DiagonalMatrixImpl(std::vector<double> d, DType dt)
    : LinearMapImpl(DIAGONAL_MATRIX), d_(std::move(d)) {
  dev_ = DVec::FromHost(d_.data(), static_cast<int64_t>(d_.size()), dt);
}
std::string DiagonalMatrixImpl::DebugString() const {
  std::ostringstream os;
  os << "diagonal matrix: n=" << d_.size();
  return os.str();
}
std::shared_ptr<const LinearMapImpl> DiagonalMatrixImpl::Transpose() const {
  return std::make_shared<DiagonalMatrixImpl>(d_, dev_.dt);
}
std::shared_ptr<const LinearMapImpl> DiagonalMatrixImpl::Inverse() const {
  std::vector<double> inv(d_.size());
  for (size_t i = 0; i < d_.size(); ++i) inv[i] = d_[i] ? 1 / d_[i] : 0;  // diagonal_matrix_impl.cc:19-21
  return std::make_shared<DiagonalMatrixImpl>(std::move(inv), dev_.dt);
}
bool DiagonalMatrixImpl::Equals(const LinearMapImpl& o) const {
  if (o.type() != DIAGONAL_MATRIX || o.m() != m()) return false;
  return static_cast<const DiagonalMatrixImpl&>(o).d_ == d_;
}
void DiagonalMatrixImpl::Apply(double alpha, const DVec& x, double beta, const DVec& y) const {
  EPS_CHECK(x.n == n() && y.n == m());
  k::DiagMul(y, alpha, dev_, x, beta);
}
std::vector<double> DiagonalMatrixImpl::AsDenseHost() const {
  const int64_t n = m();
  std::vector<double> D(n * n, 0.0);
  for (int64_t i = 0; i < n; ++i) D[i + i * n] = d_[i];
  return D;
}

// ---- Dense ----------------------------------------------------------------------------------------

std::string DenseMatrixImpl::DebugString() const {
  std::ostringstream os;
  os << "dense matrix " << m() << " x " << n() << (trans_ ? " (T)" : "") << " scale=" << scale_;
  return os.str();
}
std::shared_ptr<const LinearMapImpl> DenseMatrixImpl::Transpose() const {
  return std::make_shared<DenseMatrixImpl>(data_, rows_, cols_, !trans_, scale_, id_, symmetric_);
}
DVec DenseMatrixImpl::Materialize(bool force_copy) const {
  if (!trans_ && scale_ == 1.0 && !force_copy) return data_;
  DVec out = DVec::Empty(m() * n(), data_.dt);
  k::MatCopy(trans_, m(), n(), scale_, data_, rows_, out);
  return out;
}
namespace {
// S^-1 = V diag(1 / lambda) V^T for a symmetric S: the eigendecomposition is the SVD of the shifted
// positive definite matrix S + cI, c > ||S||_F (singular vectors = eigenvectors, sigma = lambda + c).
DVec SymmetricInverseByEig(const DVec& S, int64_t n) {
  Runtime& rt = Runtime::Get();
  rt.ResetSlots();
  const int slot = rt.NewSlot();
  k::SumSq(S, rt.SlotPtr(slot), false);
  rt.FetchSlots();
  const double fro = std::sqrt(rt.SlotValue(slot));
  EPS_CHECK_MSG(fro > 0 && std::isfinite(fro), "dense inverse: the matrix is zero or not finite");
  const double shift = fro * 1.0625;
  DVec W = S.Clone();
  k::AddDiag(W, n, n, shift, nullptr);
  DVec V = DVec::Empty(n * n, S.dt);
  k::JacobiSvd(W, n, n, V, 60, false, false);
  DVec sigma = DVec::Empty(n, S.dt);
  k::ColNorms(W, n, n, sigma, false);
  std::vector<double> lam = sigma.ToHost();
  const double tiny = (S.dt == F32 ? 1e-5 : 1e-12) * fro;
  for (double& l : lam) {
    l -= shift;
    EPS_CHECK_MSG(std::fabs(l) > tiny, "dense inverse: the matrix is singular to working precision");
    l = 1.0 / l;
  }
  DVec linv = DVec::FromHost(lam.data(), n, S.dt);
  DVec ones = DVec::Full(n, 1.0, S.dt);
  DVec T = V.Clone();
  k::ColScaleByRatio(T, n, n, ones, linv);  // T = V diag(1 / lambda)
  DVec out = DVec::Empty(n * n, S.dt);
  k::Gemm(false, true, n, n, n, 1.0, T, n, V, n, 0.0, out, n);
  // exactly symmetric, as the symmetric apply assumes
  k::SymmetrizeFromLower(out, n, n);
  return out;
}
}  // namespace

std::shared_ptr<const LinearMapImpl> DenseMatrixImpl::Inverse() const {
  // reference dense_matrix_impl.cc:21-30: symmetric assumed, LDLT + solve(I).  Here: the
  // Schur complements of the prox KKT systems are definite, so factor sign*W by Cholesky.
  EPS_CHECK_MSG(m() == n(), "inverting non-square dense matrix");
  const int64_t nn = n();
  if (nn == 0) return std::make_shared<DenseMatrixImpl>(data_, 0, 0, false, 1.0);
  // sign of the first diagonal entry decides positive / negative definite
  double d0;
  {
    DVec first = data_.Slice(0, 1);
    std::vector<double> h = first.ToHost();
    d0 = h[0] * scale_;
  }
  const double sign = d0 < 0 ? -1.0 : 1.0;
  uint64_t key = 0;
  OpCache* cache = CurrentOpCache();
  if (cache && id_) {
    key = HashDouble(HashCombine(HashCombine(id_, 0x1171), trans_ ? 2 : 1), scale_);
    if (auto hit = cache->Find(key))
      return std::make_shared<DenseMatrixImpl>(hit->data(), nn, nn, false, sign, key, true);
  }
  DVec W = DVec::Empty(nn * nn, data_.dt);
  k::MatCopy(trans_, nn, nn, sign * scale_, data_, rows_, W);
  try {
    k::SpdInverseInPlace(W, nn);
  } catch (const Error&) {
    // Not definite.  The reference's LDLT (dense_matrix_impl.cc:25-29) also inverts symmetric
    // INDEFINITE matrices; the Schur complements of the prox KKT systems never are, so this is
    // the rare path: through the eigendecomposition (no pivoting needed, same kernels as the
    // symmetric matrix proxes).
    (void)hipGetLastError();
    k::MatCopy(trans_, nn, nn, sign * scale_, data_, rows_, W);
    W = SymmetricInverseByEig(W, nn);
  }
  auto result = std::make_shared<DenseMatrixImpl>(W, nn, nn, false, sign, key, true);
  if (cache && key) cache->Put(key, result);
  return result;
}
std::shared_ptr<const LinearMapImpl> DenseMatrixImpl::InverseDistributed() const {
  Comm* comm = Runtime::Get().comm();
  const int64_t nn = n();
  // measured at n = 1e4 (tools_microbench.py): whole inverse 23.9 ms; a 1/8 slab of columns
  // 19.7 ms, a quarter 22.3 ms, a half 27.5 ms - the Cholesky (13.3 ms) is the common part, so
  // with the all-gather on top the split is marginal on one node.  Kept (and tested) for
  // larger blocks / more ranks: EPSILON_HIP_DIST_INVERSE=<min ranks> turns it on (tests use 2).
  int min_ranks = 16;
  if (const char* e = std::getenv("EPSILON_HIP_DIST_INVERSE")) min_ranks = std::atoi(e);
  if (comm == nullptr || comm->size() < std::max(2, min_ranks) || m() != nn || nn < 1024)
    return Inverse();
  double d0;
  {
    std::vector<double> h = data_.Slice(0, 1).ToHost();
    d0 = h[0] * scale_;
  }
  const double sign = d0 < 0 ? -1.0 : 1.0;
  uint64_t key = 0;
  OpCache* cache = CurrentOpCache();
  if (cache && id_) {  // same key as Inverse(): the result is the same matrix
    key = HashDouble(HashCombine(HashCombine(id_, 0x1171), trans_ ? 2 : 1), scale_);
    if (auto hit = cache->Find(key))
      return std::make_shared<DenseMatrixImpl>(hit->data(), nn, nn, false, sign, key, true);
  }
  const int G = comm->size();
  const int64_t per = (nn + G - 1) / G;
  const int64_t lo = std::min<int64_t>(nn, comm->rank() * per);
  const int64_t cnt = std::min<int64_t>(nn, lo + per) - lo;
  DVec W = DVec::Empty(nn * nn, data_.dt);
  k::MatCopy(trans_, nn, nn, sign * scale_, data_, rows_, W);
  DVec mine = DVec::Zeros(nn * per, data_.dt);  // padded to the common slab width
  k::SpdInverseColumns(W, nn, lo, cnt, mine);
  DVec full = DVec::Empty(nn * per * G, data_.dt);
  comm->AllGather(mine.data(), full.data(), static_cast<size_t>(nn * per), data_.dt);
  DVec inv = full.Slice(0, nn * nn);  // column slabs are contiguous in column-major storage
  // the columns come from independent solves: make the matrix exactly symmetric again (the
  // symmetric apply reads only its lower triangle)
  k::SymmetrizeFromLower(inv, nn, nn);
  auto result = std::make_shared<DenseMatrixImpl>(inv, nn, nn, false, sign, key, true);
  if (cache && key) cache->Put(key, result);
  return result;
}

bool DenseMatrixImpl::Equals(const LinearMapImpl& o) const {
  if (o.type() != DENSE_MATRIX || o.m() != m() || o.n() != n()) return false;
  const auto& A = static_cast<const DenseMatrixImpl&>(o);
  if (A.trans_ != trans_ || A.scale_ != scale_) return false;
  if (A.data_.data() == data_.data()) return true;
  if (A.data_.dt != data_.dt) return false;
  Runtime& rt = Runtime::Get();
  rt.ResetSlots();
  int s = rt.NewSlot();
  k::SumSqDiff(A.data_, data_, rt.SlotPtr(s), false);
  rt.FetchSlots();
  return rt.SlotValue(s) == 0.0;
}
void DenseMatrixImpl::Apply(double alpha, const DVec& x, double beta, const DVec& y) const {
  if (symmetric_ && rows_ >= 1024) {  // half the bytes; below that the two launches cost more
    // from the third apply on from a tile-packed copy (+ rows^2 / 2 values): 41 -> 31 us at 1e4
    // (EPSILON_HIP_SYMV_PACKED=0: never; =2: from the first apply - tests; read per call)
    const char* e = std::getenv("EPSILON_HIP_SYMV_PACKED");
    const bool packed = !(e && e[0] == '0');
    const int first = (e && e[0] == '2') ? 1 : 3;
    if (!packed) packed_ = DVec();
    if (packed && packed_.n == 0 && ++applies_ >= first && !Runtime::Get().capturing())
      packed_ = k::SymvPack(rows_, data_, rows_);
    if (packed_.n > 0) k::SymvPacked(rows_, alpha * scale_, packed_, x, beta, y);
    else k::Symv(rows_, alpha * scale_, data_, rows_, x, beta, y);
    return;
  }
  k::Gemv(trans_, rows_, cols_, alpha * scale_, data_, rows_, x, beta, y);
}
std::vector<double> DenseMatrixImpl::AsDenseHost() const { return Materialize(false).ToHost(); }

// ---- Kronecker ------------------------------------------------------------------------------------

std::string KroneckerProductImpl::DebugString() const {
  return "kronecker product\nA: " + A_.impl().DebugString() + "\nB: " + B_.impl().DebugString();
}
std::shared_ptr<const LinearMapImpl> KroneckerProductImpl::Transpose() const {
  return std::make_shared<KroneckerProductImpl>(A_.Transpose(), B_.Transpose());
}
std::shared_ptr<const LinearMapImpl> KroneckerProductImpl::Inverse() const {
  return std::make_shared<KroneckerProductImpl>(A_.Inverse(), B_.Inverse());
}
bool KroneckerProductImpl::Equals(const LinearMapImpl& o) const {
  if (o.type() != KRONECKER_PRODUCT || o.m() != m() || o.n() != n()) return false;
  const auto& K = static_cast<const KroneckerProductImpl&>(o);
  return K.A() == A_ && K.B() == B_;
}

namespace {
// C (M x N) = alpha * F * X + beta*C where X is (F.n x N) column-major contiguous
void LeftMultiply(const LinearMapImpl& F, double alpha, const DVec& X, int64_t N, double beta,
                  const DVec& C) {
  const int64_t M = F.m(), K = F.n();
  if (F.type() == SCALAR_MATRIX) {
    k::Axpby(C, alpha * static_cast<const ScalarMatrixImpl&>(F).alpha(), X, beta);
  } else if (F.type() == DENSE_MATRIX) {
    const auto& D = static_cast<const DenseMatrixImpl&>(F);
    k::Gemm(D.trans(), false, M, N, K, alpha * D.scale(), D.data(), D.rows(), X, K, beta, C, M);
  } else {
    auto D = ToDense(F, X.dt);
    k::Gemm(false, false, M, N, K, alpha, D->data(), D->rows(), X, K, beta, C, M);
  }
}
// C (M x N) = alpha * X * F^T + beta*C where X is (M x F.n) contiguous, N = F.m
void RightMultiplyT(const LinearMapImpl& F, double alpha, const DVec& X, int64_t M, double beta,
                    const DVec& C) {
  const int64_t N = F.m(), K = F.n();
  if (F.type() == SCALAR_MATRIX) {
    k::Axpby(C, alpha * static_cast<const ScalarMatrixImpl&>(F).alpha(), X, beta);
  } else if (F.type() == DENSE_MATRIX) {
    const auto& D = static_cast<const DenseMatrixImpl&>(F);
    // F^T = op'(data) with the transpose flag flipped
    k::Gemm(false, !D.trans(), M, N, K, alpha * D.scale(), X, M, D.data(), D.rows(), beta, C, M);
  } else {
    auto D = ToDense(F, X.dt);
    k::Gemm(false, true, M, N, K, alpha, X, M, D->data(), D->rows(), beta, C, M);
  }
}
}  // namespace

void KroneckerProductImpl::Apply(double alpha, const DVec& x, double beta, const DVec& y) const {
  // (A (x) B) vec(X) = vec(B X A^T), X is B.n x A.n column-major
  // (reference kronecker_product_impl.cc:45-58 does it as (A (B X)^T)^T with two copies)
  const LinearMapImpl& A = A_.impl();
  const LinearMapImpl& B = B_.impl();
  EPS_CHECK(x.n == n() && y.n == m());
  if (A.type() == SCALAR_MATRIX) {
    LeftMultiply(B, alpha * static_cast<const ScalarMatrixImpl&>(A).alpha(), x, A.n(), beta, y);
    return;
  }
  if (B.type() == SCALAR_MATRIX) {
    RightMultiplyT(A, alpha * static_cast<const ScalarMatrixImpl&>(B).alpha(), x, B.n(), beta, y);
    return;
  }
  DVec T = DVec::Empty(B.m() * A.n(), x.dt);
  LeftMultiply(B, 1.0, x, A.n(), 0.0, T);
  RightMultiplyT(A, alpha, T, B.m(), beta, y);
}

std::vector<double> KroneckerProductImpl::AsDenseHost() const {
  std::vector<double> A = A_.impl().AsDenseHost(), B = B_.impl().AsDenseHost();
  const int64_t mA = A_.impl().m(), nA = A_.impl().n(), mB = B_.impl().m(), nB = B_.impl().n();
  const int64_t M = mA * mB, N = nA * nB;
  std::vector<double> C(M * N);
  for (int64_t ja = 0; ja < nA; ++ja)
    for (int64_t jb = 0; jb < nB; ++jb)
      for (int64_t ia = 0; ia < mA; ++ia)
        for (int64_t ib = 0; ib < mB; ++ib)
          C[(ia * mB + ib) + (ja * nB + jb) * M] = A[ia + ja * mA] * B[ib + jb * mB];
  return C;
}

// ---- dense fallback -------------------------------------------------------------------------------

std::shared_ptr<const DenseMatrixImpl> ToDense(const LinearMapImpl& A, DType dt) {
  const int64_t m = A.m(), n = A.n();
  EPS_CHECK_MSG(m * n <= (int64_t(1) << 28), "refusing to densify a " << m << " x " << n << " "
                                                                     << ImplTypeName(A.type()));
  switch (A.type()) {
    case DENSE_MATRIX: {
      const auto& D = static_cast<const DenseMatrixImpl&>(A);
      return std::make_shared<DenseMatrixImpl>(D.Materialize(false), m, n, false, 1.0);
    }
    case SCALAR_MATRIX: {
      DVec W = DVec::Zeros(m * n, dt);
      k::AddDiag(W, n, n, static_cast<const ScalarMatrixImpl&>(A).alpha(), nullptr);
      return std::make_shared<DenseMatrixImpl>(W, m, n, false, 1.0);
    }
    case DIAGONAL_MATRIX: {
      const auto& D = static_cast<const DiagonalMatrixImpl&>(A);
      DVec W = DVec::Zeros(m * n, dt);
      k::AddDiag(W, n, n, 1.0, &D.device());
      return std::make_shared<DenseMatrixImpl>(W, m, n, false, 1.0);
    }
    case KRONECKER_PRODUCT: {
      const auto& K = static_cast<const KroneckerProductImpl&>(A);
      auto DA = ToDense(K.A().impl(), dt);
      auto DB = ToDense(K.B().impl(), dt);
      DVec W = DVec::Empty(m * n, dt);
      k::KronDense(W, DA->data(), DA->rows(), DA->cols(), DB->data(), DB->rows(), DB->cols());
      return std::make_shared<DenseMatrixImpl>(W, m, n, false, 1.0);
    }
    case SPARSE_MATRIX: {
      const auto& S = static_cast<const SparseMatrixImpl&>(A);
      DVec W = DVec::Zeros(m * n, S.dtype());
      k::ScatterAddCsc(S.csr_of_transpose(), S.scale(), W);
      return std::make_shared<DenseMatrixImpl>(W, m, n, false, 1.0);
    }
    default:
      EPS_FATAL("ToDense: unsupported type " << ImplTypeName(A.type()));
  }
}

// ---- multiply table (reference linear/linear_map_multiply.cc) -------------------------------------

namespace {

using ImplPtr = std::shared_ptr<const LinearMapImpl>;

DType PairDType(const LinearMapImpl& a, const LinearMapImpl& b) {
  return MapDType(a, MapDType(b, CurrentDType()));
}

ImplPtr MultiplyDenseDense(const DenseMatrixImpl& A, const DenseMatrixImpl& B) {
  // reference :14-37 dgemm_ with the operands' trans flags
  const int64_t M = A.m(), K = A.n(), N = B.n();
  // The product of the raw buffers is memoised by content id; the operands' lazy scalar
  // factors ride on the result as its lazy factor.
  const double scale = A.scale() * B.scale();
  uint64_t key = 0;
  OpCache* cache = CurrentOpCache();
  if (cache && A.id() && B.id()) {
    key = HashCombine(HashCombine(HashCombine(A.id(), 0x6d756c), B.id()),
                      (A.trans() ? 2 : 0) + (B.trans() ? 1 : 0));
    if (auto hit = cache->Find(key))
      return std::make_shared<DenseMatrixImpl>(hit->data(), M, N, false, scale, key);
  }
  DVec C = DVec::Empty(M * N, A.dtype());
  // A * A^T (or A^T * A) of one shared buffer: compute the lower triangle only and mirror
  const bool syrk = A.data().data() == B.data().data() && A.rows() == B.rows() &&
                    A.cols() == B.cols() && A.trans() != B.trans() && M == N;
  k::Gemm(A.trans(), B.trans(), M, N, K, 1.0, A.data(), A.rows(), B.data(), B.rows(), 0.0, C, M,
          syrk);
  if (syrk) k::SymmetrizeFromLower(C, M, M);
  auto unit = std::make_shared<DenseMatrixImpl>(C, M, N, false, 1.0, key);
  if (cache && key) cache->Put(key, unit);
  return std::make_shared<DenseMatrixImpl>(C, M, N, false, scale, key);
}

ImplPtr MultiplyViaDense(const LinearMapImpl& L, const LinearMapImpl& R) {
  DType dt = PairDType(L, R);
  return MultiplyDenseDense(*ToDense(L, dt), *ToDense(R, dt));
}

ImplPtr Multiply(const LinearMapImpl& L, const LinearMapImpl& R);

ImplPtr MultiplyScalarKron(const ScalarMatrixImpl& S, const KroneckerProductImpl& K) {
  // reference :188-198: stays Kronecker
  ScalarMatrixImpl S1(K.A().impl().m(), S.alpha());
  ScalarMatrixImpl S2(K.B().impl().m(), 1);
  return std::make_shared<KroneckerProductImpl>(LinearMap(Multiply(S1, K.A().impl())),
                                                LinearMap(Multiply(S2, K.B().impl())));
}

// Sparse result formed from the operands' host CSC structure (reference: Eigen sparse
// expressions on AsSparse() forms, linear_map_multiply.cc:79-110,119-153,208-241).
ImplPtr MultiplyDenseSparse(const DenseMatrixImpl& A, const SparseMatrixImpl& S);

ImplPtr MultiplyViaSparse(const LinearMapImpl& L, const LinearMapImpl& R) {
  // a scalar / diagonal factor only scales the rows or columns of the sparse one: one pass over
  // its values instead of a general sparse product with a diagonal matrix built for the purpose
  // (60-80 ms each on a 7.5e6-entry data matrix, five of them in one lasso setup)
  auto diag_of = [](const LinearMapImpl& D, std::vector<double>* d) {
    if (D.type() == SCALAR_MATRIX) {
      d->assign(1, static_cast<const ScalarMatrixImpl&>(D).alpha());
      return true;
    }
    if (D.type() == DIAGONAL_MATRIX) {
      *d = static_cast<const DiagonalMatrixImpl&>(D).diagonal();
      return true;
    }
    return false;
  };
  std::vector<double> d;
  if (R.type() == SPARSE_MATRIX && diag_of(L, &d)) {
    const auto& SR = static_cast<const SparseMatrixImpl&>(R);
    if (d.size() == 1) return SR.Scaled(d[0]);  // a lazy factor: shares every array
    return std::make_shared<SparseMatrixImpl>(CscScaleRows(SR.csc(), d), SR.dtype());
  }
  if (L.type() == SPARSE_MATRIX && diag_of(R, &d)) {
    const auto& SL = static_cast<const SparseMatrixImpl&>(L);
    if (d.size() == 1) return SL.Scaled(d[0]);
    return std::make_shared<SparseMatrixImpl>(CscScaleCols(SL.csc(), d), SL.dtype());
  }
  if (L.type() == SPARSE_MATRIX && R.type() == SPARSE_MATRIX) {
    // A sparse-sparse product that is really a dense contraction (the Gram matrix A A^T of a
    // 10 % dense 1500 x 50000 data matrix is 10^9 multiply-adds through Gustavson's row merges,
    // seconds on one host thread): form it on the device from the densified left operand and
    // hand the result back in the Sparse type the reference's table prescribes.
    const auto& SL = static_cast<const SparseMatrixImpl&>(L);
    const auto& SR = static_cast<const SparseMatrixImpl&>(R);
    const HostCsc& A = SL.csc_unscaled();  // (structure only: no scaled copy for an estimate)
    const HostCsc& B = SR.csc_unscaled();
    std::vector<int64_t> brow(static_cast<size_t>(B.m), 0);
    for (int32_t r : B.rowidx) ++brow[static_cast<size_t>(r)];
    double work = 0;
    for (int64_t k = 0; k < A.n; ++k)
      work += static_cast<double>(A.colptr[k + 1] - A.colptr[k]) * static_cast<double>(brow[k]);
    if (work > 5e7 && L.m() * L.n() <= (int64_t(1) << 28) && L.m() * R.n() <= (int64_t(1) << 25)) {
      const DType dt = PairDType(L, R);
      ImplPtr D = MultiplyDenseSparse(*ToDense(L, dt), SR);
      return std::make_shared<SparseMatrixImpl>(CscFromDense(D->AsDenseHost(), L.m(), R.n()), dt);
    }
  }
  return std::make_shared<SparseMatrixImpl>(CscMultiply(AsSparseHost(L), AsSparseHost(R)),
                                            PairDType(L, R));
}

ImplPtr MultiplyDenseSparse(const DenseMatrixImpl& A, const SparseMatrixImpl& S) {
  // reference :39-45 (Dense result).  Output column j is a sparse combination of A's columns.
  DVec Am = A.Materialize(false);
  DVec C = DVec::Empty(A.m() * S.n(), A.dtype());
  k::DenseSpmmCsc(S.csr_of_transpose(), S.scale(), Am, A.m(), A.m(), C);
  return std::make_shared<DenseMatrixImpl>(C, A.m(), S.n(), false, 1.0);
}

ImplPtr MultiplySparseDense(const SparseMatrixImpl& S, const DenseMatrixImpl& B) {
  // reference :71-77 (Dense result)
  DVec Bm = B.Materialize(false);
  DVec C = DVec::Empty(S.m() * B.n(), B.dtype());
  k::SpmmCsrDense(S.csr(), S.scale(), Bm, B.m(), B.n(), C);
  return std::make_shared<DenseMatrixImpl>(C, S.m(), B.n(), false, 1.0);
}

ImplPtr Multiply(const LinearMapImpl& L, const LinearMapImpl& R) {
  EPS_CHECK_MSG(L.n() == R.m(), "multiply: A: " << L.DebugString() << "\nB: " << R.DebugString());
  const ImplType a = L.type(), b = R.type();
  if (a == SPARSE_MATRIX || b == SPARSE_MATRIX) {
    if (a == DENSE_MATRIX)
      return MultiplyDenseSparse(static_cast<const DenseMatrixImpl&>(L),
                                 static_cast<const SparseMatrixImpl&>(R));
    if (b == DENSE_MATRIX)
      return MultiplySparseDense(static_cast<const SparseMatrixImpl&>(L),
                                 static_cast<const DenseMatrixImpl&>(R));
    return MultiplyViaSparse(L, R);
  }
  if (a == SCALAR_MATRIX) {
    const auto& S = static_cast<const ScalarMatrixImpl&>(L);
    switch (b) {
      case SCALAR_MATRIX:
        return std::make_shared<ScalarMatrixImpl>(
            S.n(), S.alpha() * static_cast<const ScalarMatrixImpl&>(R).alpha());
      case DIAGONAL_MATRIX: {
        const auto& D = static_cast<const DiagonalMatrixImpl&>(R);
        std::vector<double> d = D.diagonal();
        for (auto& v : d) v = S.alpha() * v;
        return std::make_shared<DiagonalMatrixImpl>(std::move(d), D.dtype());
      }
      case DENSE_MATRIX: {
        const auto& D = static_cast<const DenseMatrixImpl&>(R);
        return std::make_shared<DenseMatrixImpl>(D.data(), D.rows(), D.cols(), D.trans(),
                                                 S.alpha() * D.scale(), D.id(), D.symmetric());
      }
      case KRONECKER_PRODUCT:
        return MultiplyScalarKron(S, static_cast<const KroneckerProductImpl&>(R));
      default: break;
    }
  }
  if (b == SCALAR_MATRIX) {
    const auto& S = static_cast<const ScalarMatrixImpl&>(R);
    switch (a) {
      case DIAGONAL_MATRIX: {
        const auto& D = static_cast<const DiagonalMatrixImpl&>(L);
        std::vector<double> d = D.diagonal();
        for (auto& v : d) v = v * S.alpha();
        return std::make_shared<DiagonalMatrixImpl>(std::move(d), D.dtype());
      }
      case DENSE_MATRIX: {
        const auto& D = static_cast<const DenseMatrixImpl&>(L);
        return std::make_shared<DenseMatrixImpl>(D.data(), D.rows(), D.cols(), D.trans(),
                                                 D.scale() * S.alpha(), D.id(), D.symmetric());
      }
      case KRONECKER_PRODUCT:  // reference :223-227 delegates with swapped arguments
        return MultiplyScalarKron(S, static_cast<const KroneckerProductImpl&>(L));
      default: break;
    }
  }
  if (a == DIAGONAL_MATRIX && b == DIAGONAL_MATRIX) {
    const auto& D = static_cast<const DiagonalMatrixImpl&>(L);
    const auto& E = static_cast<const DiagonalMatrixImpl&>(R);
    std::vector<double> d(D.diagonal().size());
    for (size_t i = 0; i < d.size(); ++i) d[i] = D.diagonal()[i] * E.diagonal()[i];
    return std::make_shared<DiagonalMatrixImpl>(std::move(d), D.dtype());
  }
  if (a == DENSE_MATRIX && b == DENSE_MATRIX)
    return MultiplyDenseDense(static_cast<const DenseMatrixImpl&>(L),
                              static_cast<const DenseMatrixImpl&>(R));
  if (a == KRONECKER_PRODUCT && b == KRONECKER_PRODUCT) {  // reference :230-241
    const auto& C = static_cast<const KroneckerProductImpl&>(L);
    const auto& D = static_cast<const KroneckerProductImpl&>(R);
    if (C.A().impl().n() == D.A().impl().m() && C.B().impl().n() == D.B().impl().m())
      return std::make_shared<KroneckerProductImpl>(C.A() * D.A(), C.B() * D.B());
  }
  // Dense x {Diagonal, Kronecker}, {Diagonal, Kronecker} x Dense: Dense result (reference
  // :47-69,112-118,200-207); the Diagonal / Kronecker mixtures: Sparse result (:143-153,
  // :215-221,:240).
  if (a == DENSE_MATRIX || b == DENSE_MATRIX) return MultiplyViaDense(L, R);
  return MultiplyViaSparse(L, R);
}

// ---- add table (reference linear/linear_map_add.cc) -----------------------------------------------

ImplPtr Add(const LinearMapImpl& L, const LinearMapImpl& R);

ImplPtr AddViaDense(const LinearMapImpl& L, const LinearMapImpl& R) {
  DType dt = PairDType(L, R);
  auto A = ToDense(L, dt);
  auto B = ToDense(R, dt);
  DVec C = A->Materialize(true);
  k::Axpby(C, 1.0, B->Materialize(false), 1.0);
  return std::make_shared<DenseMatrixImpl>(C, L.m(), L.n(), false, 1.0);
}

ImplPtr AddViaSparse(const LinearMapImpl& L, const LinearMapImpl& R) {
  return std::make_shared<SparseMatrixImpl>(CscAdd(AsSparseHost(L), AsSparseHost(R)),
                                            PairDType(L, R));
}

ImplPtr AddScalarKron(const ScalarMatrixImpl& S, const KroneckerProductImpl& K) {
  // reference :167-187: kron(A, alpha*I) + beta*I = kron(A + (beta/alpha) I, alpha*I)
  if (K.A().impl().type() == SCALAR_MATRIX) {
    const auto& KS = static_cast<const ScalarMatrixImpl&>(K.A().impl());
    LinearMap S1 = LinearMap::Scalar(0, K.A().impl().n());
    LinearMap S2 = LinearMap::Scalar(S.alpha() / KS.alpha(), K.B().impl().n());
    return std::make_shared<KroneckerProductImpl>(S1 + K.A(), S2 + K.B());
  }
  if (K.B().impl().type() == SCALAR_MATRIX) {
    const auto& KS = static_cast<const ScalarMatrixImpl&>(K.B().impl());
    LinearMap S1 = LinearMap::Scalar(S.alpha() / KS.alpha(), K.A().impl().n());
    LinearMap S2 = LinearMap::Scalar(0, K.B().impl().n());
    return std::make_shared<KroneckerProductImpl>(S1 + K.A(), S2 + K.B());
  }
  return AddViaSparse(S, K);  // reference :186
}

ImplPtr Add(const LinearMapImpl& L, const LinearMapImpl& R) {
  EPS_CHECK_MSG(L.m() == R.m() && L.n() == R.n(),
                "add: A: " << L.DebugString() << "\nB: " << R.DebugString());
  const ImplType a = L.type(), b = R.type();
  if (a == SPARSE_MATRIX || b == SPARSE_MATRIX) {
    if (a == DENSE_MATRIX || b == DENSE_MATRIX) {  // reference :21-28,56-63 (Dense result)
      const auto& D = static_cast<const DenseMatrixImpl&>(a == DENSE_MATRIX ? L : R);
      const auto& S = static_cast<const SparseMatrixImpl&>(a == DENSE_MATRIX ? R : L);
      DVec C = D.Materialize(true);
      k::ScatterAddCsc(S.csr_of_transpose(), S.scale(), C);
      return std::make_shared<DenseMatrixImpl>(C, D.m(), D.n(), false, 1.0);
    }
    return AddViaSparse(L, R);  // reference :65-98
  }
  if (a == SCALAR_MATRIX && b == SCALAR_MATRIX) {
    return std::make_shared<ScalarMatrixImpl>(
        L.n(), static_cast<const ScalarMatrixImpl&>(L).alpha() +
                   static_cast<const ScalarMatrixImpl&>(R).alpha());
  }
  if (a == DIAGONAL_MATRIX && b == DIAGONAL_MATRIX) {
    const auto& D = static_cast<const DiagonalMatrixImpl&>(L);
    const auto& E = static_cast<const DiagonalMatrixImpl&>(R);
    std::vector<double> d(D.diagonal().size());
    for (size_t i = 0; i < d.size(); ++i) d[i] = D.diagonal()[i] + E.diagonal()[i];
    return std::make_shared<DiagonalMatrixImpl>(std::move(d), D.dtype());
  }
  if ((a == DIAGONAL_MATRIX && b == SCALAR_MATRIX) || (a == SCALAR_MATRIX && b == DIAGONAL_MATRIX)) {
    const auto& D = static_cast<const DiagonalMatrixImpl&>(a == DIAGONAL_MATRIX ? L : R);
    const auto& S = static_cast<const ScalarMatrixImpl&>(a == DIAGONAL_MATRIX ? R : L);
    std::vector<double> d = D.diagonal();
    for (auto& v : d) v = v + S.alpha();
    return std::make_shared<DiagonalMatrixImpl>(std::move(d), D.dtype());
  }
  if (a == DENSE_MATRIX || b == DENSE_MATRIX) {
    const LinearMapImpl& Dm = (a == DENSE_MATRIX) ? L : R;
    const LinearMapImpl& O = (a == DENSE_MATRIX) ? R : L;
    const auto& D = static_cast<const DenseMatrixImpl&>(Dm);
    if (O.type() == DENSE_MATRIX) {
      const auto& E = static_cast<const DenseMatrixImpl&>(O);
      DVec C = static_cast<const DenseMatrixImpl&>(L).Materialize(true);
      k::Axpby(C, 1.0, static_cast<const DenseMatrixImpl&>(R).Materialize(false), 1.0);
      (void)E;
      return std::make_shared<DenseMatrixImpl>(C, L.m(), L.n(), false, 1.0);
    }
    if (O.type() == SCALAR_MATRIX) {
      const double a = static_cast<const ScalarMatrixImpl&>(O).alpha();
      uint64_t key = 0;
      OpCache* cache = CurrentOpCache();
      if (cache && D.id()) {
        key = HashDouble(HashDouble(HashCombine(HashCombine(D.id(), 0xadd5), D.trans() ? 2 : 1),
                                    D.scale()), a);
        if (auto hit = cache->Find(key)) return hit;
      }
      DVec C = D.Materialize(true);
      k::AddDiag(C, D.m(), D.m(), a, nullptr);
      auto result = std::make_shared<DenseMatrixImpl>(C, D.m(), D.n(), false, 1.0, key);
      if (cache && key) cache->Put(key, result);
      return result;
    }
    if (O.type() == DIAGONAL_MATRIX) {
      DVec C = D.Materialize(true);
      k::AddDiag(C, D.m(), D.m(), 1.0, &static_cast<const DiagonalMatrixImpl&>(O).device());
      return std::make_shared<DenseMatrixImpl>(C, D.m(), D.n(), false, 1.0);
    }
    return AddViaDense(L, R);
  }
  if (a == SCALAR_MATRIX && b == KRONECKER_PRODUCT)
    return AddScalarKron(static_cast<const ScalarMatrixImpl&>(L),
                         static_cast<const KroneckerProductImpl&>(R));
  if (a == KRONECKER_PRODUCT && b == SCALAR_MATRIX)
    return AddScalarKron(static_cast<const ScalarMatrixImpl&>(R),
                         static_cast<const KroneckerProductImpl&>(L));
  if (a == KRONECKER_PRODUCT && b == KRONECKER_PRODUCT) {  // reference :213-226
    const auto& K1 = static_cast<const KroneckerProductImpl&>(L);
    const auto& K2 = static_cast<const KroneckerProductImpl&>(R);
    if (K1.A() == K2.A()) return std::make_shared<KroneckerProductImpl>(K1.A(), K1.B() + K2.B());
    if (K1.B() == K2.B()) return std::make_shared<KroneckerProductImpl>(K1.A() + K2.A(), K1.B());
  }
  return AddViaSparse(L, R);  // Diagonal + Kronecker, unmatched Kronecker sums (:129-138,:224)
}

}  // namespace

LinearMap operator*(const LinearMap& lhs, const LinearMap& rhs) {
  return LinearMap(Multiply(lhs.impl(), rhs.impl()));
}
LinearMap operator+(const LinearMap& lhs, const LinearMap& rhs) {
  return LinearMap(Add(lhs.impl(), rhs.impl()));
}

// ---- proto -> map -----------------------------------------------------------------------------------

LinearMap BuildLinearMap(const pb::LinearMap& p, DataMap* data) {
  switch (p.linear_map_type) {
    case pb::LinearMap::DENSE_MATRIX: {
      const pb::Constant& c = data->Resolve(p.constant);
      return LinearMap::Dense(data->DenseDevice(c), c.m, c.n, data->DenseId(c));
    }
    case pb::LinearMap::DIAGONAL_MATRIX:
      return LinearMap::Diagonal(data->DenseHost(p.constant), data->dtype());
    case pb::LinearMap::SCALAR:
      return LinearMap::Scalar(p.scalar, p.n);
    case pb::LinearMap::KRONECKER_PRODUCT:
      EPS_CHECK(p.arg.size() == 2);
      return LinearMap::Kronecker(BuildLinearMap(p.arg[0], data), BuildLinearMap(p.arg[1], data));
    case pb::LinearMap::TRANSPOSE:
      EPS_CHECK(p.arg.size() == 1);
      return BuildLinearMap(p.arg[0], data).Transpose();
    case pb::LinearMap::SPARSE_MATRIX: {
      const pb::Constant& c = data->Resolve(p.constant);
      const Blob& b = data->Get(c.data_location);
      EPS_CHECK_MSG(b.kind == 0, "sparse constants must be host blobs");
      return LinearMap(std::make_shared<SparseMatrixImpl>(CscFromBlob(c, b.ptr, b.len),
                                                          data->dtype()));
    }
    default:
      EPS_FATAL("No linear map function for type " << p.linear_map_type);
  }
}

std::vector<double> GetDiagonal(const LinearMap& A) {
  const LinearMapImpl& impl = A.impl();
  if (impl.type() == SCALAR_MATRIX) {
    const auto& S = static_cast<const ScalarMatrixImpl&>(impl);
    return std::vector<double>(S.n(), S.alpha());
  }
  EPS_CHECK_MSG(impl.type() == DIAGONAL_MATRIX, "Non-diagonal linear map " << impl.DebugString());
  return static_cast<const DiagonalMatrixImpl&>(impl).diagonal();
}

double GetScalar(const LinearMap& A) {
  EPS_CHECK_MSG(A.impl().type() == SCALAR_MATRIX, "Non-scalar matrix " << A.impl().DebugString());
  return static_cast<const ScalarMatrixImpl&>(A.impl()).alpha();
}

// ||A||_1 (largest absolute column sum) of a symmetric block; for the transposed view of a dense
// buffer this is the infinity norm of the map - equal for the symmetric pivot blocks this is used on.
double OneNorm(const LinearMapImpl& A) {
  switch (A.type()) {
    case SCALAR_MATRIX: return std::fabs(static_cast<const ScalarMatrixImpl&>(A).alpha());
    case DIAGONAL_MATRIX: {
      double mx = 0;
      for (double d : static_cast<const DiagonalMatrixImpl&>(A).diagonal()) mx = std::max(mx, std::fabs(d));
      return mx;
    }
    case DENSE_MATRIX: {
      const auto& D = static_cast<const DenseMatrixImpl&>(A);
      if (D.rows() == 0 || D.cols() == 0) return 0;
      auto buf = Runtime::Get().Alloc(static_cast<size_t>(D.cols()) * sizeof(double));
      k::ColAbsSums(D.data(), D.rows(), D.cols(), D.rows(), static_cast<double*>(buf->p));
      std::vector<double> h(static_cast<size_t>(D.cols()));
      EPS_HIP(hipMemcpyAsync(h.data(), buf->p, h.size() * sizeof(double), hipMemcpyDeviceToHost,
                             Runtime::Get().stream()));
      Runtime::Get().Sync();
      double mx = 0;
      for (double v : h) mx = std::max(mx, v);
      return mx * std::fabs(D.scale());
    }
    case SPARSE_MATRIX: {
      const auto& S = static_cast<const SparseMatrixImpl&>(A);
      const HostCsc& C = S.csc_unscaled();
      double mx = 0;
      for (int64_t j = 0; j < C.n; ++j) {
        double sum = 0;
        for (int32_t p = C.colptr[j]; p < C.colptr[j + 1]; ++p) sum += std::fabs(C.val[p]);
        mx = std::max(mx, sum);
      }
      return mx * std::fabs(S.scale());
    }
    case KRONECKER_PRODUCT: {
      const auto& K = static_cast<const KroneckerProductImpl&>(A);
      return OneNorm(K.A().impl()) * OneNorm(K.B().impl());
    }
    default: return 0;
  }
}

namespace {
// ||A||_2 of a symmetric map, from below, by `steps` power iterations on a fixed pseudo-random
// start; normalisation reads the norm from a device slot, the host waits once at the end.
double SpectralNormLowerBound(const LinearMapImpl& A, int steps) {
  const int64_t n = A.n();
  if (n == 0 || A.m() != n) return 0;
  const DType dt = MapDType(A, CurrentDType());
  Runtime& rt = Runtime::Get();
  DVec x = DVec::Empty(n, dt), y = DVec::Empty(n, dt);
  k::FillHash(x, 0x5eedu);
  rt.ResetSlots();
  int slot = rt.NewSlot();
  k::SumSq(x, rt.SlotPtr(slot), false);
  k::ScaleByInvNorm(x, x, rt.SlotPtr(slot));
  for (int it = 0; it < steps; ++it) {
    A.Apply(1.0, x, 0.0, y);
    slot = rt.NewSlot();
    k::SumSq(y, rt.SlotPtr(slot), false);
    k::ScaleByInvNorm(x, y, rt.SlotPtr(slot));
  }
  rt.FetchSlots();
  return std::sqrt(rt.SlotValue(slot));  // ||A x|| for the last unit x
}
}  // namespace

double ConditionEstimate(const LinearMap& B, const LinearMap& Binv) {
  if (B.impl().type() == SCALAR_MATRIX) return 1.0;
  if (B.impl().type() == DIAGONAL_MATRIX) {
    // zero entries stay zero in the reference's inverse (diagonal_matrix_impl.cc:19-21): not counted
    double mx = 0, mn = std::numeric_limits<double>::infinity();
    for (double d : static_cast<const DiagonalMatrixImpl&>(B.impl()).diagonal()) {
      if (d == 0) continue;
      mx = std::max(mx, std::fabs(d));
      mn = std::min(mn, std::fabs(d));
    }
    return mx > 0 ? mx / mn : 1.0;
  }
  // kappa_1 is an upper bound on kappa_2 for a symmetric block and costs two passes: enough to
  // clear the well-conditioned case.  It overshoots by up to ~n for large dense blocks (a random
  // 4100 x 4100 Gram matrix with kappa_2 = 9 has kappa_1 in the thousands), so above the threshold
  // the decision is taken on kappa_2 itself: ||B||_2 ||B^-1||_2 from a few power iterations on
  // the block and on its explicit inverse (a lower bound; x 1.5 for the iterations not done).
  const double k1 = OneNorm(B.impl()) * OneNorm(Binv.impl());
  if (!(k1 > 1e3)) return k1;
  const double k2 = 1.5 * SpectralNormLowerBound(B.impl(), 6) * SpectralNormLowerBound(Binv.impl(), 6);
  return std::min(k1, k2);
}

ImplType ComputeType(ImplType A, ImplType B) {  // linear_map.cc:141-149
  if (A <= SCALAR_MATRIX && B <= SCALAR_MATRIX) return A < B ? A : B;
  return DENSE_MATRIX;
}

uint64_t Nonzeros(ImplType type, uint64_t m, uint64_t n) {  // linear_map.cc:151-164
  switch (type) {
    case DENSE_MATRIX:
    case SPARSE_MATRIX: return m * n;
    case DIAGONAL_MATRIX: EPS_CHECK(m == n); return n;
    case SCALAR_MATRIX: return 1;
    default: EPS_FATAL("Nonzeros: not implemented for " << ImplTypeName(type));
  }
}

}  // namespace eps

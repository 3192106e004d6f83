// Typed linear operators with a closed algebra, resident in HBM.
//
// Mirrors the reference's interface for this layer (reference src/epsilon/linear/linear_map.h:
// 16-92): an abstract LinearMapImpl {m, n, Transpose, Inverse, Apply, ==} behind a by-value
// LinearMap wrapper with + and *, five implementation types and two 5x5 dispatch tables
// (linear_map_multiply.cc:249-299, linear_map_add.cc:234-284) that decide the *result type*.
//
// MI355X-first differences:
//   * Apply is y = alpha*A*x + beta*y on device vectors (no temporaries per operator).
//   * DenseMatrixImpl carries a lazy scalar factor and a transpose flag over a shared HBM
//     buffer, so scalar*dense and transposes never copy: the reference holds >= 5 copies of a
//     2-4 GB data matrix during Init (SURVEY.md section 7 "Memory"), this build holds one.
//   * Dense*Dense is a GEMM on the MFMA kernel (SYRK-style when it is A*A^T of one buffer);
//     Dense inverse is the on-device blocked Cholesky inverse.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "device.h"
#include "wire.h"

namespace eps {

enum ImplType {  // reference linear/linear_map.h:18-26 (order matters for ComputeType)
  DENSE_MATRIX = 0,
  SPARSE_MATRIX = 1,
  DIAGONAL_MATRIX = 2,
  SCALAR_MATRIX = 3,
  KRONECKER_PRODUCT = 4,
  NUM_IMPL_TYPES = 5,
};

const char* ImplTypeName(ImplType t);

class LinearMapImpl {
 public:
  explicit LinearMapImpl(ImplType type) : type_(type) {}
  virtual ~LinearMapImpl() {}
  ImplType type() const { return type_; }
  virtual int64_t m() const = 0;
  virtual int64_t n() const = 0;
  virtual std::string DebugString() const = 0;
  virtual std::shared_ptr<const LinearMapImpl> Transpose() const = 0;
  virtual std::shared_ptr<const LinearMapImpl> Inverse() const = 0;
  virtual bool Equals(const LinearMapImpl& other) const = 0;
  // y = alpha * (this) * x + beta * y   (x.n == n(), y.n == m()); beta == 0 never reads y.
  virtual void Apply(double alpha, const DVec& x, double beta, const DVec& y) const = 0;
  // Column-major m() x n() host copy (setup / tests only).
  virtual std::vector<double> AsDenseHost() const = 0;

 private:
  ImplType type_;
};

// A device data blob handed over the C-ABI (see include/epsilon_hip.h eps_blob).
struct Blob {
  const void* ptr = nullptr;
  size_t len = 0;   // bytes for host blobs, element count for device blobs
  int kind = 0;     // 0 host bytes, 1 device f32, 2 device f64
  // Host blobs of a solver handle are COPIED at the boundary, as the reference copies every
  // blob into its DataMap (python/epopt/solvemodule.cc:58-72): `ptr` then points into `owned`.
  std::shared_ptr<char> owned;  // new char[len]: not value-initialised (the copy is the first touch)
};

// Data map {location -> blob} plus a cache of what was already uploaded, so one constant is
// resident once however many expressions refer to it (reference DataMap: vector_util.h).
class DataMap {
 public:
  explicit DataMap(DType dt) : dtype_(dt) {}
  DType dtype() const { return dtype_; }
  // (Re)binds a location.  A device copy made from the previous blob under this key is dropped
  // and the key's generation is bumped, so content ids (OpCache) of the old data never match.
  void Insert(const std::string& key, const Blob& b) {
    blobs_[key] = b;
    uploaded_.erase(key);
    ++generation_[key];
  }
  // Same, after copying a host blob's bytes (device blobs stay borrowed).
  void InsertOwned(const std::string& key, Blob b);
  bool Has(const std::string& key) const { return blobs_.count(key) != 0; }
  const Blob& Get(const std::string& key) const;
  // Dense constant as a device vector of m*n entries (column-major), compute dtype.
  DVec DenseDevice(const pb::Constant& c);
  // Content id of a dense constant (location + blob identity), for the OpCache.
  uint64_t DenseId(const pb::Constant& c) const;
  // Dense constant as host doubles (diagonals, small vectors).
  std::vector<double> DenseHost(const pb::Constant& c);
  // Values bound to CVXPY Parameters for this call (reference solver.cc:109-116).
  void SetParameter(const std::string& id, const pb::Constant& c) { params_[id] = c; }
  const pb::Constant& Resolve(const pb::Constant& c) const;

 private:
  DType dtype_;
  std::map<std::string, Blob> blobs_;
  std::map<std::string, DVec> uploaded_;
  std::map<std::string, uint64_t> generation_;
  std::map<std::string, pb::Constant> params_;
};

class LinearMap {  // reference linear/linear_map.h:61-92
 public:
  LinearMap();  // 0 x 0 scalar, as the reference's default (linear_map.cc:13)
  explicit LinearMap(std::shared_ptr<const LinearMapImpl> impl) : impl_(std::move(impl)) {}
  const LinearMapImpl& impl() const { return *impl_; }
  const std::shared_ptr<const LinearMapImpl>& ptr() const { return impl_; }
  LinearMap Inverse() const { return LinearMap(impl_->Inverse()); }
  LinearMap Transpose() const { return LinearMap(impl_->Transpose()); }
  LinearMap& operator+=(const LinearMap& rhs);
  LinearMap& operator*=(const LinearMap& rhs);

  static LinearMap Identity(int64_t n);
  static LinearMap Scalar(double alpha, int64_t n);
  static LinearMap Diagonal(std::vector<double> d, DType dt);
  // Takes ownership of a column-major device buffer (rows x cols).
  static LinearMap Dense(DVec data, int64_t rows, int64_t cols, uint64_t id = 0);
  static LinearMap Kronecker(LinearMap A, LinearMap B);

 private:
  std::shared_ptr<const LinearMapImpl> impl_;
};

LinearMap operator+(const LinearMap& lhs, const LinearMap& rhs);
LinearMap operator*(const LinearMap& lhs, const LinearMap& rhs);
LinearMap operator*(double alpha, const LinearMap& A);
bool operator==(const LinearMap& lhs, const LinearMap& rhs);

// reference linear/linear_map.cc:83-104 (TRANSPOSE is folded at build time, :67-72)
LinearMap BuildLinearMap(const pb::LinearMap& proto, DataMap* data);

std::vector<double> GetDiagonal(const LinearMap& A);  // linear_map.cc:118-129
double GetScalar(const LinearMap& A);                 // linear_map.cc:131-139

// 1-norm of a (symmetric) map and kappa_1 = ||B||_1 ||B^-1||_1 of a pivot block given its explicit
// inverse (no reference counterpart: the reference factors in fp64 and never asks; the fp32 mode
// uses it to decide on iterative refinement of the block solve, block.cc).
double OneNorm(const LinearMapImpl& A);
double ConditionEstimate(const LinearMap& B, const LinearMap& Binv);

// Fill model of the block elimination (reference linear/linear_map.cc:141-164).
ImplType ComputeType(ImplType A, ImplType B);
uint64_t Nonzeros(ImplType type, uint64_t m, uint64_t n);

// ---- concrete implementations (exposed for the algebra tables and prox operators) ------------

class ScalarMatrixImpl final : public LinearMapImpl {  // linear/scalar_matrix_impl.h:10-42
 public:
  ScalarMatrixImpl(int64_t n, double alpha) : LinearMapImpl(SCALAR_MATRIX), n_(n), alpha_(alpha) {}
  int64_t m() const override { return n_; }
  int64_t n() const override { return n_; }
  std::string DebugString() const override;
  std::shared_ptr<const LinearMapImpl> Transpose() const override;
  std::shared_ptr<const LinearMapImpl> Inverse() const override;
  bool Equals(const LinearMapImpl& other) const override;
  void Apply(double alpha, const DVec& x, double beta, const DVec& y) const override;
  std::vector<double> AsDenseHost() const override;
  double alpha() const { return alpha_; }

 private:
  int64_t n_;
  double alpha_;
};

class DiagonalMatrixImpl final : public LinearMapImpl {  // linear/diagonal_matrix_impl.h:11-37
 public:
  DiagonalMatrixImpl(std::vector<double> d, DType dt);
  int64_t m() const override { return static_cast<int64_t>(d_.size()); }
  int64_t n() const override { return static_cast<int64_t>(d_.size()); }
  std::string DebugString() const override;
  std::shared_ptr<const LinearMapImpl> Transpose() const override;
  std::shared_ptr<const LinearMapImpl> Inverse() const override;
  bool Equals(const LinearMapImpl& other) const override;
  void Apply(double alpha, const DVec& x, double beta, const DVec& y) const override;
  std::vector<double> AsDenseHost() const override;
  const std::vector<double>& diagonal() const { return d_; }
  const DVec& device() const { return dev_; }
  DType dtype() const { return dev_.dt; }

 private:
  std::vector<double> d_;  // fp64 master copy (setup-time algebra is done on the host)
  DVec dev_;               // compute-dtype copy in HBM
};

class DenseMatrixImpl final : public LinearMapImpl {  // linear/dense_matrix_impl.h:13-60
 public:
  // op(data) * scale, data is rows x cols column-major (ld = rows)
  // `id` names the CONTENT of the buffer (0 = anonymous).  Results of the setup algebra get ids
  // derived from their operands' ids, so a later Init with the same matrices (warm start,
  // re-bound vector parameters) finds them in the OpCache instead of redoing GEMM / inverse.
  // `symmetric`: the buffer is known to be exactly symmetric (an explicit inverse of a
  // symmetric matrix); Apply then reads only half of it (k::Symv).
  DenseMatrixImpl(DVec data, int64_t rows, int64_t cols, bool trans, double scale,
                  uint64_t id = 0, bool symmetric = false)
      : LinearMapImpl(DENSE_MATRIX), data_(std::move(data)), rows_(rows), cols_(cols),
        trans_(trans), scale_(scale), id_(id), symmetric_(symmetric && rows == cols) {}
  int64_t m() const override { return trans_ ? cols_ : rows_; }
  int64_t n() const override { return trans_ ? rows_ : cols_; }
  std::string DebugString() const override;
  std::shared_ptr<const LinearMapImpl> Transpose() const override;
  std::shared_ptr<const LinearMapImpl> Inverse() const override;
  // COLLECTIVE: the same inverse for a matrix that is replicated on every rank of a sharded
  // solve - each rank solves for its own slab of columns (Cholesky + triangular solves) and the
  // slabs are all-gathered, instead of every rank forming the whole inverse.
  std::shared_ptr<const LinearMapImpl> InverseDistributed() const;
  bool Equals(const LinearMapImpl& other) const override;
  void Apply(double alpha, const DVec& x, double beta, const DVec& y) const override;
  std::vector<double> AsDenseHost() const override;

  const DVec& data() const { return data_; }
  int64_t rows() const { return rows_; }
  int64_t cols() const { return cols_; }
  bool trans() const { return trans_; }
  double scale() const { return scale_; }
  uint64_t id() const { return id_; }
  bool symmetric() const { return symmetric_; }
  DType dtype() const { return data_.dt; }
  // Contiguous m() x n() buffer holding scale*op(data) (a fresh copy unless already plain).
  DVec Materialize(bool force_copy) const;

 private:
  DVec data_;
  int64_t rows_, cols_;
  bool trans_;
  double scale_;
  uint64_t id_;
  bool symmetric_;
  // a symmetric map that is applied again and again (a cached inverse inside the sweeps) gets a
  // tile-packed copy of its lower tiles for the apply (kernels_gemv.hip: SymvPacked)
  mutable DVec packed_;
  mutable int applies_ = 0;
};

// Memo of dense setup results (Gram products, sums, explicit inverses) keyed by content id.
// One per solver handle; installed for the duration of Solver::Init by OpCacheScope.
class OpCache {
 public:
  std::shared_ptr<const DenseMatrixImpl> Find(uint64_t key) const {
    auto it = map_.find(key);
    return it == map_.end() ? nullptr : it->second;
  }
  void Put(uint64_t key, std::shared_ptr<const DenseMatrixImpl> v) { map_[key] = std::move(v); }
  size_t size() const { return map_.size(); }
  void Clear() { map_.clear(); }

 private:
  std::map<uint64_t, std::shared_ptr<const DenseMatrixImpl>> map_;
};
OpCache* CurrentOpCache();
struct OpCacheScope {
  OpCache* saved;
  explicit OpCacheScope(OpCache* c);
  ~OpCacheScope();
};
uint64_t HashCombine(uint64_t h, uint64_t v);
uint64_t HashBytes(const void* p, size_t n, uint64_t seed);
inline uint64_t HashDouble(uint64_t h, double d) {
  uint64_t b;
  static_assert(sizeof(b) == sizeof(d), "");
  __builtin_memcpy(&b, &d, 8);
  return HashCombine(h, b);
}

class KroneckerProductImpl final : public LinearMapImpl {  // linear/kronecker_product_impl.h
 public:
  KroneckerProductImpl(LinearMap A, LinearMap B)
      : LinearMapImpl(KRONECKER_PRODUCT), A_(std::move(A)), B_(std::move(B)) {}
  int64_t m() const override { return A_.impl().m() * B_.impl().m(); }
  int64_t n() const override { return A_.impl().n() * B_.impl().n(); }
  std::string DebugString() const override;
  std::shared_ptr<const LinearMapImpl> Transpose() const override;
  std::shared_ptr<const LinearMapImpl> Inverse() const override;
  bool Equals(const LinearMapImpl& other) const override;
  void Apply(double alpha, const DVec& x, double beta, const DVec& y) const override;
  std::vector<double> AsDenseHost() const override;
  const LinearMap& A() const { return A_; }
  const LinearMap& B() const { return B_; }

 private:
  LinearMap A_, B_;
};

// Dense device form of any map (setup-time fallback for rarely used type pairs).
std::shared_ptr<const DenseMatrixImpl> ToDense(const LinearMapImpl& A, DType dt);
DType MapDType(const LinearMapImpl& A, DType fallback);

}  // namespace eps

#include "block.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <sstream>
#include <unordered_set>

#include "comm.h"
#include "kernels.h"

namespace eps {

// ---- BlockVector (reference vector/block_vector.cc) -------------------------------------------------

const DVec& BlockVector::operator()(const std::string& key) const {
  auto it = data_.find(key);
  EPS_CHECK_MSG(it != data_.end(), key << " not in BlockVector");
  return it->second;
}

DVec& BlockVector::Mutable(const std::string& key) {
  auto it = data_.find(key);
  EPS_CHECK_MSG(it != data_.end(), key << " not in BlockVector");
  DVec& v = it->second;
  if (v.buf && (v.buf.use_count() > 1 || !v.buf->owned)) v = v.Clone();
  return v;
}

std::set<std::string> BlockVector::keys() const {
  std::set<std::string> r;
  for (const auto& kv : data_) r.insert(kv.first);
  return r;
}

int64_t BlockVector::n() const {
  int64_t n = 0;
  for (const auto& kv : data_) n += kv.second.n;
  return n;
}

void BlockVector::InsertOrAdd(const std::string& key, const DVec& value, double alpha) {
  auto it = data_.find(key);
  if (it == data_.end()) {  // block_vector.cc:44-49
    if (alpha == 1.0) {
      data_[key] = value;  // shared, copy-on-write protects it
    } else {
      DVec v = DVec::Empty(value.n, value.dt);
      k::Axpby(v, alpha, value, 0.0);
      data_[key] = v;
    }
    return;
  }
  EPS_CHECK_MSG(it->second.n == value.n, "block '" << key << "' has " << it->second.n
                                                   << " entries, adding " << value.n);
  // `value` may share the buffer of this block; Mutable() then clones, value stays valid
  DVec val = value;
  k::Axpby(Mutable(key), alpha, val, 1.0);
}

void BlockVector::InsertOrAddApply(const std::string& key, const LinearMapImpl& A, const DVec& x,
                                   double alpha) {
  auto it = data_.find(key);
  if (it == data_.end()) {
    DVec y = DVec::Empty(A.m(), x.dt);
    A.Apply(alpha, x, 0.0, y);
    data_[key] = y;
    return;
  }
  DVec xv = x;
  A.Apply(alpha, xv, 1.0, Mutable(key));
}

BlockVector& BlockVector::operator+=(const BlockVector& rhs) {
  for (const auto& kv : rhs.data_) InsertOrAdd(kv.first, kv.second, 1.0);
  return *this;
}

BlockVector& BlockVector::operator-=(const BlockVector& rhs) {
  for (const auto& kv : rhs.data_) InsertOrAdd(kv.first, kv.second, -1.0);
  return *this;
}

BlockVector& BlockVector::operator*=(double alpha) {
  for (auto& kv : data_) {
    DVec& v = Mutable(kv.first);
    k::Axpby(v, alpha, v, 0.0);
  }
  return *this;
}

BlockVector BlockVector::Select(const std::set<std::string>& keys) const {
  BlockVector r;
  for (const auto& key : keys) {
    auto it = data_.find(key);
    if (it != data_.end()) r.data_.insert(*it);
  }
  return r;
}

namespace {
inline double* SlotFor(Runtime& rt, int s, const std::string& key) {
  const ShardSpec& sh = ShardSpec::Get();
  return (sh.active() && sh.IsSharded(key)) ? rt.ShardSlotPtr(s) : rt.SlotPtr(s);
}
}  // namespace

int BlockVector::NormSqAsync() const {
  // slots are zeroed by ResetSlots; sharded blocks add into the slot's sharded half, which
  // FetchSlots all-reduces
  Runtime& rt = Runtime::Get();
  int s = rt.NewSlot();
  for (const auto& kv : data_) {
    if (kv.second.n == 0) continue;
    k::SumSq(kv.second, SlotFor(rt, s, kv.first), true);
  }
  return s;
}

double BlockVector::norm() const {
  Runtime& rt = Runtime::Get();
  rt.ResetSlots();
  int s = NormSqAsync();
  rt.FetchSlots();
  return std::sqrt(rt.SlotValue(s));
}

int DiffNormSqAsync(const BlockVector& a, const BlockVector& b) {
  Runtime& rt = Runtime::Get();
  int s = rt.NewSlot();
  std::set<std::string> keys = a.keys();
  for (const auto& key : b.keys()) keys.insert(key);
  for (const auto& key : keys) {
    double* slot = SlotFor(rt, s, key);
    if (a.has_key(key) && b.has_key(key)) {
      if (a(key).n == 0) continue;
      k::SumSqDiff(a(key), b(key), slot, true);
    } else {
      const DVec& v = a.has_key(key) ? a(key) : b(key);
      if (v.n == 0) continue;
      k::SumSq(v, slot, true);
    }
  }
  return s;
}

BlockVector operator+(BlockVector lhs, const BlockVector& rhs) {
  lhs += rhs;
  return lhs;
}
BlockVector operator-(BlockVector lhs, const BlockVector& rhs) {
  lhs -= rhs;
  return lhs;
}
BlockVector operator*(double alpha, BlockVector x) {
  x *= alpha;
  return x;
}

// ---- BlockMatrix (reference vector/block_matrix.cc) -------------------------------------------------

const LinearMap& BlockMatrix::operator()(const std::string& row, const std::string& col) const {
  auto c = data_.find(col);
  EPS_CHECK_MSG(c != data_.end(), "column: " << col << " not found");
  auto r = c->second.find(row);
  EPS_CHECK_MSG(r != c->second.end(), "row: " << row << " not found");
  return r->second;
}

bool BlockMatrix::has_key(const std::string& row, const std::string& col) const {
  auto c = data_.find(col);
  if (c == data_.end()) return false;
  return c->second.find(row) != c->second.end();
}

BlockMatrix BlockMatrix::Transpose() const {
  BlockMatrix t;
  for (const auto& c : data_)
    for (const auto& r : c.second) t.InsertOrAdd(c.first, r.first, r.second.Transpose());
  return t;
}

BlockMatrix BlockMatrix::Inverse() const {
  EPS_CHECK_MSG(m() == n(), "Inverting non square matrix");
  std::set<std::string> seen;
  for (const auto& c : data_) {
    EPS_CHECK_MSG(c.second.size() == 1, "Unable to invert matrix\n" << DebugString());
    const std::string& row = c.second.begin()->first;
    EPS_CHECK_MSG(seen.insert(row).second, "Unable to invert matrix\n" << DebugString());
  }
  BlockMatrix inv;
  for (const auto& c : data_) {
    const std::string& row = c.second.begin()->first;
    inv.InsertOrAdd(c.first, row, c.second.begin()->second.Inverse());
  }
  return inv;
}

BlockMatrix BlockMatrix::LeftIdentity() const {
  BlockMatrix C;
  for (const auto& c : data_)
    for (const auto& r : c.second)
      if (C.data_.find(r.first) == C.data_.end())
        C.InsertOrAdd(r.first, r.first, LinearMap::Identity(r.second.impl().m()));
  return C;
}

BlockMatrix BlockMatrix::RightIdentity() const {
  BlockMatrix C;
  for (const auto& c : data_) {
    EPS_CHECK(!c.second.empty());
    C.InsertOrAdd(c.first, c.first, LinearMap::Identity(c.second.begin()->second.impl().n()));
  }
  return C;
}

namespace {
// Sum over ranks of a partial product block.  Dense: the buffer.  Kronecker with one scalar
// factor (kron(I_k, X_g^T X_g), the multi-response Gram): the dense factor - the scalar factor is
// the same on every rank.
LinearMap AllReduceMap(const LinearMap& P, const std::string& r, const std::string& k,
                       const std::string& c) {
  Comm* comm = Runtime::Get().comm();
  if (P.impl().type() == DENSE_MATRIX) {
    const auto& D = static_cast<const DenseMatrixImpl&>(P.impl());
    DVec buf = D.Materialize(true);
    comm->AllReduceSum(buf);
    return LinearMap::Dense(buf, D.m(), D.n());
  }
  if (P.impl().type() == SPARSE_MATRIX) {
    // per-rank sparsity patterns differ: the sum is formed densely (the block is replicated,
    // hence small next to the sharded data)
    auto D = ToDense(P.impl(), MapDType(P.impl(), CurrentDType()));
    DVec buf = D->data();
    comm->AllReduceSum(buf);
    return LinearMap::Dense(buf, D->m(), D->n());
  }
  if (P.impl().type() == SCALAR_MATRIX) {
    // consensus rows (x_g - z = 0 on every rank): sum of the per-rank scalars
    double a = static_cast<const ScalarMatrixImpl&>(P.impl()).alpha();
    DVec d = DVec::FromHost(&a, 1, F64);
    comm->AllReduceSum(d);
    return LinearMap::Scalar(d.ToHost()[0], P.impl().n());
  }
  if (P.impl().type() == DIAGONAL_MATRIX) {
    const auto& D = static_cast<const DiagonalMatrixImpl&>(P.impl());
    DVec buf = DVec::FromHost(D.diagonal().data(), static_cast<int64_t>(D.diagonal().size()), F64);
    comm->AllReduceSum(buf);
    return LinearMap::Diagonal(buf.ToHost(), D.dtype());
  }
  if (P.impl().type() == KRONECKER_PRODUCT) {
    const auto& K = static_cast<const KroneckerProductImpl&>(P.impl());
    const bool a_s = K.A().impl().type() == SCALAR_MATRIX, b_s = K.B().impl().type() == SCALAR_MATRIX;
    if (a_s && K.B().impl().type() == DENSE_MATRIX) return LinearMap::Kronecker(K.A(), AllReduceMap(K.B(), r, k, c));
    if (b_s && K.A().impl().type() == DENSE_MATRIX) return LinearMap::Kronecker(AllReduceMap(K.A(), r, k, c), K.B());
  }
  EPS_FATAL("sharded contraction (" << r << "," << k << ")*(" << k << "," << c
                                    << ") has a type that cannot be summed over ranks: "
                                    << P.impl().DebugString());
}
}  // namespace

BlockMatrix operator*(const BlockMatrix& A, const BlockMatrix& B) {  // block_matrix.cc:102-127
  BlockMatrix C;
  for (const auto& bcol : B.data_) {
    for (const auto& b : bcol.second) {
      auto acol = A.data_.find(b.first);
      if (acol == A.data_.end()) continue;
      for (const auto& a : acol->second) {
        LinearMap P = a.second * b.second;
        // contraction over a sharded key into a replicated block: partial product, sum it
        const ShardSpec& sh = ShardSpec::Get();
        if (sh.active() && sh.IsSharded(b.first) && !sh.IsSharded(a.first) &&
            !sh.IsSharded(bcol.first)) {
          P = AllReduceMap(P, a.first, b.first, bcol.first);
        }
        C.InsertOrAdd(a.first, bcol.first, P);
      }
    }
  }
  return C;
}

BlockMatrix operator+(const BlockMatrix& A, const BlockMatrix& B) {
  BlockMatrix C = A;
  for (const auto& c : B.data_)
    for (const auto& r : c.second) C.InsertOrAdd(r.first, c.first, r.second);
  return C;
}

BlockMatrix operator-(const BlockMatrix& A, const BlockMatrix& B) { return A + (-1.0) * B; }

BlockMatrix operator*(double alpha, const BlockMatrix& A) {
  BlockMatrix C;
  for (const auto& c : A.data_)
    for (const auto& r : c.second) C.InsertOrAdd(r.first, c.first, alpha * r.second);
  return C;
}

BlockVector operator*(const BlockMatrix& A, const BlockVector& x) {  // block_matrix.cc:155-168
  const ShardSpec& sh = ShardSpec::Get();
  if (!sh.active()) {
    BlockVector y;
    for (const auto& xk : x.data()) {
      auto col = A.data_.find(xk.first);
      if (col == A.data_.end()) continue;
      for (const auto& blk : col->second)
        y.InsertOrAddApply(blk.first, blk.second.impl(), xk.second);
    }
    return y;
  }
  // Sharded solve: contributions of sharded columns to replicated rows are partial sums over
  // this rank's slice.  Gather them apart, all-reduce once per such row, then add the rest.
  BlockVector y, partial;
  for (const auto& xk : x.data()) {
    auto col = A.data_.find(xk.first);
    if (col == A.data_.end()) continue;
    const bool src_sharded = sh.IsSharded(xk.first);
    for (const auto& blk : col->second) {
      if (src_sharded && !sh.IsSharded(blk.first))
        partial.InsertOrAddApply(blk.first, blk.second.impl(), xk.second);
      else
        y.InsertOrAddApply(blk.first, blk.second.impl(), xk.second);
    }
  }
  for (const auto& kv : partial.data()) {
    DVec p = partial.Mutable(kv.first);
    Runtime::Get().comm()->AllReduceSum(p);
    y.InsertOrAdd(kv.first, p);
  }
  return y;
}

void BlockMatrix::InsertOrAdd(const std::string& row, const std::string& col, LinearMap value) {
  auto res = data_[col].insert(std::make_pair(row, value));
  if (!res.second) res.first->second += value;
}

int64_t BlockMatrix::m() const {
  std::unordered_set<std::string> seen;
  int64_t m = 0;
  for (const auto& c : data_)
    for (const auto& r : c.second)
      if (seen.insert(r.first).second) m += r.second.impl().m();
  return m;
}

int64_t BlockMatrix::n() const {
  int64_t n = 0;
  for (const auto& c : data_) n += c.second.begin()->second.impl().n();
  return n;
}

std::set<std::string> BlockMatrix::row_keys() const {
  std::set<std::string> r;
  for (const auto& c : data_)
    for (const auto& b : c.second) r.insert(b.first);
  return r;
}

std::set<std::string> BlockMatrix::col_keys() const {
  std::set<std::string> r;
  for (const auto& c : data_) r.insert(c.first);
  return r;
}

const std::map<std::string, LinearMap>& BlockMatrix::col(const std::string& col_key) const {
  auto it = data_.find(col_key);
  EPS_CHECK_MSG(it != data_.end(), "column " << col_key << " not found");
  return it->second;
}

std::string BlockMatrix::DebugString() const {
  std::ostringstream os;
  os << "block matrix " << m() << " x " << n();
  for (const auto& c : data_)
    for (const auto& r : c.second)
      os << "\n(" << r.first << ", " << c.first << ")\n" << r.second.impl().DebugString();
  return os.str();
}

void BlockMatrix::Remove(const std::string& row, const std::string& col) {
  auto c = data_.find(col);
  EPS_CHECK(c != data_.end());
  auto r = c->second.find(row);
  EPS_CHECK(r != c->second.end());
  c->second.erase(r);
  if (c->second.empty()) data_.erase(c);
}

// ---- BlockCholesky (reference vector/block_cholesky.cc) ---------------------------------------------

static const uint64_t kFillMax = std::numeric_limits<uint64_t>::max();
static const uint64_t kFillForbidden = kFillMax - 1;  // sharded solve: would couple ranks

// Upper bound on the non-zeros that eliminating column k adds to the remaining matrix: the
// update is V D^-1 V^T with V the off-diagonal part of column k, so every ordered pair (i, j)
// of its row keys contributes a block whose type follows from the multiply table and whose
// size is rows(i) x rows(j) (the model of reference block_cholesky.cc:11-48; its own test
// expects 4 and 25 for the two keys of block_cholesky_test.cc:69-70).
uint64_t ComputeFill(const BlockMatrix& A, const std::string& k) {
  struct Neighbour {
    const std::string* key;
    ImplType type;       // type of A(i, k)
    ImplType times_pivot;  // type of A(i, k) * A(k, k)
    uint64_t rows;
  };
  const auto& column = A.col(k);
  const auto pivot = column.find(k);
  if (pivot == column.end()) return kFillMax;  // nothing to divide by
  const ImplType pivot_type = pivot->second.impl().type();
  const ShardSpec& sh = ShardSpec::Get();
  const bool sharded_solve = sh.active();
  std::vector<Neighbour> nb;
  nb.reserve(column.size());
  bool any_sharded = false;
  for (const auto& entry : column) {
    if (entry.first == k) continue;
    Neighbour x;
    x.key = &entry.first;
    x.type = entry.second.impl().type();
    x.times_pivot = ComputeType(x.type, pivot_type);
    x.rows = static_cast<uint64_t>(entry.second.impl().m());
    if (sharded_solve && sh.IsSharded(entry.first)) {
      any_sharded = true;
      if (sh.global_dim(entry.first)) x.rows = sh.global_dim(entry.first);  // sizes every rank agrees on
    }
    nb.push_back(x);
  }
  // Sharded solve: each rank holds its own slice of every sharded key, and all maps between
  // sharded keys are block-local by construction of the per-rank problem.  Eliminating a
  // REPLICATED key k would connect the slices of two sharded keys through it (a coupling
  // across ranks that no rank can form locally, e.g. A_g^T A_h), so that order is refused;
  // it also keeps the local slice sizes from steering the order away from the single-GPU one.
  if (sharded_solve && any_sharded && !sh.IsSharded(k)) return kFillForbidden;
  uint64_t total = 0;
  for (const Neighbour& left : nb)
    for (const Neighbour& right : nb)
      total += Nonzeros(ComputeType(left.times_pivot, right.type), left.rows, right.rows);
  return total;
}

// Greedy minimum-fill pivot: the first key, in column order, with the smallest bound (the
// tie-break matters: it decides which of two equal candidates the solver eliminates first).
std::string NextKey(const BlockMatrix& A) {
  const std::set<std::string> key_set = A.col_keys();  // sorted: the order of the reference's loop
  const std::vector<std::string> keys(key_set.begin(), key_set.end());
  std::vector<uint64_t> bound(keys.size());
  for (size_t c = 0; c < keys.size(); ++c) bound[c] = ComputeFill(A, keys[c]);
  const auto best = std::min_element(bound.begin(), bound.end());
  EPS_CHECK_MSG(best != bound.end() && *best != kFillMax, "block LDL: no key with a diagonal block\n"
                                                              << A.DebugString());
  EPS_CHECK_MSG(*best != kFillForbidden,
                "sharded solve: every elimination order couples the sharded keys (shard by the "
                "other dimension)\n" << A.DebugString());
  return keys[static_cast<size_t>(best - bound.begin())];
}

// Takes row and column `key` out of the symmetric matrix; returns the column without its
// diagonal block (the V of the elimination step).
static BlockMatrix DetachKey(BlockMatrix* A, const std::string& key) {
  BlockMatrix V;
  std::vector<std::string> rows;
  for (const auto& entry : A->col(key)) {
    rows.push_back(entry.first);
    if (entry.first != key) V(entry.first, key) = entry.second;
  }
  for (const std::string& r : rows) {
    A->Remove(r, key);
    if (r != key) A->Remove(key, r);
  }
  return V;
}

namespace {

// One triangular substitution over `order` (forward: elimination order, backward: reversed).
// In a sharded solve a sharded source block contributes a partial sum to a replicated row;
// those are kept apart and all-reduced once, right before the row is itself used as a source.
template <class It>
BlockVector Substitute(const BlockMatrix& L, It begin, It end, BlockVector b) {
  const ShardSpec& sh = ShardSpec::Get();
  const bool sharded_solve = sh.active();
  BlockVector partial;
  for (It j = begin; j != end; ++j) {
    if (sharded_solve && partial.has_key(*j)) {
      DVec p = partial.Mutable(*j);
      Runtime::Get().comm()->AllReduceSum(p);
      b.InsertOrAdd(*j, p);
    }
    if (!b.has_key(*j)) continue;
    DVec bj = b(*j);
    const bool src_sharded = sharded_solve && sh.IsSharded(*j);
    for (It i = j + 1; i != end; ++i) {
      if (!L.has_key(*i, *j)) continue;
      if (src_sharded && !sh.IsSharded(*i))
        partial.InsertOrAddApply(*i, L(*i, *j).impl(), bj, -1.0);
      else
        b.InsertOrAddApply(*i, L(*i, *j).impl(), bj, -1.0);
    }
  }
  return b;
}

}  // namespace

BlockVector ForwardSub(const BlockMatrix& L, const std::vector<std::string>& keys, BlockVector b) {
  return Substitute(L, keys.begin(), keys.end(), std::move(b));  // :86-100
}

BlockVector BackSub(const BlockMatrix& LT, const std::vector<std::string>& keys, BlockVector b) {
  return Substitute(LT, keys.rbegin(), keys.rend(), std::move(b));  // :103-117
}

namespace {
// One small all-reduce: the global size of every sharded key of A (same key order on all ranks).
void RegisterGlobalDims(const BlockMatrix& A) {
  ShardSpec& sh = ShardSpec::Get();
  if (!sh.active()) return;
  std::map<std::string, double> dims;
  for (const auto& c : A.data()) {
    if (sh.IsSharded(c.first)) dims[c.first] = c.second.begin()->second.impl().n();
    for (const auto& r : c.second)
      if (sh.IsSharded(r.first)) dims[r.first] = r.second.impl().m();
  }
  if (dims.empty()) return;
  std::vector<double> loc;
  for (const auto& kv : dims) loc.push_back(kv.second);
  DVec d = DVec::FromHost(loc.data(), static_cast<int64_t>(loc.size()), F64);
  Runtime::Get().comm()->AllReduceSum(d);
  std::vector<double> g = d.ToHost();
  size_t i = 0;
  for (const auto& kv : dims) sh.set_global_dim(kv.first, static_cast<int64_t>(g[i++] + 0.5));
}
}  // namespace

BlockSolveStats& BlockSolveStats::Get() {
  static BlockSolveStats s;
  return s;
}

namespace {
// Number of refinement steps for a block solve whose worst pivot block has kappa_1 = cond, in
// fp32.  The explicit-inverse elimination is the normal-equations route: its forward error is
// ~ cond * 6e-8 per solve although the KKT system itself is far better conditioned (a
// projection has norm 1).  One step of refinement against the ORIGINAL blocks multiplies the
// error by ~ cond * 6e-8 again, down to cond(KKT) * 6e-8.  EPSILON_HIP_REFINE = <steps> forces a
// count (0 = the unrefined reference sequence of operations), "auto" / unset = by estimate.
int RefineStepsFor(double cond) {
  if (const char* e = std::getenv("EPSILON_HIP_REFINE")) {
    if (e[0] >= '0' && e[0] <= '9') return std::atoi(e);
  }
  if (!(cond > 1e3)) return 0;  // unrefined error <= ~1e-4, the fp32 mode's stated tolerance
  if (cond <= 3e4) return 1;
  if (cond <= 3e6) return 2;
  return 3;
}
}  // namespace

// Block LDL^T by successive elimination, A = L D L^T with unit block lower triangular L stored
// as L - I (reference block_cholesky.cc:119-133): pick the pivot key, invert its diagonal
// block, record the scaled column, subtract the Schur update from what is left.
void BlockCholesky::Compute(BlockMatrix A) {
  RegisterGlobalDims(A);
  const ShardSpec& sh = ShardSpec::Get();
  A_ = A;
  cond_ = 1.0;
  refine_steps_ = 0;
  bool any_f32 = false;
  for (size_t left = A.col_keys().size(); left > 0; --left) {
    const std::string pivot = NextKey(A);
    const LinearMap& block = A(pivot, pivot);
    // a replicated block of a sharded solve is the same matrix on every rank (its sharded
    // contributions were all-reduced): split the work of inverting a large dense one
    const bool split_over_ranks = sh.active() && (!sh.keys().empty() || !sh.local().empty()) &&
                                  !sh.IsSharded(pivot) && block.impl().type() == DENSE_MATRIX;
    BlockMatrix Dinv;
    Dinv(pivot, pivot) =
        split_over_ranks ? LinearMap(static_cast<const DenseMatrixImpl&>(block.impl()).InverseDistributed())
                         : block.Inverse();
    if (MapDType(block.impl(), CurrentDType()) == F32) {
      any_f32 = true;
      const double c = ConditionEstimate(block, Dinv(pivot, pivot));
      if (c > cond_ && std::isfinite(c)) cond_ = c;
    }
    const BlockMatrix V = DetachKey(&A, pivot);
    const BlockMatrix column = V * Dinv;
    A = A - column * V.Transpose();
    L_ = L_ + column;
    D_inv_ = D_inv_ + Dinv;
    p_.push_back(pivot);
  }
  LT_ = L_.Transpose();
  if (any_f32) {
    // sharded blocks differ between ranks, the step count must not (the residual's products
    // contain collectives)
    if (sh.active()) cond_ = Runtime::Get().comm()->AllReduceMaxHost(cond_);
    refine_steps_ = RefineStepsFor(cond_);
  }
  if (refine_steps_ == 0) A_ = BlockMatrix();
  BlockSolveStats& st = BlockSolveStats::Get();
  st.max_condition = std::max(st.max_condition, cond_);
  st.max_refine_steps = std::max(st.max_refine_steps, refine_steps_);
}

BlockVector BlockCholesky::SolveOnce(const BlockVector& b) const {  // :135-137
  return BackSub(LT_, p_, D_inv_ * ForwardSub(L_, p_, b));
}

BlockVector BlockCholesky::Solve(const BlockVector& b) const {
  BlockVector x = SolveOnce(b);
  // fp32 with ill-conditioned pivot blocks only: x += solve(b - A x) against the blocks as given
  for (int s = 0; s < refine_steps_; ++s) {
    BlockVector r = b - A_ * x;
    x += SolveOnce(r);
  }
  return x;
}

}  // namespace eps

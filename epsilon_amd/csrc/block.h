// Keyed block containers and the block LDL^T used by every KKT-based prox operator.
//
// Same interface as the reference (src/epsilon/vector/block_vector.h:13-75,
// block_matrix.h:33-72, block_cholesky.h:8-16): string keys, std::map iteration order
// (lexicographic, which fixes the tie-break of the min-fill ordering), InsertOrAdd semantics.
// Blocks live in HBM; BlockVector copies are shallow with copy-on-write, so the reference's
// per-iteration deep copies (`y_prev_ = y_`, prox_admm.cc:135) cost nothing here.
#pragma once

#include <map>
#include <set>
#include <string>
#include <vector>

#include "device.h"
#include "linear_map.h"

namespace eps {

class BlockMatrix;

class BlockVector {
 public:
  // Read access (throws if the key is absent, like the reference's LOG(FATAL)).
  const DVec& operator()(const std::string& key) const;
  // Replace / create a block (takes the buffer as is).
  void Set(const std::string& key, DVec v) { data_[key] = std::move(v); }
  // Block to be modified in place: un-shares the buffer first (copy-on-write).
  DVec& Mutable(const std::string& key);

  bool has_key(const std::string& key) const { return data_.count(key) != 0; }
  const std::map<std::string, DVec>& data() const { return data_; }
  std::set<std::string> keys() const;
  int64_t n() const;

  // this[key] (+)= alpha * value   (block created as alpha*value when absent)
  void InsertOrAdd(const std::string& key, const DVec& value, double alpha = 1.0);
  // this[key] (+)= alpha * A * x, without a temporary
  void InsertOrAddApply(const std::string& key, const LinearMapImpl& A, const DVec& x,
                        double alpha = 1.0);

  BlockVector& operator+=(const BlockVector& rhs);
  BlockVector& operator-=(const BlockVector& rhs);
  BlockVector& operator*=(double alpha);
  BlockVector Select(const std::set<std::string>& keys) const;

  // ||.||^2 into a device slot (no host sync); see Runtime::FetchSlots.
  int NormSqAsync() const;
  // Synchronising convenience (setup / tests).
  double norm() const;

 private:
  std::map<std::string, DVec> data_;
};

BlockVector operator+(BlockVector lhs, const BlockVector& rhs);
BlockVector operator-(BlockVector lhs, const BlockVector& rhs);
BlockVector operator*(double alpha, BlockVector x);
// ||a - b||^2 over the union of keys into a device slot.
int DiffNormSqAsync(const BlockVector& a, const BlockVector& b);

class BlockMatrix {
 public:
  LinearMap& operator()(const std::string& row, const std::string& col) { return data_[col][row]; }
  const LinearMap& operator()(const std::string& row, const std::string& col) const;
  bool has_key(const std::string& row, const std::string& col) const;

  int64_t m() const;
  int64_t n() const;
  const std::map<std::string, LinearMap>& col(const std::string& col_key) const;
  const std::map<std::string, std::map<std::string, LinearMap>>& data() const { return data_; }
  std::set<std::string> col_keys() const;
  std::set<std::string> row_keys() const;

  BlockMatrix Transpose() const;
  BlockMatrix Inverse() const;       // block diagonal only (block_matrix.cc:9-27,66-74)
  BlockMatrix LeftIdentity() const;  // block_matrix.cc:76-88
  BlockMatrix RightIdentity() const;

  void InsertOrAdd(const std::string& row, const std::string& col, LinearMap value);
  void Remove(const std::string& row, const std::string& col);
  std::string DebugString() const;

 private:
  std::map<std::string, std::map<std::string, LinearMap>> data_;  // col -> row -> value
  friend BlockMatrix operator*(const BlockMatrix& A, const BlockMatrix& B);
  friend BlockMatrix operator*(double alpha, const BlockMatrix& A);
  friend BlockMatrix operator+(const BlockMatrix& A, const BlockMatrix& B);
  friend BlockVector operator*(const BlockMatrix& A, const BlockVector& x);
};

BlockMatrix operator*(const BlockMatrix& lhs, const BlockMatrix& rhs);
BlockMatrix operator+(const BlockMatrix& lhs, const BlockMatrix& rhs);
BlockMatrix operator-(const BlockMatrix& lhs, const BlockMatrix& rhs);
BlockMatrix operator*(double alpha, const BlockMatrix& A);
BlockVector operator*(const BlockMatrix& lhs, const BlockVector& rhs);

struct AffineOperator {  // reference affine/affine.h:15-18
  BlockMatrix A;
  BlockVector b;
};

// reference vector/block_cholesky.cc
uint64_t ComputeFill(const BlockMatrix& A, const std::string& k);
std::string NextKey(const BlockMatrix& A);
BlockVector ForwardSub(const BlockMatrix& L, const std::vector<std::string>& keys, BlockVector b);
BlockVector BackSub(const BlockMatrix& LT, const std::vector<std::string>& keys, BlockVector b);

class BlockCholesky {
 public:
  void Compute(BlockMatrix A);
  BlockVector Solve(const BlockVector& b) const;
  const std::vector<std::string>& order() const { return p_; }
  const BlockMatrix& L() const { return L_; }
  const BlockMatrix& D_inv() const { return D_inv_; }
  // fp32 mode only (no reference counterpart - the reference factors in fp64): the largest
  // kappa_1 over the pivot blocks, and the number of iterative-refinement steps Solve() adds
  // because of it (0 for well-conditioned systems: Solve is then exactly the reference's
  // back-sub(D^-1 forward-sub)).  See Compute().
  double condition_estimate() const { return cond_; }
  int refine_steps() const { return refine_steps_; }

 private:
  BlockVector SolveOnce(const BlockVector& b) const;
  std::vector<std::string> p_;
  BlockMatrix D_inv_, L_, LT_;
  BlockMatrix A_;  // the matrix as given (shares every block's buffer): refinement residuals
  double cond_ = 1.0;
  int refine_steps_ = 0;
};

// Process-wide record of the last factorisations' condition estimates (tests / diagnostics):
// the largest estimate and refinement step count since the last reset.
struct BlockSolveStats {
  double max_condition = 0;
  int max_refine_steps = 0;
  static BlockSolveStats& Get();
  void Reset() { max_condition = 0; max_refine_steps = 0; }
};

}  // namespace eps

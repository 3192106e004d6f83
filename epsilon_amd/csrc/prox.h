// Proximal-operator plugin interface and registry.
//
// Same plugin point as the reference (src/epsilon/prox/prox.h:11-77): a ProxOperator is
// created by (ProxFunction::Type, epigraph), initialised once with the function's affine
// argument H and the constraint map A, then applied every sweep:
//     Apply(v) = argmin_x f(H x + g) + 1/2 ||A x - v||^2 .
// Operators register themselves with REGISTER_PROX_OPERATOR / REGISTER_EPIGRAPH_OPERATOR.
#pragma once

#include <functional>
#include <memory>
#include <string>

#include "affine.h"
#include "block.h"
#include "kernels.h"
#include "wire.h"

namespace eps {

class ProxOperatorArg {  // reference prox/prox.h:11-35
 public:
  ProxOperatorArg(const pb::ProxFunction& f, DataMap* data, const AffineOperator& affine_arg,
                  const AffineOperator& affine_constraint)
      : f_(f), data_(data), H_(affine_arg), A_(affine_constraint) {}
  const pb::ProxFunction& prox_function() const { return f_; }
  DataMap* data_map() const { return data_; }
  const AffineOperator& affine_arg() const { return H_; }
  const AffineOperator& affine_constraint() const { return A_; }

 private:
  const pb::ProxFunction& f_;
  DataMap* data_;
  const AffineOperator& H_;
  const AffineOperator& A_;
};

// What the sweep fuser (admm.cc) needs to know about an operator to replace its Apply by a
// fused kernel; an operator that cannot be described this way keeps the generic path.
struct LeastSquaresDesc {  // SumSquareProx after block elimination [constraint, variable, arg]
  std::string var_key, arg_key, constraint_key;
  std::shared_ptr<const DenseMatrixImpl> L_arg_var;  // L(arg, var): lazily scaled data matrix
  std::shared_ptr<const DenseMatrixImpl> Dinv_arg;   // cached explicit inverse (with its sign)
  DVec rhs_arg;                                      // constant part of the rhs on the arg row
};
struct ScaledZoneDesc {  // ScaledZoneProx with scalar H, A and uniform parameters
  std::string var_key, constraint_key;
  double Bs = 0, Cs = 0;  // v' = Bs*v ; x = Cs*x'
  double lam = 0, alpha = 1, beta = 1, M = 0;
  DVec alpha_vec, beta_vec;  // per-element alpha / beta (SUM_QUANTILE with data vectors); empty: uniform
};

class ProxOperator {  // reference prox/prox.h:37-43
 public:
  virtual ~ProxOperator() {}
  virtual void Init(const ProxOperatorArg& arg) {}
  virtual BlockVector Apply(const BlockVector& v) = 0;
  virtual bool DescribeLeastSquares(LeastSquaresDesc* d) const { return false; }
  virtual bool DescribeScaledZone(ScaledZoneDesc* d) const { return false; }
  // true: Apply is a fixed sequence of launches on the library's stream - no host
  // synchronisation, no decision on device data, no state carried from one call to the next -
  // so a sweep through this operator can be captured into a hipGraph and replayed (admm.cc).
  virtual bool CaptureSafe() const { return false; }
};

std::unique_ptr<ProxOperator> CreateProxOperator(int type, bool epigraph);
bool RegisterProxOperatorFactory(int type, bool epigraph,
                                 std::function<std::unique_ptr<ProxOperator>()> factory);

template <class T> bool RegisterProxOperator(int type, bool epigraph) {
  return RegisterProxOperatorFactory(type, epigraph,
                                     [] { return std::unique_ptr<ProxOperator>(new T); });
}

#define EPS_REGISTER_VAR(prefix, type, T) prefix##_##type##_##T
#define REGISTER_PROX_OPERATOR(type, T) \
  static bool EPS_REGISTER_VAR(prox, type, T) = ::eps::RegisterProxOperator<T>(pb::ProxFunction::type, false)
#define REGISTER_EPIGRAPH_OPERATOR(type, T) \
  static bool EPS_REGISTER_VAR(epi, type, T) = ::eps::RegisterProxOperator<T>(pb::ProxFunction::type, true)

// ---- VectorProx: prox with scalar / diagonal H and A reduced to a plain vector prox ------------
// reference prox/vector_prox.{h,cc}

// Slices of argument `arg` (n entries) the reference's axis loop visits (vector_prox.cc:150-177):
// columns for axis 0, rows for axis 1, the whole argument when the function has no axis.
k::Segs SegsOf(const pb::ProxFunction& f, int arg, int64_t n);

class VectorProxInput {
 public:
  double lambda() const;                       // scalar case only
  bool elementwise() const { return elementwise_; }
  const DVec& lambda_vec() const { return lambda_dev_; }
  const DVec& value_vec(int i) const;          // whole argument i (device)
  const pb::ProxFunction& prox_function() const { return f_; }

 private:
  friend class VectorProx;
  bool elementwise_ = false;
  double lambda_ = 0;
  std::vector<double> lambda_host_;
  DVec lambda_dev_;
  BlockVector v_;
  pb::ProxFunction f_;
};

class VectorProxOutput {
 public:
  void set_value(int i, DVec x);

 private:
  friend class VectorProx;
  BlockVector x_;
};

class VectorProx : public ProxOperator {
 public:
  void Init(const ProxOperatorArg& arg) override;
  BlockVector Apply(const BlockVector& v) override;

 protected:
  // Applied to whole arguments.  When the function has an axis (per-row / per-column
  // application, reference vector_prox.cc:150-177) the operator sees the full m x n argument
  // and handles the axis itself; elementwise operators are axis-agnostic.
  virtual void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) = 0;

  // For DescribeScaledZone: true iff B_, C_ are single scalar blocks, no offset, scalar lambda.
  bool ScalarForm(std::string* var_key, std::string* constraint_key, double* Bs, double* Cs,
                  double* lam) const;

 private:
  BlockMatrix B_, C_, D_;
  BlockVector g_;
  VectorProxInput input_;
  VectorProxOutput output_;
};

}  // namespace eps

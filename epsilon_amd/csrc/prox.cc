#include "prox.h"

#include <cmath>
#include <map>

#include "kernels.h"

namespace eps {

// ---- registry (reference prox/prox.cc:25-45, prox.h:51-77) -----------------------------------------

namespace {
using Factory = std::function<std::unique_ptr<ProxOperator>()>;
std::map<std::pair<int, bool>, Factory>* g_registry = nullptr;
}  // namespace

bool RegisterProxOperatorFactory(int type, bool epigraph, Factory factory) {
  if (g_registry == nullptr) g_registry = new std::map<std::pair<int, bool>, Factory>();
  (*g_registry)[std::make_pair(type, epigraph)] = std::move(factory);
  return true;
}

std::unique_ptr<ProxOperator> CreateProxOperator(int type, bool epigraph) {
  EPS_CHECK_MSG(g_registry != nullptr, "No registered operators");
  auto it = g_registry->find(std::make_pair(type, epigraph));
  if (it == g_registry->end()) {
    EPS_FATAL("No proximal operator for " << pb::ProxTypeName(type) << " (epigraph: " << epigraph
                                          << ")");
  }
  return it->second();
}

// ---- VectorProx (reference prox/vector_prox.cc) ------------------------------------------------------

namespace {

// What the argument map H and the constraint row A look like to an elementwise prox: the
// diagonals of H^T H and of H A^T A H^T, provided both are block diagonal with one and the same
// diagonal on every block (the test of reference vector_prox.cc:4-49).  `uniform` = every block
// is a scalar multiple of the identity; then a single number stands for each diagonal.
struct ArgScaling {
  bool uniform = false;
  std::vector<double> hth, haah;
};

// The common diagonal of a block-diagonal matrix whose blocks are scalar (scalar_only) or
// scalar / diagonal maps; false when a block sits off the diagonal, has another type, or the
// blocks disagree.
bool CommonDiagonal(const BlockMatrix& M, bool scalar_only, std::vector<double>* out) {
  out->clear();
  bool have = false;
  for (const auto& column : M.data()) {
    const auto& rows = column.second;
    if (rows.size() != 1 || rows.begin()->first != column.first) return false;
    const LinearMap& blk = rows.begin()->second;
    const ImplType ty = blk.impl().type();
    if (ty != SCALAR_MATRIX && (scalar_only || ty != DIAGONAL_MATRIX)) return false;
    std::vector<double> d = scalar_only ? std::vector<double>(1, GetScalar(blk)) : GetDiagonal(blk);
    if (have && d != *out) return false;
    if (!have) out->swap(d);
    have = true;
  }
  if (scalar_only && !have) out->assign(1, 0.0);  // (no blocks: the reference leaves the value unset)
  return true;
}

bool DescribeScaling(const BlockMatrix& H, const BlockMatrix& A, ArgScaling* sc) {
  const BlockMatrix Ht = H.Transpose();
  const BlockMatrix HtH = Ht * H, HAAH = H * A.Transpose() * A * Ht;
  for (const bool scalar_only : {true, false}) {
    if (CommonDiagonal(HtH, scalar_only, &sc->hth) && CommonDiagonal(HAAH, scalar_only, &sc->haah)) {
      sc->uniform = scalar_only;
      return true;
    }
  }
  return false;
}

}  // namespace

k::Segs SegsOf(const pb::ProxFunction& f, int arg, int64_t n) {
  k::Segs S;
  if (!f.has_axis) {
    S.count = 1;
    S.len = n;
    S.seg_stride = n;
    S.elem_stride = 1;
    return S;
  }
  EPS_CHECK_MSG(static_cast<int>(f.arg_size.size()) > arg && f.arg_size[arg].dim.size() == 2,
                "prox function with an axis needs arg_size for argument " << arg);
  EPS_CHECK_MSG(f.axis == 0 || f.axis == 1, "axis must be 0 or 1");
  const int64_t rows = f.arg_size[arg].dim[0], cols = f.arg_size[arg].dim[1];
  EPS_CHECK_MSG(rows * cols == n, "argument " << arg << " has " << n << " entries, arg_size says "
                                              << rows << " x " << cols);
  if (f.axis == 0) {  // slices are columns
    S.count = cols;
    S.len = rows;
    S.seg_stride = rows;
    S.elem_stride = 1;
  } else {  // slices are rows of the column-major argument
    S.count = rows;
    S.len = cols;
    S.seg_stride = 1;
    S.elem_stride = rows;
  }
  return S;
}

double VectorProxInput::lambda() const {
  EPS_CHECK_MSG(!elementwise_, "scalar lambda requested from an elementwise-scaled prox");
  return lambda_;
}

const DVec& VectorProxInput::value_vec(int i) const { return v_(affine::arg_key(i)); }

void VectorProxOutput::set_value(int i, DVec x) { x_.Set(affine::arg_key(i), std::move(x)); }

// With beta = diag(H^T H) and gamma = diag(H A^T A H^T) the update
//   argmin_x f(Hx + g) + 1/2 ||Ax - v||^2
// is an elementwise prox in the scaled variable: centre B v + g with B = H (beta / gamma) A^T,
// weight lambda = alpha beta^2 / gamma, and the result mapped back by C = (1 / beta) H^T
// (reference vector_prox.cc:51-138 derives the same three maps; uniform scalings keep them
// scalar so that the sweep can recognise the structure, ScalarForm below).
void VectorProx::Init(const ProxOperatorArg& arg) {
  ArgScaling sc;
  const BlockMatrix& H = arg.affine_arg().A;
  const BlockMatrix& A = arg.affine_constraint().A;
  if (!DescribeScaling(H, A, &sc)) EPS_FATAL("Affine transformation is not scalar or diagonal");
  const double alpha = arg.prox_function().alpha;
  const BlockMatrix Ht = H.Transpose(), At = A.Transpose();
  input_.lambda_host_.clear();
  input_.lambda_dev_ = DVec();
  D_ = BlockMatrix();
  if (sc.uniform) {
    const double beta = sc.hth[0], gamma = sc.haah[0];
    B_ = (beta / gamma) * H * At;
    C_ = (1 / beta) * Ht;
    input_.lambda_ = alpha * beta * beta / gamma;
    input_.elementwise_ = false;
    EPS_CHECK(input_.lambda_ >= 0);
  } else {
    // per-element weights.  An element that the constraint row does not see (gamma = 0) gets
    // weight 0 and is passed through from v by the extra map D (its prox centre is kept).
    const size_t n = sc.hth.size();
    EPS_CHECK(sc.haah.size() == n);
    std::vector<double> weight(n), centre(n), back(n), pass(n);
    for (size_t e = 0; e < n; ++e) {
      const bool seen = sc.haah[e] != 0;
      const double be = seen ? sc.hth[e] : 1.0, ga = seen ? sc.haah[e] : 1.0;
      weight[e] = seen ? alpha * be * be / ga : 0.0;
      centre[e] = be / ga;
      back[e] = 1 / be;
      pass[e] = seen ? 0.0 : 1.0;
    }
    const DType dt = arg.data_map()->dtype();
    auto on_columns = [&](const std::vector<double>& d) {  // the same diagonal on every column key of H
      BlockMatrix M;
      const LinearMap Dm = LinearMap::Diagonal(d, dt);
      for (const std::string& key : H.col_keys()) M(key, key) = Dm;
      return M;
    };
    B_ = H * on_columns(centre) * At;
    C_ = on_columns(back) * Ht;
    D_ = (At * A).Inverse() * on_columns(pass) * At;
    input_.lambda_host_ = weight;
    input_.lambda_dev_ = DVec::FromHost(weight.data(), static_cast<int64_t>(n), dt);
    input_.elementwise_ = true;
  }
  g_ = arg.affine_arg().b;
  input_.f_ = arg.prox_function();
}

BlockVector VectorProx::Apply(const BlockVector& v) {  // vector_prox.cc:140-183
  input_.v_ = B_ * v + g_;
  output_.x_ = BlockVector();
  ApplyVector(input_, &output_);
  BlockVector r = C_ * (output_.x_ - g_);
  if (!D_.data().empty()) r += D_ * v;
  return r;
}

bool VectorProx::ScalarForm(std::string* var_key, std::string* constraint_key, double* Bs,
                            double* Cs, double* lam) const {
  if (input_.elementwise_ || !D_.data().empty() || !g_.data().empty()) return false;
  if (B_.data().size() != 1 || B_.data().begin()->second.size() != 1) return false;
  if (C_.data().size() != 1 || C_.data().begin()->second.size() != 1) return false;
  const auto& bcol = *B_.data().begin();  // (arg:0, constraint)
  const auto& ccol = *C_.data().begin();  // (var, arg:0)
  const LinearMap& Bm = bcol.second.begin()->second;
  const LinearMap& Cm = ccol.second.begin()->second;
  if (Bm.impl().type() != SCALAR_MATRIX || Cm.impl().type() != SCALAR_MATRIX) return false;
  if (bcol.second.begin()->first != ccol.first) return false;  // both through arg:0
  *constraint_key = bcol.first;
  *var_key = ccol.second.begin()->first;
  *Bs = GetScalar(Bm);
  *Cs = GetScalar(Cm);
  *lam = input_.lambda_;
  return true;
}

// ---- ScaledZoneProx: NORM_1, SUM_DEADZONE, SUM_HINGE, SUM_QUANTILE -----------------------------------
// reference prox/scaled_zone.cc:6-121

namespace {

struct ZoneParam {  // a uniform value or a per-element vector
  double value = 0;
  DVec vec;
  bool is_vec = false;
};

ZoneParam PromoteParam(const DVec& x, int64_t n) {  // scaled_zone.cc:26-32
  ZoneParam p;
  if (x.n == n && n != 1) {
    p.vec = x;
    p.is_vec = true;
    return p;
  }
  EPS_CHECK_MSG(x.n == 1, "scaled zone parameter has " << x.n << " entries, expected " << n);
  p.value = x.ToHost()[0];
  return p;
}

// GetParams, reference prox/scaled_zone.cc:34-76
void InitZoneParams(const ProxOperatorArg& arg, ZoneParam* alpha, ZoneParam* beta, double* M,
                    int64_t* n) {
  const pb::ProxFunction& f = arg.prox_function();
  EPS_CHECK_MSG(!f.arg_size.empty() && f.arg_size[0].dim.size() == 2,
                "scaled zone prox needs arg_size");
  if (f.has_axis) *n = f.arg_size[0].dim[f.axis];
  else *n = static_cast<int64_t>(f.arg_size[0].dim[0]) * f.arg_size[0].dim[1];
  *alpha = ZoneParam();
  *beta = ZoneParam();
  alpha->value = 1;
  beta->value = 1;
  *M = 0;
  switch (f.prox_function_type) {  // scaled_zone.cc:46-72
    case pb::ProxFunction::NORM_1: break;
    case pb::ProxFunction::SUM_DEADZONE: *M = f.sz_m; break;
    case pb::ProxFunction::SUM_HINGE: beta->value = 0; break;
    case pb::ProxFunction::SUM_QUANTILE: {
      EPS_CHECK_MSG(f.sz_alpha_expr && f.sz_beta_expr, "SUM_QUANTILE needs alpha/beta exprs");
      BlockVector tmp;
      affine::BuildAffineOperator(*f.sz_alpha_expr, arg.data_map(), "alpha", nullptr, &tmp);
      affine::BuildAffineOperator(*f.sz_beta_expr, arg.data_map(), "beta", nullptr, &tmp);
      *alpha = PromoteParam(tmp("alpha"), *n);
      *beta = PromoteParam(tmp("beta"), *n);
      break;
    }
    default: EPS_FATAL("Unknown prox type: " << f.prox_function_type);
  }
}

class ScaledZoneProx final : public VectorProx {
 public:
  bool CaptureSafe() const override { return true; }

 public:
  void Init(const ProxOperatorArg& arg) override {
    VectorProx::Init(arg);
    const pb::ProxFunction& f = arg.prox_function();
    InitZoneParams(arg, &alpha_, &beta_, &M_, &n_);
    rows_ = f.arg_size[0].dim[0];
    has_axis_ = f.has_axis;
    axis_ = f.axis;
  }

 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    DVec x = DVec::Empty(v.n, v.dt);
    k::ScaledZoneArgs a;
    a.M = M_;
    a.C = 0;
    a.alpha = alpha_.value;
    a.beta = beta_.value;
    if (alpha_.is_vec) a.alpha_vec = &alpha_.vec;
    if (beta_.is_vec) a.beta_vec = &beta_.vec;
    if (input.elementwise()) a.lam_vec = &input.lambda_vec();
    else a.lam = input.lambda();
    const bool any_vec = alpha_.is_vec || beta_.is_vec || input.elementwise();
    if (has_axis_ && any_vec) {
      // per-slice parameters: slice = a column (axis 0) of the column-major argument
      EPS_CHECK_MSG(axis_ == 0, "per-element scaled-zone parameters along axis 1 not supported");
      a.period = rows_;
    }
    k::ScaledZone(x, v, a);
    output->set_value(0, x);
  }

 public:
  bool DescribeScaledZone(ScaledZoneDesc* d) const override {
    if (has_axis_) return false;
    if (!ScalarForm(&d->var_key, &d->constraint_key, &d->Bs, &d->Cs, &d->lam)) return false;
    d->alpha = alpha_.value;
    d->beta = beta_.value;
    d->alpha_vec = alpha_.is_vec ? alpha_.vec : DVec();
    d->beta_vec = beta_.is_vec ? beta_.vec : DVec();
    d->M = M_;
    return true;
  }

 private:
  ZoneParam alpha_, beta_;
  double M_ = 0;
  int64_t n_ = 0, rows_ = 0;
  bool has_axis_ = false;
  int axis_ = 0;
};
REGISTER_PROX_OPERATOR(NORM_1, ScaledZoneProx);
REGISTER_PROX_OPERATOR(SUM_DEADZONE, ScaledZoneProx);
REGISTER_PROX_OPERATOR(SUM_HINGE, ScaledZoneProx);
REGISTER_PROX_OPERATOR(SUM_QUANTILE, ScaledZoneProx);

// ---- ScaledZoneEpigraph: projection onto {(x, t): f(x) <= t} (reference scaled_zone.cc:123-284) --
// The multiplier lam >= 0 is the root of  sum_i w_i^2 max(k_i - lam, 0) = s + lam  (keys k_i,
// weights w_i as in ZoneEpigraphKeys).  The reference finds it with a randomised 3-way-partition
// selection on the host (`random()`, :198); here the active set {k_i > lam} is shrunk by
// Michelot's fixed-point iteration - a masked device reduction per step, finite and exact -
// which lands on the same lam = acc / (div + 1) (:276).

class ScaledZoneEpigraph final : public VectorProx {
 public:
  void Init(const ProxOperatorArg& arg) override {
    VectorProx::Init(arg);
    InitZoneParams(arg, &alpha_, &beta_, &M_, &n_);
  }

 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    const DVec& sv = input.value_vec(1);
    // one projection per row / column - or one short vector - is solved on chip by one launch
    // with no host round trip; a long single vector takes the device-wide reductions below
    if (input.prox_function().has_axis || v.n <= 65536) {
      DVec x = DVec::Empty(v.n, v.dt), t = DVec::Empty(sv.n, sv.dt);
      k::SegZoneEpigraph(x, t, v, sv, alpha_.value, beta_.value,
                         alpha_.is_vec ? &alpha_.vec : nullptr, beta_.is_vec ? &beta_.vec : nullptr,
                         M_, SegsOf(input.prox_function(), 0, v.n));
      output->set_value(0, x);
      output->set_value(1, t);
      return;
    }
    EPS_CHECK(sv.n == 1);
    Runtime& rt = Runtime::Get();
    const double s = sv.ToHost()[0];
    auto keys = rt.Alloc(v.n * sizeof(double)), w2 = rt.Alloc(v.n * sizeof(double));
    auto sums = rt.Alloc(4 * sizeof(double));
    double* dk = static_cast<double*>(keys->p);
    double* dw = static_cast<double*>(w2->p);
    double* ds = static_cast<double*>(sums->p);
    EPS_HIP(hipMemsetAsync(ds, 0, 4 * sizeof(double), rt.stream()));
    k::ZoneEpigraphKeys(v, alpha_.value, beta_.value, alpha_.is_vec ? &alpha_.vec : nullptr,
                        beta_.is_vec ? &beta_.vec : nullptr, M_, 0.0, dk, dw, ds + 3);
    double h[4];
    EPS_HIP(hipMemcpyAsync(h, ds, sizeof(h), hipMemcpyDeviceToHost, rt.stream()));
    rt.Sync();
    if (h[3] <= s) {  // already inside the epigraph (:191-195)
      output->set_value(0, v);
      output->set_value(1, sv);
      return;
    }
    double lam = 0, count = -1;
    for (int it = 0; it < 200; ++it) {
      EPS_HIP(hipMemsetAsync(ds, 0, 3 * sizeof(double), rt.stream()));
      k::ZoneEpigraphSums(v.n, dk, dw, lam, ds);
      EPS_HIP(hipMemcpyAsync(h, ds, 3 * sizeof(double), hipMemcpyDeviceToHost, rt.stream()));
      rt.Sync();
      // Newton step of the convex piecewise-linear phi(lam) = sum w^2 max(k - lam, 0) - s - lam
      // from the left: the iterates increase, the active sets are nested, and when the set
      // that produced lam is still the active set at lam, lam is the exact root.
      if (h[2] == count) break;
      count = h[2];
      lam = (h[0] - s) / (h[1] + 1.0);
    }
    DVec x = DVec::Empty(v.n, v.dt);
    k::ScaledZoneArgs a;
    a.M = M_;
    a.alpha = alpha_.value;
    a.beta = beta_.value;
    if (alpha_.is_vec) a.alpha_vec = &alpha_.vec;
    if (beta_.is_vec) a.beta_vec = &beta_.vec;
    a.lam = lam;
    k::ScaledZone(x, v, a);
    output->set_value(0, x);
    output->set_value(1, DVec::Full(1, s + lam, v.dt));
  }

 private:
  ZoneParam alpha_, beta_;
  double M_ = 0;
  int64_t n_ = 0;
};
REGISTER_EPIGRAPH_OPERATOR(NORM_1, ScaledZoneEpigraph);
REGISTER_EPIGRAPH_OPERATOR(SUM_DEADZONE, ScaledZoneEpigraph);
REGISTER_EPIGRAPH_OPERATOR(SUM_HINGE, ScaledZoneEpigraph);
REGISTER_EPIGRAPH_OPERATOR(SUM_QUANTILE, ScaledZoneEpigraph);

// ---- Norm2Prox (reference prox/norm_2.cc:4-19) ---------------------------------------------------------

class Norm2Prox final : public VectorProx {
 public:
  bool CaptureSafe() const override { return true; }

 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    if (input.prox_function().has_axis) {  // one group per row / column (group lasso)
      DVec x = DVec::Empty(v.n, v.dt);
      k::SegNorm2Shrink(x, v, input.lambda(), SegsOf(input.prox_function(), 0, v.n));
      output->set_value(0, x);
      return;
    }
    if (!normsq_) normsq_ = Runtime::Get().Alloc(sizeof(double));
    double* slot = static_cast<double*>(normsq_->p);
    k::SumSq(v, slot, false);
    DVec x = DVec::Empty(v.n, v.dt);
    k::Norm2Shrink(x, v, input.lambda(), slot);
    output->set_value(0, x);
  }

 private:
  std::shared_ptr<Buffer> normsq_;
};
REGISTER_PROX_OPERATOR(NORM_2, Norm2Prox);

// ---- NonNegativeProx (reference prox/non_negative.cc:3-11) ---------------------------------------------

class NonNegativeProx final : public VectorProx {
 public:
  bool CaptureSafe() const override { return true; }

 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    DVec x = DVec::Empty(v.n, v.dt);
    k::MaxZero(x, v);
    output->set_value(0, x);
  }
};
REGISTER_PROX_OPERATOR(NON_NEGATIVE, NonNegativeProx);

// ---- TotalVariation1DProx (reference prox/total_variation_1d.cc:7-25) ---------------------------------

class TotalVariation1DProx final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    DVec x = DVec::Empty(v.n, v.dt);
    k::Tv1d(x, v, input.lambda());
    output->set_value(0, x);
  }
};
REGISTER_PROX_OPERATOR(TOTAL_VARIATION_1D, TotalVariation1DProx);

// ---- SumSquareProx: ||H(x)||_2^2 (reference prox/sum_square.cc:10-40) ---------------------------------

class SumSquareProx final : public ProxOperator {
 public:
  bool CaptureSafe() const override { return true; }

 public:
  void Init(const ProxOperatorArg& arg) override {
    const BlockMatrix& H = arg.affine_arg().A;
    const BlockVector& g = arg.affine_arg().b;
    const BlockMatrix& A = arg.affine_constraint().A;
    const double alpha = std::sqrt(2 * arg.prox_function().alpha);
    // [ 0   H'  A'][ x ] = [ 0 ]
    // [ H  -I   0 ][ y ]   [-g ]
    // [ A   0  -I ][ z ]   [ v ]
    BlockMatrix M = alpha * (H + H.Transpose()) + (A + A.Transpose()) - H.LeftIdentity() -
                    A.LeftIdentity();
    chol_.Compute(M);
    b_ = (-alpha) * g;
    var_keys_ = H.col_keys();
  }
  BlockVector Apply(const BlockVector& v) override {
    return chol_.Solve(b_ + v).Select(var_keys_);
  }

  // Pattern of the compiled lasso: elimination order [constraint, var, arg],
  // L(var, constraint) = -1, Dinv(constraint) = -1, Dinv(var) = 1, L(arg, var) and Dinv(arg)
  // dense.  Then Solve(b_ + v)[var] = v_c + kappa * A^T (Dinv_arg (rhs_arg + kappa * A v_c)).
  bool DescribeLeastSquares(LeastSquaresDesc* d) const override {
    const std::vector<std::string>& p = chol_.order();
    // a refined solve is not the closed form below (fp32 on ill-conditioned data: block.cc)
    if (chol_.refine_steps() > 0) return false;
    if (p.size() == 2 && var_keys_.size() == 1) {
      // Two-block driver: A = I on the term's own variable, whose id is then both a column and a
      // row key (prox_admm_two_block.cc:70-75), so the KKT matrix has the two keys [var, arg],
      // M(var, var) = I, and Solve(b_ + v)[var] = v + kappa A^T (Dinv_arg (rhs_arg + kappa A v)) again.
      const std::string &vk = p[0], &ak = p[1];
      if (vk != *var_keys_.begin()) return false;
      const BlockMatrix& L = chol_.L();
      const BlockMatrix& Di = chol_.D_inv();
      if (!L.has_key(ak, vk) || !Di.has_key(vk, vk) || !Di.has_key(ak, ak)) return false;
      if (Di(vk, vk).impl().type() != SCALAR_MATRIX || GetScalar(Di(vk, vk)) != 1.0) return false;
      if (L(ak, vk).impl().type() != DENSE_MATRIX || Di(ak, ak).impl().type() != DENSE_MATRIX)
        return false;
      for (const auto& kv : b_.data())
        if (kv.first != ak) return false;
      d->constraint_key = vk;
      d->var_key = vk;
      d->arg_key = ak;
      d->L_arg_var = std::static_pointer_cast<const DenseMatrixImpl>(L(ak, vk).ptr());
      d->Dinv_arg = std::static_pointer_cast<const DenseMatrixImpl>(Di(ak, ak).ptr());
      if (b_.has_key(ak)) d->rhs_arg = b_(ak);
      return true;
    }
    if (p.size() != 3 || var_keys_.size() != 1) return false;
    const std::string &ck = p[0], &vk = p[1], &ak = p[2];
    if (vk != *var_keys_.begin()) return false;
    const BlockMatrix& L = chol_.L();
    const BlockMatrix& Di = chol_.D_inv();
    auto is_scalar = [](const LinearMap& m, double a) {
      return m.impl().type() == SCALAR_MATRIX && GetScalar(m) == a;
    };
    if (!L.has_key(vk, ck) || !L.has_key(ak, vk) || L.has_key(ak, ck)) return false;
    if (!Di.has_key(ck, ck) || !Di.has_key(vk, vk) || !Di.has_key(ak, ak)) return false;
    if (!is_scalar(L(vk, ck), -1.0) || !is_scalar(Di(ck, ck), -1.0) || !is_scalar(Di(vk, vk), 1.0))
      return false;
    if (L(ak, vk).impl().type() != DENSE_MATRIX || Di(ak, ak).impl().type() != DENSE_MATRIX)
      return false;
    for (const auto& kv : b_.data())
      if (kv.first != ak) return false;
    d->constraint_key = ck;
    d->var_key = vk;
    d->arg_key = ak;
    d->L_arg_var = std::static_pointer_cast<const DenseMatrixImpl>(L(ak, vk).ptr());
    d->Dinv_arg = std::static_pointer_cast<const DenseMatrixImpl>(Di(ak, ak).ptr());
    if (b_.has_key(ak)) d->rhs_arg = b_(ak);
    return true;
  }

 private:
  BlockCholesky chol_;
  BlockVector b_;
  std::set<std::string> var_keys_;
};
REGISTER_PROX_OPERATOR(SUM_SQUARE, SumSquareProx);

// ---- SumSquareEpigraph: ||x||^2 <= t (reference prox/sum_square.cc:42-57) ------------------------------

class SumSquareEpigraph final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& u = input.value_vec(0);
    const DVec& s = input.value_vec(1);
    if (!scratch_) scratch_ = Runtime::Get().Alloc(2 * sizeof(double));
    double* normsq = static_cast<double*>(scratch_->p);
    k::SumSq(u, normsq, false);
    DVec x = DVec::Empty(u.n, u.dt), t = DVec::Empty(1, u.dt);
    k::SumSquareEpigraph(x, t, u, s, normsq, normsq + 1);
    output->set_value(0, x);
    output->set_value(1, t);
  }

 private:
  std::shared_ptr<Buffer> scratch_;
};
REGISTER_EPIGRAPH_OPERATOR(SUM_SQUARE, SumSquareEpigraph);

// ---- ZeroProx: I(H(x) = 0) (reference prox/zero.cc:10-36) ---------------------------------------------

class ZeroProx final : public ProxOperator {
 public:
  bool CaptureSafe() const override { return true; }

 public:
  void Init(const ProxOperatorArg& arg) override {
    const BlockMatrix& H = arg.affine_arg().A;
    const BlockVector& g = arg.affine_arg().b;
    const BlockMatrix& A = arg.affine_constraint().A;
    BlockMatrix M = H + H.Transpose() + A + A.Transpose() - A.LeftIdentity();
    chol_.Compute(M);
    b_ = (-1.0) * g;
    var_keys_ = H.col_keys();
  }
  BlockVector Apply(const BlockVector& v) override {
    return chol_.Solve(b_ + v).Select(var_keys_);
  }

 private:
  BlockCholesky chol_;
  BlockVector b_;
  std::set<std::string> var_keys_;
};
REGISTER_PROX_OPERATOR(ZERO, ZeroProx);

// ---- AffineProx: c'x (reference prox/affine.cc:8-49) ---------------------------------------------------

class AffineProx final : public ProxOperator {
 public:
  bool CaptureSafe() const override { return true; }

 public:
  void Init(const ProxOperatorArg& arg) override {
    const BlockMatrix& A = arg.affine_constraint().A;
    const BlockVector& b = arg.affine_constraint().b;
    const double alpha = arg.prox_function().alpha;
    BlockVector c;
    if (arg.prox_function().prox_function_type == pb::ProxFunction::AFFINE) {
      // GetLinear (affine.cc:8-17): the 1 x n maps of H as vectors
      const DType dt = arg.data_map()->dtype();
      for (const auto& col : arg.affine_arg().A.data()) {
        for (const auto& row : col.second) {
          EPS_CHECK_MSG(row.second.impl().m() == 1, "AFFINE prox expects 1 x n argument maps");
          auto D = ToDense(row.second.impl(), dt);
          c.Set(col.first, D->Materialize(false));  // 1 x n contiguous == the transposed row
        }
      }
      c = alpha * c;
    }
    BlockMatrix M = A + A.Transpose() - A.LeftIdentity();
    chol_.Compute(M);
    g_ = (-1.0) * b - c;
  }
  BlockVector Apply(const BlockVector& v) override { return chol_.Solve(g_ + v); }

 private:
  BlockCholesky chol_;
  BlockVector g_;
};
REGISTER_PROX_OPERATOR(AFFINE, AffineProx);
REGISTER_PROX_OPERATOR(CONSTANT, AffineProx);

}  // namespace

}  // namespace eps

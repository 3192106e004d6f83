// Remaining proximal / epigraph operators of the reference's library:
// sort-based vector functions, the smooth Newton family, log-sum-exp, KL divergence, the
// second-order cone and the orthogonally invariant matrix functions.  Each class keeps the
// reference's plugin interface (prox/prox.h:37-77) and cites the file it restates; the numeric
// work is in kernels_segprox.hip / kernels_svd.hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "comm.h"
#include "kernels.h"
#include "prox.h"

namespace eps {
namespace {

DVec NewLike(const DVec& v) { return DVec::Empty(v.n, v.dt); }

// ---- MAX (reference prox/max.cc) ----------------------------------------------------------------

class MaxProx final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    DVec x = NewLike(v);
    k::SegMaxProx(x, v, input.lambda(), SegsOf(input.prox_function(), 0, v.n));
    output->set_value(0, x);
  }
};
REGISTER_PROX_OPERATOR(MAX, MaxProx);

class MaxEpigraph final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    const DVec& s = input.value_vec(1);
    DVec x = NewLike(v), t = NewLike(s);
    k::SegMaxEpigraph(x, t, v, s, SegsOf(input.prox_function(), 0, v.n));
    output->set_value(0, x);
    output->set_value(1, t);
  }
};
REGISTER_EPIGRAPH_OPERATOR(MAX, MaxEpigraph);

// ---- SUM_LARGEST (reference prox/sum_largest.cc; epigraph: BisectionEpigraph newton.cc:239-288) --

class SumLargestProx final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    DVec x = NewLike(v);
    k::SegSumLargestProx(x, v, input.lambda(), input.prox_function().sum_largest_k,
                         SegsOf(input.prox_function(), 0, v.n));
    output->set_value(0, x);
  }
};
REGISTER_PROX_OPERATOR(SUM_LARGEST, SumLargestProx);

class SumLargestEpigraph final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    const DVec& s = input.value_vec(1);
    DVec x = NewLike(v), t = NewLike(s);
    k::SegSumLargestEpigraph(x, t, v, s, input.prox_function().sum_largest_k,
                             SegsOf(input.prox_function(), 0, v.n));
    output->set_value(0, x);
    output->set_value(1, t);
  }
};
REGISTER_EPIGRAPH_OPERATOR(SUM_LARGEST, SumLargestEpigraph);

// ---- LOG_SUM_EXP (reference prox/log_sum_exp.cc) -----------------------------------------------------

class LogSumExpProx final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    DVec x = NewLike(v);
    k::SegLogSumExpProx(x, v, input.lambda(), SegsOf(input.prox_function(), 0, v.n));
    output->set_value(0, x);
  }
};
REGISTER_PROX_OPERATOR(LOG_SUM_EXP, LogSumExpProx);

class LogSumExpEpigraph final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    const DVec& s = input.value_vec(1);
    DVec x = NewLike(v), t = NewLike(s);
    k::SegLogSumExpEpigraph(x, t, v, s, SegsOf(input.prox_function(), 0, v.n));
    output->set_value(0, x);
    output->set_value(1, t);
  }
};
REGISTER_EPIGRAPH_OPERATOR(LOG_SUM_EXP, LogSumExpEpigraph);

// ---- smooth separable family (reference prox/newton.cc + sum_exp.cc, sum_logistic.cc,
//      sum_neg_entr.cc, sum_inv_pos.cc, sum_neg_log.cc) --------------------------------------------------

template <k::SmoothFn FN> class SmoothProxOp final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    DVec x = NewLike(v);
    if (input.elementwise()) {
      EPS_CHECK_MSG(input.lambda_vec().n == v.n, "elementwise lambda does not match the argument");
      k::SmoothProx(FN, x, v, 0.0, &input.lambda_vec());
    } else {
      k::SmoothProx(FN, x, v, input.lambda(), nullptr);
    }
    output->set_value(0, x);
  }
};

template <k::SmoothFn FN> class SmoothEpigraphOp final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    const DVec& s = input.value_vec(1);
    DVec x = NewLike(v), t = NewLike(s);
    k::SegSmoothEpigraph(FN, x, t, v, s, SegsOf(input.prox_function(), 0, v.n));
    output->set_value(0, x);
    output->set_value(1, t);
  }
};

using SumExpProx = SmoothProxOp<k::SMOOTH_EXP>;
using SumExpEpigraph = SmoothEpigraphOp<k::SMOOTH_EXP>;
using SumLogisticProx = SmoothProxOp<k::SMOOTH_LOGISTIC>;
using SumLogisticEpigraph = SmoothEpigraphOp<k::SMOOTH_LOGISTIC>;
using SumNegEntrProx = SmoothProxOp<k::SMOOTH_NEG_ENTR>;
using SumNegEntrEpigraph = SmoothEpigraphOp<k::SMOOTH_NEG_ENTR>;
using SumInvPosProx = SmoothProxOp<k::SMOOTH_INV_POS>;
using SumInvPosEpigraph = SmoothEpigraphOp<k::SMOOTH_INV_POS>;
using SumNegLogProx = SmoothProxOp<k::SMOOTH_NEG_LOG>;
using SumNegLogEpigraph = SmoothEpigraphOp<k::SMOOTH_NEG_LOG>;
REGISTER_PROX_OPERATOR(SUM_EXP, SumExpProx);
REGISTER_EPIGRAPH_OPERATOR(SUM_EXP, SumExpEpigraph);
REGISTER_PROX_OPERATOR(SUM_LOGISTIC, SumLogisticProx);
REGISTER_EPIGRAPH_OPERATOR(SUM_LOGISTIC, SumLogisticEpigraph);
REGISTER_PROX_OPERATOR(SUM_NEG_ENTR, SumNegEntrProx);
REGISTER_EPIGRAPH_OPERATOR(SUM_NEG_ENTR, SumNegEntrEpigraph);
REGISTER_PROX_OPERATOR(SUM_INV_POS, SumInvPosProx);
REGISTER_EPIGRAPH_OPERATOR(SUM_INV_POS, SumInvPosEpigraph);
REGISTER_PROX_OPERATOR(SUM_NEG_LOG, SumNegLogProx);
REGISTER_EPIGRAPH_OPERATOR(SUM_NEG_LOG, SumNegLogEpigraph);

// ---- SUM_KL_DIV (reference prox/sum_kl_div.cc) ----------------------------------------------------------

class SumKLDivProx final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& u = input.value_vec(0);
    const DVec& v = input.value_vec(1);
    DVec x = NewLike(u), y = NewLike(v);
    if (input.elementwise()) {
      // lambda_vec() covers both arguments (vector_prox.cc:108-116); the kernel wants the first
      EPS_CHECK_MSG(input.lambda_vec().n >= u.n, "elementwise lambda shorter than the argument");
      DVec lv = input.lambda_vec().Slice(0, u.n);
      k::KlDivProx(x, y, u, v, 0.0, &lv);
    } else {
      k::KlDivProx(x, y, u, v, input.lambda(), nullptr);
    }
    output->set_value(0, x);
    output->set_value(1, y);
  }
};
REGISTER_PROX_OPERATOR(SUM_KL_DIV, SumKLDivProx);

class SumKLDivEpigraph final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& u = input.value_vec(0);
    const DVec& v = input.value_vec(1);
    const DVec& s = input.value_vec(2);
    DVec x = NewLike(u), y = NewLike(v), t = NewLike(s);
    k::SegKlDivEpigraph(x, y, t, u, v, s, SegsOf(input.prox_function(), 0, u.n));
    output->set_value(0, x);
    output->set_value(1, y);
    output->set_value(2, t);
  }
};
REGISTER_EPIGRAPH_OPERATOR(SUM_KL_DIV, SumKLDivEpigraph);

// ---- EXP epigraph, elementwise (reference prox/exp.cc) -----------------------------------------------

class ExpEpigraph final : public VectorProx {
 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    const DVec& v = input.value_vec(0);
    const DVec& s = input.value_vec(1);
    DVec x = NewLike(v), t = NewLike(s);
    k::ExpEpigraph(x, t, v, s);
    output->set_value(0, x);
    output->set_value(1, t);
  }
};
REGISTER_EPIGRAPH_OPERATOR(EXP, ExpEpigraph);

// ---- SECOND_ORDER_CONE (reference prox/second_order_cone.cc:21-124) ------------------------------------
// ||ax*X_i + bx||_2 <= at*t_i + bt_i for the rows X_i of the m x n argument; "t" is argument 0.

class SecondOrderConeProx final : public ProxOperator {
 public:
  void Init(const ProxOperatorArg& arg) override {
    const pb::ProxFunction& f = arg.prox_function();
    EPS_CHECK_MSG(f.arg_size.size() == 2 && f.arg_size[1].dim.size() == 2,
                  "SECOND_ORDER_CONE needs two sized arguments");
    m_ = f.arg_size[1].dim[0];
    n_ = f.arg_size[1].dim[1];
    const DType dt = arg.data_map()->dtype();
    // InitArgs (:90-107)
    const BlockMatrix& H = arg.affine_arg().A;
    for (const auto& col : H.data()) {  // GetArgKeys (:6-19)
      for (const auto& row : col.second) {
        if (row.first == affine::arg_key(0)) t_key_ = col.first;
        else if (row.first == affine::arg_key(1)) x_key_ = col.first;
        else EPS_FATAL("Unknown row key " << row.first);
      }
    }
    EPS_CHECK_MSG(!t_key_.empty() && !x_key_.empty(), "SECOND_ORDER_CONE: missing argument");
    const double at = GetScalar(H(affine::arg_key(0), t_key_));
    const double ax = GetScalar(H(affine::arg_key(1), x_key_));
    const BlockVector& g = arg.affine_arg().b;
    a_ = at / std::fabs(ax);
    // bx_ = g(arg 1) / ax ; bt_/a_ = g(arg 0) / |ax| / a_   (zeros when absent, block_vector.cc:62-67)
    bx_ = DVec::Zeros(m_ * n_, dt);
    if (g.has_key(affine::arg_key(1))) k::Axpby(bx_, 1 / ax, g(affine::arg_key(1)), 0.0);
    bt_over_a_ = DVec::Zeros(m_, dt);
    if (g.has_key(affine::arg_key(0)))
      k::Axpby(bt_over_a_, 1 / std::fabs(ax) / a_, g(affine::arg_key(0)), 0.0);
    // InitConstraints (:109-122): A'A must be one scalar
    const BlockMatrix& A = arg.affine_constraint().A;
    AT_ = A.Transpose();
    BlockMatrix ATA = AT_ * A;
    const double alphat = GetScalar(ATA(t_key_, t_key_));
    const double alphax = GetScalar(ATA(x_key_, x_key_));
    EPS_CHECK_MSG(alphat == alphax, "A'A not scalar matrix");
    BlockMatrix D;
    D(x_key_, x_key_) = LinearMap::Scalar(1 / alphat, m_ * n_);
    D(t_key_, t_key_) = LinearMap::Scalar(1 / alphat, m_);
    AT_ = D * AT_;
  }

  BlockVector Apply(const BlockVector& v) override {  // :46-56
    BlockVector u = AT_ * v;
    DVec X = u(x_key_).Clone();
    k::Axpby(X, 1.0, bx_, 1.0);
    DVec t = u(t_key_).Clone();
    k::Axpby(t, 1.0, bt_over_a_, 1.0);
    EPS_CHECK(X.n == m_ * n_ && t.n == m_);
    k::Segs S;  // rows of the column-major m x n matrix
    S.count = m_;
    S.len = n_;
    S.seg_stride = 1;
    S.elem_stride = m_;
    DVec Xo = NewLike(X), to = NewLike(t);
    k::SegSocProject(Xo, to, X, t, a_, S);
    k::Axpby(Xo, -1.0, bx_, 1.0);
    k::Axpby(to, -1.0, bt_over_a_, 1.0);
    BlockVector x;
    x.Set(x_key_, Xo);
    x.Set(t_key_, to);
    return x;
  }

 private:
  BlockMatrix AT_;
  double a_ = 1;
  DVec bx_, bt_over_a_;
  std::string t_key_, x_key_;
  int64_t m_ = 0, n_ = 0;
};
REGISTER_PROX_OPERATOR(SECOND_ORDER_CONE, SecondOrderConeProx);

// ---- OrthoInvariantProx: F(X) = f(spectrum of X) (reference prox/ortho_invariant.{h,cc}) -----------
// The reference diagonalises Y^T Y + 1e-15 I (singular values) or (Y + Y^T)/2 (eigenvalues) with
// Eigen's SelfAdjointEigenSolver on the host and hands the spectrum to a nested vector prox.
// Here both spectra come from the one-sided Jacobi SVD on the device (kernels_svd.hip); the
// symmetric eigenproblem is solved as the SVD of the shifted matrix S + cI with c > ||S||_F,
// which is positive definite, so its singular vectors ARE the eigenvectors and its singular
// values the eigenvalues + c (no +-sigma ambiguity).

class OrthoInvariantProx : public VectorProx {
 public:
  OrthoInvariantProx(int eigen_prox_type, bool symmetric_part = false, bool add_residual = false,
                     bool epigraph = false)
      : eigen_prox_type_(eigen_prox_type), symmetric_part_(symmetric_part),
        add_residual_(add_residual), epigraph_(epigraph) {}

  void Init(const ProxOperatorArg& arg) override {
    VectorProx::Init(arg);
    const pb::ProxFunction& f = arg.prox_function();
    EPS_CHECK_MSG(!f.arg_size.empty() && f.arg_size[0].dim.size() == 2,
                  "matrix prox needs arg_size");
    m_ = f.arg_size[0].dim[0];
    n_ = f.arg_size[0].dim[1];
    EPS_CHECK_MSG(!symmetric_part_ || m_ == n_, "symmetric matrix function of a non-square argument");
    dtype_ = arg.data_map()->dtype();
    eigen_prox_.reset();
    V_prev_ = DVec();
    basis_prev_ = DVec();
    basis_is_left_ = false;
    {
      const char* pe2 = std::getenv("EPSILON_HIP_SVD_POLAR");
      polar_enabled_ = !(pe2 && pe2[0] == '0');
    }
    calls_ = 0;
    const char* e = std::getenv("EPSILON_HIP_SVD_WARM");
    warm_start_ = !(e && e[0] == '0');
    const char* pe = std::getenv("EPSILON_HIP_SVD_PARTIAL");
    partial_enabled_ = !(pe && pe[0] == '0');
    partial_backoff_ = 0;
    partial_fail_streak_ = 0;
    last_rank_ = -1;
  }

 protected:
  void ApplyVector(const VectorProxInput& input, VectorProxOutput* output) override {
    if (!eigen_prox_) InitEigenProx(epigraph_ ? 1.0 : input.lambda());
    const DVec& y = input.value_vec(0);
    EPS_CHECK(y.n == m_ * n_);
    // Sharded solve with the matrix argument split by rows over the ranks (m_ = this rank's
    // rows): the decomposition runs in its row-sharded form - panel Grams and column norms are
    // all-reduced, the n x n factor V and the singular values are replicated.
    const ShardSpec& shard = ShardSpec::Get();
    const bool row_sharded = shard.active() && shard.IsSharded(affine::arg_key(0));
    EPS_CHECK_MSG(!(row_sharded && symmetric_part_),
                  "symmetric matrix functions are not available on a row-sharded argument");
    // Nuclear-norm prox: only the singular values above lambda survive the nested soft threshold
    // (ortho_invariant.cc:76-105), so when they are few the leading block is all that is needed.
    if (eigen_prox_type_ == pb::ProxFunction::NORM_1 && !symmetric_part_ && !epigraph_ && !row_sharded &&
        partial_enabled_ && std::min(m_, n_) >= kPartialMinSize) {
      if (partial_backoff_ > 0) {
        --partial_backoff_;
      } else {
        DVec X;
        static const bool trace = std::getenv("EPSILON_HIP_SVD_TRACE") != nullptr;
        const bool ok = ThresholdedPartialSvd(y, input.lambda(), &X);
        if (trace)
          std::fprintf(stderr, "[svd] call %lld: partial route %s (values above lambda: %lld)\n",
                       static_cast<long long>(calls_), ok ? "taken" : "gave up", static_cast<long long>(last_rank_));
        if (ok) {
          partial_fail_streak_ = 0;
          output->set_value(0, X);
          return;
        }
        // The spectrum is (numerically) full above lambda: stop trying for a while.  A failed
        // attempt costs a few per cent of the full decomposition, and the first applications of
        // an ADMM run (an argument far from its low-rank limit) say little about the later ones:
        // skip 1, 2, 4 ... 16 applications after consecutive failures.
        partial_backoff_ = std::min(16, 1 << std::min(partial_fail_streak_, 4));
        if (const char* e = std::getenv("EPSILON_HIP_SVD_PARTIAL_BACKOFF")) partial_backoff_ = std::atoi(e);
        ++partial_fail_streak_;
      }
    }
    // Full rank above the threshold: the GEMM-only polar route while no warm basis exists.  At
    // n = 1e4 a polar application costs 0.65 s, a cold Jacobi one 1.6 s, a warm-started one 0.25 s:
    // a run that stops after one application (the reference's robust-PCA benchmark at its default
    // tolerance) should never pay for the decomposition, a long run should reach the warm starts.
    // Ski-rental rule: stay on the polar route until its extra cost over warm starts (0.4 s per
    // application) has reached the price of the cold decomposition - kPolarCalls = 4 applications.
    if (eigen_prox_type_ == pb::ProxFunction::NORM_1 && !symmetric_part_ && !epigraph_ && !row_sharded &&
        polar_enabled_ && m_ >= n_ && n_ >= kPolarMinSize && basis_prev_.n == 0 && V_prev_.n == 0 &&
        calls_ < kPolarCalls) {
      DVec X;
      if (PolarNuclearProx(y, input.lambda(), &X)) {
        ++calls_;
        output->set_value(0, X);
        return;
      }
      polar_enabled_ = false;  // rejected once: this operator stays on the decomposition
    }
    if (!symmetric_part_ && !row_sharded && m_ >= n_ && k::JacobiSvdCanSkipV(m_, n_, y.dt)) {
      ApplyOneSided(input, output, y);
      return;
    }
    DVec W, R;
    double shift = 0;
    if (symmetric_part_) {
      W = DVec::Empty(y.n, y.dt);
      k::MatCopy(true, n_, n_, 0.5, y, n_, W);  // W = Y^T / 2
      if (add_residual_) {
        R = DVec::Empty(y.n, y.dt);
        k::Axpby(R, 0.5, y, 0.0);
        k::Axpby(R, -1.0, W, 1.0);  // (Y - Y^T) / 2
      }
      k::Axpby(W, 0.5, y, 1.0);  // (Y + Y^T) / 2
      Runtime& rt = Runtime::Get();
      rt.ResetSlots();
      const int slot = rt.NewSlot();
      k::SumSq(W, rt.SlotPtr(slot), false);
      rt.FetchSlots();
      // c > ||S||_2 makes S + cI positive definite; a (numerically) zero S is shifted by 1 so
      // that the Jacobi sweep has well-scaled columns to work on
      const double fro = std::sqrt(rt.SlotValue(slot));
      shift = fro > 1e-150 ? fro * 1.0625 : 1.0;
      k::AddDiag(W, n_, n_, shift, nullptr);
    } else {
      W = y.Clone();
    }
    // Successive applications inside an ADMM run see slowly changing matrices: continue from
    // the right singular vectors of the previous call (W <- Y0 V_prev; any orthogonal start is
    // exact) and the Jacobi iteration needs 4-5 sweeps instead of 10-13; every 64th call starts
    // from the identity anyway.
    const DVec Y0 = symmetric_part_ ? W : y;  // the matrix being decomposed (W is overwritten)
    DVec V;
    const bool warm = warm_start_ && V_prev_.n == n_ * n_ && V_prev_.dt == y.dt && (calls_ % 64) != 0;
    ++calls_;
    if (warm) {
      V = V_prev_;  // rotated in place; nothing else holds it
      DVec W0 = DVec::Empty(m_ * n_, y.dt);
      k::Gemm(false, false, m_, n_, n_, 1.0, Y0, m_, V, n_, 0.0, W0, m_);
      W = W0;
    } else {
      if (symmetric_part_) W = W.Clone();
      V = DVec::Empty(n_ * n_, y.dt);
    }
    k::JacobiSvd(W, m_, n_, V, 40, warm, row_sharded);
    {
      // W and V have been rotated separately thousands of times: in fp32 their rounding is a
      // random walk that leaves V orthogonal, and W equal to Y0 V, only to ~1e-5 (n = 100) ..
      // 2e-4 (n = 10^4) - enough to move the stopping sweep of an fp32 solve and to show as a
      // 0.6 % defect in the optimality certificate at full size.  One Newton-Schulz step
      // V <- V (3 I - V^T V) / 2 squares the orthogonality defect, and W is re-formed from it.
      DVec T = DVec::Empty(n_ * n_, y.dt);
      k::Gemm(true, false, n_, n_, n_, -1.0, V, n_, V, n_, 0.0, T, n_);
      k::AddDiag(T, n_, n_, 3.0, nullptr);
      DVec V2 = DVec::Empty(n_ * n_, y.dt);
      k::Gemm(false, false, n_, n_, n_, 0.5, V, n_, T, n_, 0.0, V2, n_);
      V = V2;
      k::Gemm(false, false, m_, n_, n_, 1.0, Y0, m_, V, n_, 0.0, W, m_);
    }
    V_prev_ = V;
    DVec sigma = DVec::Empty(n_, y.dt);
    k::ColNorms(W, m_, n_, sigma, row_sharded);
    DVec d = sigma;
    if (symmetric_part_) {
      d = DVec::Full(n_, -shift, y.dt);
      k::Axpby(d, 1.0, sigma, 1.0);
    }
    BlockVector in;
    DVec xt, t;
    // the nested prox works on the replicated singular values: nothing in it is sharded
    LocalShardScope replicated_scope{std::set<std::string>()};
    if (epigraph_) {  // ApplyEigenEpigraph (:107-116)
      in.Set(affine::arg_key(0), d);
      in.Set(affine::arg_key(1), input.value_vec(1));
      BlockVector out = eigen_prox_->Apply(in);
      xt = out(affine::arg_key(0));
      t = out(affine::arg_key(1));
    } else {  // ApplyEigenProx (:100-105): input alpha*d
      DVec scaled = DVec::Empty(n_, y.dt);
      k::Axpby(scaled, alpha_, d, 0.0);
      in.Set(affine::arg_key(0), scaled);
      xt = eigen_prox_->Apply(in)(affine::arg_key(0));
    }
    k::ColScaleByRatio(W, m_, n_, sigma, xt);  // U diag(x~), U = W diag(1/sigma)
    DVec X = DVec::Empty(m_ * n_, y.dt);
    k::Gemm(false, true, m_, n_, n_, 1.0, W, m_, V, n_, 0.0, X, m_);
    if (!epigraph_) {
      if (add_residual_) {
        k::Axpby(X, 1.0, R, 1.0);
      } else if (symmetric_part_) {
        DVec Xt = DVec::Empty(X.n, X.dt);
        k::MatCopy(true, n_, n_, 0.5, X, n_, Xt);
        k::Axpby(Xt, 0.5, X, 1.0);
        X = Xt;
      }
    }
    output->set_value(0, X);
    if (epigraph_) output->set_value(1, t);
  }

 private:
  // The decomposition WITHOUT an accumulated factor (fp32 block Jacobi on the matrix cores, m >= n).
  // X = U diag(x~) V^T needs only ONE orthonormal side: with Q the normalised columns of the
  // converged W (left vectors when W = Y V0, right vectors when W = Y^T U0) and Z the other side
  // re-formed from Y (Z = Y^T Q or Y Q, columns sigma_i v_i / sigma_i u_i),
  //     X = Q diag(x~ / sigma) Z^T      resp.     X = Z diag(x~ / sigma) Q^T .
  // So the Jacobi sweeps rotate W alone - a step streams 1.2 GB instead of 2.0 GB at n = 1e4 - and
  // the basis Q of one application is the orthogonal start of the next one, applied from the
  // OTHER side (W = Y^T Q_left, then W = Y Q_right, ...): warm starts survive although no factor
  // is ever accumulated.  Q's columns are orthonormal to the accuracy of the sweeps whatever the
  // singular value (a column of noise is still a unit vector orthogonal to the others), one
  // Newton-Schulz step squares the defect, and sigma comes from the re-formed Z - the same clean-up
  // the two-sided path applies to V and W.  (reference prox/ortho_invariant.cc:36-66 forms all
  // three factors from eig(Y^T Y).)
  void ApplyOneSided(const VectorProxInput& input, VectorProxOutput* output, const DVec& y) {
    const int64_t m = m_, n = n_;
    const DType dt = y.dt;
    const bool have = warm_start_ && basis_prev_.dt == dt && (calls_ % 64) != 0 &&
                      basis_prev_.n == (basis_is_left_ ? m * n : n * n);
    ++calls_;
    // left == true: this application starts from left vectors (W = Y^T U0, n x n) and produces Q = V
    const bool left = have && basis_is_left_;
    const int64_t wr = left ? n : m;  // rows of W
    DVec W;
    if (!have) {
      W = y.Clone();
    } else if (left) {
      W = DVec::Empty(n * n, dt);
      k::Gemm(true, false, n, n, m, 1.0, y, m, basis_prev_, m, 0.0, W, n);
    } else {
      W = DVec::Empty(m * n, dt);
      k::Gemm(false, false, m, n, n, 1.0, y, m, basis_prev_, n, 0.0, W, m);
    }
    basis_prev_ = DVec();
    k::JacobiSvdNoV(W, wr, n, 40);
    DVec one = DVec::Full(n, 1.0, dt), s0 = DVec::Empty(n, dt);
    k::ColNorms(W, wr, n, s0, false);
    k::ColScaleByRatio(W, wr, n, s0, one);  // Q0 = W diag(1 / sigma) (a zero column stays zero)
    // Newton-Schulz: Q = Q0 (3 I - Q0^T Q0) / 2
    DVec T = DVec::Empty(n * n, dt);
    k::Gemm(true, false, n, n, wr, -1.0, W, wr, W, wr, 0.0, T, n);
    k::AddDiag(T, n, n, 3.0, nullptr);
    DVec Q = DVec::Empty(wr * n, dt);
    k::Gemm(false, false, wr, n, n, 0.5, W, wr, T, n, 0.0, Q, wr);
    W = DVec();
    // the other side, re-formed from Y: Z = Y Q (m x n) or Y^T Q (n x n); columns of length sigma_i
    const int64_t zr = left ? m : n;
    DVec Z = left ? DVec::Empty(m * n, dt) : T;
    if (left) k::Gemm(false, false, m, n, n, 1.0, y, m, Q, n, 0.0, Z, m);
    else k::Gemm(true, false, n, n, m, 1.0, y, m, Q, m, 0.0, Z, n);
    DVec sigma = DVec::Empty(n, dt);
    k::ColNorms(Z, zr, n, sigma, false);
    // nested prox on the singular values (ortho_invariant.cc:100-116)
    BlockVector in;
    DVec xt, t;
    LocalShardScope replicated_scope{std::set<std::string>()};
    if (epigraph_) {
      in.Set(affine::arg_key(0), sigma);
      in.Set(affine::arg_key(1), input.value_vec(1));
      BlockVector out = eigen_prox_->Apply(in);
      xt = out(affine::arg_key(0));
      t = out(affine::arg_key(1));
    } else {
      DVec scaled = DVec::Empty(n, dt);
      k::Axpby(scaled, alpha_, sigma, 0.0);
      in.Set(affine::arg_key(0), scaled);
      xt = eigen_prox_->Apply(in)(affine::arg_key(0));
    }
    // keep Q for the next application (before it is scaled); a rank-deficient argument (a zero
    // column) leaves no orthogonal basis: the next application starts cold
    {
      Runtime& rt = Runtime::Get();
      std::vector<double> sh = s0.ToHost();
      double smin = sh.empty() ? 0.0 : sh[0];
      for (double v : sh) smin = std::min(smin, v);
      (void)rt;
      if (smin > 0.0 && warm_start_) {
        basis_prev_ = Q.Clone();
        basis_is_left_ = !left;  // Q holds left vectors when this application started from the right
      }
    }
    DVec X = DVec::Empty(m * n, dt);
    if (left) {  // X = Z diag(x~ / sigma) Q^T
      k::ColScaleByRatio(Z, m, n, sigma, xt);
      k::Gemm(false, true, m, n, n, 1.0, Z, m, Q, n, 0.0, X, m);
    } else {     // X = Q diag(x~ / sigma) Z^T
      k::ColScaleByRatio(Q, m, n, sigma, xt);
      k::Gemm(false, true, m, n, n, 1.0, Q, m, Z, n, 0.0, X, m);
    }
    output->set_value(0, X);
    if (epigraph_) output->set_value(1, t);
  }

  // ---- nuclear-norm prox WITHOUT a decomposition: GEMMs only (round 3) ---------------------------
  // With the polar decomposition Y = Q H (Q^T Q = I, H = V Sigma V^T):
  //     prox(Y) = U (Sigma - tau)_+ V^T = Q (H - tau I)_+ ,   (A)_+ = (A + |A|) / 2 ,  |A| = A sign(A).
  // Both matrix functions come from Newton-Schulz iterations - odd cubics p(x) = a x + b x^3 applied
  // to the singular values (X <- X (a I + b X^T X)) resp. eigenvalues (S <- S (a I + b S^2)) - with
  // the equioscillating cubic of the current interval [l, 1] (p(l) = p(1), max p = 1: the lower end
  // grows ~2.6x per step, l' = p(l) is known without looking at the matrix) and plain (3/2, -1/2)
  // steps at the end.  What makes it exact enough in the presence of tiny singular values: the
  // polar iteration only has to converge for sigma >= tau (a direction with sigma < tau ends with a
  // factor q <= 1 in Q, its eigenvalue q sigma - tau of H - tau I is negative, the positive part
  // drops it whatever q is), and the sign iteration only for |sigma - tau| above a resolution r
  // (an unconverged sign s in (-1, 1) leaves an error <= r / 2 in that direction).  The few
  // dominant, well separated triplets are taken out first by a randomized block (their prox is a
  // rank-k update; prox(Y_top + Y_rest) = prox(Y_top) + prox(Y_rest) for orthogonal singular
  // subspaces): the rounding of a product scales with ||Y_rest||_2, not with sigma_1.
  // ~115 GEMMs of 1-2 n^3 flop on the split-f16 kernel at n = 1e4 (fp32: mostly monotone steps, see
  // CubicOn), against 18 HBM-bound Jacobi sweeps;
  // the result is held to the optimality condition of the prox afterwards (||P||_2 <= 1 and
  // <X, P> = ||X||_* with P = (Y - X) / tau; ||X||_* = trace((H - tau I)_+) + the block's part):
  // false sends the call to the Jacobi route.  (reference prox/ortho_invariant.cc:36-105 thresholds
  // the singular values of an eigendecomposition of Y^T Y.)
  // The equioscillating cubic is not monotone (it sends the top of the interval to the bottom),
  // and a non-monotone step multiplies the rounding noise BETWEEN eigen-directions by its largest
  // divided difference (~2.6) while the iteration only restores S^2 = I, not S = sign of the matrix
  // it was given: in fp32 three such steps already cost 3e-4 ||Y_rest||_2 (measured in a strict
  // float32 emulation and on the device), two cost nothing.  `amp` carries the product of the
  // steps' bounds; past the budget (10 in fp32, 1e8 in fp64) the monotone (3/2, -1/2) step is used.
  static void CubicOn(double l, double budget, double* amp, double* a, double* b) {
    *a = 1.5;
    *b = -0.5;
    if (l > 0.95 || *amp * 2.6 > budget) return;
    const double q = 1.0 + l + l * l, xs = std::sqrt(q / 3.0);
    *a = 3.0 / (2.0 * xs);
    *b = -*a / q;
    *amp *= std::max(*a, std::fabs(*a + 3.0 * *b));
  }

  bool PolarNuclearProx(const DVec& y, double tau, DVec* Xout) {
    Runtime& rt = Runtime::Get();
    const int64_t m = m_, n = n_;
    const DType dt = y.dt;
    ProfScope prof("polar_prox", m, n);
    static const bool trace = std::getenv("EPSILON_HIP_SVD_TRACE") != nullptr;
    // ---- 1. the dominant block: randomized range finder (two power steps), small SVD
    const int64_t k = 32;
    DVec Z = DVec::Empty(n * k, dt), Q0 = DVec::Empty(m * k, dt);
    k::FillHash(Z, 0x90A12ull);
    k::Gemm(false, false, m, k, n, 1.0, y, m, Z, n, 0.0, Q0, m);
    for (int it = 0; it < 2; ++it) {
      Orthonormalise(Q0, m, k);
      k::Gemm(true, false, n, k, m, 1.0, y, m, Q0, m, 0.0, Z, n);
      Orthonormalise(Z, n, k);
      k::Gemm(false, false, m, k, n, 1.0, y, m, Z, n, 0.0, Q0, m);
    }
    Orthonormalise(Q0, m, k);
    DVec Wb = DVec::Empty(n * k, dt), Vb = DVec::Empty(k * k, dt);
    k::Gemm(true, false, n, k, m, 1.0, y, m, Q0, m, 0.0, Wb, n);  // B^T = Y^T Q0 = Wb Vb^T
    k::JacobiSvd(Wb, n, k, Vb, 40, false, false);
    DVec sig = DVec::Empty(k, dt);
    k::ColNorms(Wb, n, k, sig, false);
    DVec U = DVec::Empty(m * k, dt);
    k::Gemm(false, false, m, k, k, 1.0, Q0, m, Vb, k, 0.0, U, m);
    std::vector<double> sh = sig.ToHost();
    // residuals || Y (Wb_i / sigma_i) - sigma_i u_i || of the pairs
    std::vector<double> keep(static_cast<size_t>(k), 0.0);
    int64_t kd = 0;
    {
      DVec T = DVec::Empty(m * k, dt), U2 = U.Clone();
      k::Gemm(false, false, m, k, n, 1.0, y, m, Wb, n, 0.0, T, m);
      DVec one = DVec::Full(k, 1.0, dt), sig2 = DVec::Empty(k, dt);
      k::DiagMul(sig2, 1.0, sig, sig, 0.0);
      k::ColScaleByRatio(U2, m, k, one, sig2);
      k::Axpby(T, -1.0, U2, 1.0);
      DVec rn = DVec::Empty(k, dt);
      k::ColNorms(T, m, k, rn, false);
      const std::vector<double> rh = rn.ToHost();
      double smax0 = 0, smin0 = sh[0];
      for (double sv : sh) {
        smax0 = std::max(smax0, sv);
        smin0 = std::min(smin0, sv);
      }
      const double res_tol = dt == F32 ? 2e-5 : 1e-10;
      // taken out: converged pairs that stand clear of the rest of the block
      for (int64_t i = 0; i < k; ++i)
        if (sh[i] > 2.0 * smin0 && sh[i] > 0 && rh[i] / sh[i] <= res_tol * smax0) {
          keep[static_cast<size_t>(i)] = 1.0;
          ++kd;
        }
    }
    DVec mask = DVec::FromHost(keep.data(), k, dt), onek = DVec::Full(k, 1.0, dt);
    k::ColScaleByRatio(U, m, k, onek, mask);  // columns of the pairs that stay in Y_rest become zero
    DVec Yr = y.Clone();
    if (kd > 0) k::Gemm(false, true, m, n, k, -1.0, U, m, Wb, n, 1.0, Yr, m);
    // ---- 2. an upper bound of ||Y_rest||_2: power iteration (a lower bound) with a factor 1.5
    // in hand - a factor costs log_2.6(1.5) = 0.4 steps, an underestimate would flip the sign of
    // the top singular value under the aggressive cubic - capped by sqrt(||.||_1 ||.||_inf)
    double est = 0;
    {
      DVec v = DVec::Empty(n, dt), w = DVec::Empty(m, dt);
      k::FillHash(v, 0xC0FFEEull);
      int slot = 0;
      rt.ResetSlots();
      slot = rt.NewSlot();
      k::SumSq(v, rt.SlotPtr(slot), false);
      k::ScaleByInvNorm(v, v, rt.SlotPtr(slot));
      for (int it = 0; it < 30; ++it) {
        k::Gemv(false, m, n, 1.0, Yr, m, v, 0.0, w);
        k::Gemv(true, m, n, 1.0, Yr, m, w, 0.0, v);
        slot = rt.NewSlot();
        k::SumSq(v, rt.SlotPtr(slot), false);
        k::ScaleByInvNorm(v, v, rt.SlotPtr(slot));
      }
      rt.FetchSlots();
      est = std::sqrt(std::sqrt(rt.SlotValue(slot)));  // ||Y^T Y v|| -> sigma^2
    }
    if (!(est > 0) || !std::isfinite(est)) return false;
    double smax = 1.5 * est;
    {
      auto col = rt.Alloc(static_cast<size_t>(std::max(m, n)) * sizeof(double));
      std::vector<double> h(static_cast<size_t>(n));
      k::ColAbsSums(Yr, m, n, m, static_cast<double*>(col->p));
      EPS_HIP(hipMemcpyAsync(h.data(), col->p, h.size() * sizeof(double), hipMemcpyDeviceToHost, rt.stream()));
      rt.Sync();
      double l1 = 0;
      for (double v : h) l1 = std::max(l1, v);
      DVec Yt = DVec::Empty(m * n, dt);
      k::MatCopy(true, n, m, 1.0, Yr, m, Yt);
      h.resize(static_cast<size_t>(m));
      k::ColAbsSums(Yt, n, m, n, static_cast<double*>(col->p));
      EPS_HIP(hipMemcpyAsync(h.data(), col->p, h.size() * sizeof(double), hipMemcpyDeviceToHost, rt.stream()));
      rt.Sync();
      double linf = 0;
      for (double v : h) linf = std::max(linf, v);
      const double bound = std::sqrt(l1 * linf);
      if (bound > 0 && bound < smax) smax = std::max(bound, est);
    }
    // fp32: the rounding of the products is ~6e-6 ||Y_rest||_2 per direction; beyond 1e4 tau it
    // would show against the threshold (the Jacobi route resolves directions relative to sigma_i)
    if (dt == F32 && smax > 2e4 * tau) return false;
    if (trace)
      std::fprintf(stderr, "[svd] polar route: %lld dominant triplets taken out, ||rest||_2 ~ %.4g (bound %.4g), tau %.4g\n",
                   static_cast<long long>(kd), est, smax, tau);
    // ---- 3. polar factor of Y_rest
    const double noise_budget = dt == F32 ? 10.0 : 1e8;
    DVec Xc = DVec::Empty(m * n, dt), Xn = DVec::Empty(m * n, dt);
    k::Axpby(Xc, 1.0 / smax, Yr, 0.0);
    DVec G = DVec::Empty(n * n, dt);
    int steps = 0;
    {
      double l = 0.9 * tau / smax, amp = 1.0;
      int tail = 0;
      while (tail < 2 && steps < 80) {
        double a, b;
        CubicOn(l, noise_budget, &amp, &a, &b);
        k::Gemm(true, false, n, n, m, b, Xc, m, Xc, m, 0.0, G, n, true);  // b X^T X (lower tiles)
        k::SymmetrizeFromLower(G, n, n);
        k::AddDiag(G, n, n, a, nullptr);
        k::Gemm(false, false, m, n, n, 1.0, Xc, m, G, n, 0.0, Xn, m);
        std::swap(Xc, Xn);
        l = a * l + b * l * l * l;
        if (a == 1.5 && l > 0.9999) ++tail;
        ++steps;
      }
    }
    Xn = DVec();
    if (trace) {  // orthogonality of the polar factor (on everything that converged)
      k::Gemm(true, false, n, n, m, 1.0, Xc, m, Xc, m, 0.0, G, n);
      k::AddDiag(G, n, n, -1.0, nullptr);
      rt.ResetSlots();
      const int sl = rt.NewSlot();
      k::SumSq(G, rt.SlotPtr(sl), false);
      rt.FetchSlots();
      std::fprintf(stderr, "[svd] polar route: ||Q^T Q - I||_F / sqrt(n) = %.3e after %d steps\n",
                   std::sqrt(rt.SlotValue(sl) / static_cast<double>(n)), steps);
    }
    // ---- 4. A = sym(Q^T Y_rest) - tau I
    DVec A = DVec::Empty(n * n, dt);
    k::Gemm(true, false, n, n, m, 1.0, Xc, m, Yr, m, 0.0, A, n);
    {
      DVec At = DVec::Empty(n * n, dt);
      k::MatCopy(true, n, n, 0.5, A, n, At);
      k::Axpby(A, 1.0, At, 0.5);
    }
    k::AddDiag(A, n, n, -tau, nullptr);
    // ---- 5. S = sign(A), resolution r around the threshold
    const double amax = smax + tau;
    const double res = std::max(1e-3 * tau, (dt == F32 ? 2e-6 : 1e-12) * amax);
    DVec S = DVec::Empty(n * n, dt), Sn = DVec::Empty(n * n, dt);
    k::Axpby(S, 1.0 / amax, A, 0.0);
    int ssteps = 0;
    {
      double l = res / amax, amp = 1.0;
      int tail = 0;
      while (tail < 2 && ssteps < 120) {
        double a, b;
        CubicOn(l, noise_budget, &amp, &a, &b);
        k::Gemm(true, false, n, n, n, b, S, n, S, n, 0.0, G, n, true);  // b S^2 = b S^T S (lower tiles)
        k::SymmetrizeFromLower(G, n, n);
        k::AddDiag(G, n, n, a, nullptr);
        // S G = a S + b S^3 is symmetric: the lower tiles only, mirrored (half the product, and S
        // stays exactly symmetric)
        k::Gemm(false, false, n, n, n, 1.0, S, n, G, n, 0.0, Sn, n, true);
        k::SymmetrizeFromLower(Sn, n, n);
        std::swap(S, Sn);
        l = a * l + b * l * l * l;
        if (a == 1.5 && l > 0.9999) ++tail;
        ++ssteps;
      }
    }
    if (trace) {
      k::Gemm(false, false, n, n, n, 1.0, S, n, S, n, 0.0, G, n);
      k::AddDiag(G, n, n, -1.0, nullptr);
      rt.ResetSlots();
      const int sl = rt.NewSlot();
      k::SumSq(G, rt.SlotPtr(sl), false);
      rt.FetchSlots();
      std::fprintf(stderr, "[svd] polar route: ||S^2 - I||_F / sqrt(n) = %.3e after %d steps\n",
                   std::sqrt(rt.SlotValue(sl) / static_cast<double>(n)), ssteps);
    }
    // ---- 6. X_rest = Q (A + A S) / 2, plus the block's part
    k::Gemm(false, false, n, n, n, 0.5, A, n, S, n, 0.0, Sn, n, true);  // A S = |A|: symmetric
    k::SymmetrizeFromLower(Sn, n, n);
    k::Axpby(Sn, 0.5, A, 1.0);  // (A)_+
    S = DVec();
    G = DVec();
    DVec X = DVec::Empty(m * n, dt);
    k::Gemm(false, false, m, n, n, 1.0, Xc, m, Sn, n, 0.0, X, m);
    // trace((A)_+) = the nuclear norm of X_rest
    double nuc = 0;
    {
      DVec dg = DVec::Empty(n, dt);
      EPS_HIP(hipMemcpy2DAsync(dg.data(), DTypeSize(dt), Sn.data(), static_cast<size_t>(n + 1) * DTypeSize(dt),
                               DTypeSize(dt), static_cast<size_t>(n), hipMemcpyDeviceToDevice, rt.stream()));
      for (double v : dg.ToHost()) nuc += v;
    }
    if (kd > 0) {
      DVec shr = DVec::Full(k, -tau, dt), pos = DVec::Empty(k, dt);
      k::Axpby(shr, 1.0, sig, 1.0);
      k::MaxZero(pos, shr);
      k::ColScaleByRatio(U, m, k, sig, pos);  // u_i (sigma_i - tau)_+ / sigma_i (zero columns stay zero)
      k::Gemm(false, true, m, n, k, 1.0, U, m, Wb, n, 1.0, X, m);
      for (int64_t i = 0; i < k; ++i)
        if (keep[static_cast<size_t>(i)] != 0.0) nuc += std::max(sh[static_cast<size_t>(i)] - tau, 0.0);
    }
    // ---- 7. the optimality condition of the result
    double inner = 0, pnorm = 0;
    {
      DVec P = y.Clone();
      k::Axpby(P, -1.0 / tau, X, 1.0 / tau);  // P = (Y - X) / tau
      rt.ResetSlots();
      const int sd = rt.NewSlot();
      k::Dot(X, P, rt.SlotPtr(sd), false);
      DVec v = DVec::Empty(n, dt), w = DVec::Empty(m, dt);
      k::FillHash(v, 0xFACEull);
      int slot = rt.NewSlot();
      k::SumSq(v, rt.SlotPtr(slot), false);
      k::ScaleByInvNorm(v, v, rt.SlotPtr(slot));
      for (int it = 0; it < 20; ++it) {
        k::Gemv(false, m, n, 1.0, P, m, v, 0.0, w);
        k::Gemv(true, m, n, 1.0, P, m, w, 0.0, v);
        slot = rt.NewSlot();
        k::SumSq(v, rt.SlotPtr(slot), false);
        k::ScaleByInvNorm(v, v, rt.SlotPtr(slot));
      }
      rt.FetchSlots();
      inner = rt.SlotValue(sd);
      pnorm = std::sqrt(std::sqrt(rt.SlotValue(slot)));
    }
    const double ptol = dt == F32 ? 2e-2 : 1e-6, itol = dt == F32 ? 2e-3 : 1e-8;
    const bool ok = std::isfinite(pnorm) && std::isfinite(inner) && pnorm <= 1.0 + ptol &&
                    std::fabs(inner - nuc) <= itol * std::max(nuc, tau);
    if (trace)
      std::fprintf(stderr, "[svd] polar route: %d + %d steps, ||P||_2 >= %.5f, <X,P> %.6g, ||X||_* %.6g: %s\n", steps,
                   ssteps, pnorm, inner, nuc, ok ? "accepted" : "REJECTED");
    if (!ok) return false;
    *Xout = X;
    return true;
  }

  // Columns of Q (rows x k) <- an orthonormal basis of their span (one-sided Jacobi on the k
  // columns: Q = W V^T, the columns of W orthogonal; a numerically null column becomes zero).
  static void Orthonormalise(const DVec& Q, int64_t rows, int64_t k) {
    DVec V = DVec::Empty(k * k, Q.dt);
    k::JacobiSvd(Q, rows, k, V, 40, false, false);
    DVec sig = DVec::Empty(k, Q.dt), one = DVec::Full(k, 1.0, Q.dt);
    k::ColNorms(Q, rows, k, sig, false);
    k::ColScaleByRatio(Q, rows, k, sig, one);
  }

  // X = sum over sigma_i > tau of (sigma_i - tau) u_i v_i^T from the leading singular block of
  // Y alone: randomized block subspace iteration on the GEMM kernels (range finder + two power
  // steps), the small problem by the Jacobi SVD, and an a-posteriori certificate - the block
  // reaches below tau with room to spare AND the remainder Y - U S V^T has spectral norm <= tau
  // (power iteration).  false: the numerical rank above tau is large (or the certificate
  // fails); the caller computes the full decomposition.
  bool ThresholdedPartialSvd(const DVec& y, double tau, DVec* X) {
    Runtime& rt = Runtime::Get();
    const int64_t m = m_, n = n_, kmax = std::min<int64_t>(512, std::min(m, n) / 4);
    const int64_t guard = 8;  // captured values that must lie below tau
    int64_t k = last_rank_ >= 0 ? (last_rank_ + guard + 8 + 31) / 32 * 32 : 32;
    k = std::min(std::max<int64_t>(k, 32), kmax);
    ProfScope prof("partial_svd", m, n);
    rt.ResetSlots();
    const int fro_slot = rt.NewSlot();
    k::SumSq(y, rt.SlotPtr(fro_slot), false);
    rt.FetchSlots();
    const double fro2 = rt.SlotValue(fro_slot);  // sum of all squared singular values
    int power = 2;  // power steps of the range finder
    // a singular pair above the threshold counts as converged when || Y v - sigma u || is at the
    // rounding level of the decomposition the full route would deliver
    const double res_tol = y.dt == F32 ? 2e-5 : 1e-10;
    for (int attempt = 0;; ++attempt) {
      // range finder with `power` power steps: Q = orth(Y (Y^T Y)^power Omega)
      DVec Z = DVec::Empty(n * k, y.dt), Q = DVec::Empty(m * k, y.dt);
      k::FillHash(Z, 0x5EEDull + static_cast<uint64_t>(attempt));
      k::Gemm(false, false, m, k, n, 1.0, y, m, Z, n, 0.0, Q, m);
      for (int it = 0; it < power; ++it) {
        Orthonormalise(Q, m, k);
        k::Gemm(true, false, n, k, m, 1.0, y, m, Q, m, 0.0, Z, n);
        Orthonormalise(Z, n, k);
        k::Gemm(false, false, m, k, n, 1.0, y, m, Z, n, 0.0, Q, m);
      }
      Orthonormalise(Q, m, k);
      // B = Q^T Y; its SVD through B^T (n x k): B^T = Wb Vb^T, columns of Wb = sigma_i v_i
      DVec Wb = DVec::Empty(n * k, y.dt), Vb = DVec::Empty(k * k, y.dt);
      k::Gemm(true, false, n, k, m, 1.0, y, m, Q, m, 0.0, Wb, n);
      k::JacobiSvd(Wb, n, k, Vb, 40, false, false);
      DVec sig = DVec::Empty(k, y.dt);
      k::ColNorms(Wb, n, k, sig, false);
      const std::vector<double> sh = sig.ToHost();
      int64_t above = 0;
      for (double sv : sh) above += sv > tau ? 1 : 0;
      const bool reaches_below = above + guard <= k;
      if (!reaches_below) {
        if (k >= kmax) return false;
        // Give up early when the energy outside the block alone implies more values above tau
        // than the largest block holds: every value outside is (about) at most the block's
        // smallest one, so their count is at least (outside energy - what values <= tau could
        // carry) / that bound squared.  (A wrong guess here only sends a matrix the block could
        // have handled to the full decomposition.)
        double inside = 0, smin = sh[0];
        for (double sv : sh) {
          inside += sv * sv;
          smin = std::min(smin, sv);
        }
        const double outside = fro2 - inside - static_cast<double>(std::min(m, n) - k) * tau * tau;
        if (smin > 0 && outside / (2.25 * smin * smin) > static_cast<double>(kmax)) return false;
        k = std::min(kmax, std::max(2 * k, (above + guard + 31) / 32 * 32));
        continue;
      }
      DVec U = DVec::Empty(m * k, y.dt);
      k::Gemm(false, false, m, k, k, 1.0, Q, m, Vb, k, 0.0, U, m);  // left singular vectors of the block
      {
        // convergence of the pairs that matter: Wb_i = Y^T u_i holds by construction, the other
        // half || Y v_i - sigma_i u_i || = || Y Wb_i - sigma_i^2 u_i || / sigma_i is measured
        DVec T = DVec::Empty(m * k, y.dt), U2 = U.Clone();
        k::Gemm(false, false, m, k, n, 1.0, y, m, Wb, n, 0.0, T, m);
        DVec one = DVec::Full(k, 1.0, y.dt), sig2 = DVec::Empty(k, y.dt);
        k::DiagMul(sig2, 1.0, sig, sig, 0.0);
        k::ColScaleByRatio(U2, m, k, one, sig2);
        k::Axpby(T, -1.0, U2, 1.0);
        DVec rn = DVec::Empty(k, y.dt);
        k::ColNorms(T, m, k, rn, false);
        const std::vector<double> rh = rn.ToHost();
        double smax = 0;
        for (double sv : sh) smax = std::max(smax, sv);
        bool converged = true;
        for (int64_t i = 0; i < k; ++i)
          if (sh[i] > 0.99 * tau && !(rh[i] / sh[i] <= res_tol * smax)) converged = false;
        if (!converged) {
          if (power < 6) {
            power += 2;
          } else if (k < kmax) {
            k = std::min(kmax, 2 * k);
            power = 2;
          } else {
            return false;
          }
          continue;
        }
      }
      // certificate: || Y - U Wb^T ||_2 by power iteration on R^T R (R x = Y x - U (Wb^T x))
      DVec v = DVec::Empty(n, y.dt), w = DVec::Empty(m, y.dt), z = DVec::Empty(n, y.dt), c = DVec::Empty(k, y.dt);
      k::FillHash(v, 0xC0FFEEull);
      double est = 0;
      for (int it = 0; it < 24; ++it) {
        rt.ResetSlots();
        const int slot = rt.NewSlot();
        k::SumSq(v, rt.SlotPtr(slot), false);
        rt.FetchSlots();
        const double nv = std::sqrt(rt.SlotValue(slot));
        if (!(nv > 0)) break;
        k::Axpby(v, 1.0 / nv, v, 0.0);
        k::Gemv(false, m, n, 1.0, y, m, v, 0.0, w);   // w = Y v
        k::Gemv(true, n, k, 1.0, Wb, n, v, 0.0, c);   // c = Wb^T v
        k::Gemv(false, m, k, -1.0, U, m, c, 1.0, w);  // w -= U c
        rt.ResetSlots();
        const int s2 = rt.NewSlot();
        k::SumSq(w, rt.SlotPtr(s2), false);
        rt.FetchSlots();
        est = std::sqrt(rt.SlotValue(s2));           // ||R v|| with ||v|| = 1: a lower bound, increasing
        k::Gemv(true, m, n, 1.0, y, m, w, 0.0, z);    // z = Y^T w
        k::Gemv(true, m, k, 1.0, U, m, w, 0.0, c);    // c = U^T w
        k::Gemv(false, n, k, -1.0, Wb, n, c, 1.0, z); // z -= Wb c
        std::swap(v, z);
      }
      // `est` is a LOWER bound on ||R||_2 (24 steps from a fixed start resolve the top of a flat
      // bulk to a few per cent, a separated value to rounding): accept only with that margin in
      // hand, so that a remainder whose top sits just above tau is never kept out of the result
      if (!(est * 1.05 <= tau)) {
        if (k >= kmax) return false;
        k = std::min(kmax, 2 * k);
        continue;
      }
      // X = U diag(max(sigma - tau, 0) / sigma) Wb^T
      DVec shr = DVec::Full(k, -tau, y.dt);
      k::Axpby(shr, 1.0, sig, 1.0);
      DVec pos = DVec::Empty(k, y.dt);
      k::MaxZero(pos, shr);
      k::ColScaleByRatio(U, m, k, sig, pos);  // U[:, i] *= pos_i / sigma_i (0 where sigma_i = 0)
      DVec Xo = DVec::Empty(m * n, y.dt);
      k::Gemm(false, true, m, n, k, 1.0, U, m, Wb, n, 0.0, Xo, m);
      *X = Xo;
      last_rank_ = above;
      return true;
    }
  }

  void InitEigenProx(double lambda) {  // ortho_invariant.cc:76-98
    // The reference sizes the nested prox with min(m, n) and feeds it n values (:36-50,:77); n is
    // the well-defined reading.
    const int num_args = epigraph_ ? 2 : 1;
    alpha_ = epigraph_ ? 1.0 : 1 / std::sqrt(lambda);
    eigen_f_ = pb::ProxFunction();
    eigen_f_.prox_function_type = eigen_prox_type_;
    eigen_f_.alpha = 1;
    pb::Size sz;
    sz.dim = {static_cast<int32_t>(n_), 1};
    eigen_f_.arg_size.push_back(sz);
    eigen_H_ = AffineOperator();
    eigen_A_ = AffineOperator();
    for (int i = 0; i < num_args; ++i) {
      const std::string key = affine::arg_key(i);
      const int64_t ni = i == 0 ? n_ : 1;
      eigen_H_.A(key, key) = LinearMap::Identity(ni);
      eigen_A_.A(key, key) = LinearMap::Scalar(alpha_, ni);
    }
    eigen_data_.reset(new DataMap(dtype_));
    eigen_prox_ = CreateProxOperator(eigen_prox_type_, epigraph_);
    eigen_prox_->Init(ProxOperatorArg(eigen_f_, eigen_data_.get(), eigen_H_, eigen_A_));
  }

  int eigen_prox_type_;
  bool symmetric_part_, add_residual_, epigraph_;
  int64_t m_ = 0, n_ = 0;
  DType dtype_ = F32;
  double alpha_ = 1;
  pb::ProxFunction eigen_f_;
  AffineOperator eigen_H_, eigen_A_;
  std::unique_ptr<DataMap> eigen_data_;
  std::unique_ptr<ProxOperator> eigen_prox_;
  DVec V_prev_;  // right singular vectors of the previous application (warm start)
  // one-sided form (fp32 block Jacobi without an accumulated factor, ApplyOneSided): the orthonormal
  // basis the previous application produced - right vectors (n x n) or left vectors (m x n)
  DVec basis_prev_;
  bool basis_is_left_ = false;
  // polar route (PolarNuclearProx)
  static constexpr int64_t kPolarMinSize = 1024;
  static constexpr int64_t kPolarCalls = 4;
  bool polar_enabled_ = true;
  int64_t calls_ = 0;
  bool warm_start_ = true;
  // thresholded partial SVD (nuclear-norm prox of a large matrix)
  static constexpr int64_t kPartialMinSize = 512;
  bool partial_enabled_ = true;
  int partial_backoff_ = 0;   // applications left before the partial route is tried again
  int partial_fail_streak_ = 0;
  int64_t last_rank_ = -1;    // singular values above lambda at the last successful partial call
};

#define EPS_ORTHO_OPERATOR(NAME, ...)                         \
  class NAME final : public OrthoInvariantProx {              \
   public:                                                    \
    NAME() : OrthoInvariantProx(__VA_ARGS__) {}               \
  }

EPS_ORTHO_OPERATOR(NormNuclearProx, pb::ProxFunction::NORM_1);                          // norm_nuclear.cc
EPS_ORTHO_OPERATOR(NormNuclearEpigraph, pb::ProxFunction::NORM_1, false, false, true);
EPS_ORTHO_OPERATOR(LambdaMaxProx, pb::ProxFunction::MAX, true);                         // lambda_max.cc
EPS_ORTHO_OPERATOR(LambdaMaxEpigraph, pb::ProxFunction::MAX, true, false, true);
EPS_ORTHO_OPERATOR(NegLogDetProx, pb::ProxFunction::SUM_NEG_LOG, true);                 // neg_log_det.cc
EPS_ORTHO_OPERATOR(NegLogDetEpigraph, pb::ProxFunction::SUM_NEG_LOG, true, false, true);
EPS_ORTHO_OPERATOR(SemidefiniteProx, pb::ProxFunction::NON_NEGATIVE, true, true);       // semidefinite.cc
REGISTER_PROX_OPERATOR(NORM_NUCLEAR, NormNuclearProx);
REGISTER_EPIGRAPH_OPERATOR(NORM_NUCLEAR, NormNuclearEpigraph);
REGISTER_PROX_OPERATOR(LAMBDA_MAX, LambdaMaxProx);
REGISTER_EPIGRAPH_OPERATOR(LAMBDA_MAX, LambdaMaxEpigraph);
REGISTER_PROX_OPERATOR(NEG_LOG_DET, NegLogDetProx);
REGISTER_EPIGRAPH_OPERATOR(NEG_LOG_DET, NegLogDetEpigraph);
REGISTER_PROX_OPERATOR(SEMIDEFINITE, SemidefiniteProx);

}  // namespace
}  // namespace eps

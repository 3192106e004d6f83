#include "admm.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "comm.h"
#include "kernels.h"

namespace eps {

GraphStats& GraphStats::Get() {
  static GraphStats g;
  return g;
}


namespace {

double Now() {
  using clock = std::chrono::steady_clock;
  return std::chrono::duration<double>(clock::now().time_since_epoch()).count();
}

// Sharded solves: the "arg:<k>" rows of a prox operator's H are private to that operator.  An
// arg row fed from a sharded variable through an elementwise map is itself sharded; through a
// dense / Kronecker map it is a contraction into a replicated row (comm.h).
std::set<std::string> InferShardedArgs(const BlockMatrix& H) {
  std::set<std::string> out;
  const ShardSpec& sh = ShardSpec::Get();
  if (!sh.active()) return out;
  if (sh.consensus_terms() && !H.data().empty()) {
    // consensus form: a term over sharded variables only is this rank's own term, and so are
    // its argument rows, whatever the type of the maps into them
    bool all_sharded = true;
    for (const auto& col : H.data()) all_sharded = all_sharded && sh.IsSharded(col.first);
    if (all_sharded) return H.row_keys();
  }
  std::map<std::string, int> vote;  // 1 sharded, 2 replicated
  for (const auto& col : H.data()) {
    if (!sh.IsSharded(col.first)) continue;
    for (const auto& row : col.second) {
      const ImplType t = row.second.impl().type();
      const int v = (t == SCALAR_MATRIX || t == DIAGONAL_MATRIX) ? 1 : 2;
      int& cur = vote[row.first];
      EPS_CHECK_MSG(cur == 0 || cur == v, "row " << row.first
                                                  << " mixes elementwise and dense maps of sharded variables");
      cur = v;
    }
  }
  for (const auto& kv : vote)
    if (kv.second == 1) out.insert(kv.first);
  return out;
}

// Global row / column counts of a block matrix whose sharded keys hold per-rank slices.
void GlobalDims(const BlockMatrix& A, int64_t* m, int64_t* n) {
  const ShardSpec& sh = ShardSpec::Get();
  if (!sh.active()) {
    *m = A.m();
    *n = A.n();
    return;
  }
  double loc[2] = {0, 0}, rep[2] = {0, 0};
  std::set<std::string> seen;
  for (const auto& col : A.data()) {
    const int64_t cn = col.second.begin()->second.impl().n();
    (sh.IsSharded(col.first) ? loc : rep)[1] += cn;
    for (const auto& row : col.second)
      if (seen.insert(row.first).second)
        (sh.IsSharded(row.first) ? loc : rep)[0] += row.second.impl().m();
  }
  DVec d = DVec::FromHost(loc, 2, F64);
  Runtime::Get().comm()->AllReduceSum(d);
  std::vector<double> g = d.ToHost();
  *m = static_cast<int64_t>(g[0] + rep[0] + 0.5);
  *n = static_cast<int64_t>(g[1] + rep[1] + 0.5);
}

}  // namespace

Solver::Solver(pb::Problem problem, std::shared_ptr<DataMap> data, pb::SolverParams params)
    : problem_(std::move(problem)), data_(std::move(data)), params_(params) {}

void Solver::LogStatus() {  // reference prox_admm.cc:219-230
  if (!params_.verbose || !log_) return;
  char buf[256];
  std::snprintf(buf, sizeof(buf), "iter=%d residuals primal=%.2e [%.2e] dual=%.2e [%.2e]",
                status_.num_iterations, status_.r_norm, status_.epsilon_primal, status_.s_norm,
                status_.epsilon_dual);
  log_(buf);
}

void Solver::FinishResiduals(double r, double s, double eps_pri, double eps_dual) {
  status_.r_norm = r;
  status_.s_norm = s;
  status_.epsilon_primal = eps_pri;
  status_.epsilon_dual = eps_dual;
  if (r <= eps_pri && s <= eps_dual && !params_.ignore_stopping_criteria)
    status_.state = pb::SolverStatus::OPTIMAL;
  else
    status_.state = pb::SolverStatus::RUNNING;
  status_.num_iterations = iter_;
}

int Solver::Run(int max_sweeps) {
  EPS_CHECK_MSG(initialized_, "Solver::Run before Init");
  SetCurrentDType(data_->dtype());
  const double t0 = Now();
  int done = 0;
  const int epoch = params_.epoch_iterations > 0 ? params_.epoch_iterations : 1;
  const int log_every = params_.log_iterations > 0 ? params_.log_iterations : 1;
  // the sweeps from iteration `it` up to and including the next one that is followed by a host
  // decision (residual check; log line when verbose), clipped to the limits; 0 = none left
  auto batch_size = [&](int it, int done_so_far) {
    if (it >= params_.max_iterations || (max_sweeps >= 0 && done_so_far >= max_sweeps)) return 0;
    int batch = 1;
    while ((it + batch - 1) % epoch != 0 && !(params_.verbose && (it + batch - 1) % log_every == 0))
      ++batch;
    if (batch > params_.max_iterations - it) batch = params_.max_iterations - it;
    if (max_sweeps >= 0 && batch > max_sweeps - done_so_far) batch = max_sweeps - done_so_far;
    return batch;
  };
  int speculated = 0;  // sweeps of the NEXT batch already enqueued behind a pending check
  // A speculative batch is wasted work when its check says OPTIMAL, so none is started once the
  // last check was within a factor 4 of both tolerances (the residuals fall geometrically, the
  // next check is then likely the last): the steady state gets the overlap, the time to OPTIMAL
  // does not pay for it.
  auto far_from_optimal = [&] {
    if (status_.epsilon_primal <= 0 || status_.epsilon_dual <= 0) return true;  // no check yet
    return status_.r_norm > 4 * status_.epsilon_primal || status_.s_norm > 4 * status_.epsilon_dual;
  };
  while (!finished_ && iter_ < params_.max_iterations && (max_sweeps < 0 || done < max_sweeps)) {
    int batch = speculated;
    if (batch == 0) {
      batch = batch_size(iter_, done);
      SweepBatch(batch);
    }
    speculated = 0;
    done += batch;
    iter_ += batch - 1;  // index of the last sweep of the batch
    if (iter_ % epoch == 0) {
      const int next = (PipelinedChecks() && (params_.ignore_stopping_criteria || far_from_optimal()))
                           ? batch_size(iter_ + 1, done)
                           : 0;
      if (next > 0) {
        BeginResiduals();
        SaveSnapshot();
        SweepBatch(next);  // runs on the device while the host waits for the check's scalars
        speculated = next;
        EndResiduals();
        if (status_.state == pb::SolverStatus::OPTIMAL) {
          RestoreSnapshot();  // the speculative sweeps are discarded (and were never counted)
          finished_ = true;
          break;
        }
      } else {
        ComputeResiduals();
        if (status_.state == pb::SolverStatus::OPTIMAL) {
          finished_ = true;
          break;
        }
      }
    }
    if (iter_ % log_every == 0) LogStatus();
    ++iter_;
  }
  if (!finished_ && iter_ == params_.max_iterations) {
    ComputeResiduals();
    status_.state = pb::SolverStatus::MAX_ITERATIONS_REACHED;
    finished_ = true;
  }
  Runtime::Get().Sync();
  if (Runtime::Get().peer()) Runtime::Get().peer()->CheckError();
  loop_seconds_ += Now() - t0;
  if (finished_) LogStatus();
  status_.init_time = init_seconds_;
  status_.total_time = init_seconds_ + loop_seconds_;
  return done;
}

void Solver::Solve() {
  const double t0 = Now();
  Init();
  Runtime::Get().Sync();
  init_seconds_ = Now() - t0;
  Run(-1);
}

// ---------------------------------------------------------------------------------------------------
// The generic operator path's sweeps between two residual checks as ONE hipGraph (small problems
// are bound by launch latency: 30-60 launches of a few microseconds per sweep).  BlockVector
// blocks are replaced, never mutated, so a sweep ends in other buffers than it started from; the
// capture therefore runs on CANONICAL copies of the state (shared with this object, hence
// copy-on-write keeps the sweep from writing into them) and ends with device-to-device copies of
// the final blocks back into them: every replay starts and ends at the same addresses.  All
// buffers the captured launches touch are fenced off by a Runtime hold for the graph's life.
// Only for problems whose operators are all ProxOperator::CaptureSafe and whose state is small.
// ---------------------------------------------------------------------------------------------------
class SweepGraph {
 public:
  ~SweepGraph() { Reset(); }

  void Reset() {
    if (exec_) (void)hipGraphExecDestroy(exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    exec_ = nullptr;
    graph_ = nullptr;
    len_ = 0;
    if (!held_.empty()) Runtime::Get().ReturnHeld(&held_);  // (stream order keeps the reuse safe)
    canon_.clear();
  }
  void ClearFailure() { failed_ = false; }

  // MEASURED, AND OFF BY DEFAULT (round 3, ROCm 7.2, MI355X): replaying the generic sweeps from a
  // graph is bit-identical to the eager launches (tests) and buys nothing - 1000 sweeps of the
  // reference's lasso_sparse / mnist / mv_lasso problems take 0.102 / 0.312 / 0.104 s replayed
  // against 0.093 / 0.320 / 0.104 s eager: the host already runs far ahead of the stream, and the
  // gap between two dependent kernels is the same inside a graph - while capturing and
  // instantiating a batch costs 4-5 ms, more than most solves of these problems take in all.
  // EPSILON_HIP_GRAPH_GENERIC (option "graph_generic") = 1: once a run has lasted kEagerFirst
  // sweeps; = 2: after the first sweep (tests); 0 / unset: never.  Read per call.
  static constexpr int kEagerFirst = 50;
  static int Mode() {
    const char* e = std::getenv("EPSILON_HIP_GRAPH_GENERIC");
    return e ? std::atoi(e) : 0;
  }
  static int EagerSweepsFirst() { return Mode() >= 2 ? 1 : kEagerFirst; }

  bool Wanted(int count, const std::vector<BlockVector*>& state) const {
    const int gmode = Mode();
    Runtime& rt = Runtime::Get();
    if (gmode == 0 || count < 2 || failed_) return false;
    if (rt.profiling() || rt.capturing() || rt.holding() || ShardSpec::Get().active()) return false;
    // launch-bound problems only: the copies back cost a pass over the state per batch
    int64_t bytes = 0;
    for (const BlockVector* v : state)
      for (const auto& kv : v->data()) bytes += static_cast<int64_t>(kv.second.bytes());
    return bytes <= (int64_t(64) << 20);
  }

  // Replays `count` sweeps (capturing first when there is no graph of that length, or when
  // somebody re-bound the state since).  false: nothing was enqueued - the caller launches eagerly.
  // The first `n_prev` handles are "previous iterate" copies that every sweep overwrites before it
  // reads them (y_prev = y): their layout before the capture does not matter.
  template <class SweepFn>
  bool Run(int count, const std::vector<BlockVector*>& state, size_t n_prev, SweepFn sweep) {
    if (exec_ == nullptr || len_ != count || !IsCanonical(state)) Capture(count, state, n_prev, sweep);
    if (exec_ == nullptr) return false;
    EPS_HIP(hipGraphLaunch(exec_, Runtime::Get().stream()));
    GraphStats::Get().replayed_sweeps += count;
    return true;
  }

 private:
  static bool SameBuffers(const BlockVector& a, const BlockVector& b) {
    if (a.data().size() != b.data().size()) return false;
    auto ia = a.data().begin();
    auto ib = b.data().begin();
    for (; ia != a.data().end(); ++ia, ++ib)
      if (ia->first != ib->first || ia->second.data() != ib->second.data() || ia->second.n != ib->second.n)
        return false;
    return true;
  }
  static BlockVector CloneBlocks(const BlockVector& v) {
    BlockVector c;
    for (const auto& kv : v.data()) c.Set(kv.first, kv.second.Clone());
    return c;
  }
  // dst (canonical) <- src, block by block, on the stream; false: the layouts differ
  static bool CopyBlocksBack(const BlockVector& dst, const BlockVector& src) {
    if (dst.data().size() != src.data().size()) return false;
    hipStream_t s = Runtime::Get().stream();
    auto id = dst.data().begin();
    auto is = src.data().begin();
    for (; id != dst.data().end(); ++id, ++is) {
      if (id->first != is->first || id->second.n != is->second.n || id->second.dt != is->second.dt) return false;
      if (id->second.data() == is->second.data() || id->second.n == 0) continue;
      EPS_HIP(hipMemcpyAsync(const_cast<void*>(static_cast<const void*>(id->second.data())), is->second.data(),
                             id->second.bytes(), hipMemcpyDeviceToDevice, s));
    }
    return true;
  }
  bool IsCanonical(const std::vector<BlockVector*>& state) const {
    if (state.size() != canon_.size()) return false;
    for (size_t i = 0; i < state.size(); ++i)
      if (!SameBuffers(*state[i], canon_[i])) return false;
    return true;
  }

  // `state`: every BlockVector a sweep reads from the one before it, "previous iterate" handles
  // FIRST (a block the last sweep left alone may still be the canonical buffer of its successor,
  // which the copies back overwrite afterwards).
  template <class SweepFn>
  void Capture(int count, const std::vector<BlockVector*>& state, size_t n_prev, SweepFn sweep) {
    Runtime& rt = Runtime::Get();
    hipStream_t s = rt.stream();
    Reset();
    std::vector<BlockVector> saved;  // the handles as they are, should the capture be abandoned
    for (BlockVector* v : state) {
      saved.push_back(*v);
      canon_.push_back(CloneBlocks(*v));
    }
    rt.BeginHold();
    for (size_t i = 0; i < state.size(); ++i) *state[i] = canon_[i];
    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess;
    hipGraph_t graph = nullptr;
    std::string why;
    if (ok) {
      rt.set_capturing(true);
      try {
        for (int i = 0; i < count; ++i) sweep();
        for (size_t i = 0; i < state.size() && ok; ++i) {
          ok = CopyBlocksBack(canon_[i], *state[i]);
          if (!ok && i < n_prev) {  // written before read: its canonical buffers take today's layout
            BlockVector fresh;
            for (const auto& kv : state[i]->data()) fresh.Set(kv.first, DVec::Empty(kv.second.n, kv.second.dt));
            canon_[i] = fresh;
            ok = CopyBlocksBack(canon_[i], *state[i]);
          }
          if (!ok) {
            why = "the block layout of state handle " + std::to_string(i) + " changed during the sweeps: had";
            for (const auto& kv : canon_[i].data()) why += " " + kv.first + ":" + std::to_string(kv.second.n);
            why += "; has";
            for (const auto& kv : state[i]->data()) why += " " + kv.first + ":" + std::to_string(kv.second.n);
          }
        }
      } catch (const std::exception& e) {
        ok = false;
        why = e.what();
      } catch (...) {
        ok = false;
        why = "exception";
      }
      rt.set_capturing(false);
      const hipError_t ee = hipStreamEndCapture(s, &graph);
      if (ee != hipSuccess || graph == nullptr) {
        if (ok) why = std::string("hipStreamEndCapture: ") + hipGetErrorString(ee);
        ok = false;
      }
    } else {
      why = "hipStreamBeginCapture failed";
    }
    if (ok) {
      const hipError_t ei = hipGraphInstantiate(&exec_, graph, nullptr, nullptr, 0);
      if (ei != hipSuccess) {
        ok = false;
        why = std::string("hipGraphInstantiate: ") + hipGetErrorString(ei);
      }
    }
    if (!ok) {
      static const bool trace = std::getenv("EPSILON_HIP_GRAPH_TRACE") != nullptr;
      if (trace) std::fprintf(stderr, "[graph] capture of %d generic sweeps abandoned: %s\n", count, why.c_str());
      (void)hipGetLastError();
      if (graph) (void)hipGraphDestroy(graph);
      exec_ = nullptr;
      for (size_t i = 0; i < state.size(); ++i) *state[i] = saved[i];  // nothing ran: the state is where it was
      std::vector<std::pair<size_t, void*>> held = rt.EndHold();
      rt.ReturnHeld(&held);
      Reset();
      failed_ = true;
      return;
    }
    graph_ = graph;
    len_ = count;
    ++GraphStats::Get().captures;
    for (size_t i = 0; i < state.size(); ++i) *state[i] = canon_[i];  // the last sweep's handles go to the held pool
    saved.clear();
    held_ = rt.EndHold();
  }

  hipGraph_t graph_ = nullptr;
  hipGraphExec_t exec_ = nullptr;
  int len_ = 0;
  bool failed_ = false;  // a capture did not work out: the solver stays on eager launches
  std::vector<std::pair<size_t, void*>> held_;
  std::vector<BlockVector> canon_;
};


// ---------------------------------------------------------------------------------------------------
// ProxADMMSolver (reference algorithms/prox_admm.cc)
// ---------------------------------------------------------------------------------------------------

class ProxADMMSolver final : public Solver {
 public:
  using Solver::Solver;
  ~ProxADMMSolver() override { ResetGraph(); }

  void Init() override {  // :110-129
    SetCurrentDType(data_->dtype());
    const double t0 = Now();
    if (op_cache_.size() > 64) op_cache_.Clear();
    OpCacheScope cache_scope(&op_cache_);
    static const bool trace = std::getenv("EPSILON_HIP_INIT_TRACE") != nullptr;
    auto mark = [&](const char* what) {  // host wall clock + device drain, debugging aid only
      if (!trace) return;
      const double th = Now();
      Runtime::Get().Sync();
      std::fprintf(stderr, "[init] %-16s host %.2f ms  drained %.2f ms\n", what, 1e3 * (th - t0),
                   1e3 * (Now() - t0));
    };
    InitConstraints();
    mark("constraints");
    InitProxOperators();
    mark("prox operators");
    if (!params_.warm_start || !vars_initialized_) {
      InitVariables();
      vars_initialized_ = true;
    }
    iter_ = 0;
    finished_ = false;
    status_ = pb::SolverStatus();
    initialized_ = true;
    TryEnableFused();
    mark("fused state");
    capture_safe_ = true;
    for (const auto& op : prox_) capture_safe_ = capture_safe_ && op->CaptureSafe();
    eager_sweeps_ = 0;
    gg_.ClearFailure();
    if (params_.verbose && log_) {
      char buf[128];
      std::snprintf(buf, sizeof(buf), "constraints, m = %lld, variables, n = %lld",
                    static_cast<long long>(m_), static_cast<long long>(n_));
      log_(buf);
    }
    Runtime::Get().Sync();
    init_seconds_ = Now() - t0;
  }

  BlockVector GetSolution() override {  // :171-176
    BlockVector r;
    for (int i = 0; i < N_; ++i) r += x_[i];
    return r;
  }

 protected:
  void InitConstraints() {  // :25-43
    A_ = BlockMatrix();
    b_ = BlockVector();
    for (size_t i = 0; i < problem_.constraint.size(); ++i) {
      const pb::Expression& constr = problem_.constraint[i];
      EPS_CHECK_MSG(constr.expression_type == pb::Expression::INDICATOR, "constraint is not an indicator");
      EPS_CHECK_MSG(constr.cone_type == 1, "constraint cone is not ZERO");
      EPS_CHECK(constr.arg.size() == 1);
      affine::BuildAffineOperator(constr.arg[0], data_.get(), affine::constraint_key(i), &A_, &b_);
    }
    AT_ = A_.Transpose();
    GlobalDims(A_, &m_, &n_);
  }

  void InitProxOperators() {  // :45-94
    EPS_CHECK_MSG(problem_.objective.expression_type == pb::Expression::ADD, "objective is not ADD");
    N_ = static_cast<int>(problem_.objective.arg.size());
    EPS_CHECK_MSG(params_.rho == 1, "rho != 1 is not supported (reference prox_admm.cc:50)");
    const double sqrt_rho = std::sqrt(params_.rho);
    prox_.clear();
    AiT_.clear();
    arg_shards_.clear();
    term_per_rank_.clear();
    std::set<std::string> constr_vars = A_.col_keys();
    for (int i = 0; i < N_; ++i) {
      const pb::Expression& f_expr = problem_.objective.arg[i];
      EPS_CHECK_MSG(f_expr.expression_type == pb::Expression::PROX_FUNCTION,
                    "objective term " << i << " is not a PROX_FUNCTION");
      AffineOperator H;
      for (size_t k = 0; k < f_expr.arg.size(); ++k)
        affine::BuildAffineOperator(f_expr.arg[k], data_.get(), affine::arg_key(k), &H.A, &H.b);
      AffineOperator A;
      std::map<std::string, const pb::Expression*> vars;
      GetVariables(f_expr, &vars);
      for (const auto& var : vars) {
        if (constr_vars.find(var.first) == constr_vars.end()) continue;
        for (const auto& it : A_.col(var.first)) A.A(it.first, var.first) = sqrt_rho * it.second;
      }
      prox_.emplace_back(CreateProxOperator(f_expr.prox_function.prox_function_type,
                                            f_expr.prox_function.epigraph));
      arg_shards_.push_back(InferShardedArgs(H.A));
      {
        // consensus form: is this one of the per-rank terms f_g(x_g)?
        const ShardSpec& sh = ShardSpec::Get();
        bool own = sh.active() && sh.consensus_terms() && !vars.empty();
        for (const auto& var : vars) own = own && sh.IsSharded(var.first);
        term_per_rank_.push_back(own);
      }
      {
        LocalShardScope scope(arg_shards_.back());
        prox_.back()->Init(ProxOperatorArg(f_expr.prox_function, data_.get(), H, A));
      }
      AiT_.push_back(A.A.Transpose());
    }
  }

  void InitVariables() {  // :96-108
    x_.assign(N_, BlockVector());
    y_.assign(N_, BlockVector());
    u_ = BlockVector();
    for (size_t i = 0; i < problem_.constraint.size(); ++i) {
      u_.Set(affine::constraint_key(i),
             DVec::Zeros(GetDimension(problem_.constraint[i].arg[0]), data_->dtype()));
    }
  }

  // ---- fused sweep: "least squares + separable threshold" (kernels_fused.hip) ------------------
  // Recognised structure (the compiled lasso): two terms [SUM_SQUARE with a dense
  // argument map, scaled-zone prox with scalar maps], one consensus constraint a0 x' + a1 x = 0
  // with a0 = 1 and no constant.  The sweep is then: one fused pass over A (back substitution of
  // this sweep, elementwise chain, forward substitution of the next sweep), a partial-sum
  // reduction (+ the all-reduce when sharded) and the apply of the cached inverse.
  void TryEnableFused() {
    fused_ = false;
    ResetGraph();
    const char* env = std::getenv("EPSILON_HIP_FUSED");
    if (env && env[0] == '0') return;
    if (N_ != 2 || problem_.constraint.size() != 1) return;
    if (!b_.data().empty()) return;
    // consensus form: the threshold step averages over the ranks, which the fused pass does not
    if (ShardSpec::Get().active() && ShardSpec::Get().consensus_terms()) return;
    FusedState f;
    if (!prox_[0]->DescribeLeastSquares(&f.ls) || !prox_[1]->DescribeScaledZone(&f.sz)) return;
    if ((f.sz.alpha_vec.n > 0 && f.sz.alpha_vec.dt != data_->dtype()) ||
        (f.sz.beta_vec.n > 0 && f.sz.beta_vec.dt != data_->dtype()))
      return;
    const std::string ck = affine::constraint_key(0);
    if (f.ls.constraint_key != ck || f.sz.constraint_key != ck) return;
    if (A_.data().size() != 2 || !A_.has_key(ck, f.ls.var_key) || !A_.has_key(ck, f.sz.var_key))
      return;
    const LinearMap& A0 = A_(ck, f.ls.var_key);
    const LinearMap& A1 = A_(ck, f.sz.var_key);
    if (A0.impl().type() != SCALAR_MATRIX || A1.impl().type() != SCALAR_MATRIX) return;
    if (GetScalar(A0) != 1.0) return;
    f.a1 = GetScalar(A1);
    const DenseMatrixImpl& L = *f.ls.L_arg_var;
    if (L.trans()) return;
    f.m = L.rows();
    f.n = L.cols();
    if (A0.impl().n() != f.n || A1.impl().n() != f.n) return;
    if (!k::LassoFusedSupported(f.m, f.n, L.data(), L.rows())) return;
    if (f.ls.rhs_arg.n != 0 && f.ls.rhs_arg.n != f.m) return;
    const DType dt = data_->dtype();
    // the six state vectors are slices of ONE buffer, so that a residual check can snapshot the
    // iterates with a single copy (pipelined checks, Solver::Run)
    const int64_t npad = (f.n + 63) / 64 * 64;
    f.state_all = DVec::Zeros(6 * npad, dt);
    f.snapshot = DVec::Empty(6 * npad, dt);
    f.norm_work = DVec::Zeros(64 * 5 + 1, F64);
    int next_slice = 0;
    auto state = [&](const BlockVector& src, const std::string& key) {
      DVec v = f.state_all.Slice(static_cast<int64_t>(next_slice++) * npad, f.n);
      if (src.has_key(key)) k::Copy(v, src(key));
      return v;
    };
    f.x0 = state(x_[0], f.ls.var_key);
    f.x1 = state(x_[1], f.sz.var_key);
    f.y0 = state(y_[0], ck);
    f.y1 = state(y_[1], ck);
    f.u = state(u_, ck);
    f.y1prev = state(BlockVector(), ck);
    f.p = DVec::Zeros(f.m, dt);
    f.grid = k::LassoFusedGrid(f.m, f.n, dt);
    f.tpart = DVec::Empty(static_cast<int64_t>(f.grid) * f.m, dt);
    {
      Comm* comm = Runtime::Get().comm();
      PeerExchange* px = Runtime::Get().peer();
      const ShardSpec& sh = ShardSpec::Get();
      const bool sharded = sh.active() && sh.IsSharded(f.ls.var_key);
      // one-shot peer-write exchange inside the sweep's own kernels (kernels_peer.hip) when the
      // ranks share a window and the m-float message fits its slots; RCCL collectives otherwise
      // (a granule carries 32 value bits: an f64 value takes two)
      f.use_peer = sharded && px != nullptr && f.m * (dt == F64 ? 2 : 1) <= px->slot() && f.ls.Dinv_arg != nullptr &&
                   !f.ls.Dinv_arg->trans() && f.ls.Dinv_arg->rows() == f.m;
      const int G = f.use_peer ? px->view().G : (comm ? comm->size() : 1);
      f.slab = ((f.m + G - 1) / G + 3) / 4 * 4;
      f.wpad = DVec::Zeros(f.slab * G, dt);
      f.wslice = DVec::Zeros(f.slab, dt);
      f.w = f.wpad.Slice(0, f.m);  // the gathered vector IS w (first m entries)
      // the inverse is applied by row slabs + all-gather from 3 ranks up; with 2 ranks the
      // symmetric apply of the whole matrix reads the same m^2/2 entries and needs no exchange
      const char* e = std::getenv("EPSILON_HIP_SHARDED_APPLY");
      const bool want_slab = e ? e[0] != 'r' : G >= 3;
      f.peer_slab = f.use_peer && want_slab &&
                    k::PeerSlabApplySupported(px->view(), f.m, f.slab, f.ls.Dinv_arg->data(), f.m);
    }
    {
      const DenseMatrixImpl& D = *f.ls.Dinv_arg;
      if (D.symmetric() && D.rows() == f.m && D.rows() >= 1024 && !D.trans()) {
        f.symv_work = DVec::Empty(k::SymvWorkspace(f.m), dt);
        // the apply reads a tile-packed copy of the lower tiles (EPSILON_HIP_SYMV_PACKED=0: the
        // matrix as it lies): +m^2/2 values of memory for a tenth of a millisecond at Init
        static const bool packed = [] {
          const char* e = std::getenv("EPSILON_HIP_SYMV_PACKED");
          return !(e && e[0] == '0');
        }();
        if (packed) f.symv_packed = k::SymvPack(f.m, D.data(), f.m);
      }
    }
    ResetGraph();
    fs_ = f;
    // the generic containers become views of the fused state
    x_[0] = BlockVector();
    x_[0].Set(fs_.ls.var_key, fs_.x0);
    x_[1] = BlockVector();
    x_[1].Set(fs_.sz.var_key, fs_.x1);
    y_[0] = BlockVector();
    y_[0].Set(ck, fs_.y0);
    y_[1] = BlockVector();
    y_[1].Set(ck, fs_.y1);
    u_ = BlockVector();
    u_.Set(ck, fs_.u);
    y_prev_.assign(2, BlockVector());
    y_prev_[1].Set(ck, fs_.y1prev);
    fused_ = true;
    FusedForward(/*from_state=*/true);
  }

  // p = rhs_arg - L(arg,var) v0 (all-reduced when sharded), w = Dinv_arg p.
  void FusedForward(bool from_state) {
    FusedState& f = fs_;
    const DenseMatrixImpl& L = *f.ls.L_arg_var;
    if (from_state) {
      // v0 = ((u - y0) - y1) + y0 of the current state, then the generic forward product
      DVec v0 = f.u.Clone();
      k::Axpby(v0, -1.0, f.y0, 1.0);
      k::Axpby(v0, -1.0, f.y1, 1.0);
      k::Axpby(v0, 1.0, f.y0, 1.0);
      L.Apply(-1.0, v0, 0.0, f.p);
    }
    const ShardSpec& sh = ShardSpec::Get();
    const bool sharded = sh.active() && sh.IsSharded(f.ls.var_key);
    bool rhs_added = false;
    if (!from_state) {
      // the constant part of the rhs rides in the reduction kernel (same rounding order as the
      // separate axpy: sum first, then + rhs); in a sharded run rank 0 alone contributes it to
      // the sum over ranks - one launch less in a sweep that is launch-latency-bound at N = 8
      const bool have_rhs = f.ls.rhs_arg.n != 0;
      const bool fold = have_rhs && (!sharded || Runtime::Get().comm()->rank() == 0);
      k::ReducePartials(f.m, f.grid, f.tpart, -L.scale(), 0.0, f.p, fold ? &f.ls.rhs_arg : nullptr);
      rhs_added = have_rhs;  // folded here, or by rank 0 into the all-reduced sum
    }
    if (sharded) Runtime::Get().comm()->AllReduceSum(f.p);
    if (f.ls.rhs_arg.n != 0 && !rhs_added) k::Axpby(f.p, 1.0, f.ls.rhs_arg, 1.0);
    const DenseMatrixImpl& D = *f.ls.Dinv_arg;
    Comm* comm = Runtime::Get().comm();
    // EPSILON_HIP_SHARDED_APPLY=replicated: every rank applies the whole inverse instead (no
    // all-gather; m^2 bytes per rank) - the cheaper form when the collective's latency exceeds
    // the apply, to be decided on the machine
    static const bool replicated_apply = [] {
      const char* e = std::getenv("EPSILON_HIP_SHARDED_APPLY");
      return e && e[0] == 'r';
    }();
    if (sharded && comm->size() > 1 && !D.trans() && D.rows() == f.m && !replicated_apply) {
      // The cached inverse is replicated and symmetric: each rank applies only its slab of rows
      // (= columns, read contiguously) and the slices are all-gathered, so the m^2 bytes of the
      // apply are split over the ranks like the data matrix is.
      const int G = comm->size();
      const int64_t per = f.slab;  // multiple of 4, G*per >= m
      const int64_t lo = std::min<int64_t>(f.m, comm->rank() * per);
      const int64_t cnt = std::min<int64_t>(f.m, lo + per) - lo;
      DVec mine = f.wslice;
      if (cnt < per) k::Fill(mine, 0.0);
      if (cnt > 0) {
        DVec slab = D.data().Slice(lo * f.m, cnt * f.m);
        k::Gemv(true, f.m, cnt, D.scale(), slab, f.m, f.p, 0.0, mine.Slice(0, cnt));
      }
      comm->AllGather(mine.data(), f.wpad.data(), static_cast<size_t>(per), f.wpad.dt);
      (void)G;
    } else {
      ApplyInverseFixed();
    }
  }

  // The sharded sweep's tail on the peer window: 2 launches, no collective call.
  void FusedForwardPeer() {
    FusedState& f = fs_;
    const DenseMatrixImpl& L = *f.ls.L_arg_var;
    const DenseMatrixImpl& D = *f.ls.Dinv_arg;
    const PeerView& pv = Runtime::Get().peer()->view();
    k::PeerReduceExchange(pv, f.m, f.grid, f.tpart, -L.scale(),
                          f.ls.rhs_arg.n != 0 ? &f.ls.rhs_arg : nullptr, f.p);
    if (f.peer_slab) {
      const int64_t lo = std::min<int64_t>(f.m, static_cast<int64_t>(pv.rank) * f.slab);
      k::PeerSlabApplyExchange(pv, f.m, f.slab, lo, D.data(), f.m, D.scale(), f.p, f.wpad);
    } else {
      ApplyInverseFixed();
    }
  }

  // w = Dinv p with every buffer at a fixed address (what a captured launch needs)
  void ApplyInverseFixed() {
    FusedState& f = fs_;
    const DenseMatrixImpl& D = *f.ls.Dinv_arg;
    if (f.symv_packed.n > 0) k::SymvPacked(f.m, D.scale(), f.symv_packed, f.p, 0.0, f.w, &f.symv_work);
    else if (f.symv_work.n > 0) k::Symv(f.m, D.scale(), D.data(), f.m, f.p, 0.0, f.w, &f.symv_work);
    else D.Apply(1.0, f.p, 0.0, f.w);
  }

  void ResetGraph() {
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    graph_exec_ = nullptr;
    graph_ = nullptr;
    graph_len_ = 0;
    ResetGenericGraph();
  }

  void ResetGenericGraph() { gg_.Reset(); }

  std::vector<BlockVector*> StateHandles() {  // (the "previous" handles first: see SweepGraph::Capture)
    std::vector<BlockVector*> h;
    for (int i = 0; i < N_; ++i) h.push_back(&y_prev_[i]);
    h.push_back(&u_);
    for (int i = 0; i < N_; ++i) h.push_back(&x_[i]);
    for (int i = 0; i < N_; ++i) h.push_back(&y_[i]);
    return h;
  }

  bool GenericGraphWanted(int count) {
    if (fused_ || !capture_safe_ || eager_sweeps_ < SweepGraph::EagerSweepsFirst() ||
        static_cast<int>(y_prev_.size()) != N_)
      return false;
    return gg_.Wanted(count, StateHandles());
  }

  // The sweeps between two residual checks replayed from one hipGraph: a sharded sweep is 3
  // short dependent launches (~60 us of kernels at 8 ranks), so the host's per-launch cost and
  // jitter would otherwise sit on the critical path.  Iterates are bit-identical to the eager
  // launches (same kernels, same arguments; the exchange tags come from a device counter).
  void SweepBatch(int count) override {
    static const int mode = [] {  // EPSILON_HIP_GRAPH: 0 never, 1 always (fused), default: peer mode
      const char* e = std::getenv("EPSILON_HIP_GRAPH");
      return e ? std::atoi(e) : -1;
    }();
    Runtime& rt = Runtime::Get();
    if (GenericGraphWanted(count) && gg_.Run(count, StateHandles(), static_cast<size_t>(N_), [this] { Sweep(); })) return;
    const ShardSpec& sh = ShardSpec::Get();
    const bool rccl_in_sweep = fused_ && !fs_.use_peer && sh.active() && sh.IsSharded(fs_.ls.var_key);
    const bool fixed_buffers = (fs_.use_peer && fs_.peer_slab) || fs_.symv_work.n > 0;
    const bool want = fused_ && !rccl_in_sweep && fixed_buffers &&
                      (mode == 1 || (mode != 0 && fs_.use_peer));
    if (!want || count < 2 || rt.profiling()) {
      for (int i = 0; i < count; ++i) Sweep();
      if (!fused_) eager_sweeps_ += count;
      return;
    }
    if (graph_exec_ == nullptr || graph_len_ != count) {
      ResetGraph();
      hipStream_t s = rt.stream();
      EPS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      rt.set_capturing(true);
      try {
        for (int i = 0; i < count; ++i) FusedSweep();
      } catch (...) {
        rt.set_capturing(false);
        hipGraph_t dead = nullptr;
        (void)hipStreamEndCapture(s, &dead);
        if (dead) (void)hipGraphDestroy(dead);
        throw;
      }
      rt.set_capturing(false);
      EPS_HIP(hipStreamEndCapture(s, &graph_));
      EPS_HIP(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0));
      graph_len_ = count;
    }
    EPS_HIP(hipGraphLaunch(graph_exec_, rt.stream()));
  }

  void FusedSweep() {
    FusedState& f = fs_;
    const DenseMatrixImpl& L = *f.ls.L_arg_var;
    k::LassoFusedArgs a;
    a.m = f.m;
    a.n = f.n;
    a.lda = L.rows();
    a.A = L.data();
    a.w = f.w;
    a.kappa = -L.scale();
    a.Bs = f.sz.Bs;
    a.Cs = f.sz.Cs;
    a.a1 = f.a1;
    a.lam = f.sz.lam;
    a.sz_alpha = f.sz.alpha;
    a.sz_beta = f.sz.beta;
    a.sz_alpha_vec = f.sz.alpha_vec;
    a.sz_beta_vec = f.sz.beta_vec;
    a.sz_M = f.sz.M;
    a.u = f.u;
    a.x0 = f.x0;
    a.x1 = f.x1;
    a.y0 = f.y0;
    a.y1 = f.y1;
    a.y1prev = f.y1prev;
    a.tpart = f.tpart;
    if (f.use_peer) {
      a.epoch = Runtime::Get().peer()->view().epoch;
      k::LassoFusedPass(a);
      FusedForwardPeer();
      return;
    }
    k::LassoFusedPass(a);
    FusedForward(/*from_state=*/false);
  }

  void Sweep() override {  // :135-147
    if (fused_) {
      FusedSweep();
      return;
    }
    y_prev_ = y_;  // shallow: blocks are replaced, never mutated, below
    u_ -= b_;
    for (int i = 0; i < N_; ++i) u_ -= y_[i];
    for (int i = 0; i < N_; ++i) {
      u_ += y_[i];
      {
        LocalShardScope scope(arg_shards_[i]);
        x_[i] = prox_[i]->Apply(u_);
      }
      y_[i] = A_ * x_[i];
      u_ -= y_[i];
    }
  }

  // ---- residual check of the fused structure: one launch, splittable for pipelining ------------
  // With A_ = [a0 I, a1 I] (a0 = 1), b_ empty and N = 2 the quantities of :178-217 are
  //   ||A x_i|| = ||y_i||,  r = ||y0 + y1||,  s = rho ||A_0^T (y1 - y1_prev)|| = rho ||y1 - y1_prev||,
  //   ||A^T u||^2 = (a0^2 + a1^2) ||u||^2.
  bool PipelinedChecks() const override {
    static const bool off = [] {
      const char* e = std::getenv("EPSILON_HIP_PIPELINE_CHECKS");
      return e && e[0] == '0';
    }();
    const ShardSpec& sh = ShardSpec::Get();
    return fused_ && !off && !(sh.active() && sh.consensus_terms());
  }
  void BeginResiduals() override {
    Runtime& rt = Runtime::Get();
    FusedState& f = fs_;
    rt.ResetSlots();
    norm_slot_ = rt.NewSlot();
    for (int k = 1; k < 6; ++k) rt.NewSlot();
    const ShardSpec& sh = ShardSpec::Get();
    const bool sharded = sh.active() && sh.IsSharded(f.ls.var_key);
    k::LassoFusedNorms(f.u, f.y0, f.y1, f.y1prev,
                       sharded ? rt.ShardSlotPtr(norm_slot_) : rt.SlotPtr(norm_slot_), f.norm_work,
                       f.use_peer ? rt.peer()->device_error_word() : nullptr);
    rt.FetchSlotsAsync();
  }
  void EndResiduals() override {
    Runtime& rt = Runtime::Get();
    rt.WaitSlots();
    // a timed-out exchange on ANY rank shows in the all-reduced sixth value: every rank raises at
    // the same check, none is left waiting in a collective the others never enter
    if (rt.SlotValue(norm_slot_ + 5) > 0) {
      rt.Sync();
      if (rt.peer()) rt.peer()->ClearError();
      EPS_FATAL("peer exchange: a poll timed out on at least one rank (a peer did not deliver its part)");
    }
    const double ny0 = rt.SlotValue(norm_slot_), ny1 = rt.SlotValue(norm_slot_ + 1),
                 nr = rt.SlotValue(norm_slot_ + 2), ns = rt.SlotValue(norm_slot_ + 3),
                 nu = rt.SlotValue(norm_slot_ + 4);
    const double rho = params_.rho;
    const double max_norm = std::fmax(std::sqrt(ny0), std::sqrt(ny1));
    FinishResiduals(std::sqrt(nr), rho * std::sqrt(ns),
                    params_.abs_tol * std::sqrt(static_cast<double>(m_)) + params_.rel_tol * max_norm,
                    params_.abs_tol * std::sqrt(static_cast<double>(n_)) +
                        params_.rel_tol * rho * std::sqrt((1.0 + fs_.a1 * fs_.a1) * nu));
  }
  void SaveSnapshot() override { k::Copy(fs_.snapshot, fs_.state_all); }
  void RestoreSnapshot() override {
    Runtime::Get().Sync();  // let the discarded sweeps drain
    if (Runtime::Get().peer()) Runtime::Get().peer()->CheckError();
    k::Copy(fs_.state_all, fs_.snapshot);
  }

  void ComputeResiduals() override {  // :178-217
    if (fused_) {
      BeginResiduals();
      EndResiduals();
      return;
    }
    Runtime& rt = Runtime::Get();
    rt.ResetSlots();
    const int s_b = b_.NormSqAsync();
    std::vector<int> s_Ax(N_);
    BlockVector Ax_b = b_;
    for (int i = 0; i < N_; ++i) {
      // A_*x_[i] is y_[i], computed by the sweep (the reference recomputes it, :186)
      s_Ax[i] = y_[i].NormSqAsync();
      Ax_b += y_[i];
    }
    const int s_r = Ax_b.NormSqAsync();
    std::vector<int> s_s;
    BlockVector Ax_diff;
    for (int i = N_ - 2; i >= 0; --i) {
      Ax_diff += y_[i + 1] - y_prev_[i + 1];
      s_s.push_back((AiT_[i] * Ax_diff).NormSqAsync());
    }
    const int s_u = (AT_ * u_).NormSqAsync();
    rt.FetchSlots();
    if (rt.peer()) rt.peer()->CheckError();

    double max_norm = std::sqrt(rt.SlotValue(s_b));
    double own_max = -1;
    for (int i = 0; i < N_; ++i) {
      if (term_per_rank_[i])  // one term per rank: the reference's max runs over all of them
        own_max = std::fmax(own_max, std::sqrt(rt.SlotLocalValue(s_Ax[i])));
      else
        max_norm = std::fmax(max_norm, std::sqrt(rt.SlotValue(s_Ax[i])));
    }
    if (ShardSpec::Get().active() && ShardSpec::Get().consensus_terms())
      max_norm = std::fmax(max_norm, rt.comm()->AllReduceMaxHost(own_max));
    double s2 = 0;
    for (int s : s_s) {
      const double si = std::sqrt(rt.SlotValue(s));
      s2 += si * si;
    }
    const double rho = params_.rho;
    FinishResiduals(std::sqrt(rt.SlotValue(s_r)), rho * std::sqrt(s2),
                    params_.abs_tol * std::sqrt(static_cast<double>(m_)) + params_.rel_tol * max_norm,
                    params_.abs_tol * std::sqrt(static_cast<double>(n_)) +
                        params_.rel_tol * rho * std::sqrt(rt.SlotValue(s_u)));
  }

 private:
  int64_t m_ = 0, n_ = 0;
  int N_ = 0;
  bool vars_initialized_ = false;
  BlockMatrix A_, AT_;
  BlockVector b_;
  std::vector<BlockMatrix> AiT_;
  std::vector<std::unique_ptr<ProxOperator>> prox_;
  std::vector<std::set<std::string>> arg_shards_;
  std::vector<bool> term_per_rank_;
  struct FusedState {
    LeastSquaresDesc ls;
    ScaledZoneDesc sz;
    double a1 = 0;
    int64_t m = 0, n = 0;
    int grid = 0;
    int64_t slab = 0;  // rows of the cached inverse applied per rank (sharded runs)
    bool use_peer = false;   // exchanges ride in the sweep's kernels (peer window), not in RCCL
    bool peer_slab = false;  // ... and the inverse is applied by row slabs
    DVec u, x0, x1, y0, y1, y1prev, w, p, tpart, wpad, wslice;
    DVec symv_work;  // fixed workspace of the symmetric inverse apply (empty: not that form)
    DVec symv_packed;  // the cached inverse's lower tiles, each contiguous (empty: apply from the matrix)
    DVec state_all, snapshot;  // u, x0, x1, y0, y1, y1prev in one buffer; its copy at a check
    DVec norm_work;            // partials + ticket of the one-launch residual norms
  };
  bool fused_ = false;
  FusedState fs_;
  int norm_slot_ = 0;
  hipGraph_t graph_ = nullptr;
  hipGraphExec_t graph_exec_ = nullptr;
  int graph_len_ = 0;
  SweepGraph gg_;  // the generic operator path's sweeps between two checks as one hipGraph
  bool capture_safe_ = false;  // every prox operator of the problem is (ProxOperator::CaptureSafe)
  int eager_sweeps_ = 0;       // operators build lazily on their first Apply: one eager sweep first
  BlockVector u_;
  std::vector<BlockVector> x_, y_, y_prev_;
};

// ---------------------------------------------------------------------------------------------------
// ProxADMMTwoBlockSolver (reference algorithms/prox_admm_two_block.cc)
// ---------------------------------------------------------------------------------------------------

class ProxADMMTwoBlockSolver final : public Solver {
 public:
  using Solver::Solver;

  void Init() override {  // :21-94
    SetCurrentDType(data_->dtype());
    const double t0 = Now();
    if (op_cache_.size() > 64) op_cache_.Clear();
    OpCacheScope cache_scope(&op_cache_);
    const double sqrt_rho = std::sqrt(params_.rho);
    const DType dt = data_->dtype();
    AffineOperator H, A;
    BlockVector z0;
    for (size_t i = 0; i < problem_.constraint.size(); ++i) {
      const pb::Expression& constr = problem_.constraint[i];
      EPS_CHECK_MSG(constr.expression_type == pb::Expression::INDICATOR, "constraint is not an indicator");
      EPS_CHECK_MSG(constr.cone_type == 1, "constraint cone is not ZERO");
      EPS_CHECK(constr.arg.size() == 1);
      affine::BuildAffineOperator(constr.arg[0], data_.get(), affine::constraint_key(i), &H.A, &H.b);
      std::map<std::string, const pb::Expression*> vars;
      GetVariables(constr, &vars);
      for (const auto& var : vars) {
        const int64_t dim = GetDimension(*var.second);
        A.A(var.first, var.first) = sqrt_rho * LinearMap::Identity(dim);
        z0.Set(var.first, DVec::Zeros(dim, dt));
      }
    }
    constr_prox_ = CreateProxOperator(pb::ProxFunction::ZERO, false);
    zero_f_ = pb::ProxFunction();
    constr_prox_->Init(ProxOperatorArg(zero_f_, data_.get(), H, A));
    GlobalDims(H.A, &m_, &n_);
    constr_H_ = H;

    EPS_CHECK_MSG(problem_.objective.expression_type == pb::Expression::ADD, "objective is not ADD");
    N_ = static_cast<int>(problem_.objective.arg.size());
    prox_.clear();
    arg_shards_.clear();
    for (int i = 0; i < N_; ++i) {
      const pb::Expression& f_expr = problem_.objective.arg[i];
      EPS_CHECK_MSG(f_expr.expression_type == pb::Expression::PROX_FUNCTION,
                    "objective term " << i << " is not a PROX_FUNCTION");
      AffineOperator Hi, Ai;
      for (size_t k = 0; k < f_expr.arg.size(); ++k)
        affine::BuildAffineOperator(f_expr.arg[k], data_.get(), affine::arg_key(k), &Hi.A, &Hi.b);
      std::map<std::string, const pb::Expression*> vars;
      GetVariables(f_expr, &vars);
      for (const auto& var : vars)
        Ai.A(var.first, var.first) = sqrt_rho * LinearMap::Identity(GetDimension(*var.second));
      prox_.emplace_back(CreateProxOperator(f_expr.prox_function.prox_function_type,
                                            f_expr.prox_function.epigraph));
      arg_shards_.push_back(InferShardedArgs(Hi.A));
      LocalShardScope scope(arg_shards_.back());
      prox_.back()->Init(ProxOperatorArg(f_expr.prox_function, data_.get(), Hi, Ai));
    }
    if (!params_.warm_start || !vars_initialized_) {
      z_ = z0;
      u_ = BlockVector();
      x_ = BlockVector();
      vars_initialized_ = true;
    }
    iter_ = 0;
    finished_ = false;
    status_ = pb::SolverStatus();
    initialized_ = true;
    TryEnableFused();
    gg_.Reset();
    gg_.ClearFailure();
    capture_safe_ = constr_prox_ != nullptr && constr_prox_->CaptureSafe();
    for (const auto& op : prox_) capture_safe_ = capture_safe_ && op->CaptureSafe();
    eager_sweeps_ = 0;
    Runtime::Get().Sync();
    init_seconds_ = Now() - t0;
  }

  BlockVector GetSolution() override { return x_; }

  // the sweeps between two residual checks: one hipGraph where the operators allow it (SweepGraph)
  void SweepBatch(int count) override {
    if (!fused_ && capture_safe_ && eager_sweeps_ >= SweepGraph::EagerSweepsFirst()) {
      const std::vector<BlockVector*> state = {&z_prev_, &x_, &z_, &u_};
      if (gg_.Wanted(count, state) && gg_.Run(count, state, 1, [this] { Sweep(); })) return;
    }
    for (int i = 0; i < count; ++i) Sweep();
    if (!fused_) eager_sweeps_ += count;
  }

 protected:
  // ---- fused sweep for "least squares + separable threshold" problems, two-block form -----------
  // [SUM_SQUARE with a dense argument map, scaled-zone prox], one constraint a0 x0 + a1 x1 = 0
  // without a constant: the x-updates are the same two operators as in the multi-block driver, the
  // z-update is the closed-form projection onto the constraint, so one pass over the data matrix
  // does a whole sweep (kernels_fused.hip, chain 1).  f32 and f64, single GPU.
  void TryEnableFused() {
    fused_ = false;
    const char* env = std::getenv("EPSILON_HIP_FUSED");
    if (env && env[0] == '0') return;
    if (N_ != 2 || problem_.constraint.size() != 1) return;
    if (ShardSpec::Get().active()) return;
    if (!constr_H_.b.data().empty()) return;
    FusedState f;
    if (!prox_[0]->DescribeLeastSquares(&f.ls) || !prox_[1]->DescribeScaledZone(&f.sz)) return;
    if (f.ls.var_key == f.sz.var_key) return;
    const std::string ck = affine::constraint_key(0);
    const BlockMatrix& H = constr_H_.A;
    if (H.data().size() != 2 || !H.has_key(ck, f.ls.var_key) || !H.has_key(ck, f.sz.var_key)) return;
    const LinearMap& H0 = H(ck, f.ls.var_key);
    const LinearMap& H1 = H(ck, f.sz.var_key);
    if (H0.impl().type() != SCALAR_MATRIX || H1.impl().type() != SCALAR_MATRIX) return;
    f.a0 = GetScalar(H0);
    f.a1 = GetScalar(H1);
    if (f.a0 == 0 || f.a1 == 0) return;
    const DenseMatrixImpl& L = *f.ls.L_arg_var;
    if (L.trans()) return;
    f.m = L.rows();
    f.n = L.cols();
    if (H0.impl().n() != f.n || H1.impl().n() != f.n) return;
    if (!k::LassoFusedSupported(f.m, f.n, L.data(), L.rows())) return;
    if (f.ls.rhs_arg.n != 0 && f.ls.rhs_arg.n != f.m) return;
    const DenseMatrixImpl& D = *f.ls.Dinv_arg;
    if (D.trans() || D.rows() != f.m) return;
    const DType dt = data_->dtype();
    if ((f.sz.alpha_vec.n > 0 && f.sz.alpha_vec.dt != dt) || (f.sz.beta_vec.n > 0 && f.sz.beta_vec.dt != dt)) return;
    if (L.dtype() != dt || D.dtype() != dt) return;
    auto state = [&](const BlockVector& src, const std::string& key) {
      DVec v = DVec::Zeros(f.n, dt);
      if (src.has_key(key)) k::Copy(v, src(key));
      return v;
    };
    f.x0 = state(x_, f.ls.var_key);
    f.x1 = state(x_, f.sz.var_key);
    f.z0 = state(z_, f.ls.var_key);
    f.z1 = state(z_, f.sz.var_key);
    f.u0 = state(u_, f.ls.var_key);
    f.u1 = state(u_, f.sz.var_key);
    f.z0p = DVec::Zeros(f.n, dt);
    f.z1p = DVec::Zeros(f.n, dt);
    f.p = DVec::Zeros(f.m, dt);
    f.w = DVec::Zeros(f.m, dt);
    f.grid = k::LassoFusedGrid(f.m, f.n, dt);
    f.tpart = DVec::Empty(static_cast<int64_t>(f.grid) * f.m, dt);
    fs_ = f;
    // the generic containers become views of the fused state
    auto two = [&](const DVec& a, const DVec& b) {
      BlockVector v;
      v.Set(fs_.ls.var_key, a);
      v.Set(fs_.sz.var_key, b);
      return v;
    };
    x_ = two(fs_.x0, fs_.x1);
    z_ = two(fs_.z0, fs_.z1);
    u_ = two(fs_.u0, fs_.u1);
    z_prev_ = two(fs_.z0p, fs_.z1p);
    fused_ = true;
    // p = rhs_arg - L(arg, var) (z0 - u0) of the current state, w = Dinv_arg p
    DVec v0 = fs_.z0.Clone();
    k::Axpby(v0, -1.0, fs_.u0, 1.0);
    L.Apply(-1.0, v0, 0.0, fs_.p);
    if (fs_.ls.rhs_arg.n != 0) k::Axpby(fs_.p, 1.0, fs_.ls.rhs_arg, 1.0);
    D.Apply(1.0, fs_.p, 0.0, fs_.w);
  }

  void FusedSweep() {
    FusedState& f = fs_;
    const DenseMatrixImpl& L = *f.ls.L_arg_var;
    k::LassoFusedArgs a;
    a.m = f.m;
    a.n = f.n;
    a.lda = L.rows();
    a.A = L.data();
    a.w = f.w;
    a.kappa = -L.scale();
    a.Bs = f.sz.Bs;
    a.Cs = f.sz.Cs;
    a.a1 = f.a1;
    a.a0 = f.a0;
    a.lam = f.sz.lam;
    a.sz_alpha = f.sz.alpha;
    a.sz_beta = f.sz.beta;
    a.sz_alpha_vec = f.sz.alpha_vec;
    a.sz_beta_vec = f.sz.beta_vec;
    a.sz_M = f.sz.M;
    a.chain = 1;
    a.u = f.u0;
    a.x0 = f.x0;
    a.x1 = f.x1;
    a.y0 = f.z0;
    a.y1 = f.z1;
    a.y1prev = f.z0p;
    a.e0 = f.u1;
    a.e1 = f.z1p;
    a.tpart = f.tpart;
    k::LassoFusedPass(a);
    k::ReducePartials(f.m, f.grid, f.tpart, -L.scale(), 0.0, f.p,
                      f.ls.rhs_arg.n != 0 ? &f.ls.rhs_arg : nullptr);
    f.ls.Dinv_arg->Apply(1.0, f.p, 0.0, f.w);
  }

  void Sweep() override {  // :97-112
    if (fused_) {
      FusedSweep();
      return;
    }
    z_prev_ = z_;
    BlockVector zu = z_ - u_;
    x_ = BlockVector();
    for (int i = 0; i < N_; ++i) {
      LocalShardScope scope(arg_shards_[i]);
      x_ += prox_[i]->Apply(zu);
    }
    z_ = constr_prox_->Apply(x_ + u_);
    u_ += x_ - z_;
  }

  void ComputeResiduals() override {  // :135-156
    Runtime& rt = Runtime::Get();
    rt.ResetSlots();
    const int s_r = DiffNormSqAsync(x_, z_);
    const int s_s = DiffNormSqAsync(z_, z_prev_);
    const int s_x = x_.NormSqAsync();
    const int s_z = z_.NormSqAsync();
    const int s_u = u_.NormSqAsync();
    rt.FetchSlots();
    const double rho = params_.rho;
    const double sq_n = std::sqrt(static_cast<double>(n_));
    FinishResiduals(std::sqrt(rt.SlotValue(s_r)), rho * std::sqrt(rt.SlotValue(s_s)),
                    params_.abs_tol * sq_n + params_.rel_tol * std::fmax(std::sqrt(rt.SlotValue(s_x)),
                                                                         std::sqrt(rt.SlotValue(s_z))),
                    params_.abs_tol * sq_n + params_.rel_tol * rho * std::sqrt(rt.SlotValue(s_u)));
  }

 private:
  int64_t m_ = 0, n_ = 0;
  int N_ = 0;
  bool vars_initialized_ = false;
  pb::ProxFunction zero_f_;
  std::vector<std::unique_ptr<ProxOperator>> prox_;
  std::vector<std::set<std::string>> arg_shards_;
  std::unique_ptr<ProxOperator> constr_prox_;
  AffineOperator constr_H_;
  struct FusedState {
    LeastSquaresDesc ls;
    ScaledZoneDesc sz;
    double a0 = 1, a1 = -1;
    int64_t m = 0, n = 0;
    int grid = 0;
    DVec x0, x1, z0, z1, u0, u1, z0p, z1p, p, w, tpart;
  };
  bool fused_ = false;
  FusedState fs_;
  BlockVector x_, z_, u_, z_prev_;
  SweepGraph gg_;
  bool capture_safe_ = false;
  int eager_sweeps_ = 0;
};

std::unique_ptr<Solver> CreateSolver(pb::Problem problem, std::shared_ptr<DataMap> data,
                                     pb::SolverParams params) {  // solvemodule.cc:74-87
  if (params.solver == pb::SolverParams::PROX_ADMM)
    return std::unique_ptr<Solver>(new ProxADMMSolver(std::move(problem), std::move(data), params));
  if (params.solver == pb::SolverParams::PROX_ADMM_TWO_BLOCK)
    return std::unique_ptr<Solver>(
        new ProxADMMTwoBlockSolver(std::move(problem), std::move(data), params));
  EPS_FATAL("Unknown solver: " << params.solver);
}

BlockVector EvalProx(const pb::Expression& f_expr, double lambda, DataMap* data,
                     const BlockVector& v_in) {  // solvemodule.cc:189-242
  EPS_CHECK_MSG(f_expr.expression_type == pb::Expression::PROX_FUNCTION,
                "eval_prox: expression is not a PROX_FUNCTION");
  SetCurrentDType(data->dtype());
  AffineOperator H, A;
  for (size_t i = 0; i < f_expr.arg.size(); ++i)
    affine::BuildAffineOperator(f_expr.arg[i], data, affine::arg_key(i), &H.A, &H.b);
  std::map<std::string, const pb::Expression*> vars;
  GetVariables(f_expr, &vars);
  int i = 0;
  for (const auto& var : vars) {
    A.A(affine::constraint_key(i++), var.first) =
        (1 / std::sqrt(lambda)) * LinearMap::Identity(GetDimension(*var.second));
  }
  BlockVector v = A.A * v_in;
  std::unique_ptr<ProxOperator> op = CreateProxOperator(f_expr.prox_function.prox_function_type,
                                                        f_expr.prox_function.epigraph);
  op->Init(ProxOperatorArg(f_expr.prox_function, data, H, A));
  return op->Apply(v);
}

}  // namespace eps

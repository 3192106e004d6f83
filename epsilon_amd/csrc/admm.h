// ADMM drivers.
//
// Same drivers as the reference: `ProxADMMSolver` (Gauss-Seidel multi-block, reference
// src/epsilon/algorithms/prox_admm.cc:131-217) and `ProxADMMTwoBlockSolver` (Jacobi x-updates +
// projection z-update, prox_admm_two_block.cc:96-156) behind the `Solver` base
// (algorithms/solver.h:42-102).  All iterate state stays in HBM; the only host round trip in
// the loop is one copy of the residual scalars every `epoch_iterations` sweeps.
//
// Unlike the reference, a solver can be driven in pieces (Init / Run(k) / status / variable
// read-back) so a caller can keep it - and its cached factorisation - alive across calls
// (warm start, reference solvemodule.cc:142-156) and so bench.py can time exactly K sweeps.
#pragma once

#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "block.h"
#include "prox.h"
#include "wire.h"

namespace eps {

class Solver {
 public:
  Solver(pb::Problem problem, std::shared_ptr<DataMap> data, pb::SolverParams params);
  virtual ~Solver() {}

  // Build operators and factorisations (everything the reference does in Init()).
  virtual void Init() = 0;
  // Run sweeps until OPTIMAL / max_iterations, or at most `max_sweeps` more sweeps when
  // max_sweeps >= 0.  Returns the number of sweeps executed by this call.
  int Run(int max_sweeps);
  // Full solve as the reference's Solve(): Init() then Run(-1).
  void Solve();

  virtual BlockVector GetSolution() = 0;
  const pb::SolverStatus& status() const { return status_; }
  const pb::Problem& problem() const { return problem_; }
  pb::SolverParams& params() { return params_; }
  DataMap* data() { return data_.get(); }
  bool initialized() const { return initialized_; }
  double init_seconds() const { return init_seconds_; }
  double loop_seconds() const { return loop_seconds_; }
  void set_log(std::function<void(const std::string&)> log) { log_ = std::move(log); }

 protected:
  virtual void Sweep() = 0;
  // `count` sweeps with no host decision in between (Run() cuts the iteration into such runs at
  // the residual checks); a driver whose sweep is launch-bound replays them from a hipGraph.
  virtual void SweepBatch(int count) {
    for (int i = 0; i < count; ++i) Sweep();
  }
  virtual void ComputeResiduals() = 0;
  // Pipelined residual checks: a driver that can (a) split a check into device work that is only
  // ENQUEUED (BeginResiduals) and the host decision (EndResiduals), and (b) save / restore its
  // iterates, lets Run() enqueue the next batch of sweeps before it waits for the check's scalars,
  // so the host round trip (and, sharded, the all-reduce of the scalars) hides behind useful work.
  // If the check says OPTIMAL the iterates of the check are restored: results are exactly those
  // of the unpipelined loop.
  virtual bool PipelinedChecks() const { return false; }
  virtual void BeginResiduals() {}
  virtual void EndResiduals() {}
  virtual void SaveSnapshot() {}
  virtual void RestoreSnapshot() {}
  void LogStatus();
  void FinishResiduals(double r2, double s2, double eps_pri, double eps_dual);

  pb::Problem problem_;
  std::shared_ptr<DataMap> data_;
  pb::SolverParams params_;
  pb::SolverStatus status_;
  int iter_ = 0;
  bool initialized_ = false;
  bool finished_ = false;
  double init_seconds_ = 0, loop_seconds_ = 0;
  std::function<void(const std::string&)> log_;
  // Gram products / inverses of previous Inits of this solver, by content id (warm start).
  OpCache op_cache_;
};

std::unique_ptr<Solver> CreateSolver(pb::Problem problem, std::shared_ptr<DataMap> data,
                                     pb::SolverParams params);

// One prox evaluation (reference python/epopt/solvemodule.cc:189-242).
BlockVector EvalProx(const pb::Expression& f_expr, double lambda, DataMap* data,
                     const BlockVector& v);

// generic-path graph replay counters (eps_graph_stats)
struct GraphStats {
  long long replayed_sweeps = 0, captures = 0;
  static GraphStats& Get();
};

}  // namespace eps

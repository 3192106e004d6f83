// extern "C" boundary: see include/epsilon_hip.h for the contract of each function and the
// reference interface it replaces (python/epopt/solvemodule.cc).
#include "../../include/epsilon_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "admm.h"
#include "comm.h"
#include "kernels.h"
#include "linear_map.h"
#include "wire.h"

using namespace eps;

struct eps_result {
  std::string status;
  std::vector<std::string> ids;
  std::vector<HostArray> values;  // not value-initialised, huge pages from 64 MB
};

struct eps_solver {
  std::shared_ptr<DataMap> data;
  std::unique_ptr<Solver> solver;
  DType dtype;
};

namespace {

thread_local std::string g_last_error;
int g_dtype_option = -1;  // -1: take EPSILON_HIP_DTYPE / default

DType ConfiguredDType() {
  if (g_dtype_option >= 0) return static_cast<DType>(g_dtype_option);
  const char* e = std::getenv("EPSILON_HIP_DTYPE");
  if (e && (std::strcmp(e, "f64") == 0 || std::strcmp(e, "float64") == 0)) return F64;
  return F32;
}

template <class F> int Guard(F f) {
  try {
    g_last_error.clear();
    f();
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
  } catch (...) {
    g_last_error = "unknown error";
  }
  // leave no half-finished GPU work behind
  (void)hipGetLastError();
  return 1;
}

// own_host: copy host blobs (solver handles outlive the call that passes them; the one-shot entry
// points read the caller's memory in place, it is theirs for the duration of the call).
void InsertBlobs(DataMap* dm, const eps_blob* data, size_t ndata, bool own_host) {
  bool device_blobs = false;
  for (size_t i = 0; i < ndata; ++i) {
    EPS_CHECK_MSG(data[i].key != nullptr, "data blob without a key");
    Blob b;
    b.ptr = data[i].ptr;
    b.len = data[i].len;
    b.kind = data[i].kind;
    EPS_CHECK_MSG(b.kind >= 0 && b.kind <= 2, "bad blob kind " << b.kind);
    if (b.kind != 0) device_blobs = true;
    if (own_host) dm->InsertOwned(data[i].key, b);
    else dm->Insert(data[i].key, b);
  }
  // Borrowed device memory is read on this library's own non-blocking stream, which has no
  // ordering with whatever stream produced it: wait for the device once, here.
  if (device_blobs) {
    Runtime::Get();
    EPS_HIP(hipDeviceSynchronize());
  }
}

std::shared_ptr<DataMap> MakeDataMap(const eps_blob* data, size_t ndata, DType dt,
                                     bool own_host = false) {
  auto dm = std::make_shared<DataMap>(dt);
  InsertBlobs(dm.get(), data, ndata, own_host);
  return dm;
}

void FillResult(Solver* solver, eps_result* r) {
  r->status = solver->status().Serialize();
  BlockVector x = solver->GetSolution();
  for (const auto& var : GetVariables(solver->problem())) {  // solvemodule.cc:166-176
    r->ids.push_back(var.first);
    r->values.push_back(x(var.first).ToHostArray());
  }
}

void LogToStdout(const std::string& msg) {  // reference util/logging.cc:10-13 -> PySys_WriteStdout
  std::fputs(msg.c_str(), stdout);
  std::fputc('\n', stdout);
  std::fflush(stdout);
}

LinearMap ParseMap(const void* bytes, size_t len, DataMap* dm) {
  pb::LinearMap p = pb::ParseLinearMap(bytes, len);
  return BuildLinearMap(p, dm);
}

}  // namespace

extern "C" {

const char* eps_last_error(void) { return g_last_error.c_str(); }

const char* eps_version(void) { return "epsilon_hip 0.1 gfx950"; }

int eps_set_option(const char* key, const char* value) {
  return Guard([&] {
    EPS_CHECK(key != nullptr && value != nullptr);
    if (std::strcmp(key, "dtype") == 0) {
      if (std::strcmp(value, "f32") == 0) g_dtype_option = F32;
      else if (std::strcmp(value, "f64") == 0) g_dtype_option = F64;
      else EPS_FATAL("dtype must be f32 or f64, got " << value);
    } else if (std::strcmp(key, "device") == 0) {
      setenv("EPSILON_HIP_DEVICE", value, 1);
    } else if (std::strcmp(key, "gemm") == 0) {
      setenv("EPSILON_HIP_GEMM", value, 1);
    } else if (std::strcmp(key, "profile_filter") == 0) {
      Runtime::Get().set_prof_filter(value);
    } else if (std::strcmp(key, "fused") == 0) {
      setenv("EPSILON_HIP_FUSED", value, 1);
    } else if (std::strcmp(key, "graph_generic") == 0) {
      setenv("EPSILON_HIP_GRAPH_GENERIC", value, 1);
    } else if (std::strcmp(key, "refine") == 0) {
      if (std::strcmp(value, "auto") == 0) unsetenv("EPSILON_HIP_REFINE");
      else setenv("EPSILON_HIP_REFINE", value, 1);
    } else {
      EPS_FATAL("unknown option " << key);
    }
  });
}

int eps_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return count;
}

int eps_solve(const void* problem, size_t problem_len, const void* solver_params,
              size_t solver_params_len, const eps_blob* data, size_t ndata,
              const eps_param* params, size_t nparams, eps_result** out) {
  return Guard([&] {
    EPS_CHECK(out != nullptr);
    *out = nullptr;
    const DType dt = ConfiguredDType();
    SetCurrentDType(dt);
    pb::Problem p = pb::ParseProblem(problem, problem_len);
    pb::SolverParams sp = pb::ParseSolverParams(solver_params, solver_params_len);
    auto dm = MakeDataMap(data, ndata, dt);
    for (size_t i = 0; i < nparams; ++i)
      dm->SetParameter(params[i].id, pb::ParseConstant(params[i].constant_proto, params[i].len));
    std::unique_ptr<Solver> solver = CreateSolver(std::move(p), dm, sp);
    solver->set_log(LogToStdout);
    solver->Solve();
    std::unique_ptr<eps_result> r(new eps_result);
    FillResult(solver.get(), r.get());
    *out = r.release();
  });
}

int eps_eval_prox(const void* f_expr, size_t f_expr_len, double lambda, const eps_blob* data,
                  size_t ndata, const eps_blob* v, size_t nv, eps_result** out) {
  return Guard([&] {
    EPS_CHECK(out != nullptr);
    *out = nullptr;
    const DType dt = ConfiguredDType();
    SetCurrentDType(dt);
    pb::Expression e = pb::ParseExpression(f_expr, f_expr_len);
    auto dm = MakeDataMap(data, ndata, dt);
    BlockVector vin;
    for (size_t i = 0; i < nv; ++i) {
      EPS_CHECK_MSG(v[i].kind == EPS_BLOB_HOST && v[i].len % sizeof(double) == 0,
                    "eval_prox: v must be host float64 bytes");
      vin.Set(v[i].key, DVec::FromHost(static_cast<const double*>(v[i].ptr),
                                       static_cast<int64_t>(v[i].len / sizeof(double)), dt));
    }
    BlockVector x = EvalProx(e, lambda, dm.get(), vin);
    std::unique_ptr<eps_result> r(new eps_result);
    for (const auto& kv : x.data()) {  // GetVariableMap, solvemodule.cc:45-56
      r->ids.push_back(kv.first);
      r->values.push_back(kv.second.ToHostArray());
    }
    *out = r.release();
  });
}

int eps_result_status(const eps_result* r, const void** bytes, size_t* len) {
  if (!r || !bytes || !len) return 1;
  *bytes = r->status.data();
  *len = r->status.size();
  return 0;
}

size_t eps_result_num_vars(const eps_result* r) { return r ? r->ids.size() : 0; }

int eps_result_var(const eps_result* r, size_t i, const char** id, const double** values,
                   size_t* count) {
  if (!r || i >= r->ids.size()) return 1;
  if (id) *id = r->ids[i].c_str();
  if (values) *values = r->values[i].data();
  if (count) *count = r->values[i].size();
  return 0;
}

int eps_result_copy_var(const eps_result* r, size_t i, void* dst, size_t bytes) {
  if (!r || i >= r->ids.size() || dst == nullptr || bytes != r->values[i].size() * sizeof(double)) return 1;
  ParallelHostCopy(dst, r->values[i].data(), bytes);
  return 0;
}

void eps_result_free(eps_result* r) { delete r; }

int eps_solver_create(const void* problem, size_t problem_len, const void* solver_params,
                      size_t solver_params_len, const eps_blob* data, size_t ndata,
                      eps_solver** out) {
  return Guard([&] {
    EPS_CHECK(out != nullptr);
    *out = nullptr;
    const DType dt = ConfiguredDType();
    SetCurrentDType(dt);
    std::unique_ptr<eps_solver> s(new eps_solver);
    s->dtype = dt;
    s->data = MakeDataMap(data, ndata, dt, /*own_host=*/true);
    s->solver = CreateSolver(pb::ParseProblem(problem, problem_len), s->data,
                             pb::ParseSolverParams(solver_params, solver_params_len));
    s->solver->set_log(LogToStdout);
    *out = s.release();
  });
}

int eps_solver_set_parameter(eps_solver* s, const char* id, const void* constant_proto,
                             size_t len, const eps_blob* data, size_t ndata) {
  return Guard([&] {
    EPS_CHECK(s != nullptr && id != nullptr);
    InsertBlobs(s->data.get(), data, ndata, /*own_host=*/true);
    s->data->SetParameter(id, pb::ParseConstant(constant_proto, len));
  });
}

int eps_solver_init(eps_solver* s) {
  return Guard([&] {
    EPS_CHECK(s != nullptr);
    SetCurrentDType(s->dtype);
    s->solver->Init();
  });
}

int eps_solver_run(eps_solver* s, int max_sweeps, int* sweeps_done) {
  return Guard([&] {
    EPS_CHECK(s != nullptr);
    SetCurrentDType(s->dtype);
    int done = s->solver->Run(max_sweeps);
    if (sweeps_done) *sweeps_done = done;
  });
}

int eps_solver_result(eps_solver* s, eps_result** out) {
  return Guard([&] {
    EPS_CHECK(s != nullptr && out != nullptr);
    *out = nullptr;
    SetCurrentDType(s->dtype);
    std::unique_ptr<eps_result> r(new eps_result);
    FillResult(s->solver.get(), r.get());
    *out = r.release();
  });
}

int eps_solver_timing(const eps_solver* s, double* init_seconds, double* loop_seconds) {
  if (!s) return 1;
  if (init_seconds) *init_seconds = s->solver->init_seconds();
  if (loop_seconds) *loop_seconds = s->solver->loop_seconds();
  return 0;
}

void eps_solver_destroy(eps_solver* s) {
  try {
    delete s;
  } catch (...) {
  }
}

int eps_comm_unique_id(void* out128) {
  return Guard([&] {
    EPS_CHECK(out128 != nullptr);
    GetRcclUniqueId(out128);
  });
}

int eps_comm_init_rccl(int rank, int world, const void* id128) {
  return Guard([&] {
    EPS_CHECK(id128 != nullptr && world >= 1 && rank >= 0 && rank < world);
    Runtime& rt = Runtime::Get();
    delete rt.peer();
    rt.set_peer(nullptr);
    delete rt.comm();
    rt.set_comm(nullptr);
    rt.set_comm(NewRcclComm(rank, world, id128));
  });
}

int eps_comm_init_callback(int rank, int world, eps_allreduce_fn fn, void* ctx) {
  return Guard([&] {
    EPS_CHECK(fn != nullptr && world >= 1 && rank >= 0 && rank < world);
    Runtime& rt = Runtime::Get();
    delete rt.peer();
    rt.set_peer(nullptr);
    delete rt.comm();
    rt.set_comm(nullptr);
    rt.set_comm(NewHostCallbackComm(rank, world, fn, ctx));
  });
}

int eps_comm_warmup(size_t count) {
  return Guard([&] {
    Runtime& rt = Runtime::Get();
    EPS_CHECK_MSG(rt.comm() != nullptr, "eps_comm_warmup: no communicator");
    // one all-reduce and one all-gather of `count` floats: RCCL sets up its rings / buffers on
    // the first collective of each kind, which should not land inside a timed solver Init
    const int64_t n = static_cast<int64_t>(count < 1 ? 1 : count);
    DVec a = DVec::Full(n, 1.0, F32);
    rt.comm()->AllReduceSum(a);
    DVec g = DVec::Zeros(n * rt.comm()->size(), F32);
    rt.comm()->AllGather(a.data(), g.data(), static_cast<size_t>(n), F32);
    rt.Sync();
    std::vector<double> h = a.Slice(0, 1).ToHost();
    EPS_CHECK_MSG(h[0] == static_cast<double>(rt.comm()->size()),
                  "eps_comm_warmup: all-reduce of ones gave " << h[0] << " on "
                                                              << rt.comm()->size() << " ranks");
  });
}

int eps_comm_enable_peer(size_t slot_floats, int rehearse_ranks, int* enabled) {
  return Guard([&] {
    Runtime& rt = Runtime::Get();
    EPS_CHECK_MSG(rt.comm() != nullptr, "eps_comm_enable_peer: no communicator");
    if (enabled) *enabled = 0;
    rt.Sync();
    delete rt.peer();
    rt.set_peer(nullptr);
    std::string why;
    PeerExchange* px = PeerExchange::Create(rt.comm(), static_cast<int64_t>(slot_floats ? slot_floats : 16384),
                                            rehearse_ranks, &why);
    if (px == nullptr) {
      g_last_error = "peer window not available: " + why;  // informational: the call succeeds
      return;
    }
    rt.set_peer(px);
    if (enabled) *enabled = 1;
  });
}

int eps_comm_disable_peer(void) {
  return Guard([&] {
    Runtime& rt = Runtime::Get();
    rt.Sync();
    delete rt.peer();
    rt.set_peer(nullptr);
  });
}

int eps_comm_shutdown(void) {
  return Guard([&] {
    Runtime& rt = Runtime::Get();
    rt.Sync();
    delete rt.peer();
    rt.set_peer(nullptr);
    delete rt.comm();
    rt.set_comm(nullptr);
    ShardSpec::Get().Clear();
  });
}

int eps_shard_keys(const char* const* keys, size_t nkeys) {
  return Guard([&] {
    ShardSpec::Get().Clear();
    for (size_t i = 0; i < nkeys; ++i) ShardSpec::Get().Add(keys[i]);
  });
}

int eps_shard_consensus_terms(int on) {
  return Guard([&] { ShardSpec::Get().set_consensus_terms(on != 0); });
}

int eps_block_solve_stats(double* max_condition, int* max_refine_steps, int reset) {
  return Guard([&] {
    BlockSolveStats& st = BlockSolveStats::Get();
    if (max_condition) *max_condition = st.max_condition;
    if (max_refine_steps) *max_refine_steps = st.max_refine_steps;
    if (reset) st.Reset();
  });
}

int eps_graph_stats(long long* replayed_sweeps, long long* captures, int reset) {
  return Guard([&] {
    GraphStats& g = GraphStats::Get();
    if (replayed_sweeps) *replayed_sweeps = g.replayed_sweeps;
    if (captures) *captures = g.captures;
    if (reset) g = GraphStats();
  });
}

int eps_profile_enable(int on) {
  return Guard([&] { Runtime::Get().set_profiling(on != 0); });
}

int eps_profile_reset(void) {
  return Guard([&] { Runtime::Get().ProfReset(); });
}

int eps_profile_dump(char* buf, size_t cap) {
  return Guard([&] {
    EPS_CHECK(buf != nullptr && cap > 0);
    Runtime& rt = Runtime::Get();
    rt.ProfCollect();
    std::string out;
    for (const auto& kv : rt.prof_totals()) {
      char line[256];
      std::snprintf(line, sizeof(line), "%s %lld %.6f\n", kv.first.c_str(),
                    static_cast<long long>(kv.second.count), kv.second.ms);
      out += line;
    }
    std::strncpy(buf, out.c_str(), cap - 1);
    buf[cap - 1] = 0;
  });
}

int eps_linear_map_apply(const void* linear_map, size_t len, const eps_blob* data, size_t ndata,
                         int transpose, const double* x, size_t nx, double* y, size_t ny) {
  return Guard([&] {
    const DType dt = ConfiguredDType();
    SetCurrentDType(dt);
    auto dm = MakeDataMap(data, ndata, dt);
    LinearMap A = ParseMap(linear_map, len, dm.get());
    if (transpose & 1) A = A.Transpose();
    if (transpose & 2) A = A.Inverse();  // apply the (cached, explicit) inverse map
    EPS_CHECK_MSG(static_cast<int64_t>(nx) == A.impl().n() && static_cast<int64_t>(ny) == A.impl().m(),
                  "map is " << A.impl().m() << " x " << A.impl().n() << ", got x of " << nx
                            << " and y of " << ny);
    DVec xd = DVec::FromHost(x, nx, dt);
    DVec yd = DVec::Empty(ny, dt);
    A.impl().Apply(1.0, xd, 0.0, yd);
    yd.ToHost(y);
  });
}

int eps_linear_map_binary(char op, const void* a, size_t a_len, int ta, const void* b,
                          size_t b_len, int tb, const eps_blob* data, size_t ndata,
                          int* result_type, int64_t* m, int64_t* n, double* dense,
                          size_t dense_capacity) {
  return Guard([&] {
    const DType dt = ConfiguredDType();
    SetCurrentDType(dt);
    auto dm = MakeDataMap(data, ndata, dt);
    LinearMap A = ParseMap(a, a_len, dm.get());
    LinearMap B = ParseMap(b, b_len, dm.get());
    if (ta) A = A.Transpose();
    if (tb) B = B.Transpose();
    LinearMap C;
    if (op == '+') C = A + B;
    else if (op == '*') C = A * B;
    else EPS_FATAL("op must be '+' or '*'");
    if (result_type) *result_type = static_cast<int>(C.impl().type());
    if (m) *m = C.impl().m();
    if (n) *n = C.impl().n();
    if (dense) {
      std::vector<double> D = C.impl().AsDenseHost();
      EPS_CHECK_MSG(D.size() <= dense_capacity, "dense output buffer too small");
      std::memcpy(dense, D.data(), D.size() * sizeof(double));
    }
  });
}

int eps_linear_map_inverse(const void* linear_map, size_t len, const eps_blob* data,
                           size_t ndata, double* dense, size_t dense_capacity) {
  return Guard([&] {
    const DType dt = ConfiguredDType();
    SetCurrentDType(dt);
    auto dm = MakeDataMap(data, ndata, dt);
    LinearMap A = ParseMap(linear_map, len, dm.get()).Inverse();
    std::vector<double> D = A.impl().AsDenseHost();
    EPS_CHECK_MSG(D.size() <= dense_capacity, "dense output buffer too small");
    std::memcpy(dense, D.data(), D.size() * sizeof(double));
  });
}

}  // extern "C"

namespace {
template <class F> double TimeLaunches(int iters, F f) {
  Runtime& rt = Runtime::Get();
  for (int i = 0; i < 2; ++i) f();
  hipEvent_t a, b;
  EPS_HIP(hipEventCreate(&a));
  EPS_HIP(hipEventCreate(&b));
  EPS_HIP(hipEventRecord(a, rt.stream()));
  for (int i = 0; i < iters; ++i) f();
  EPS_HIP(hipEventRecord(b, rt.stream()));
  EPS_HIP(hipEventSynchronize(b));
  float ms = 0;
  EPS_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return ms / iters;
}
// deterministic non-trivial fill: v[i] = ((i * 2654435761) mod 1024) / 1024 - 0.5
DVec Synthetic(int64_t n, DType dt, double scale) {
  std::vector<double> h(std::min<int64_t>(n, 1 << 20));
  for (size_t i = 0; i < h.size(); ++i)
    h[i] = scale * (static_cast<double>((i * 2654435761ull) & 1023) / 1024.0 - 0.5);
  DVec chunk = DVec::FromHost(h.data(), static_cast<int64_t>(h.size()), dt);
  if (n <= static_cast<int64_t>(h.size())) return chunk;
  DVec v = DVec::Empty(n, dt);
  for (int64_t off = 0; off < n; off += chunk.n) {
    int64_t len = std::min<int64_t>(chunk.n, n - off);
    k::Copy(v.Slice(off, len), chunk.Slice(0, len));
  }
  return v;
}
}  // namespace

extern "C" {

int eps_bench_gemv(int trans, int64_t rows, int64_t cols, int iters, double* ms_avg) {
  return Guard([&] {
    const DType dt = ConfiguredDType();
    DVec A = Synthetic(rows * cols, dt, 1.0);
    DVec x = Synthetic(trans ? rows : cols, dt, 1.0);
    DVec y = DVec::Zeros(trans ? cols : rows, dt);
    if (trans == 2) {  // symmetric apply (reads the lower tiles only)
      EPS_CHECK(rows == cols);
      *ms_avg = TimeLaunches(iters, [&] { k::Symv(rows, 1.0, A, rows, x, 0.0, y); });
      return;
    }
    *ms_avg = TimeLaunches(iters, [&] { k::Gemv(trans != 0, rows, cols, 1.0, A, rows, x, 0.0, y); });
  });
}

int eps_bench_stream(const void* device_ptr, size_t bytes, int mode, int grid, int iters,
                     double* ms_avg) {
  return Guard([&] {
    EPS_CHECK_MSG(bytes >= 16 && mode >= 0 && mode <= 2, "eps_bench_stream: bad arguments");
    Runtime& rt = Runtime::Get();
    DVec own;
    const void* src = device_ptr;
    if (src == nullptr) {
      own = Synthetic(static_cast<int64_t>(bytes / 4), F32, 1.0);
      src = own.data();
    }
    EPS_HIP(hipDeviceSynchronize());  // a borrowed pointer may still be written by another stream
    DVec dst;
    if (mode == 2) dst = DVec::Empty(static_cast<int64_t>(bytes / 4), F32);
    if (grid <= 0) grid = 2048;
    DVec scratch = DVec::Empty(grid, F32);
    *ms_avg = TimeLaunches(iters, [&] {
      k::StreamProbe(mode, src, mode == 2 ? dst.data() : nullptr, static_cast<int64_t>(bytes),
                     scratch.as<float>(), grid);
    });
    (void)rt;
  });
}

int eps_bench_gemm(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, int lower_only,
                   int iters, double* ms_avg) {
  return Guard([&] {
    const DType dt = ConfiguredDType();
    // EPSILON_HIP_BENCH_RANDOM=1: operands with independent pseudo-random entries instead of the
    // repeating 1024-value pattern (the Gram product inside a solve multiplies random data and
    // runs 4.8 ms slower than this microbenchmark on the pattern: is it the data?)
    const bool rnd = std::getenv("EPSILON_HIP_BENCH_RANDOM") != nullptr;
    auto operand = [&](int64_t count, uint64_t seed) {
      if (!rnd) return Synthetic(count, dt, 1.0);
      DVec v = DVec::Empty(count, dt);
      k::FillHash(v, seed);
      return v;
    };
    DVec A = operand(M * K, 0xA11CEull);
    // lower_only == 2: SYRK proper, both operands are the same buffer (the Gram product)
    DVec B = lower_only == 2 ? A : operand(K * N, 0xB0Bull);
    DVec C = DVec::Zeros(M * N, dt);
    const int64_t lda = trans_a ? K : M, ldb = trans_b ? N : K;
    *ms_avg = TimeLaunches(iters, [&] {
      k::Gemm(trans_a != 0, trans_b != 0, M, N, K, 1.0, A, lda, B, ldb, 0.0, C, M, lower_only != 0);
    });
  });
}

int eps_bench_spd_inverse(int64_t n, int iters, double* ms_avg) {
  return Guard([&] {
    const DType dt = ConfiguredDType();
    DVec G = Synthetic(n * n, dt, 1.0);
    DVec W0 = DVec::Zeros(n * n, dt);
    k::Gemm(false, true, n, n, n, 1.0 / n, G, n, G, n, 0.0, W0, n, false);
    k::AddDiag(W0, n, n, 1.0, nullptr);
    DVec W = DVec::Empty(n * n, dt);
    Runtime& rt = Runtime::Get();
    double total = 0;
    for (int i = 0; i < iters + 1; ++i) {
      k::Copy(W, W0);
      rt.Sync();
      hipEvent_t a, b;
      EPS_HIP(hipEventCreate(&a));
      EPS_HIP(hipEventCreate(&b));
      EPS_HIP(hipEventRecord(a, rt.stream()));
      k::SpdInverseInPlace(W, n);
      EPS_HIP(hipEventRecord(b, rt.stream()));
      EPS_HIP(hipEventSynchronize(b));
      float e = 0;
      EPS_HIP(hipEventElapsedTime(&e, a, b));
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
      if (i > 0) total += e;
    }
    *ms_avg = total / iters;
  });
}

int eps_test_spd_inverse_repeat(int64_t n, int form_a, int form_b, double* diff_fro, double* norm_fro) {
  return Guard([&] {
    EPS_CHECK(n > 0 && diff_fro != nullptr && norm_fro != nullptr);
    const DType dt = ConfiguredDType();
    DVec G = Synthetic(n * n, dt, 1.0);
    DVec W0 = DVec::Zeros(n * n, dt);
    k::Gemm(false, true, n, n, n, 1.0 / n, G, n, G, n, 0.0, W0, n, false);
    k::AddDiag(W0, n, n, 1.0, nullptr);
    G = DVec();
    DVec Xa = W0.Clone(), Xb = W0.Clone();
    struct Restore { ~Restore() { k::SetPotrfForm(-1); } } restore;
    k::SetPotrfForm(form_a);
    k::SpdInverseInPlace(Xa, n);
    k::SetPotrfForm(form_b);
    k::SpdInverseInPlace(Xb, n);
    Runtime& rt = Runtime::Get();
    rt.ResetSlots();
    const int s0 = rt.NewSlot(), s1 = rt.NewSlot();
    k::SumSqDiff(Xa, Xb, rt.SlotPtr(s0), false);
    k::SumSq(Xa, rt.SlotPtr(s1), false);
    rt.FetchSlots();
    *diff_fro = std::sqrt(rt.SlotValue(s0));
    *norm_fro = std::sqrt(rt.SlotValue(s1));
  });
}

int eps_bench_spd_inverse_columns(int64_t n, int64_t cnt, int iters, double* ms_avg) {
  return Guard([&] {
    const DType dt = ConfiguredDType();
    DVec G = Synthetic(n * n, dt, 1.0);
    DVec W0 = DVec::Zeros(n * n, dt);
    k::Gemm(false, true, n, n, n, 1.0 / n, G, n, G, n, 0.0, W0, n, false);
    k::AddDiag(W0, n, n, 1.0, nullptr);
    DVec W = DVec::Empty(n * n, dt);
    DVec Out = DVec::Empty(n * cnt, dt);
    Runtime& rt = Runtime::Get();
    double total = 0;
    for (int i = 0; i < iters + 1; ++i) {
      k::Copy(W, W0);
      rt.Sync();
      hipEvent_t a, b;
      EPS_HIP(hipEventCreate(&a));
      EPS_HIP(hipEventCreate(&b));
      EPS_HIP(hipEventRecord(a, rt.stream()));
      k::SpdInverseColumns(W, n, 0, cnt, Out);
      EPS_HIP(hipEventRecord(b, rt.stream()));
      EPS_HIP(hipEventSynchronize(b));
      float e = 0;
      EPS_HIP(hipEventElapsedTime(&e, a, b));
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
      if (i > 0) total += e;
    }
    *ms_avg = total / iters;
  });
}

int eps_bench_prox(int kind, int64_t n, int iters, double* ms_avg) {
  return Guard([&] {
    EPS_CHECK(n > 0 && iters > 0 && ms_avg != nullptr);
    const DType dt = ConfiguredDType();
    DVec v = Synthetic(n, dt, 4.0);
    DVec x = DVec::Empty(n, dt);
    k::ScaledZoneArgs a;
    a.lam = 0.7;
    DVec lamv;
    if (kind == 1) {  // per-element thresholds (weighted norm_1, quantile)
      lamv = Synthetic(n, dt, 1.0);
      k::Axpby(lamv, 0.0, lamv, 1.0);
      k::Fill(lamv, 0.7);
      a.lam_vec = &lamv;
    }
    *ms_avg = TimeLaunches(iters, [&] {
      if (kind == 2) k::MaxZero(x, v);
      else k::ScaledZone(x, v, a);
    });
  });
}

int eps_bench_svd(int64_t m, int64_t n, int rank, int max_sweeps, double perturb, double* ms_cold,
                  int* sweeps_cold, double* ms_warm, int* sweeps_warm, double* defects) {
  return Guard([&] {
    EPS_CHECK(m > 0 && n > 0 && ms_cold != nullptr);
    const DType dt = ConfiguredDType();
    Runtime& rt = Runtime::Get();
    // the reference's robust-PCA generator (python/epopt/problems/robust_pca.py:5-22):
    // rank-r part + 10 % sparse part of 10 * randn; a cheap host generator (sum of uniforms)
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto uni = [&] {
      st ^= st << 13;
      st ^= st >> 7;
      st ^= st << 17;
      return static_cast<double>(st >> 11) * (1.0 / 9007199254740992.0);
    };
    auto gauss = [&] { return (uni() + uni() + uni() + uni() - 2.0) * 1.7320508075688772; };
    std::vector<double> h(static_cast<size_t>(m) * std::max<int64_t>(rank, 1));
    for (auto& v : h) v = gauss();
    DVec A = DVec::FromHost(h.data(), m * std::max<int64_t>(rank, 1), dt);
    h.resize(static_cast<size_t>(n) * std::max<int64_t>(rank, 1));
    for (auto& v : h) v = gauss();
    DVec B = DVec::FromHost(h.data(), n * std::max<int64_t>(rank, 1), dt);
    DVec Y = DVec::Empty(m * n, dt);
    {
      std::vector<float> sp(static_cast<size_t>(m) * n);
      for (auto& v : sp) v = uni() < 0.1 ? static_cast<float>(10.0 * gauss()) : 0.0f;
      std::vector<double> col(m);
      // upload column by column through the fp64 staging the DVec helpers take
      std::vector<double> all(sp.begin(), sp.end());
      sp.clear();
      sp.shrink_to_fit();
      Y = DVec::FromHost(all.data(), m * n, dt);
    }
    if (rank > 0) k::Gemm(false, true, m, n, rank, 1.0, A, m, B, n, 1.0, Y, m);
    auto timed = [&](const DVec& W, const DVec& V, bool warm, double* ms, int* sweeps) {
      rt.Sync();
      hipEvent_t a, b;
      EPS_HIP(hipEventCreate(&a));
      EPS_HIP(hipEventCreate(&b));
      EPS_HIP(hipEventRecord(a, rt.stream()));
      const int sw = k::JacobiSvd(W, m, n, V, max_sweeps, warm, false);
      EPS_HIP(hipEventRecord(b, rt.stream()));
      EPS_HIP(hipEventSynchronize(b));
      float e = 0;
      EPS_HIP(hipEventElapsedTime(&e, a, b));
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
      *ms = e;
      if (sweeps) *sweeps = sw;
    };
    auto measure = [&](const DVec& Y0, const DVec& W, const DVec& V, double* out) {
      // out[0] = ||V^T V - I||_F / sqrt(n), out[1] = ||W V^T - Y||_F / ||Y||_F,
      // out[2] = ||offdiag(W^T W)||_F / ||diag(W^T W)||_F
      DVec T = DVec::Empty(n * n, dt);
      k::Gemm(true, false, n, n, n, 1.0, V, n, V, n, 0.0, T, n);
      k::AddDiag(T, n, n, -1.0, nullptr);
      rt.ResetSlots();
      const int s0 = rt.NewSlot(), s1 = rt.NewSlot(), s2 = rt.NewSlot(), s3 = rt.NewSlot(), s4 = rt.NewSlot();
      k::SumSq(T, rt.SlotPtr(s0), false);
      DVec R = Y0.Clone();
      k::Gemm(false, true, m, n, n, 1.0, W, m, V, n, -1.0, R, m);
      k::SumSq(R, rt.SlotPtr(s1), false);
      k::SumSq(Y0, rt.SlotPtr(s2), false);
      k::Gemm(true, false, n, n, m, 1.0, W, m, W, m, 0.0, T, n);
      k::SumSq(T, rt.SlotPtr(s3), false);
      DVec sig = DVec::Empty(n, dt);
      k::ColNorms(W, m, n, sig, false);
      DVec sq = DVec::Empty(n, dt);
      k::DiagMul(sq, 1.0, sig, sig, 0.0);
      k::SumSq(sq, rt.SlotPtr(s4), false);
      rt.FetchSlots();
      out[0] = std::sqrt(rt.SlotValue(s0) / static_cast<double>(n));
      out[1] = std::sqrt(rt.SlotValue(s1) / rt.SlotValue(s2));
      const double all = rt.SlotValue(s3), diag = rt.SlotValue(s4);
      out[2] = std::sqrt(std::max(0.0, all - diag) / diag);
    };
    DVec W = Y.Clone();
    DVec V = DVec::Empty(n * n, dt);
    timed(W, V, false, ms_cold, sweeps_cold);
    if (defects) measure(Y, W, V, defects);
    if (ms_warm != nullptr && perturb >= 0) {
      // a nearby matrix, as the next ADMM sweep would present: Y2 = Y + perturb * (scaled fill)
      DVec Y2 = Y.Clone();
      DVec N = Synthetic(m * n, dt, 1.0);
      k::Axpby(Y2, perturb, N, 1.0);
      DVec W2 = DVec::Empty(m * n, dt);
      k::Gemm(false, false, m, n, n, 1.0, Y2, m, V, n, 0.0, W2, m);
      timed(W2, V, true, ms_warm, sweeps_warm);
      if (defects) measure(Y2, W2, V, defects + 3);
    }
  });
}

int eps_bench_svd_device(const void* y_dev, int64_t m, int64_t n, int max_sweeps, double* ms, int* sweeps) {
  return Guard([&] {
    EPS_CHECK(y_dev != nullptr && m > 0 && n > 0 && ms != nullptr);
    Runtime& rt = Runtime::Get();
    EPS_HIP(hipDeviceSynchronize());  // the caller's stream may still be writing the matrix
    DVec Y = DVec::Borrow(const_cast<void*>(y_dev), m * n, F32);
    DVec W = Y.Clone();
    DVec V = DVec::Empty(n * n, F32);
    rt.Sync();
    hipEvent_t a, b;
    EPS_HIP(hipEventCreate(&a));
    EPS_HIP(hipEventCreate(&b));
    EPS_HIP(hipEventRecord(a, rt.stream()));
    const int sw = k::JacobiSvd(W, m, n, V, max_sweeps, false, false);
    EPS_HIP(hipEventRecord(b, rt.stream()));
    EPS_HIP(hipEventSynchronize(b));
    float e = 0;
    EPS_HIP(hipEventElapsedTime(&e, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms = e;
    if (sweeps) *sweeps = sw;
  });
}

int eps_tv1d(const double* v, size_t n, double lam, double* x) {
  return Guard([&] {
    const DType dt = ConfiguredDType();
    DVec vd = DVec::FromHost(v, n, dt);
    DVec xd = DVec::Empty(n, dt);
    k::Tv1d(xd, vd, lam);
    xd.ToHost(x);
  });
}

int eps_tv1d_device(const void* v_dev, void* x_dev, size_t n, int kind, double lam, int* levels) {
  return Guard([&] {
    EPS_CHECK(v_dev != nullptr && x_dev != nullptr);
    EPS_CHECK_MSG(kind == EPS_BLOB_DEVICE_F32 || kind == EPS_BLOB_DEVICE_F64, "bad kind");
    const DType dt = kind == EPS_BLOB_DEVICE_F32 ? F32 : F64;
    DVec v = DVec::Borrow(const_cast<void*>(v_dev), static_cast<int64_t>(n), dt);
    DVec x = DVec::Borrow(x_dev, static_cast<int64_t>(n), dt);
    Runtime::Get();
    EPS_HIP(hipDeviceSynchronize());  // v may have been produced on any stream of the caller
    k::Tv1d(x, v, lam);
    Runtime::Get().Sync();
    if (levels) *levels = k::Tv1dLastLevels();
  });
}

}  // extern "C"

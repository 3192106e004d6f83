/* epsilon_hip.h - C ABI of the MI355X-native prox-ADMM solver core for Epsilon.
 *
 * This library replaces the solver side of Epsilon that sits behind the CPython extension
 * `epopt._solve` (reference python/epopt/solvemodule.cc).  The payloads are the ones the
 * unchanged frontend already produces:
 *
 *   problem / f_expr   protobuf wire bytes of `Problem` / `Expression`
 *                      (reference proto/epsilon/expression.proto:205-346)
 *   solver_params      protobuf wire bytes of `SolverParams` (proto/epsilon/solver_params.proto)
 *   data               {location -> raw bytes}: dense = float64 column-major
 *                      (reference python/epopt/constant.py:12-17)
 *   variable values    float64, column-major, m*n per variable (solvemodule.cc:24-56,166-176)
 *   status             protobuf wire bytes of `SolverStatus` (proto/epsilon/solver.proto:4-60)
 *
 * Every function returns 0 on success.  A non-zero return is what the reference reports as
 * `_solve.error("CHECK failed")` (solvemodule.cc:158,185,245-248); the message is available
 * from eps_last_error() (thread-local).  There is no CPU fallback: without a HIP device every
 * compute entry point fails with an error.
 *
 * Not re-entrant per solver handle (the reference is not either: solvemodule.cc:17-22,
 * prox/vector_prox.h:75-76); one process drives one GPU.
 */
#ifndef EPSILON_HIP_H_
#define EPSILON_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One entry of the data map / of a variable-value map.
 * Replaces: PyDict {str: str} walked by WriteConstants / GetVariableVector
 * (reference solvemodule.cc:24-43,58-72). */
enum {
  EPS_BLOB_HOST = 0,       /* ptr -> host bytes, len = byte count (the reference's only form) */
  /* Device blobs are borrowed, not copied, and are read on the library's own non-blocking HIP
   * stream.  The entry point that receives them waits for the device once (hipDeviceSynchronize),
   * so work already enqueued by the caller is complete; they must stay alive and unmodified
   * while a solver handle built from them exists. */
  EPS_BLOB_DEVICE_F32 = 1, /* ptr -> device memory, float,  len = element count (borrowed)   */
  EPS_BLOB_DEVICE_F64 = 2  /* ptr -> device memory, double, len = element count (borrowed)   */
};
typedef struct eps_blob {
  const char* key;  /* NUL-terminated location / variable id */
  const void* ptr;  /* host: read during the call only - the solver-handle entry points copy the
                     * bytes, as the reference copies every blob (solvemodule.cc:58-72), so the
                     * caller may free or overwrite them as soon as the call returns;
                     * device: borrowed while a solver handle built from it exists */
  size_t len;
  int kind;
} eps_blob;

/* A CVXPY Parameter binding: (parameter_id, serialized `Constant`).
 * Replaces: the `parameters` iterable of `_solve.solve` (solvemodule.cc:89-106). */
typedef struct eps_param {
  const char* id;
  const void* constant_proto;
  size_t len;
} eps_param;

typedef struct eps_result eps_result; /* status bytes + {variable_id -> float64 bytes} */
typedef struct eps_solver eps_solver; /* a live solver: operators, factorisation, iterates in HBM */

/* ---- process-level ------------------------------------------------------------------------ */

/* Message of the last failed call on this thread ("" if none). */
const char* eps_last_error(void);
/* "epsilon_hip <version> gfx950" */
const char* eps_version(void);
/* Options (all optional): "dtype" = "f32" (default) | "f64"  compute type of subsequent solves
 * (env EPSILON_HIP_DTYPE);  "device" = ordinal, before first use (env EPSILON_HIP_DEVICE /
 * LOCAL_RANK).  Rides outside SolverParams because the frontend passes only its own kwargs
 * (reference python/epopt/cvxpy_solver.py:69).
 * "refine" = "auto" (default) | "<steps>"  fp32 mode only: iterative refinement of the block
 * LDL^T solve behind SUM_SQUARE / ZERO / AFFINE.  The elimination inverts its pivot blocks
 * explicitly (reference vector/block_cholesky.cc:119-133, linear/dense_matrix_impl.cc:21-30); in
 * fp64, as the reference runs, that is harmless, in fp32 the forward error is kappa * 6e-8 per
 * solve.  "auto" estimates the condition of every pivot block at Init (kappa_1 from two passes
 * over the block and its inverse; where that exceeds 1e3, kappa_2 from six power iterations on
 * each) and adds 1 / 2 / 3 refinement steps (residual against the blocks as given) above
 * 1e3 / 3e4 / 3e6; below 1e3 - every BASELINE.json lasso - the solve is the reference's
 * sequence of operations unchanged.
 * "0" switches it off (env EPSILON_HIP_REFINE).
 * "graph_generic" = "0" (default) | "1" | "2"  replay the sweeps of the generic operator path
 * between two residual checks from a hipGraph (1: once a run has lasted 50 sweeps, 2: from the
 * second sweep on); bit-identical to eager launches and, as measured, no faster (env
 * EPSILON_HIP_GRAPH_GENERIC). */
int eps_set_option(const char* key, const char* value);
/* Number of visible HIP devices (0 if none); never fails. */
int eps_device_count(void);

/* ---- the two entry points of `epopt._solve` ----------------------------------------------- */

/* Replaces `_solve.solve(problem, parameters, solver_params, data)` (solvemodule.cc:110-187).
 * On success *out holds the SolverStatus bytes and one float64 column-major vector per
 * variable reachable from the problem, in lexicographic id order (solvemodule.cc:166). */
int eps_solve(const void* problem, size_t problem_len, const void* solver_params,
              size_t solver_params_len, const eps_blob* data, size_t ndata,
              const eps_param* params, size_t nparams, eps_result** out);

/* Replaces `_solve.eval_prox(f_expr, lam, data, v)` (solvemodule.cc:189-242):
 * argmin_x lam*f(x) + 1/2||x - v||^2 for one PROX_FUNCTION expression.  `v` holds one
 * float64 host blob per variable id.  The result has an empty status. */
int eps_eval_prox(const void* f_expr, size_t f_expr_len, double lambda, const eps_blob* data,
                  size_t ndata, const eps_blob* v, size_t nv, eps_result** out);

/* Result accessors (pointers stay valid until eps_result_free). */
int eps_result_status(const eps_result* r, const void** bytes, size_t* len);
size_t eps_result_num_vars(const eps_result* r);
int eps_result_var(const eps_result* r, size_t i, const char** id, const double** values,
                   size_t* count);
/* Copies variable i's values (count * 8 bytes) into the caller's buffer with the library's host
 * threads - the binding's replacement for one thread's memcpy of a 0.8 GB iterate into a fresh
 * Python bytes object (solvemodule.cc:166-176 builds each output string the same way, one copy).
 * 1 if i is out of range or `bytes` is not the variable's size. */
int eps_result_copy_var(const eps_result* r, size_t i, void* dst, size_t bytes);
void eps_result_free(eps_result* r);

/* ---- solver handles: warm start, staged runs, timing --------------------------------------- */
/* Replaces the process-global warm-start cache (solvemodule.cc:22,142-156): the caller keeps the
 * handle, so the data matrix, the cached factorisation and x/y/u stay resident in HBM across
 * calls.  Host blobs are copied by create / set_parameter (free them when the call returns); the
 * copy lives until eps_solver_destroy, as the reference's DataMap does for the life of its
 * Solver (solvemodule.cc:58-72) - a re-Init after a parameter change may have to upload it
 * again - so a handle costs host memory equal to its host blobs (1.9 GB for the 60000 x 4000
 * feature matrix of BASELINE.json configs[3]) for as long as it is cached; device blobs are
 * borrowed until destroy.  Re-binding a location that is already bound replaces its contents:
 * the next eps_solver_init rebuilds everything that depended on it. */
int eps_solver_create(const void* problem, size_t problem_len, const void* solver_params,
                      size_t solver_params_len, const eps_blob* data, size_t ndata,
                      eps_solver** out);
/* (Re)bind a CVXPY Parameter value; takes effect at the next eps_solver_init
 * (reference algorithms/solver.cc:109-116). */
int eps_solver_set_parameter(eps_solver* s, const char* id, const void* constant_proto,
                             size_t len, const eps_blob* data, size_t ndata);
/* Build operators / Gram matrix / factorisation (the reference's Solver::Init()).  With
 * warm_start set in the params, x/y/u of a previous run are kept (prox_admm.cc:115-120). */
int eps_solver_init(eps_solver* s);
/* Run ADMM sweeps: until OPTIMAL / max_iterations if max_sweeps < 0, else at most max_sweeps
 * more.  *sweeps_done may be NULL. */
int eps_solver_run(eps_solver* s, int max_sweeps, int* sweeps_done);
/* Snapshot of status + variables (same layout as eps_solve's result). */
int eps_solver_result(eps_solver* s, eps_result** out);
/* Wall-clock seconds spent in init and in the sweep loop (device-synchronised). */
int eps_solver_timing(const eps_solver* s, double* init_seconds, double* loop_seconds);
void eps_solver_destroy(eps_solver* s);

/* ---- sharded solves over the GPUs of one node (one process per GPU) ------------------------- */
/* No reference counterpart: the reference has no distributed mode.  Each rank
 * passes its LOCAL problem: sharded variables / constraint rows hold this rank's slice, a data
 * matrix feeding a replicated row from a sharded variable holds this rank's column slab.
 * Sharded keys are declared once with eps_shard_keys; the solver then all-reduces exactly the
 * contractions over sharded keys (lasso: m floats per sweep + a few doubles per residual
 * check) and returns each rank's slice of the sharded variables. */
#define EPS_UNIQUE_ID_BYTES 128
/* rank 0: RCCL unique id to be distributed to the other ranks by the caller. */
int eps_comm_unique_id(void* out128);
/* all ranks: join the RCCL communicator (collective call). */
int eps_comm_init_rccl(int rank, int world, const void* id128);
/* all ranks: host-staged collective through a caller-provided function that sums `count`
 * elements (dtype 0 = float, 1 = double) in place across ranks; used by tests that run the
 * ranks on one GPU or over gloo. */
typedef void (*eps_allreduce_fn)(void* host_buf, size_t count, int dtype, void* ctx);
int eps_comm_init_callback(int rank, int world, eps_allreduce_fn fn, void* ctx);
/* One all-reduce + one all-gather of `count` floats over the communicator, checked (sum of ones
 * == world size): a barrier that also takes RCCL's first-use setup out of a timed Init. */
int eps_comm_warmup(size_t count);
/* COLLECTIVE, optional, after eps_comm_init_*: a one-shot peer-write exchange window over the
 * direct xGMI links (HIP IPC) for the per-sweep messages of a column-sharded solve - m floats
 * each, latency-bound.  With it the sharded fused sweep makes no collective call
 * at all: every rank writes its part straight into every peer's window from inside the sweep's
 * own kernels and sums what it received in rank order (csrc/kernels_peer.hip), and the sweeps
 * between two residual checks are replayed from one hipGraph.  The communicator above stays in
 * charge of the large setup messages and of the residual scalars.  `slot_floats` = largest
 * message in 32-bit words (an fp64 solve needs two per row of the m-vector; 0: 16384); `rehearse_ranks` > 1 (single-rank communicator only) makes this process
 * play ONE rank of that many for timing rehearsals on one GPU (results are not a solve's).
 * *enabled = 1 if every rank has the window and its self test passed, else 0 - then nothing
 * changes (RCCL per sweep) and eps_last_error() says why.  The reference has no counterpart
 * (it has no distributed mode). */
int eps_comm_enable_peer(size_t slot_floats, int rehearse_ranks, int* enabled);
/* Drop the window again (every rank, or none): later solves use the communicator's collectives. */
int eps_comm_disable_peer(void);
int eps_comm_shutdown(void);
/* Replace the set of sharded block keys (variable ids and "constraint:<i>" rows). */
int eps_shard_keys(const char* const* keys, size_t nkeys);
/* Consensus form (call after eps_shard_keys, which resets it): every rank passes
 *   minimise f_g(x_g) + h(z)  subject to  x_g - z = 0
 * with x_g and the constraint row sharded and z replicated.  An objective term over sharded
 * variables only is then this rank's OWN term (its argument rows stay on the rank; nothing of
 * it is all-reduced); the update of z all-reduces the n entries of sum_g (x_g + u_g) - the
 * z-averaging step - and the residual norms / the max over per-term norms are reduced across
 * ranks.  The iterates are those of the single-process solve of the stacked problem
 * (terms f_1..f_G, h; G consensus constraints). */
int eps_shard_consensus_terms(int on);

/* Largest condition estimate of a pivot block and refinement step count of the block
 * factorisations set up since the last reset (fp32 mode; see the "refine" option).  Either
 * pointer may be NULL; reset != 0 clears the record afterwards.  Diagnostics / tests. */
int eps_block_solve_stats(double* max_condition, int* max_refine_steps, int reset);

/* Sweeps of the generic operator path that were replayed from a captured hipGraph, and the
 * captures made, since the last reset (EPSILON_HIP_GRAPH_GENERIC=0 turns the replay off).
 * Either pointer may be NULL.  Diagnostics / tests. */
int eps_graph_stats(long long* replayed_sweeps, long long* captures, int reset);

/* ---- live kernel timing ---------------------------------------------------------------------- */
/* When enabled, every hot kernel launch is bracketed by HIP events on the solver's stream.
 * eps_profile_dump writes one line per tag "tag count total_ms\n" (NUL-terminated, truncated
 * to cap) after synchronising; eps_profile_reset clears the totals. */
int eps_profile_enable(int on);
int eps_profile_reset(void);
int eps_profile_dump(char* buf, size_t cap);

/* ---- per-operator entry points (parity tests pin each kernel through these) ----------------- */

/* y = op(A) x for a serialized `LinearMap` (reference linear/linear_map.cc:83-104 +
 * LinearMapImpl::Apply); `transpose` is a bit set: 1 applies the adjoint, 2 applies the map's
 * Inverse() (reference LinearMapImpl::Inverse, e.g. dense_matrix_impl.cc:21-30).
 * x, y: host float64. */
int eps_linear_map_apply(const void* linear_map, size_t len, const eps_blob* data, size_t ndata,
                         int transpose, const double* x, size_t nx, double* y, size_t ny);
/* C = A op B with op = '+' or '*' through the type-dispatch tables (reference
 * linear/linear_map_add.cc:234-284, linear_map_multiply.cc:249-299).  *result_type receives
 * the ImplType of the result (0 dense, 1 sparse, 2 diagonal, 3 scalar, 4 kronecker);
 * dense (m*n float64, column-major) receives its values; ta / tb transpose the operand first. */
int eps_linear_map_binary(char op, const void* a, size_t a_len, int ta, const void* b,
                          size_t b_len, int tb, const eps_blob* data, size_t ndata,
                          int* result_type, int64_t* m, int64_t* n, double* dense,
                          size_t dense_capacity);
/* Inverse of a serialized map (reference LinearMapImpl::Inverse), dense values out. */
int eps_linear_map_inverse(const void* linear_map, size_t len, const eps_blob* data,
                           size_t ndata, double* dense, size_t dense_capacity);
/* Micro-benchmark of the dense mat-vec kernels on device-resident synthetic data: average
 * milliseconds per launch over `iters` launches (HIP events on the solver stream). */
int eps_bench_gemv(int trans, int64_t rows, int64_t cols, int iters, double* ms_avg);
/* HBM ceiling probe on `bytes` of device memory (device_ptr, 16-byte aligned, or NULL for a
 * synthetic buffer): mode 0 = read-only with non-temporal loads, 1 = read-only, 2 = copy to a
 * scratch buffer (bytes read + bytes written); grid = workgroups of 256 threads (0: 2048).
 * Measurement only: bench.py reports the sweep as
 * a fraction of these beside the 8 TB/s vendor peak. */
int eps_bench_stream(const void* device_ptr, size_t bytes, int mode, int grid, int iters,
                     double* ms_avg);
/* Same for C = op(A) op(B) (M x N x K); lower_only = SYRK-style. */
int eps_bench_gemm(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, int lower_only,
                   int iters, double* ms_avg);
/* SPD inverse of an n x n synthetic matrix. */
int eps_bench_spd_inverse(int64_t n, int iters, double* ms_avg);
int eps_bench_spd_inverse_columns(int64_t n, int64_t cnt, int iters, double* ms_avg);
/* Test entry: the explicit inverse (reference linear/dense_matrix_impl.cc:21-30) of one synthetic
 * n x n positive definite matrix computed twice, with the Cholesky step in form_a and form_b
 * (0 = the fused step of the f32 mode, 1 = separate diagonal / panel launches, -1 = default):
 * ||X_a - X_b||_F and ||X_a||_F.  form_a == form_b checks run-to-run determinism (a data race
 * between the workgroups of a launch shows as a non-zero difference, above all on a busy GPU). */
int eps_test_spd_inverse_repeat(int64_t n, int form_a, int form_b, double* diff_fro, double* norm_fro);

/* Microbenchmark: average milliseconds of one launch of a standalone elementwise prox kernel on
 * n synthetic elements of the configured dtype - kind 0 scaled-zone (NORM_1) with scalar
 * parameters, 1 with a per-element threshold vector, 2 projection onto R+ (reference
 * prox/scaled_zone.cc:90-101, prox/non_negative.cc:8).  Algorithmic bytes 2 n s (3 n s for kind 1). */
int eps_bench_prox(int kind, int64_t n, int iters, double* ms_avg);

/* Microbenchmark of the singular value decomposition behind the orthogonally-invariant prox
 * operators (reference prox/ortho_invariant.cc:36-50) on the reference's robust-PCA matrix
 * (problems/robust_pca.py:5-22: rank-`rank` part + 10 % sparse part) of the configured dtype:
 * milliseconds and Jacobi sweeps of a cold decomposition and of a warm-started one of a matrix
 * `perturb` away (perturb < 0: skipped).  defects (6 doubles, may be NULL), cold then warm:
 * ||V^T V - I||_F / sqrt(n), ||W V^T - Y||_F / ||Y||_F, ||offdiag(W^T W)||_F / ||diag(W^T W)||_F. */
int eps_bench_svd(int64_t m, int64_t n, int rank, int max_sweeps, double perturb, double* ms_cold,
                  int* sweeps_cold, double* ms_warm, int* sweeps_warm, double* defects);

/* The same decomposition (cold) of a caller-provided m x n column-major float32 matrix in HBM. */
int eps_bench_svd_device(const void* y_dev, int64_t m, int64_t n, int max_sweeps, double* ms, int* sweeps);

/* Exact 1-D total-variation prox of v (n float64) with weight lam
 * (reference prox/total_variation_1d.cc:21 -> glmgen tf_dp). */
int eps_tv1d(const double* v, size_t n, double lam, double* x);

/* The same on device-resident data (the work runs on the library's own non-blocking stream; the
 * call waits for the device first, so v may come straight from the caller's streams): v, x are
 * device pointers to n elements of `kind`
 * (EPS_BLOB_DEVICE_F32 / EPS_BLOB_DEVICE_F64); *levels (may be NULL) receives the depth of the
 * level-set recursion.  Synchronises before returning. */
int eps_tv1d_device(const void* v_dev, void* x_dev, size_t n, int kind, double lam, int* levels);

#ifdef __cplusplus
}
#endif

#endif /* EPSILON_HIP_H_ */

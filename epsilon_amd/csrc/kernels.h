// Launchers for the hand-written gfx950 kernels.  All launch on Runtime::Get().stream().
// DVec arguments carry their dtype (f32 / f64); scalars are passed as double and narrowed in
// the kernel.  Reference call sites replaced are cited per group.
#pragma once

#include "device.h"

namespace eps {
struct PeerView;
namespace k {

// ---- K7: BlockVector += -= *=, scalar / diagonal Apply -------------------------------------
// (reference vector/block_vector.cc:9-48, linear/scalar_matrix_impl.h:24,
//  linear/diagonal_matrix_impl.h:23)
void Fill(const DVec& y, double v);
void Copy(const DVec& dst, const DVec& src);
// y[i] = deterministic pseudo-random value in (-1, 1) (hash of seed and index)
void FillHash(const DVec& y, uint64_t seed);
// y = a*x + b*y   (b == 0 never reads y)
void Axpby(const DVec& y, double a, const DVec& x, double b);
// y = a * d .* x + b*y
void DiagMul(const DVec& y, double a, const DVec& d, const DVec& x, double b);
// host f64 staging buffer (device memory, doubles) -> dst (dst.dt)
void ConvertFromF64(const DVec& dst, const double* src_dev);
// src (any dt) -> device doubles
void ConvertToF64(double* dst_dev, const DVec& src);
void ConvertFromF32(const DVec& dst, const float* src_dev);

// ---- K8: norms (reference vector/block_vector.cc:87-93) -----------------------------------
// *slot = (accumulate ? *slot : 0) + sum_i x_i^2, accumulated in double, deterministic.
void SumSq(const DVec& x, double* slot, bool accumulate);
// *slot (+)= sum_i (x_i - y_i)^2
void SumSqDiff(const DVec& x, const DVec& y, double* slot, bool accumulate);
void Dot(const DVec& x, const DVec& y, double* slot, bool accumulate);

// ---- K1/K2/K3: dense mat-vec (reference linear/dense_matrix_impl.cc:55-67 dgemv_) --------
// y = alpha * op(A) x + beta*y ; A is rows x cols column-major with leading dimension lda.
void Gemv(bool trans, int64_t rows, int64_t cols, double alpha, const DVec& A, int64_t lda,
          const DVec& x, double beta, const DVec& y);

// y = alpha * S x + beta * y for a symmetric S in full storage (n x n, leading dimension lds):
// reads only the lower-triangle tiles, half the bytes of Gemv.
// `work` (optional, SymvWorkspace(n) elements of S's dtype): per-tile partials at a fixed
// address instead of a pool allocation per call (launches captured in a hipGraph need that).
void Symv(int64_t n, double alpha, const DVec& S, int64_t lds, const DVec& x, double beta,
          const DVec& y, const DVec* work = nullptr);
int64_t SymvWorkspace(int64_t n);
// The same apply from a tile-packed copy of the lower tiles (every 128 x 128 tile contiguous,
// zero-padded at the edge): SymvPack builds it once (SymvPackedSize(n) values), SymvPacked applies it.
int64_t SymvPackedSize(int64_t n);
DVec SymvPack(int64_t n, const DVec& S, int64_t lds);
void SymvPacked(int64_t n, double alpha, const DVec& P, const DVec& x, double beta, const DVec& y,
                const DVec* work = nullptr);

// y[r] = alpha * sum_{k < nparts} partial[k*rows + r] + beta*y[r], fixed summation order.
// `add` (optional) is added to the result afterwards: y = (alpha*sum + beta*y) + add.
void ReducePartials(int64_t rows, int nparts, const DVec& partial, double alpha, double beta,
                    const DVec& y, const DVec* add = nullptr);

// ---- fused lasso sweep: one pass over A per ADMM sweep (kernels_fused.hip) ------------------
struct LassoFusedArgs {
  int64_t m = 0, n = 0, lda = 0;
  DVec A;        // m x n column-major
  DVec w;        // m: the block-diagonal-scaled forward-substitution result of this sweep
  double kappa = 0;                 // x0 = v0 + kappa * (A^T w)
  double Bs = 0, Cs = 0, a1 = 0;    // prox-1 pre / post scaling, y1 = a1 * x1
  double lam = 0, sz_alpha = 1, sz_beta = 1, sz_M = 0;
  DVec sz_alpha_vec, sz_beta_vec;   // optional per-column alpha / beta (n entries; f32 pass only)
  DVec u, x0, x1, y0, y1, y1prev;   // n each, updated in place
  DVec tpart;                       // LassoFusedGrid(m, n) * m: per-workgroup partials of A v0'
  unsigned* epoch = nullptr;        // optional device counter, incremented once per launch
  // chain = 1: the two-block driver's sweep (prox_admm_two_block.cc:97-112); the arrays then mean
  // u -> u0, y0 -> z0, y1 -> z1, y1prev -> z0_prev, e0 -> u1, e1 -> z1_prev; a0, a1: the consensus
  // constraint a0 x0 + a1 x1 = 0 the z-update projects onto.  f32 only.
  int chain = 0;
  double a0 = 1;
  DVec e0, e1;
};
bool LassoFusedSupported(int64_t m, int64_t n, const DVec& A, int64_t lda);
int LassoFusedGrid(int64_t m, int64_t n, DType dt = F32);
void LassoFusedPass(const LassoFusedArgs& args);
// out6 = {||y0||^2, ||y1||^2, ||y0 + y1||^2, ||y1 - y1prev||^2, ||u||^2, peer_err ? 1 : 0} (device
// doubles), one launch; `work`: 64 * 5 + 1 doubles, zero-initialised once (the last double is a
// ticket counter); `peer_err` (optional): the device-side error word of the peer exchange.
void LassoFusedNorms(const DVec& u, const DVec& y0, const DVec& y1, const DVec& y1prev, double* out6,
                     const DVec& work, const unsigned* peer_err = nullptr);

// ---- one-shot peer-write exchange (kernels_peer.hip; PeerView in comm.h) ----------------------
void PeerBumpEpoch(const PeerView& pv);
// y = (sum over ranks, in rank order, of alpha * sum_k partial[k*rows + r]) + add
void PeerReduceExchange(const PeerView& pv, int64_t rows, int nparts, const DVec& partial,
                        double alpha, const DVec* add, const DVec& y);
bool PeerSlabApplySupported(const PeerView& pv, int64_t m, int64_t slab, const DVec& D, int64_t ldd);
// wpad[q*slab + j] = scale * D[:, q*slab + j] . p for every rank q (this rank computes q = rank,
// lo = rank*slab, pushes it to the peers and gathers theirs); columns >= m give 0.
void PeerSlabApplyExchange(const PeerView& pv, int64_t m, int64_t slab, int64_t lo, const DVec& D,
                           int64_t ldd, double scale, const DVec& p, const DVec& wpad);

// ---- K4: dense mat-mat (reference linear/linear_map_multiply.cc:14-37 dgemm_) -------------
// C (M x N, ldc) = alpha * op(A) (M x K) * op(B) (K x N) + beta * C ; column-major.
// lower_only: compute only tiles touching the lower triangle (SYRK-style); the caller
// mirrors with SymmetrizeFromLower.
void Gemm(bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha,
          const DVec& A, int64_t lda, const DVec& B, int64_t ldb, double beta, const DVec& C,
          int64_t ldc, bool lower_only = false);
// The same for batch * outer problems (blockIdx.z = z): operands at element offsets
// (z % batch) * s + (z / batch) * s2, results at z * sC.
void GemmBatched(bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha,
                 const DVec& A, int64_t lda, int64_t sA, const DVec& B, int64_t ldb, int64_t sB,
                 double beta, const DVec& C, int64_t ldc, int64_t sC, int64_t batch,
                 bool lower_only = false, int64_t outer = 1, int64_t sA2 = 0, int64_t sB2 = 0);
void SymmetrizeFromLower(const DVec& C, int64_t n, int64_t ldc);
// Large f32 products on the f16 matrix cores with two-term split operands
// (kernels_gemm_f16split.hip): f32 accuracy at 3/16 of the f32 MFMA time.  Gemm routes there by
// itself (f32, M, N >= 2048, K >= 256, >= 8e9 multiply-adds); false = not eligible, nothing done.
bool GemmSplitF16(bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha, const DVec& A,
                  int64_t lda, const DVec& B, int64_t ldb, double beta, const DVec& C, int64_t ldc,
                  bool lower_only);
// fp64 products on the software-pipelined f64 MFMA kernel (kernels_gemm_f64.hip), arguments as
// GemmBatched (n1 = inner batch count, batch = n1 * outer); false if the operands are not
// 16-byte aligned (nothing is done).
bool GemmF64Pipe(bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha, const DVec& A,
                 int64_t lda, int64_t sA, const DVec& B, int64_t ldb, int64_t sB, double beta,
                 const DVec& C, int64_t ldc, int64_t sC, int64_t n1, int64_t batch, bool lower_only,
                 int64_t sA2, int64_t sB2);
// C = alpha A B (no transposes, beta = 0) with B (kmode 3) or A (kmode 4) lower triangular, on the
// split-f16 kernel with a k range per tile; false: not eligible, nothing done.
bool GemmSplitF16KRange(int kmode, int64_t M, int64_t N, int64_t K, double alpha, const DVec& A, int64_t lda,
                        const DVec& B, int64_t ldb, const DVec& C, int64_t ldc);
// Lower tiles of C = X^T X for a lower-triangular X (zeros stored above the diagonal) in one launch
// of the split-f16 kernel with a k range per tile; false: not eligible (f32, n >= 2048), nothing done.
bool SyrkSplitF16LowerTriangular(int64_t n, const DVec& X, int64_t ldx, const DVec& C, int64_t ldc);
// C (M x N, ldc == M, N <= 16) = alpha op(A) B + beta C as a mat-vec with N right-hand sides
// (kernels_gemv_multi.hip); false if the shape / alignment is not covered (nothing is done).
bool MultiGemv(bool transA, int64_t M, int64_t N, int64_t K, double alpha, const DVec& A, int64_t lda,
               const DVec& B, int64_t ldb, double beta, const DVec& C, int64_t ldc);
// HBM ceiling probes: mode 0 read (non-temporal), 1 read, 2 copy; scratch holds `grid` floats.
void StreamProbe(int mode, const void* src, void* dst, int64_t bytes, float* scratch, int grid);

// dst (rows x cols, ld = rows) = alpha * op(src)
void MatCopy(bool trans, int64_t rows, int64_t cols, double alpha, const DVec& src,
             int64_t lds, const DVec& dst);
// W[i,i] += alpha (d undefined) or W[i,i] += alpha*d[i]
void AddDiag(const DVec& W, int64_t n, int64_t ld, double alpha, const DVec* d);
// colsum_dev[j] = sum_i |A(i, j)| as device doubles (max_j = the 1-norm of A): condition estimates
void ColAbsSums(const DVec& A, int64_t rows, int64_t cols, int64_t lda, double* colsum_dev);
// y = x / sqrt(*normsq_dev) (device scalar; 0 leaves x unscaled): power-iteration normalisation
void ScaleByInvNorm(const DVec& y, const DVec& x, const double* normsq_dev);
// dst (mA*mB x nA*nB) = kron(A, B), all column-major contiguous
void KronDense(const DVec& dst, const DVec& A, int64_t mA, int64_t nA, const DVec& B,
               int64_t mB, int64_t nB);
// y = A x for W = diag? helpers used by Kronecker apply are composed from Gemm.

// ---- K5: symmetric definite inverse (reference linear/dense_matrix_impl.cc:21-30) ---------
// W (n x n, ld = n, symmetric positive definite, full storage) -> W^{-1} in place.
// Blocked Cholesky + triangular inverse + X^T X, all on device.  Throws if a pivot is <= 0.
void SpdInverseInPlace(const DVec& W, int64_t n);
// Cholesky step form of the next factorisations: -1 by environment (default), 0 the fused f32 step,
// 1 the diagonal + panel launches (tests hold the two forms to each other)
void SetPotrfForm(int form);
// Columns [lo, lo + cnt) of W^-1 into Out (n x cnt, ld n); W is overwritten by its Cholesky
// factor.  Cholesky + two blocked triangular solves on the cnt unit columns.
void SpdInverseColumns(const DVec& W, int64_t n, int64_t lo, int64_t cnt, const DVec& Out);

// ---- K6 / K12: elementwise and group prox kernels ------------------------------------------
// reference prox/scaled_zone.cc:78-104 ; lam / alpha / beta are either uniform scalars or
// per-element device vectors (pass defined DVecs to use the vector form).
struct ScaledZoneArgs {
  double lam = 0, alpha = 1, beta = 1, M = 0, C = 0;
  const DVec* lam_vec = nullptr;
  const DVec* alpha_vec = nullptr;
  const DVec* beta_vec = nullptr;
  int64_t period = 0;  // >0: alpha/beta/lam vectors are indexed by (i % period)
};
void ScaledZone(const DVec& x, const DVec& v, const ScaledZoneArgs& args);
// reference prox/non_negative.cc:8
void MaxZero(const DVec& x, const DVec& v);
// reference prox/norm_2.cc:11-16 ; normsq is a device slot holding ||v||^2
void Norm2Shrink(const DVec& x, const DVec& v, double lam, const double* normsq);
// x = soft-threshold of singular values etc. is composed from ScaledZone.

// ---- epigraph projections (reference prox/sum_square.cc:42-57, prox/scaled_zone.cc:152-279) ----
// SUM_SQUARE epigraph: lam = max(0, largest real root of the cubic of newton.cc:293-323) from
// ||u||^2 (device slot) and s (1 element); x = u / (1 + 2 lam), t = s + lam.  No host sync.
void SumSquareEpigraph(const DVec& x, const DVec& t, const DVec& u, const DVec& s,
                       const double* normsq, double* lam_scratch);
// Scaled-zone epigraph, pass 1: keys k_i = (|y_i| - M) / w_i and weights w_i^2 for the samples
// that can move (w = alpha for y > 0, beta for y < 0), zero weight otherwise; *fval += f(y).
void ZoneEpigraphKeys(const DVec& v, double alpha, double beta, const DVec* alpha_vec,
                      const DVec* beta_vec, double M, double C, double* key, double* w2,
                      double* fval);
// pass 2: sums[0] += sum w2*k, sums[1] += sum w2, sums[2] += count over {k_i > lam, w2_i > 0}
void ZoneEpigraphSums(int64_t n, const double* key, const double* w2, double lam, double* sums);

// ---- K11: SVD for the orthogonally-invariant proxes (reference prox/ortho_invariant.cc) ------
// One-sided Jacobi on W (m x n, ld = m): on return W = U*Sigma (orthogonal columns) and the
// input equals W V^T; V (n x n) is overwritten.  Returns the number of sweeps used.
// warm: V holds an orthogonal matrix on entry and W has already been multiplied by it (the
// decomposition continues from there: Y = W V^T holds throughout).
// row_sharded: W holds this rank's block of rows (V replicated); the column inner products are
// all-reduced over the communicator (block form only).
int JacobiSvd(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps = 40,
              bool warm = false, bool row_sharded = false);
// The same decomposition by the block algorithm (pairs of 32-column panels: batched Gram on the
// MFMA kernel, 64 x 64 eigenproblems on chip, batched GEMM updates); JacobiSvd switches to it
// from 1536 columns up (measured crossover on MI355X; EPSILON_HIP_SVD=block|scalar forces one).
int BlockJacobiSvd(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps = 40,
                   bool warm = false, bool row_sharded = false);
// The same with the rotations applied to W only (no right factor is accumulated: a step moves
// 0.6 of the bytes); fp32 block form.  JacobiSvdCanSkipV says whether it applies.
bool JacobiSvdCanSkipV(int64_t m, int64_t n, DType dt);
int JacobiSvdNoV(const DVec& W, int64_t m, int64_t n, int max_sweeps = 40);
void ColNorms(const DVec& W, int64_t m, int64_t n, const DVec& sigma, bool row_sharded = false);
// W[:, j] *= xt[j] / sigma[j]   (0 where sigma[j] == 0, as ortho_invariant.cc:44-49)
void ColScaleByRatio(const DVec& W, int64_t m, int64_t n, const DVec& sigma, const DVec& xt);

// ---- batched ("segmented") and Newton-family operators (kernels_segprox.hip) ------------------
// A segment is one slice of the argument the reference's axis loop would visit
// (prox/vector_prox.cc:150-177): entry p of segment s lives at s*seg_stride + p*elem_stride.
struct Segs {
  int64_t count = 1, len = 0, seg_stride = 0, elem_stride = 1;
};
void SegNorm2Shrink(const DVec& x, const DVec& v, double lam, const Segs& S);
void SegMaxProx(const DVec& x, const DVec& v, double lam, const Segs& S);          // prox/max.cc:7-43
void SegMaxEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s, const Segs& S);
void SegSumLargestProx(const DVec& x, const DVec& v, double lam, int k, const Segs& S);
void SegSumLargestEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s, int k,
                           const Segs& S);
// scaled-zone epigraph per segment; alpha/beta vectors (if given) are indexed by the position
// within the segment (scaled_zone.cc:34-44 sizes them with the slice length)
void SegZoneEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s, double alpha,
                     double beta, const DVec* alpha_vec, const DVec* beta_vec, double M,
                     const Segs& S);
// one second-order cone per segment: (x_s, t_s) = proj{||x|| <= beta t} (second_order_cone.cc:58-79)
void SegSocProject(const DVec& x, const DVec& t, const DVec& v, const DVec& tin, double beta,
                   const Segs& S);
void SegLogSumExpProx(const DVec& x, const DVec& v, double lam, const Segs& S);
void SegLogSumExpEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s,
                          const Segs& S);
enum SmoothFn { SMOOTH_EXP, SMOOTH_LOGISTIC, SMOOTH_NEG_ENTR, SMOOTH_INV_POS, SMOOTH_NEG_LOG };
// elementwise argmin lam f(x) + 1/2 (x - v)^2 (prox/newton.cc:49-112, sum_neg_log.cc:9-24)
void SmoothProx(SmoothFn fn, const DVec& x, const DVec& v, double lam, const DVec* lam_vec);
void SegSmoothEpigraph(SmoothFn fn, const DVec& x, const DVec& t, const DVec& v, const DVec& s,
                       const Segs& S);
void KlDivProx(const DVec& x, const DVec& y, const DVec& u, const DVec& v, double lam,
               const DVec* lam_vec);
void SegKlDivEpigraph(const DVec& x, const DVec& y, const DVec& t, const DVec& u, const DVec& v,
                      const DVec& s, const Segs& S);
void ExpEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s);  // prox/exp.cc

// reference prox/total_variation_1d.cc:21 (glmgen tf_dp): exact 1-D TV prox
void Tv1d(const DVec& x, const DVec& v, double lam);
int Tv1dLastLevels();  // depth of the level-set recursion of the last Tv1d call
int Tv1dBinary(const DVec& x, const DVec& v, double lam);  // round-2 form (kernels_tv.hip), returns the depth
// the same by Johnson's sequential DP on one lane (cross-check of the parallel kernel)
void Tv1dSerial(const DVec& x, const DVec& v, double lam);

}  // namespace k
}  // namespace eps

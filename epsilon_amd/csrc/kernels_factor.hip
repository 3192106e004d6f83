// K5: explicit inverse of a symmetric positive-definite matrix, entirely on the device.
//
// The reference inverts the Schur complement of the least-squares prox with Eigen::LDLT +
// solve(Identity) on one CPU thread (reference src/epsilon/linear/dense_matrix_impl.cc:21-30)
// and then applies the explicit inverse by dgemv every iteration.  Same contract here (the
// cached operator is an explicit inverse, because a GEMV is the GPU-friendly apply), built as
//
//   1. blocked right-looking Cholesky  W = L L^T        (64-wide panels; the panel solve and the
//      trailing update are GEMMs on the MFMA kernel, the 64x64 diagonal block is factored
//      and inverted inside one workgroup's LDS),
//   2. X = L^-1 by recursive doubling: inv([L11 0; L21 L22]) = [X11 0; -X22 L21 X11, X22],
//      bottom-up from the 64x64 diagonal inverses -- every step is two GEMMs,
//   3. W^-1 = X^T X  (lower tiles, then mirrored).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int NB = 64;

// Broadcast of lane `L`'s value (L is a compile-time constant after unrolling): v_readlane_b32,
// one scalar-unit instruction, no LDS.
template <int L> __device__ inline float ReadLane(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), L));
}
template <int L> __device__ inline double ReadLane(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane(static_cast<int>(b), L);
  const int hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), L);
  return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) |
                                        static_cast<unsigned int>(lo));
}

// Compile-time loops so that every register index and every lane index is a constant.
template <int I, int N> struct StaticFor {
  template <class F> __device__ static inline void Run(F&& f) {
    f(std::integral_constant<int, I>());
    StaticFor<I + 1, N>::Run(f);
  }
};
template <int N> struct StaticFor<N, N> {
  template <class F> __device__ static inline void Run(F&&) {}
};

// Factor the kb x kb block at W (ld) in place (lower Cholesky) and write inv(L) (dense NB x NB,
// zeros above the diagonal, ld = NB) to Dinv.  ONE WAVE, register resident: lane i owns row i
// of the block (64 registers); the pivot column is broadcast lane by lane with v_readlane, so
// the 64 elimination steps need no LDS and no barriers (the previous LDS version spent ~190 us
// per block in ~400 barriers; this one ~4000 readlane + FMA pairs).  *flag != 0 on a bad pivot.
template <class T>
__global__ __launch_bounds__(64) void PotrfDiagKernel(T* W, int64_t ld, int kb, T* Dinv,
                                                      int* flag) {
  const int lane = threadIdx.x;
  T r[NB];  // row `lane` of the block; identity padding beyond kb
  StaticFor<0, NB>::Run([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    T v = (lane == c) ? T(1) : T(0);
    if (lane < kb && c < kb && lane >= c) v = W[lane + static_cast<int64_t>(c) * ld];
    r[c] = v;
  });
  bool bad = false;
  // right-looking Cholesky: after step j, r[j] holds L[lane][j]
  StaticFor<0, NB>::Run([&](auto jj) {
    constexpr int j = decltype(jj)::value;
    T d = ReadLane<j>(r[j]);
    if (!(d > T(0))) {
      bad = true;
      d = T(1);
    }
    const T dj = sqrt(d);
    const T l = lane > j ? r[j] / dj : (lane == j ? dj : T(0));
    r[j] = l;
    StaticFor<j + 1, NB>::Run([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      r[c] -= l * ReadLane<c>(l);  // only lanes >= c hold live entries of column c
    });
  });
  if (bad && lane == 0) *flag = 1;
  StaticFor<0, NB>::Run([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (lane < kb && c < kb && lane >= c) W[lane + static_cast<int64_t>(c) * ld] = r[c];
  });
  // X = inv(L), columns last to first (trti2 order):
  //   X[j][j] = 1/L[j][j] ;  X[j+1:, j] = -X[j+1:, j+1:] * L[j+1:, j] * X[j][j]
  // lane t accumulates row t of the product; L[kk][j] is lane kk's r[j].
  T x[NB];
  StaticFor<0, NB>::Run([&](auto ii) {
    constexpr int j = NB - 1 - decltype(ii)::value;
    const T ajj = T(1) / ReadLane<j>(r[j]);
    T acc = T(0);
    StaticFor<j + 1, NB>::Run([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      // x[kk] of lanes < kk is zero (strictly upper part), so no predicate is needed
      acc += x[kk] * ReadLane<kk>(r[j]);
    });
    x[j] = lane > j ? -acc * ajj : (lane == j ? ajj : T(0));
  });
  StaticFor<0, NB>::Run([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    Dinv[lane + c * NB] = (lane < kb && c < kb && lane >= c) ? x[c] : T(0);
  });
}

DVec Sub(const DVec& W, int64_t i, int64_t j, int64_t ld) {
  const int64_t off = i + j * ld;
  return W.Slice(off, W.n - off);
}

}  // namespace

namespace {
// Blocked Cholesky of W (n x n, ld n) in place (lower factor; the strictly upper part of the
// diagonal blocks is scratch) and the inverses of the 64 x 64 diagonal blocks of L in dinv.
void PotrfBlocked(const DVec& W, int64_t n, const DVec& dinv, int* flag);
void CheckFlag(int* flag);
// X (n x n, zero on entry) receives the inverses of the diagonal blocks of size `cap` of the
// factor L held in W (cap = n: all of inv(L)), grown from the 64 x 64 inverses in dinv by
// inv([L11 0; L21 L22]) = [X11 0; -X22 L21 X11, X22].
void DoublingInverse(const DVec& W, const DVec& X, int64_t n, const DVec& dinv, int64_t cap);
}  // namespace

void SpdInverseInPlace(const DVec& W, int64_t n) {
  EPS_CHECK(W.n >= n * n);
  if (n == 0) return;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("spd_inverse", n);
  const DType dt = W.dt;
  const int64_t ld = n;
  const int64_t nb = (n + NB - 1) / NB;

  DVec dinv = DVec::Empty(nb * NB * NB, dt);
  auto flag_buf = rt.Alloc(sizeof(int));
  int* flag = static_cast<int*>(flag_buf->p);
  EPS_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
  PotrfBlocked(W, n, dinv, flag);

  // ---- 2. X = inv(L) by recursive doubling ---------------------------------------------------
  DVec X = DVec::Zeros(n * n, dt);
  DoublingInverse(W, X, n, dinv, n);

  // ---- 3. W^-1 = X^T X ------------------------------------------------------------------------
  // Row block p of the lower-triangular X is zero right of column (p+1)*B, so it only touches
  // the leading (p+1)B x (p+1)B corner of X^T X: a third of the flops of the dense product.
  {
    const int64_t B = 1024;
    Fill(W.Slice(0, n * n), 0.0);
    for (int64_t p0 = 0; p0 < n; p0 += B) {
      const int64_t pb = std::min<int64_t>(B, n - p0);
      const int64_t cp = p0 + pb;  // columns with non-zeros in these rows
      DVec Xp = Sub(X, p0, 0, ld);
      Gemm(true, false, cp, cp, pb, 1.0, Xp, ld, Xp, ld, 1.0, W, ld, true);
    }
  }
  SymmetrizeFromLower(W, n, ld);

  CheckFlag(flag);
}

namespace {

void CheckFlag(int* flag) {
  hipStream_t s = Runtime::Get().stream();
  int host_flag = 0;
  EPS_HIP(hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, s));
  EPS_HIP(hipStreamSynchronize(s));
  EPS_CHECK_MSG(host_flag == 0, "dense inverse: matrix is not positive definite");
}

void DoublingInverse(const DVec& W, const DVec& X, int64_t n, const DVec& dinv, int64_t cap) {
  hipStream_t s = Runtime::Get().stream();
  const DType dt = W.dt;
  const int64_t ld = n;
  const int64_t nb = (n + NB - 1) / NB;
  for (int64_t kblk = 0; kblk < nb; ++kblk) {
    const int64_t k0 = kblk * NB;
    const int64_t kb = std::min<int64_t>(NB, n - k0);
    // copy the kb x kb inverse block (ld NB) into X's diagonal block (ld n)
    EPS_HIP(hipMemcpy2DAsync(Sub(X, k0, k0, ld).data(), ld * DTypeSize(dt),
                             dinv.Slice(kblk * NB * NB, NB * NB).data(), NB * DTypeSize(dt),
                             kb * DTypeSize(dt), kb, hipMemcpyDeviceToDevice, s));
  }
  const int64_t half = std::min<int64_t>(cap, n) / 2 + NB;
  DVec tmp2 = DVec::Empty(std::max<int64_t>(1, half * half), dt);
  for (int64_t sz = NB; sz < cap; sz *= 2) {
    for (int64_t r0 = 0; r0 + sz < n; r0 += 2 * sz) {
      const int64_t s1 = sz;                                  // rows/cols of block 1
      const int64_t s2 = std::min<int64_t>(sz, n - (r0 + sz));  // rows of block 2
      DVec L21 = Sub(W, r0 + s1, r0, ld);
      DVec X11 = Sub(X, r0, r0, ld);
      DVec X22 = Sub(X, r0 + s1, r0 + s1, ld);
      DVec X21 = Sub(X, r0 + s1, r0, ld);
      EPS_CHECK(tmp2.n >= s2 * s1);
      DVec T = tmp2.Slice(0, s2 * s1);
      // X11 and X22 are lower triangular: column block j of X11 is zero above row j*cb and row
      // block i of X22 is zero right of column (i+1)*rb, so the products only run over the
      // non-zero part of K (62 % of the dense flops with four blocks).
      const int64_t nsplit = sz >= 1024 ? 4 : 1;
      const int64_t cb = (s1 + nsplit - 1) / nsplit;
      for (int64_t c0 = 0; c0 < s1; c0 += cb) {
        const int64_t cw = std::min<int64_t>(cb, s1 - c0);
        Gemm(false, false, s2, cw, s1 - c0, 1.0, Sub(L21, 0, c0, ld), ld, Sub(X11, c0, c0, ld), ld,
             0.0, T.Slice(c0 * s2, cw * s2), s2);
      }
      const int64_t rb = (s2 + nsplit - 1) / nsplit;
      for (int64_t r1 = 0; r1 < s2; r1 += rb) {
        const int64_t rw = std::min<int64_t>(rb, s2 - r1);
        Gemm(false, false, rw, s1, r1 + rw, -1.0, Sub(X22, r1, 0, ld), ld, T, s2, 0.0,
             Sub(X21, r1, 0, ld), ld);
      }
    }
  }

}

void PotrfBlocked(const DVec& W, int64_t n, const DVec& dinv, int* flag) {
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  const DType dt = W.dt;
  const int64_t ld = n;
  DVec panel = DVec::Empty(std::max<int64_t>(n, 1) * NB, dt);
  // ---- 1. blocked Cholesky ------------------------------------------------------------------
  // Two-level blocking: 64-wide steps update only the rest of their 256-wide outer panel; the
  // trailing matrix sees one rank-256 update per outer panel (a rank-64 update of the whole
  // trailing matrix is HBM-bound: it re-reads and re-writes up to n^2 entries for 64 columns).
  const int64_t OB = 4 * NB;
  for (int64_t K0 = 0; K0 < n; K0 += OB) {
    const int64_t KB = std::min<int64_t>(OB, n - K0);
    for (int64_t k0 = K0; k0 < K0 + KB; k0 += NB) {
      const int64_t kblk = k0 / NB;
      const int kb = static_cast<int>(std::min<int64_t>(NB, n - k0));
      DVec Wkk = Sub(W, k0, k0, ld);
      DVec Dk = dinv.Slice(kblk * NB * NB, NB * NB);
      if (dt == F32) {
        hipLaunchKernelGGL(PotrfDiagKernel<float>, dim3(1), dim3(64), 0, s, Wkk.as<float>(), ld,
                           kb, Dk.as<float>(), flag);
      } else {
        hipLaunchKernelGGL(PotrfDiagKernel<double>, dim3(1), dim3(64), 0, s, Wkk.as<double>(),
                           ld, kb, Dk.as<double>(), flag);
      }
      const int64_t rem = n - (k0 + kb);
      if (rem <= 0) continue;
      DVec W21 = Sub(W, k0 + kb, k0, ld);
      DVec tmp = panel.Slice(0, rem * kb);
      MatCopy(false, rem, kb, 1.0, W21, ld, tmp);
      // L21 = W21 * inv(L11)^T
      Gemm(false, true, rem, kb, kb, 1.0, tmp, rem, Dk, NB, 0.0, W21, ld);
      // the remaining columns of this outer panel: W[k0+kb:, k0+kb : K0+KB] -= L21 L21[0:pc]^T
      const int64_t pc = K0 + KB - (k0 + kb);
      if (pc > 0) {
        DVec Wp = Sub(W, k0 + kb, k0 + kb, ld);
        Gemm(false, true, rem, pc, kb, -1.0, W21, ld, W21, ld, 1.0, Wp, ld);
      }
    }
    // trailing matrix: W22 -= L21 L21^T with the whole outer panel (lower tiles only)
    const int64_t rem2 = n - (K0 + KB);
    if (rem2 > 0) {
      DVec L21 = Sub(W, K0 + KB, K0, ld);
      DVec W22 = Sub(W, K0 + KB, K0 + KB, ld);
      Gemm(false, true, rem2, rem2, KB, -1.0, L21, ld, L21, ld, 1.0, W22, ld, true);
    }
  }

}

}  // namespace

// Columns [lo, lo + cnt) of W^-1 (n x cnt, ld n) without forming the rest: Cholesky, then the two
// triangular solves L Y = E, L^T Z = Y on the cnt unit columns, 64 rows at a time with the
// inverted diagonal blocks (every step is two small GEMMs).  W is overwritten by its factor.
// Used when the inverse of a replicated matrix is split over the ranks of a sharded solve: each
// rank solves for its own slab of columns and the slabs are all-gathered.
void SpdInverseColumns(const DVec& W, int64_t n, int64_t lo, int64_t cnt, const DVec& Out) {
  EPS_CHECK(W.n >= n * n && Out.n >= n * cnt && Out.dt == W.dt && lo >= 0 && lo + cnt <= n);
  if (n == 0) return;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("spd_inverse_columns", n, cnt);
  const DType dt = W.dt;
  const int64_t ld = n;
  const int64_t nb = (n + NB - 1) / NB;
  DVec dinv = DVec::Empty(nb * NB * NB, dt);
  auto flag_buf = rt.Alloc(sizeof(int));
  int* flag = static_cast<int*>(flag_buf->p);
  EPS_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
  PotrfBlocked(W, n, dinv, flag);
  if (cnt == 0) {
    CheckFlag(flag);
    return;
  }
  // B = E[:, lo : lo + cnt]
  Fill(Out.Slice(0, n * cnt), 0.0);
  AddDiag(Sub(Out, lo, 0, ld), cnt, ld, 1.0, nullptr);
  // inverses of the BS x BS diagonal blocks of L (a few doubling levels), then BS rows per step
  const int64_t BS = 512;
  DVec X = DVec::Zeros(n * n, dt);
  DoublingInverse(W, X, n, dinv, BS);
  DVec tmp = DVec::Empty(BS * cnt, dt);
  const size_t es = DTypeSize(dt);
  const int64_t nbs = (n + BS - 1) / BS;
  auto store_rows = [&](int64_t k0, int64_t kb) {  // Out[k0 : k0 + kb, :] = tmp (kb x cnt)
    EPS_HIP(hipMemcpy2DAsync(Sub(Out, k0, 0, ld).data(), ld * es, tmp.data(), kb * es, kb * es, cnt,
                             hipMemcpyDeviceToDevice, s));
  };
  // forward L Y = E: rows above lo stay zero
  for (int64_t kblk = lo / BS; kblk < nbs; ++kblk) {
    const int64_t k0 = kblk * BS;
    const int64_t kb = std::min<int64_t>(BS, n - k0);
    Gemm(false, false, kb, cnt, kb, 1.0, Sub(X, k0, k0, ld), ld, Sub(Out, k0, 0, ld), ld, 0.0, tmp,
         kb);
    store_rows(k0, kb);  // Y_k = inv(L_kk) B_k
    const int64_t rem = n - (k0 + kb);
    if (rem > 0)  // B[k+1:, :] -= L[k+1:, k] Y_k
      Gemm(false, false, rem, cnt, kb, -1.0, Sub(W, k0 + kb, k0, ld), ld, tmp, kb, 1.0,
           Sub(Out, k0 + kb, 0, ld), ld);
  }
  // backward L^T Z = Y
  for (int64_t kblk = nbs - 1; kblk >= 0; --kblk) {
    const int64_t k0 = kblk * BS;
    const int64_t kb = std::min<int64_t>(BS, n - k0);
    Gemm(true, false, kb, cnt, kb, 1.0, Sub(X, k0, k0, ld), ld, Sub(Out, k0, 0, ld), ld, 0.0, tmp,
         kb);
    store_rows(k0, kb);  // Z_k = inv(L_kk)^T Y_k
    if (k0 > 0)  // Y[0:k, :] -= L[k, 0:k]^T Z_k
      Gemm(true, false, k0, cnt, kb, -1.0, Sub(W, k0, 0, ld), ld, tmp, kb, 1.0, Out, ld);
  }
  CheckFlag(flag);
}

}  // namespace k
}  // namespace eps

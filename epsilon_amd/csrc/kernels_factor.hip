// K5: explicit inverse of a symmetric positive-definite matrix, entirely on the device.
//
// The reference inverts the Schur complement of the least-squares prox with Eigen::LDLT +
// solve(Identity) on one CPU thread (reference src/epsilon/linear/dense_matrix_impl.cc:21-30)
// and then applies the explicit inverse by dgemv every iteration.  Same contract here (the
// cached operator is an explicit inverse, because a GEMV is the GPU-friendly apply), built as
//
//   1. blocked Cholesky  W = L L^T: 256-wide outer panels, right-looking between panels (one
//      rank-256 update of the trailing matrix per panel on the MFMA kernel) and left-looking in
//      64-column steps inside a panel - per step one workgroup updates and factors the 64 x 64
//      diagonal block (register resident, one wave) and one launch updates the rows below and
//      solves them against the new block by forward substitution,
//   2. X = L^-1 by recursive doubling: inv([L11 0; L21 L22]) = [X11 0; -X22 L21 X11, X22],
//      bottom-up from the 64x64 diagonal inverses (one batched launch) -- every level is two
//      GEMMs, batched over the pairs of the level below 4096 rows,
//   3. W^-1 = X^T X  (lower tiles, then mirrored).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
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

constexpr int KC = 32;  // columns of the outer panel consumed per LDS stage

// Forces `v` to be computed at this point of the program (an empty instruction that "modifies" it).
template <class T> __device__ inline void Pin(T& v) { asm volatile("" : "+v"(v)); }

// Row `lane` of a 64 x 64 lower-triangular block in registers -> its Cholesky factor in place.
// ONE WAVE, register resident: the pivot column is broadcast lane by lane with v_readlane, so the
// 64 elimination steps need no LDS and no barriers (an LDS version spent ~190 us per block in
// ~400 barriers).  Returns true on a non-positive pivot.
template <class T> __device__ __forceinline__ bool CholRows(T (&r)[NB], int lane) {
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
    // four broadcasts, then their four updates: a readlane result needs two idle cycles before
    // a vector instruction may use it, which the grouping fills with the other readlanes
    StaticFor<0, (NB - 1 - j + 3) / 4>::Run([&](auto gg) {
      constexpr int c0 = j + 1 + 4 * decltype(gg)::value;
      T b[4];
      StaticFor<0, 4>::Run([&](auto ii) {
        constexpr int c = c0 + decltype(ii)::value;
        if constexpr (c < NB) b[decltype(ii)::value] = ReadLane<c>(l);
      });
      StaticFor<0, 4>::Run([&](auto ii) {
        constexpr int c = c0 + decltype(ii)::value;
        // only lanes >= c hold live entries of column c
        if constexpr (c < NB) r[c] -= l * b[decltype(ii)::value];
      });
      // Pin the updates here.  The whole factorisation is one basic block, and left alone the
      // compiler defers every update of column c to step c: it parks the ~2000 broadcast values
      // in spare VGPR lanes (v_writelane) and fetches them back later, twice the lane traffic.
      StaticFor<0, 4>::Run([&](auto ii) {
        constexpr int c = c0 + decltype(ii)::value;
        if constexpr (c < NB) Pin(r[c]);
      });
    });
  });
  return bad;
}

// The same elimination carried along a second row per lane: t (row `lane` of a 64-column block T
// of the rows below) becomes row `lane` of T L^-T.  The forward substitution x_j = t_j / L_jj,
// t_c -= x_j L_cj (c > j) needs exactly the scalars the factorisation broadcasts at step j, so it
// rides on them: 2016 extra multiply-adds per lane, no second pass over L through LDS.
template <class T> __device__ __forceinline__ bool CholRowsSolve(T (&r)[NB], T (&t)[NB], int lane) {
  bool bad = false;
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
    const T x = t[j] / dj;
    t[j] = x;
    StaticFor<0, (NB - 1 - j + 3) / 4>::Run([&](auto gg) {
      constexpr int c0 = j + 1 + 4 * decltype(gg)::value;
      T b[4];
      StaticFor<0, 4>::Run([&](auto ii) {
        constexpr int c = c0 + decltype(ii)::value;
        if constexpr (c < NB) b[decltype(ii)::value] = ReadLane<c>(l);
      });
      StaticFor<0, 4>::Run([&](auto ii) {
        constexpr int c = c0 + decltype(ii)::value;
        if constexpr (c < NB) {
          r[c] -= l * b[decltype(ii)::value];
          t[c] -= x * b[decltype(ii)::value];
        }
      });
      StaticFor<0, 4>::Run([&](auto ii) {  // (see CholRows)
        constexpr int c = c0 + decltype(ii)::value;
        if constexpr (c < NB) {
          Pin(r[c]);
          Pin(t[c]);
        }
      });
    });
  });
  return bad;
}

// One 64-column step of the left-looking factorisation inside an outer panel [K0, K0 + 256):
// the diagonal block D = W[k0:k0+kb, k0:k0+kb] first receives the update of the panel columns
// already factored, D -= P P^T with P = W[k0:k0+kb, K0:k0], then wave 0 factors it in registers.
// One workgroup; the rows below are PotrfPanelStepKernel's.
template <class T>
__global__ __launch_bounds__(256) void PotrfDiagStepKernel(T* W, int64_t ld, int64_t K0, int64_t k0,
                                                           int kb, int* flag) {
  __shared__ T D[NB][NB];   // [column][row]
  __shared__ T Pc[KC][NB];  // [k][row]
  const int t = threadIdx.x, r = t & 63, q = t >> 6;
  T acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int col = q * 16 + c;
    T v = (r == col) ? T(1) : T(0);
    if (r < kb && col < kb && r >= col) v = W[(k0 + r) + (k0 + col) * ld];
    acc[c] = v;
  }
  // the next stage's global loads are issued before this stage's arithmetic (the loop is
  // otherwise a chain of exposed memory latencies: one workgroup, nothing else to switch to)
  constexpr int PF = KC * NB / 256;
  T pf[PF];
  auto fetch = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = t + i * 256, rr = idx & 63, k = idx >> 6;
      pf[i] = rr < kb ? W[(k0 + rr) + (kk + k) * ld] : T(0);
    }
  };
  if (K0 < k0) fetch(K0);
  for (int64_t kk = K0; kk < k0; kk += KC) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = t + i * 256;
      Pc[idx >> 6][idx & 63] = pf[i];
    }
    __syncthreads();
    if (kk + KC < k0) fetch(kk + KC);
#pragma unroll 4
    for (int k = 0; k < KC; ++k) {
      const T a = Pc[k][r];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] -= a * Pc[k][q * 16 + c];
    }
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) D[q * 16 + c][r] = acc[c];
  __syncthreads();
  if (q != 0) return;
  T row[NB];
  StaticFor<0, NB>::Run([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    row[c] = r >= c ? D[c][r] : T(0);
  });
  const bool bad = CholRows(row, r);
  if (bad && r == 0) *flag = 1;
  StaticFor<0, NB>::Run([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (r < kb && c < kb && r >= c) W[(k0 + r) + (k0 + c) * ld] = row[c];
  });
}

// The rows below the diagonal block in the same step: T = W[rows, k0:k0+kb] - L[rows, K0:k0] P^T,
// then L[rows, k0:k0+kb] = T L11^-T by forward substitution (thread per row, L11 broadcast from
// LDS).  64 rows per workgroup: four waves split the columns of the update, wave 0 substitutes.
template <class T>
__global__ __launch_bounds__(256) void PotrfPanelStepKernel(T* W, int64_t ld, int64_t n, int64_t K0,
                                                            int64_t k0, int kb) {
  __shared__ T Tt[NB][NB];          // [column][row]
  __shared__ T buf[2 * KC * NB];    // stage: Lr[k][row] | Pc[k][col]; afterwards L11 as [j][k]
  T* Lr = buf;
  T* Pc = buf + KC * NB;
  const int t = threadIdx.x, r = t & 63, q = t >> 6;
  const int64_t row0 = k0 + kb + static_cast<int64_t>(blockIdx.x) * NB;
  const int64_t row = row0 + r;
  const bool live = row < n;
  T acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int col = q * 16 + c;
    acc[c] = (live && col < kb) ? W[row + (k0 + col) * ld] : T(0);
  }
  constexpr int PF = KC * NB / 256;
  T pfl[PF], pfp[PF];  // next stage, in flight during this stage's arithmetic
  auto fetch = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = t + i * 256, rr = idx & 63, k = idx >> 6;
      pfl[i] = (row0 + rr < n) ? W[(row0 + rr) + (kk + k) * ld] : T(0);
      pfp[i] = rr < kb ? W[(k0 + rr) + (kk + k) * ld] : T(0);
    }
  };
  if (K0 < k0) fetch(K0);
  for (int64_t kk = K0; kk < k0; kk += KC) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = t + i * 256;
      Lr[idx] = pfl[i];
      Pc[idx] = pfp[i];
    }
    __syncthreads();
    if (kk + KC < k0) fetch(kk + KC);
#pragma unroll 4
    for (int k = 0; k < KC; ++k) {
      const T a = Lr[k * NB + r];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] -= a * Pc[k * NB + q * 16 + c];
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 16; ++c) Tt[q * 16 + c][r] = acc[c];
  // L11 (identity padding beyond kb), stored [j][k] so that the k run of one j is contiguous
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int j = idx & 63, k = idx >> 6;
    T v = (j == k) ? T(1) : T(0);
    if (j < kb && k < kb && j >= k) v = W[(k0 + j) + (k0 + k) * ld];
    buf[j * NB + k] = v;
  }
  __syncthreads();
  if (q == 0) {
    // forward substitution in 8 x 8 blocks, x written back over T: the outer loops stay rolled
    // (fully unrolled, the compiler hoists all 2016 broadcast reads and spills them)
#pragma unroll 1
    for (int jb = 0; jb < NB; jb += 8) {
      T tv[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) tv[jj] = Tt[jb + jj][r];
#pragma unroll 1
      for (int k8 = 0; k8 < jb; k8 += 8) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const T xk = Tt[k8 + kk][r];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) tv[jj] -= xk * buf[(jb + jj) * NB + k8 + kk];
        }
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
        for (int kk = 0; kk < jj; ++kk) tv[jj] -= tv[kk] * buf[(jb + jj) * NB + jb + kk];
        tv[jj] /= buf[(jb + jj) * NB + jb + jj];
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) Tt[jb + jj][r] = tv[jj];
    }
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int col = q * 16 + c;
      if (col < kb) W[row + (k0 + col) * ld] = Tt[col][r];
    }
  }
}

// fp32: the two launches of a step fused.  Every workgroup re-derives the factor of the diagonal
// block itself (the same instructions on the same data: identical bits) next to its own rows'
// update - the panel columns it stages serve both - so the 157 steps of a 10^4 matrix are one
// launch each and the single-workgroup diagonal kernel leaves the critical path.  Workgroup 0
// writes the factor of the diagonal block to a side buffer (nothing later in the factorisation
// reads a diagonal block; writing it into W here would race with the workgroups of the same
// launch that have yet to read the unfactored block).  (fp64 keeps the two kernels: the third
// 64 x 64 tile does not fit the 64 KB of static LDS.)
__global__ __launch_bounds__(256) void PotrfFusedStepKernel(float* W, int64_t ld, int64_t n, int64_t K0,
                                                            int64_t k0, int kb, int* flag,
                                                            float* __restrict__ dfac) {
  using T = float;
  __shared__ T Tt[NB][NB];        // [column][row], this workgroup's rows
  __shared__ T Dd[NB][NB];        // [column][row], the diagonal block
  __shared__ T buf[2 * KC * NB];  // stage: Lr[k][row] | Pc[k][col]
  T* Lr = buf;
  T* Pc = buf + KC * NB;
  const int t = threadIdx.x, r = t & 63, q = t >> 6;
  const int64_t row0 = k0 + kb + static_cast<int64_t>(blockIdx.x) * NB;
  const int64_t row = row0 + r;
  const bool live = row < n;
  T acc[16], dac[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int col = q * 16 + c;
    acc[c] = (live && col < kb) ? W[row + (k0 + col) * ld] : T(0);
    T v = (r == col) ? T(1) : T(0);
    if (r < kb && col < kb && r >= col) v = W[(k0 + r) + (k0 + col) * ld];
    dac[c] = v;
  }
  constexpr int PF = KC * NB / 256;
  T pfl[PF], pfp[PF];
  auto fetch = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = t + i * 256, rr = idx & 63, k = idx >> 6;
      pfl[i] = (row0 + rr < n) ? W[(row0 + rr) + (kk + k) * ld] : T(0);
      pfp[i] = rr < kb ? W[(k0 + rr) + (kk + k) * ld] : T(0);
    }
  };
  if (K0 < k0) fetch(K0);
  for (int64_t kk = K0; kk < k0; kk += KC) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = t + i * 256;
      Lr[idx] = pfl[i];
      Pc[idx] = pfp[i];
    }
    __syncthreads();
    if (kk + KC < k0) fetch(kk + KC);
#pragma unroll 4
    for (int k = 0; k < KC; ++k) {
      const T a = Lr[k * NB + r], d = Pc[k * NB + r];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const T pc = Pc[k * NB + q * 16 + c];
        acc[c] -= a * pc;
        dac[c] -= d * pc;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    Tt[q * 16 + c][r] = acc[c];
    Dd[q * 16 + c][r] = dac[c];
  }
  __syncthreads();
  if (q == 0) {
    // wave 0: lane r holds row r of the diagonal block AND row r of this workgroup's rows; one
    // elimination factors the first and forward-substitutes the second (CholRowsSolve)
    T rowv[NB], tv[NB];
    StaticFor<0, NB>::Run([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      rowv[c] = r >= c ? Dd[c][r] : T(0);
      tv[c] = Tt[c][r];
    });
    const bool bad = CholRowsSolve(rowv, tv, r);
    if (bad && r == 0) *flag = 1;
    StaticFor<0, NB>::Run([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      Tt[c][r] = tv[c];
      // NOT into W: the other workgroups of this launch read the unfactored block from there,
      // whenever they happen to start (PotrfBlocked scatters the side buffer at the end);
      // zero above the diagonal by construction, identity rows beyond kb by the padding of D
      if (blockIdx.x == 0) dfac[c * NB + r] = rowv[c];
    });
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int col = q * 16 + c;
      if (col < kb) W[row + (k0 + col) * ld] = Tt[col][r];
    }
  }
}

// inv(L11) of every 64 x 64 diagonal block of the factor (one wave per block): Dinv block b is
// dense NB x NB (ld NB), zeros above the diagonal and beyond the matrix.
template <class T>
__global__ __launch_bounds__(64) void TrtriDiagBlocksKernel(const T* W, int64_t ld, int64_t n, T* Dinv) {
  const int lane = threadIdx.x;
  const int64_t k0 = static_cast<int64_t>(blockIdx.x) * NB;
  const int kb = static_cast<int>(n - k0 < NB ? n - k0 : NB);
  T r[NB];
  StaticFor<0, NB>::Run([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    T v = (lane == c) ? T(1) : T(0);
    if (lane < kb && c < kb && lane >= c) v = W[(k0 + lane) + (k0 + c) * ld];
    r[c] = v;
  });
  // X = inv(L), columns last to first (trti2 order):
  //   X[j][j] = 1/L[j][j] ;  X[j+1:, j] = -X[j+1:, j+1:] * L[j+1:, j] * X[j][j]
  // lane t accumulates row t of the product; L[kk][j] is lane kk's r[j].
  T x[NB];
  StaticFor<0, NB>::Run([&](auto ii) {
    constexpr int j = NB - 1 - decltype(ii)::value;
    const T ajj = T(1) / ReadLane<j>(r[j]);
    T acc = T(0);
    // x[kk] of lanes < kk is zero (strictly upper part), so no predicate is needed; broadcasts
    // in groups of four (see CholRows)
    StaticFor<0, (NB - 1 - j + 3) / 4>::Run([&](auto gg) {
      constexpr int c0 = j + 1 + 4 * decltype(gg)::value;
      T b[4];
      StaticFor<0, 4>::Run([&](auto ii) {
        constexpr int kk = c0 + decltype(ii)::value;
        if constexpr (kk < NB) b[decltype(ii)::value] = ReadLane<kk>(r[j]);
      });
      StaticFor<0, 4>::Run([&](auto ii) {
        constexpr int kk = c0 + decltype(ii)::value;
        if constexpr (kk < NB) acc += x[kk] * b[decltype(ii)::value];
      });
      Pin(acc);
    });
    x[j] = lane > j ? -acc * ajj : (lane == j ? ajj : T(0));
  });
  T* out = Dinv + static_cast<int64_t>(blockIdx.x) * NB * NB;
  StaticFor<0, NB>::Run([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    out[lane + c * NB] = (lane < kb && c < kb && lane >= c) ? x[c] : T(0);
  });
}

template <class T>
__global__ __launch_bounds__(256) void ScatterDiagBlocksKernel(const T* Dinv, T* X, int64_t ld, int64_t n) {
  const int64_t k0 = static_cast<int64_t>(blockIdx.x) * NB;
  const T* src = Dinv + static_cast<int64_t>(blockIdx.x) * NB * NB;
  for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
    const int r = idx & 63, c = idx >> 6;
    if (k0 + r < n && k0 + c < n) X[(k0 + r) + (k0 + c) * ld] = src[idx];
  }
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
  {
    ProfScope p1("potrf", n);
    PotrfBlocked(W, n, dinv, flag);
  }

  // ---- 2. X = inv(L) by recursive doubling ---------------------------------------------------
  DVec X = DVec::Zeros(n * n, dt);
  {
    ProfScope p2("trtri", n);
    DoublingInverse(W, X, n, dinv, n);
  }
  ProfScope p3("lauum", n);

  // ---- 3. W^-1 = X^T X ------------------------------------------------------------------------
  // Row block p of the lower-triangular X is zero right of column (p+1)*B, so it only touches
  // the leading (p+1)B x (p+1)B corner of X^T X: a third of the flops of the dense product.
  if (!SyrkSplitF16LowerTriangular(n, X, ld, W, ld)) {
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
  // the 64 x 64 inverse blocks (ld NB) go onto X's diagonal (ld n), all in one launch
  if (dt == F32)
    hipLaunchKernelGGL(ScatterDiagBlocksKernel<float>, dim3(static_cast<unsigned>(nb)), dim3(256), 0, s,
                       dinv.as<float>(), X.as<float>(), ld, n);
  else
    hipLaunchKernelGGL(ScatterDiagBlocksKernel<double>, dim3(static_cast<unsigned>(nb)), dim3(256), 0, s,
                       dinv.as<double>(), X.as<double>(), ld, n);
  const int64_t half = std::min<int64_t>(cap, n) / 2 + NB;
  const int64_t kBatchBelow = 4096;  // levels below this size: all pairs of a level in one launch
  DVec tmp2 = DVec::Empty(std::max<int64_t>({1, half * half, n * std::min(cap, kBatchBelow) / 4}), dt);
  for (int64_t sz = NB; sz < cap; sz *= 2) {
    int64_t r_first = 0;
    if (sz < kBatchBelow) {
      // the pairs of one level are independent and (but for a ragged last one) of equal shape:
      // two batched launches per level instead of two per pair (sz = 64: 156 -> 2)
      const int64_t P = n / (2 * sz);  // full pairs
      if (P > 0) {
        const int64_t stride = 2 * sz * (ld + 1);
        EPS_CHECK(tmp2.n >= P * sz * sz);
        GemmBatched(false, false, sz, sz, sz, 1.0, Sub(W, sz, 0, ld), ld, stride, X, ld, stride, 0.0,
                    tmp2, sz, sz * sz, P);
        GemmBatched(false, false, sz, sz, sz, -1.0, Sub(X, sz, sz, ld), ld, stride, tmp2, sz, sz * sz,
                    0.0, Sub(X, sz, 0, ld), ld, stride, P);
      }
      r_first = P * 2 * sz;
    }
    for (int64_t r0 = r_first; r0 + sz < n; r0 += 2 * sz) {
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
      // ... as long as every piece still fills the chip (>= ~400 tiles of 128 x 128)
      const int64_t tiles = ((s1 + 127) / 128) * ((s2 + 127) / 128);
      const int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>(4, tiles / 400));
      const int64_t cb = (s1 + nsplit - 1) / nsplit;
      // (large f32 levels: one launch each with a k range per tile, kernels_gemm_f16split.hip)
      const bool t_done = GemmSplitF16KRange(3, s2, s1, s1, 1.0, L21, ld, X11, ld, T, s2);
      for (int64_t c0 = 0; c0 < s1 && !t_done; c0 += cb) {
        const int64_t cw = std::min<int64_t>(cb, s1 - c0);
        Gemm(false, false, s2, cw, s1 - c0, 1.0, Sub(L21, 0, c0, ld), ld, Sub(X11, c0, c0, ld), ld,
             0.0, T.Slice(c0 * s2, cw * s2), s2);
      }
      const int64_t rb = (s2 + nsplit - 1) / nsplit;
      const bool x_done = GemmSplitF16KRange(4, s2, s1, s2, -1.0, X22, ld, T, s2, X21, ld);
      for (int64_t r1 = 0; r1 < s2 && !x_done; r1 += rb) {
        const int64_t rw = std::min<int64_t>(rb, s2 - r1);
        Gemm(false, false, rw, s1, r1 + rw, -1.0, Sub(X22, r1, 0, ld), ld, T, s2, 0.0,
             Sub(X21, r1, 0, ld), ld);
      }
    }
  }

}

int g_potrf_form = -1;  // -1: by environment, 0: fused step (f32), 1: diagonal + panel launches

void PotrfBlocked(const DVec& W, int64_t n, const DVec& dinv, int* flag) {
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  const DType dt = W.dt;
  const int64_t ld = n;
  // ---- 1. blocked Cholesky ------------------------------------------------------------------
  // Two-level blocking: 64-wide steps update only the rest of their 256-wide outer panel; the
  // trailing matrix sees one rank-256 update per outer panel (a rank-64 update of the whole
  // trailing matrix is HBM-bound: it re-reads and re-writes up to n^2 entries for 64 columns).
  const int64_t OB = 4 * NB;  // measured with the fused step: 3 ... 8 blocks are within 0.3 ms at n = 10^4
  {
    // Left-looking inside the outer panel: a 64-column step is two launches - the diagonal block
    // (update with the panel columns already done + register-resident factorisation) and the
    // rows below it (same update + forward substitution against the new L11) - instead of the
    // diagonal kernel, a copy and two small GEMMs; the 64 x 64 inverses the later stages want are
    // formed for all blocks at once at the end.
    // (read per call: tests compare the two forms in one process, eps_test_spd_inverse_repeat)
    const bool two_kernels = g_potrf_form == 1 ||
                             (g_potrf_form < 0 && std::getenv("EPSILON_HIP_POTRF_TWO_KERNELS") != nullptr);
    const unsigned nblk = static_cast<unsigned>((n + NB - 1) / NB);
    const bool fused = dt == F32 && !two_kernels;
    std::shared_ptr<Buffer> dfac_buf;
    if (fused) dfac_buf = rt.Alloc(static_cast<size_t>(nblk) * NB * NB * sizeof(float));
    float* dfac = fused ? static_cast<float*>(dfac_buf->p) : nullptr;
    for (int64_t K0 = 0; K0 < n; K0 += OB) {
      const int64_t KB = std::min<int64_t>(OB, n - K0);
      for (int64_t k0 = K0; k0 < K0 + KB; k0 += NB) {
        const int kb = static_cast<int>(std::min<int64_t>(NB, n - k0));
        const int64_t rem = n - (k0 + kb);
        const unsigned blocks = static_cast<unsigned>((rem + NB - 1) / NB);
        if (dt == F32) {
          if (fused) {
            hipLaunchKernelGGL(PotrfFusedStepKernel, dim3(blocks ? blocks : 1), dim3(256), 0, s,
                               W.as<float>(), ld, n, K0, k0, kb, flag, dfac + (k0 / NB) * NB * NB);
          } else {
            hipLaunchKernelGGL(PotrfDiagStepKernel<float>, dim3(1), dim3(256), 0, s, W.as<float>(), ld,
                               K0, k0, kb, flag);
            if (blocks)
              hipLaunchKernelGGL(PotrfPanelStepKernel<float>, dim3(blocks), dim3(256), 0, s,
                                 W.as<float>(), ld, n, K0, k0, kb);
          }
        } else {
          hipLaunchKernelGGL(PotrfDiagStepKernel<double>, dim3(1), dim3(256), 0, s, W.as<double>(),
                             ld, K0, k0, kb, flag);
          if (blocks)
            hipLaunchKernelGGL(PotrfPanelStepKernel<double>, dim3(blocks), dim3(256), 0, s,
                               W.as<double>(), ld, n, K0, k0, kb);
        }
      }
      const int64_t rem2 = n - (K0 + KB);
      if (rem2 > 0) {
        DVec L21 = Sub(W, K0 + KB, K0, ld);
        DVec W22 = Sub(W, K0 + KB, K0 + KB, ld);
        Gemm(false, true, rem2, rem2, KB, -1.0, L21, ld, L21, ld, 1.0, W22, ld, true);
      }
    }
    if (fused)  // the factors of the diagonal blocks, from the side buffer (zeros above the diagonal)
      hipLaunchKernelGGL(ScatterDiagBlocksKernel<float>, dim3(nblk), dim3(256), 0, s, dfac, W.as<float>(), ld, n);
    if (dt == F32)
      hipLaunchKernelGGL(TrtriDiagBlocksKernel<float>, dim3(nblk), dim3(64), 0, s, W.as<float>(), ld, n,
                         dinv.as<float>());
    else
      hipLaunchKernelGGL(TrtriDiagBlocksKernel<double>, dim3(nblk), dim3(64), 0, s, W.as<double>(), ld,
                         n, dinv.as<double>());
  }
}

}  // namespace

void SetPotrfForm(int form) { g_potrf_form = form; }

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

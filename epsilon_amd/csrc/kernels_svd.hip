// K11: singular value decomposition for the orthogonally-invariant prox operators
// (nuclear norm: reference src/epsilon/prox/ortho_invariant.cc:13-73, norm_nuclear.cc:3-14).
//
// The reference forms Y^T Y (+1e-15 I), calls Eigen::SelfAdjointEigenSolver and rebuilds
// U = Y V D^-1 (ortho_invariant.cc:36-50).  Here: one-sided Jacobi (Hestenes) directly on the
// columns of Y - no Gram matrix, no squaring of the condition number.  Columns are paired by a
// round-robin tournament, so the n/2 rotations of a step touch disjoint columns and run as one
// launch (one workgroup per pair: three fp64 wave-shuffle reductions, then the rotation applied
// to the column pair of W and of V).  After convergence W = U Sigma and Y = W V^T.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "comm.h"
#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;

__device__ inline double WaveSumD(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// pair (p, q) number `pair` of round `step` in a round-robin tournament over npad players
__device__ inline void TournamentPair(int64_t npad, int64_t step, int64_t pair, int64_t* p,
                                      int64_t* q) {
  auto player = [&](int64_t pos) -> int64_t {
    if (pos == 0) return 0;
    return 1 + ((pos - 1 + step) % (npad - 1));
  };
  *p = player(pair);
  *q = player(npad - 1 - pair);
}

template <class T>
__global__ __launch_bounds__(kBlock) void JacobiStepKernel(T* W, int64_t m, int64_t n, T* V,
                                                           int64_t npad, int64_t step, double tol,
                                                           int* rotated) {
  __shared__ double red[kBlock / 64][3];
  __shared__ double cs[2];
  int64_t p, q;
  TournamentPair(npad, step, blockIdx.x, &p, &q);
  if (p >= n || q >= n) return;  // the dummy player of an odd tournament
  if (p > q) {
    int64_t t = p;
    p = q;
    q = t;
  }
  T* wp = W + p * m;
  T* wq = W + q * m;
  double a = 0, b = 0, g = 0;
  for (int64_t i = threadIdx.x; i < m; i += kBlock) {
    const double x = static_cast<double>(wp[i]), y = static_cast<double>(wq[i]);
    a += x * x;
    b += y * y;
    g += x * y;
  }
  a = WaveSumD(a);
  b = WaveSumD(b);
  g = WaveSumD(g);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave][0] = a;
    red[wave][1] = b;
    red[wave][2] = g;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double A = 0, B = 0, G = 0;
    for (int w = 0; w < kBlock / 64; ++w) {
      A += red[w][0];
      B += red[w][1];
      G += red[w][2];
    }
    double c = 1.0, s = 0.0;
    if (fabs(G) > tol * sqrt(A * B) && G != 0.0) {
      const double zeta = (B - A) / (2.0 * G);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      c = 1.0 / sqrt(1.0 + t * t);
      s = c * t;
      *rotated = 1;
    }
    cs[0] = c;
    cs[1] = s;
  }
  __syncthreads();
  const double c = cs[0], s = cs[1];
  if (s == 0.0) return;
  for (int64_t i = threadIdx.x; i < m; i += kBlock) {
    const double x = static_cast<double>(wp[i]), y = static_cast<double>(wq[i]);
    wp[i] = static_cast<T>(c * x - s * y);
    wq[i] = static_cast<T>(s * x + c * y);
  }
  T* vp = V + p * n;
  T* vq = V + q * n;
  for (int64_t i = threadIdx.x; i < n; i += kBlock) {
    const double x = static_cast<double>(vp[i]), y = static_cast<double>(vq[i]);
    vp[i] = static_cast<T>(c * x - s * y);
    vq[i] = static_cast<T>(s * x + c * y);
  }
}

// Small matrices (W and V together fit in the LDS of one CU): the whole decomposition - every
// rotation step of every sweep - in ONE launch of one workgroup.  The launch-per-step form above
// costs ~5 us per step whatever the size (99 launches and a host round trip per sweep at
// n = 100); here a step is a wave per column pair on LDS-resident columns and a barrier.
constexpr int kSmallThreads = 1024;
constexpr size_t kSmallLdsBytes = 150 * 1024;

// NI > 0: every lane keeps its (at most NI) entries of the two columns in registers between
// the dot products and the rotation - one batch of LDS reads per pair instead of two dependent
// loops (the step is latency-bound: ~7 reads deep per loop at n = 100); NI = 0: plain loops.
template <class T, int NI>
__global__ __launch_bounds__(kSmallThreads) void SmallJacobiSvdKernel(T* Wg, int m, int n, T* Vg,
                                                                      int npad, double tol,
                                                                      int max_sweeps, int warm, int G,
                                                                      int* sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char small_svd_lds[];
  T* W = reinterpret_cast<T*>(small_svd_lds);  // m x n, column-major
  T* V = W + m * n;                            // n x n
  __shared__ int rotated;
  // a group of G lanes per column pair, every pair of a step with its own group (the host
  // chooses G: 8 lanes at n = 100), so a step is one pass and a barrier.
  // 32-bit indices and a modulo-free tournament: the step is instruction-bound (16 waves on
  // one CU), and 64-bit index arithmetic was most of it.
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & (G - 1), grp = tid / G, ngrp = nthreads / G;
  for (int i = tid; i < m * n; i += nthreads) W[i] = Wg[i];
  if (warm) {
    for (int i = tid; i < n * n; i += nthreads) V[i] = Vg[i];
  } else {
    for (int i = tid; i < n * n; i += nthreads) V[i] = (i % (n + 1) == 0) ? T(1) : T(0);
  }
  const int ring = npad - 1;
  int sweeps = 0;
  for (; sweeps < max_sweeps; ++sweeps) {
    if (tid == 0) rotated = 0;
    __syncthreads();
    for (int step = 0; step < ring; ++step) {
      for (int pair = grp; pair < npad / 2; pair += ngrp) {
        // round-robin tournament: player 0 is fixed, the others rotate (TournamentPair)
        auto player = [&](int pos) {
          if (pos == 0) return 0;
          int r = pos - 1 + step;
          if (r >= ring) r -= ring;
          return 1 + r;
        };
        int p = player(pair), q = player(npad - 1 - pair);
        if (p >= n || q >= n) continue;  // the dummy player of an odd tournament
        if (p > q) {
          const int t = p;
          p = q;
          q = t;
        }
        T* wp = W + p * m;
        T* wq = W + q * m;
        T* vp = V + p * n;
        T* vq = V + q * n;
        double a = 0, b = 0, g = 0;
        T xw[NI > 0 ? NI : 1], yw[NI > 0 ? NI : 1];
        if constexpr (NI > 0) {
#pragma unroll
          for (int k = 0; k < NI; ++k) {
            const int i = lane + k * G;
            const bool in = i < m;
            xw[k] = in ? wp[i] : T(0);
            yw[k] = in ? wq[i] : T(0);
          }
#pragma unroll
          for (int k = 0; k < NI; ++k) {
            const double x = static_cast<double>(xw[k]), y = static_cast<double>(yw[k]);
            a += x * x;
            b += y * y;
            g += x * y;
          }
        } else {
          for (int i = lane; i < m; i += G) {
            const double x = static_cast<double>(wp[i]), y = static_cast<double>(wq[i]);
            a += x * x;
            b += y * y;
            g += x * y;
          }
        }
        for (int off = G >> 1; off > 0; off >>= 1) {
          a += __shfl_xor(a, off, 64);
          b += __shfl_xor(b, off, 64);
          g += __shfl_xor(g, off, 64);
        }
        // The sums are fp64; the rotation itself is formed and applied in the storage precision.
        const T gt = static_cast<T>(g), dt = static_cast<T>(b - a);
        const T lim = static_cast<T>(tol) * (sqrt(static_cast<T>(a)) * sqrt(static_cast<T>(b)));
        if (!(fabs(gt) > lim) || gt == T(0)) continue;  // same decision in the group
        T xv[NI > 0 ? NI : 1], yv[NI > 0 ? NI : 1];
        if constexpr (NI > 0) {  // in flight while the rotation is being formed
#pragma unroll
          for (int k = 0; k < NI; ++k) {
            const int i = lane + k * G;
            const bool in = i < n;
            xv[k] = in ? vp[i] : T(0);
            yv[k] = in ? vq[i] : T(0);
          }
        }
        const T zeta = dt / (T(2) * gt);
        const T t = (zeta >= T(0) ? T(1) : T(-1)) / (fabs(zeta) + sqrt(T(1) + zeta * zeta));
        const T c = T(1) / sqrt(T(1) + t * t), sn = c * t;
        if (lane == 0) rotated = 1;
        if constexpr (NI > 0) {
#pragma unroll
          for (int k = 0; k < NI; ++k) {
            const int i = lane + k * G;
            if (i < m) {
              wp[i] = c * xw[k] - sn * yw[k];
              wq[i] = sn * xw[k] + c * yw[k];
            }
            if (i < n) {
              vp[i] = c * xv[k] - sn * yv[k];
              vq[i] = sn * xv[k] + c * yv[k];
            }
          }
        } else {
          for (int i = lane; i < m; i += G) {
            const T x = wp[i], y = wq[i];
            wp[i] = c * x - sn * y;
            wq[i] = sn * x + c * y;
          }
          for (int i = lane; i < n; i += G) {
            const T x = vp[i], y = vq[i];
            vp[i] = c * x - sn * y;
            vq[i] = sn * x + c * y;
          }
        }
      }
      __syncthreads();
    }
    const int any = rotated;
    __syncthreads();  // everyone has read the flag before the next sweep clears it
    if (any == 0) break;
  }
  for (int i = tid; i < m * n; i += nthreads) Wg[i] = W[i];
  for (int i = tid; i < n * n; i += nthreads) Vg[i] = V[i];
  if (tid == 0) *sweeps_out = sweeps;
}

template <class T, int NI>
bool LaunchSmallJacobi(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps, double tol,
                       bool warm, size_t bytes, int G, int threads, int* sweeps) {
  static const bool big_lds_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&SmallJacobiSvdKernel<T, NI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(kSmallLdsBytes)) == hipSuccess;
  }();
  if (!big_lds_ok && bytes > 60 * 1024) return false;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  auto out = rt.Alloc(sizeof(int));
  const int64_t npad = n + (n & 1);
  hipLaunchKernelGGL((SmallJacobiSvdKernel<T, NI>), dim3(1), dim3(threads), bytes, s, W.as<T>(),
                     static_cast<int>(m), static_cast<int>(n), V.as<T>(), static_cast<int>(npad), tol,
                     max_sweeps, warm ? 1 : 0, G, static_cast<int*>(out->p));
  EPS_HIP(hipGetLastError());
  EPS_HIP(hipMemcpyAsync(sweeps, out->p, sizeof(int), hipMemcpyDeviceToHost, s));
  EPS_HIP(hipStreamSynchronize(s));
  return true;
}

// fp32 form of the on-chip kernel rebuilt around the latency of a rotation step, the same way as
// PairEigFastKernel further down (the 121-sweep robust-PCA solve of the reference's benchmark
// spent 188 of its 225 ms in 121 calls of the kernel above: 3.9 us per rotation step).  A group
// of 8 lanes owns a column pair, a lane up to 16 CONSECUTIVE rows of it (16-byte LDS accesses,
// column stride rows + 4), the columns of V are fetched together with those of W so that their
// latency hides behind the dot products, and the 8-lane sums run on DPP row operations.  The sums
// stay fp64 and the rotation is formed with the same correctly rounded operations as above: only
// the order of the additions inside a dot product differs.  Rows are padded with zeros to a
// multiple of 32 (they contribute nothing and stay zero under rotations).
template <int CTRL> __device__ __forceinline__ double DppMoveD(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(b), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(b >> 32), CTRL, 0xF, 0xF, false);
  return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ __forceinline__ double Sum8D(double v) {
  v += DppMoveD<0xB1>(v);   // quad_perm [1,0,3,2]
  v += DppMoveD<0x4E>(v);   // quad_perm [2,3,0,1]
  v += DppMoveD<0x141>(v);  // row_half_mirror
  return v;
}

__global__ __launch_bounds__(512) void SmallJacobiSvdFastKernel(float* Wg, int m, int n, float* Vg, int npad,
                                                                int mp, int np, double tol, int max_sweeps,
                                                                int warm, int* sweeps_out, int keep_v) {
  extern __shared__ __attribute__((aligned(16))) unsigned char small_svd_lds[];
  const int ldw = mp + 4, ldv = np + 4;
  float* W = reinterpret_cast<float*>(small_svd_lds);  // n columns of ldw floats
  float* V = W + n * ldw;                              // n columns of ldv floats
  __shared__ int rotated;
  __shared__ unsigned rotmax;  // largest |cos| rotated away in the sweep (float bits)
  const int tid = threadIdx.x, nthreads = blockDim.x;
  for (int i = tid; i < n * ldw; i += nthreads) {
    const int c = i / ldw, r = i - c * ldw;
    W[i] = r < m ? Wg[r + c * m] : 0.0f;
  }
  // keep_v == 0: the rotations go to W alone (the caller needs one orthonormal side only and
  // re-forms the other from the matrix: JacobiSvdNoV) - no V in LDS, half the LDS traffic of a step
  if (keep_v)
    for (int i = tid; i < n * ldv; i += nthreads) {
      const int c = i / ldv, r = i - c * ldv;
      V[i] = r < n ? (warm ? Vg[r + c * n] : (r == c ? 1.0f : 0.0f)) : 0.0f;
    }
  const int pair = tid >> 3, sub = tid & 7;
  const int rw4 = mp / 32, rv4 = keep_v ? np / 32 : 0;  // float4 per lane and column (<= 4)
  const int ring = npad - 1;
  int sweeps = 0;
  for (; sweeps < max_sweeps; ++sweeps) {
    if (tid == 0) {
      rotated = 0;
      rotmax = 0;
    }
    __syncthreads();
    for (int step = 0; step < ring; ++step) {
      if (pair < npad / 2) {
        int pp = pair - 1 + step;
        if (pp >= ring) pp -= ring;
        int p = pair == 0 ? 0 : 1 + pp;
        int qq = npad - 2 - pair + step;
        if (qq >= ring) qq -= ring;
        int q = 1 + qq;
        if (p < n && q < n) {  // (the dummy player of an odd tournament sits out)
          if (p > q) {
            const int t = p;
            p = q;
            q = t;
          }
          float4* wp = reinterpret_cast<float4*>(W + p * ldw + sub * (mp / 8));
          float4* wq = reinterpret_cast<float4*>(W + q * ldw + sub * (mp / 8));
          float4* vp = reinterpret_cast<float4*>(V + p * ldv + sub * (np / 8));
          float4* vq = reinterpret_cast<float4*>(V + q * ldv + sub * (np / 8));
          float4 x[4], y[4], u[4], w[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < rw4) {
              x[i] = wp[i];
              y[i] = wq[i];
            }
            if (i < rv4) {
              u[i] = vp[i];
              w[i] = vq[i];
            }
          }
          double a = 0, b = 0, g = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < rw4) {
              const float xs[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
              const float ys[4] = {y[i].x, y[i].y, y[i].z, y[i].w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const double xd = static_cast<double>(xs[e]), yd = static_cast<double>(ys[e]);
                a += xd * xd;
                b += yd * yd;
                g += xd * yd;
              }
            }
          }
          a = Sum8D(a);
          b = Sum8D(b);
          g = Sum8D(g);
          // the sums are fp64; the rotation itself is formed and applied in the storage precision
          const float gt = static_cast<float>(g), dt = static_cast<float>(b - a);
          const float lim = static_cast<float>(tol) * (sqrtf(static_cast<float>(a)) * sqrtf(static_cast<float>(b)));
          if (fabsf(gt) > lim && gt != 0.0f) {
            const float zeta = dt / (2.0f * gt);
            const float t = (zeta >= 0.0f ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
            const float c = 1.0f / sqrtf(1.0f + t * t), sn = c * t;
            if (sub == 0) {
              rotated = 1;
              atomicMax(&rotmax, __float_as_uint(fabsf(gt) / (sqrtf(static_cast<float>(a)) * sqrtf(static_cast<float>(b)))));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (i < rw4) {
                float4 xo, yo;
                xo.x = c * x[i].x - sn * y[i].x;
                xo.y = c * x[i].y - sn * y[i].y;
                xo.z = c * x[i].z - sn * y[i].z;
                xo.w = c * x[i].w - sn * y[i].w;
                yo.x = sn * x[i].x + c * y[i].x;
                yo.y = sn * x[i].y + c * y[i].y;
                yo.z = sn * x[i].z + c * y[i].z;
                yo.w = sn * x[i].w + c * y[i].w;
                wp[i] = xo;
                wq[i] = yo;
              }
              if (i < rv4) {
                float4 uo, wo;
                uo.x = c * u[i].x - sn * w[i].x;
                uo.y = c * u[i].y - sn * w[i].y;
                uo.z = c * u[i].z - sn * w[i].z;
                uo.w = c * u[i].w - sn * w[i].w;
                wo.x = sn * u[i].x + c * w[i].x;
                wo.y = sn * u[i].y + c * w[i].y;
                wo.z = sn * u[i].z + c * w[i].z;
                wo.w = sn * u[i].w + c * w[i].w;
                vp[i] = uo;
                vq[i] = wo;
              }
            }
          }
        }
      }
      __syncthreads();
    }
    const int any = rotated;
    const float worst = __uint_as_float(rotmax);
    __syncthreads();
    if (any == 0) break;
    // quadratic convergence: a sweep whose largest rotated |cos| was below 3e-4 leaves ~1e-7,
    // under the 2e-7 rotation threshold - the next sweep would rotate nothing (it is the
    // confirming sweep, a quarter of a warm-started decomposition) and is not run
    if (worst <= 3e-4f) {
      ++sweeps;
      break;
    }
  }
  for (int i = tid; i < m * n; i += nthreads) {
    const int c = i / m, r = i - c * m;
    Wg[i] = W[c * ldw + r];
  }
  if (keep_v)
    for (int i = tid; i < n * n; i += nthreads) {
      const int c = i / n, r = i - c * n;
      Vg[i] = V[c * ldv + r];
    }
  if (tid == 0) *sweeps_out = sweeps;
}

bool SmallFastEnabled() {
  static const bool off = [] {
    const char* e = std::getenv("EPSILON_HIP_SVD_SMALL_FAST");
    return e && e[0] == '0';
  }();
  return !off;
}

// false: shape outside the fast kernel's range (the general kernel takes it).  keep_v == false: V is
// not touched (may be an empty DVec).
bool SmallJacobiSvdFast(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps, double tol,
                        bool warm, int* sweeps, bool keep_v = true) {
  if (!SmallFastEnabled() || m > 128 || n > 128 || n < 2) return false;
  const int64_t npad = n + (n & 1);
  const int mp = static_cast<int>((m + 31) / 32 * 32), np = static_cast<int>((n + 31) / 32 * 32);
  const size_t bytes = static_cast<size_t>(n) * (mp + 4 + (keep_v ? np + 4 : 0)) * sizeof(float);
  if (bytes > kSmallLdsBytes) return false;
  static const bool big_lds_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&SmallJacobiSvdFastKernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(kSmallLdsBytes)) == hipSuccess;
  }();
  if (!big_lds_ok && bytes > 60 * 1024) return false;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  auto out = rt.Alloc(sizeof(int));
  int threads = static_cast<int>(((npad / 2) * 8 + 63) / 64 * 64);
  threads = std::max(64, std::min(threads, 512));
  hipLaunchKernelGGL(SmallJacobiSvdFastKernel, dim3(1), dim3(threads), bytes, s, W.as<float>(),
                     static_cast<int>(m), static_cast<int>(n), keep_v ? V.as<float>() : static_cast<float*>(nullptr),
                     static_cast<int>(npad), mp, np, tol, max_sweeps, warm ? 1 : 0, static_cast<int*>(out->p),
                     keep_v ? 1 : 0);
  EPS_HIP(hipGetLastError());
  if (sweeps != nullptr) {  // (nullptr: nobody wants the count - no host synchronisation)
    EPS_HIP(hipMemcpyAsync(sweeps, out->p, sizeof(int), hipMemcpyDeviceToHost, s));
    EPS_HIP(hipStreamSynchronize(s));
  }
  return true;
}

template <class T> bool SmallJacobiSvd(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps,
                                       double tol, bool warm, int* sweeps) {
  const size_t bytes = static_cast<size_t>(m * n + n * n) * sizeof(T);
  if (bytes > kSmallLdsBytes) return false;
  if constexpr (sizeof(T) == 4) {
    if (SmallJacobiSvdFast(W, m, n, V, max_sweeps, tol, warm, sweeps)) return true;
  }
  // a group of G lanes per column pair; every pair of a step has its own group
  const int64_t npad = n + (n & 1);
  // (measured at n = 100: G = 8 on 448 threads 0.208 s per 121-sweep solve, G = 16 on 832
  // threads 0.218 s, G = 4 and G = 32 0.32 s)
  int G = 64;
  while (G > 8 && (npad / 2) * G > 512) G >>= 1;
  int threads = static_cast<int>(((npad / 2) * G + 63) / 64 * 64);
  threads = std::max(64, std::min(threads, kSmallThreads));
  const int64_t need = (std::max(m, n) + G - 1) / G;
  if (need <= 8) return LaunchSmallJacobi<T, 8>(W, m, n, V, max_sweeps, tol, warm, bytes, G, threads, sweeps);
  if constexpr (sizeof(T) == 4) {  // 4 x 16 doubles per lane would spill
    if (need <= 16) return LaunchSmallJacobi<T, 16>(W, m, n, V, max_sweeps, tol, warm, bytes, G, threads, sweeps);
  }
  return LaunchSmallJacobi<T, 0>(W, m, n, V, max_sweeps, tol, warm, bytes, G, threads, sweeps);
}

// sigma[j] = ||W[:, j]||_2
template <class T>
__global__ __launch_bounds__(kBlock) void ColNormKernel(const T* W, int64_t m, int64_t n,
                                                        T* sigma, int squared = 0) {
  __shared__ double red[kBlock / 64];
  const int64_t j = blockIdx.x;
  double a = 0;
  for (int64_t i = threadIdx.x; i < m; i += kBlock) {
    const double x = static_cast<double>(W[i + j * m]);
    a += x * x;
  }
  a = WaveSumD(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < kBlock / 64; ++w) t += red[w];
    sigma[j] = static_cast<T>(squared ? t : sqrt(t));
  }
}

template <class T> __global__ void SqrtInPlaceKernel(T* x, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(256) + threadIdx.x;
  if (i < n) x[i] = sqrt(x[i]);
}

// W[:, j] *= (sigma[j] != 0 ? xt[j] / sigma[j] : 0)
template <class T>
__global__ __launch_bounds__(kBlock) void ColScaleKernel(T* W, int64_t m, int64_t n,
                                                         const T* sigma, const T* xt) {
  const int64_t j = blockIdx.x;
  const T s = sigma[j];
  const T f = (s != T(0)) ? xt[j] / s : T(0);
  for (int64_t i = threadIdx.x; i < m; i += kBlock) W[i + j * m] *= f;
}

template <class T> __global__ void EyeKernel(T* V, int64_t n) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx < n * n) V[idx] = (idx % n == idx / n) ? T(1) : T(0);
}

// ---- block one-sided Jacobi ------------------------------------------------------------------------
// For more than a few hundred columns the scalar algorithm above streams the whole matrix once
// per rotation step (n - 1 steps per sweep).  The block form works on pairs of 32-column panels:
//   G_k = P_k^T P_k           (64 x 64 Gram of the panel pair, batched GEMM on the MFMA kernel)
//   G_k = J_k D J_k^T         (on chip: one workgroup per pair, one-sided Jacobi in LDS)
//   P_k <- P_k J_k, V_k <- V_k J_k   (batched GEMMs)
// so a sweep is nb - 1 steps of GEMMs instead of n - 1 streaming passes.  Panels are kept
// physically adjacent (pair k = column blocks 2k, 2k+1) and moved by the round-robin permutation
// after every step.

constexpr int kJB = 32;         // panel width
constexpr int kJN = 2 * kJB;    // order of the pair problems
constexpr int kJLd = kJN + 1;   // LDS row stride

// One workgroup per pair: G (sum of `nsplit` partial Grams) -> eigenvectors J (kJN x kJN).
// Also folds max_{i != j} |G_ij| / sqrt(G_ii G_jj) into *offmax (float bits, atomicMax).
template <class T>
__global__ __launch_bounds__(kBlock) void PairEigKernel(const T* __restrict__ G, int nsplit,
                                                        int64_t split_stride, T* __restrict__ J,
                                                        int inner_sweeps, double tol,
                                                        double skip_below, unsigned int* offmax) {
  __shared__ T A[kJN * kJLd];  // column-major: A[c * kJLd + r]
  __shared__ T E[kJN * kJLd];
  __shared__ float wmax[kBlock / 64];
  const int t = threadIdx.x;
  const T* g = G + static_cast<int64_t>(blockIdx.x) * kJN * kJN;
  for (int idx = t; idx < kJN * kJN; idx += kBlock) {
    const int r = idx % kJN, c = idx / kJN;
    T v = T(0);
    for (int sp = 0; sp < nsplit; ++sp) v += g[sp * split_stride + idx];
    A[c * kJLd + r] = v;
    E[c * kJLd + r] = (r == c) ? T(1) : T(0);
  }
  __syncthreads();
  // symmetrise (the partial sums of the two triangles differ in rounding) and measure
  float mx = 0.0f;
  for (int idx = t; idx < kJN * kJN; idx += kBlock) {
    const int r = idx % kJN, c = idx / kJN;
    if (r > c) {
      const double v = 0.5 * (static_cast<double>(A[c * kJLd + r]) + static_cast<double>(A[r * kJLd + c]));
      const double d = static_cast<double>(A[r * kJLd + r]) * static_cast<double>(A[c * kJLd + c]);
      if (d > 0) mx = fmaxf(mx, static_cast<float>(fabs(v) / sqrt(d)));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((t & 63) == 0) wmax[t >> 6] = mx;
  __syncthreads();
  const float m4 = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  if (t == 0) atomicMax(offmax, __float_as_uint(m4));
  if (m4 <= static_cast<float>(skip_below)) {
    // this pair of panels is already orthogonal to working accuracy: J = I, no sweeps
    T* jj = J + static_cast<int64_t>(blockIdx.x) * kJN * kJN;
    for (int idx = t; idx < kJN * kJN; idx += kBlock) jj[idx] = (idx % kJN == idx / kJN) ? T(1) : T(0);
    return;
  }
  // one-sided Jacobi on the columns of A; kJN / 2 = 32 pairs per step, 8 lanes per pair
  constexpr int kTpp = kBlock / (kJN / 2);
  const int pair = t / kTpp, sub = t % kTpp;
  for (int sw = 0; sw < inner_sweeps; ++sw) {
    for (int step = 0; step < kJN - 1; ++step) {
      // round-robin tournament, 32-bit and modulo-free (the 64-bit remainders of TournamentPair
      // were a sizeable part of a step that is otherwise ~40 instructions)
      auto player = [&](int pos) {
        if (pos == 0) return 0;
        int r = pos - 1 + step;
        if (r >= kJN - 1) r -= kJN - 1;
        return 1 + r;
      };
      const int p = player(pair), q = player(kJN - 1 - pair);
      T* ap = A + p * kJLd;
      T* aq = A + q * kJLd;
      // rotations in the storage precision (these are 64 x 64 problems whose result only has to
      // reduce the off-diagonal mass: the outer iteration corrects what an inner one leaves)
      T a = 0, b = 0, gm = 0;
      for (int r = sub; r < kJN; r += kTpp) {
        const T x = ap[r], y = aq[r];
        a += x * x;
        b += y * y;
        gm += x * y;
      }
#pragma unroll
      for (int off = kTpp / 2; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
        gm += __shfl_xor(gm, off, 64);
      }
      T c = 1, s = 0;
      if (fabs(gm) > static_cast<T>(tol) * sqrt(a * b) && gm != T(0)) {
        const T zeta = (b - a) / (T(2) * gm);
        const T tt = (zeta >= 0 ? T(1) : T(-1)) / (fabs(zeta) + sqrt(T(1) + zeta * zeta));
        c = T(1) / sqrt(T(1) + tt * tt);
        s = c * tt;
      }
      if (s != T(0)) {
        T* ep = E + p * kJLd;
        T* eq = E + q * kJLd;
        for (int r = sub; r < kJN; r += kTpp) {
          const T x = ap[r], y = aq[r];
          ap[r] = c * x - s * y;
          aq[r] = s * x + c * y;
          const T u = ep[r], w = eq[r];
          ep[r] = c * u - s * w;
          eq[r] = s * u + c * w;
        }
      }
      __syncthreads();
    }
  }
  T* j = J + static_cast<int64_t>(blockIdx.x) * kJN * kJN;
  for (int idx = t; idx < kJN * kJN; idx += kBlock) {
    const int r = idx % kJN, c = idx / kJN;
    j[idx] = E[c * kJLd + r];
  }
}

// fp32 form of PairEigKernel, rebuilt around the latency of a rotation step (the kernel is a chain
// of 63 dependent steps on one workgroup: at 2.1 us per step it was 131 us per launch, a third of a
// block-Jacobi step once the products run on the matrix cores).  A lane owns 8 CONSECUTIVE rows of
// its pair's two columns: two 16-byte LDS reads per column instead of eight scalar ones (column
// stride 68 floats keeps them aligned), the eigenvector columns are fetched with them so their
// latency hides behind the dot products, the 8-lane sums run on DPP row operations instead of
// ds_bpermute, and the rotation is formed with v_rcp / v_rsq / v_sqrt (1 ulp) instead of the
// correctly rounded division and square-root sequences.
constexpr int kELd = 68;

template <int CTRL> __device__ __forceinline__ float DppMove(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 8 lanes of a group (aligned to 8 within a row of 16)
__device__ __forceinline__ float Sum8(float v) {
  v += DppMove<0xB1>(v);   // quad_perm [1,0,3,2]
  v += DppMove<0x4E>(v);   // quad_perm [2,3,0,1]
  v += DppMove<0x141>(v);  // row_half_mirror: lane i <-> 7 - i of the 8
  return v;
}
__device__ __forceinline__ float Dot4(const float4& a, const float4& b, float acc) {
  acc = __builtin_fmaf(a.x, b.x, acc);
  acc = __builtin_fmaf(a.y, b.y, acc);
  acc = __builtin_fmaf(a.z, b.z, acc);
  return __builtin_fmaf(a.w, b.w, acc);
}
__device__ __forceinline__ void Rot4(float4& x, float4& y, float c, float s) {
  const float4 u = x, w = y;
  x.x = __builtin_fmaf(c, u.x, -s * w.x);
  x.y = __builtin_fmaf(c, u.y, -s * w.y);
  x.z = __builtin_fmaf(c, u.z, -s * w.z);
  x.w = __builtin_fmaf(c, u.w, -s * w.w);
  y.x = __builtin_fmaf(s, u.x, c * w.x);
  y.y = __builtin_fmaf(s, u.y, c * w.y);
  y.z = __builtin_fmaf(s, u.z, c * w.z);
  y.w = __builtin_fmaf(s, u.w, c * w.w);
}

__global__ __launch_bounds__(kBlock) void PairEigFastKernel(const float* __restrict__ G, int nsplit,
                                                            int64_t split_stride, float* __restrict__ J,
                                                            int inner_sweeps, float tol, float skip_below,
                                                            unsigned int* offmax, int32_t* __restrict__ skip) {
  __shared__ __attribute__((aligned(16))) float A[kJN * kELd];  // column-major, stride 68
  __shared__ __attribute__((aligned(16))) float E[kJN * kELd];
  __shared__ float wmax[kBlock / 64];
  const int t = threadIdx.x;
  const float* g = G + static_cast<int64_t>(blockIdx.x) * kJN * kJN;
  for (int q4 = t; q4 < kJN * kJN / 4; q4 += kBlock) {
    const int c = q4 >> 4, r = (q4 & 15) * 4;
    float4 v = *reinterpret_cast<const float4*>(g + 4 * q4);
    for (int sp = 1; sp < nsplit; ++sp) {
      const float4 w = *reinterpret_cast<const float4*>(g + sp * split_stride + 4 * q4);
      v.x += w.x;
      v.y += w.y;
      v.z += w.z;
      v.w += w.w;
    }
    *reinterpret_cast<float4*>(&A[c * kELd + r]) = v;
    float4 e = {0.f, 0.f, 0.f, 0.f};
    if (c >= r && c < r + 4) (&e.x)[c - r] = 1.0f;
    *reinterpret_cast<float4*>(&E[c * kELd + r]) = e;
  }
  __syncthreads();
  float mx = 0.0f;
  for (int idx = t; idx < kJN * kJN; idx += kBlock) {
    const int r = idx & 63, c = idx >> 6;
    if (r > c) {
      const float d = A[r * kELd + r] * A[c * kELd + c];
      if (d > 0.0f) mx = fmaxf(mx, fabsf(A[c * kELd + r]) * __builtin_amdgcn_rsqf(d));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((t & 63) == 0) wmax[t >> 6] = mx;
  __syncthreads();
  const float m4 = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  if (t == 0) {
    atomicMax(offmax, __float_as_uint(m4));
    skip[blockIdx.x] = m4 <= skip_below ? 1 : 0;  // the updates of this pair are not launched into memory
  }
  float* jj = J + static_cast<int64_t>(blockIdx.x) * kJN * kJN;
  if (m4 <= skip_below) {  // already orthogonal to working accuracy: J = I (E holds it)
    for (int q4 = t; q4 < kJN * kJN / 4; q4 += kBlock)
      *reinterpret_cast<float4*>(jj + 4 * q4) = *reinterpret_cast<const float4*>(&E[(q4 >> 4) * kELd + (q4 & 15) * 4]);
    return;
  }
  const int pair = t >> 3, r0 = (t & 7) * 8;
  for (int sw = 0; sw < inner_sweeps; ++sw) {
    for (int step = 0; step < kJN - 1; ++step) {
      int pp = pair - 1 + step;  // player(pos) = pos == 0 ? 0 : 1 + (pos - 1 + step) mod 63
      if (pp >= kJN - 1) pp -= kJN - 1;
      const int p = pair == 0 ? 0 : 1 + pp;
      int qq = kJN - 2 - pair + step;
      if (qq >= kJN - 1) qq -= kJN - 1;
      const int q = 1 + qq;
      float4* ap = reinterpret_cast<float4*>(&A[p * kELd + r0]);
      float4* aq = reinterpret_cast<float4*>(&A[q * kELd + r0]);
      float4* ep = reinterpret_cast<float4*>(&E[p * kELd + r0]);
      float4* eq = reinterpret_cast<float4*>(&E[q * kELd + r0]);
      float4 x0 = ap[0], x1 = ap[1], y0 = aq[0], y1 = aq[1];
      float4 u0 = ep[0], u1 = ep[1], w0 = eq[0], w1 = eq[1];
      float a = Dot4(x1, x1, Dot4(x0, x0, 0.0f));
      float b = Dot4(y1, y1, Dot4(y0, y0, 0.0f));
      float gm = Dot4(x1, y1, Dot4(x0, y0, 0.0f));
      a = Sum8(a);
      b = Sum8(b);
      gm = Sum8(gm);
      if (fabsf(gm) > tol * (__builtin_amdgcn_sqrtf(a) * __builtin_amdgcn_sqrtf(b))) {
        const float zeta = (b - a) * __builtin_amdgcn_rcpf(2.0f * gm);
        const float den = fabsf(zeta) + __builtin_amdgcn_sqrtf(__builtin_fmaf(zeta, zeta, 1.0f));
        const float tt = __builtin_copysignf(__builtin_amdgcn_rcpf(den), zeta);
        const float c = __builtin_amdgcn_rsqf(__builtin_fmaf(tt, tt, 1.0f));
        const float s = c * tt;
        if (s != 0.0f) {
          Rot4(x0, y0, c, s);
          Rot4(x1, y1, c, s);
          Rot4(u0, w0, c, s);
          Rot4(u1, w1, c, s);
          ap[0] = x0;
          ap[1] = x1;
          aq[0] = y0;
          aq[1] = y1;
          ep[0] = u0;
          ep[1] = u1;
          eq[0] = w0;
          eq[1] = w1;
        }
      }
      __syncthreads();
    }
  }
  for (int q4 = t; q4 < kJN * kJN / 4; q4 += kBlock)
    *reinterpret_cast<float4*>(jj + 4 * q4) = *reinterpret_cast<const float4*>(&E[(q4 >> 4) * kELd + (q4 & 15) * 4]);
}

// dst[:, dst_block[b]] = src[:, b] for column blocks of kJB columns (rows x nblocks*kJB)
template <class T>
__global__ __launch_bounds__(kBlock) void PermuteBlocksKernel(T* __restrict__ dst,
                                                              const T* __restrict__ src,
                                                              int64_t rows,
                                                              const int32_t* __restrict__ dst_block) {
  const int64_t b = blockIdx.y;
  const int64_t per = rows * kJB;
  const T* s = src + b * per;
  T* d = dst + static_cast<int64_t>(dst_block[b]) * per;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x; i < per;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    d[i] = s[i];
}

// Partial Gram matrices of a step: G[split][pair] = P^T P over the row chunk `split` of the
// 64-column pair.  Same reasoning as PanelUpdateKernel: a 64 x 64 result per pair wastes three
// quarters of a 128 x 128 MFMA tile.  A workgroup streams its row chunk through LDS in tiles of
// 128 rows; a thread owns a 4 x 4 block of the result.
template <class T>
__global__ __launch_bounds__(kBlock) void PanelGramKernel(const T* __restrict__ W, int64_t rows,
                                                          int64_t chunk, int64_t h,
                                                          T* __restrict__ G) {
  constexpr int TR = 128, LD = TR + 1;
  __shared__ T Ts[kJN * LD];  // [column][row]
  const int t = threadIdx.x;
  const int64_t pair = blockIdx.y, split = blockIdx.x;
  const int64_t r0 = split * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
  const T* src = W + pair * kJN * rows;
  const int a0 = (t & 15) * 4, b0 = (t >> 4) * 4;
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = T(0);
  for (int64_t rt = r0; rt < r1; rt += TR) {
    __syncthreads();
    for (int idx = t; idx < TR * kJN; idx += kBlock) {
      const int r = idx % TR, c = idx / TR;
      Ts[c * LD + r] = (rt + r < r1) ? src[(rt + r) + static_cast<int64_t>(c) * rows] : T(0);
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < TR; ++r) {
      T av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = Ts[(a0 + i) * LD + r];
        bv[i] = Ts[(b0 + i) * LD + r];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
  }
  T* g = G + (split * h + pair) * kJN * kJN;  // column-major 64 x 64
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) g[(a0 + i) + (b0 + j) * kJN] = acc[i][j];
}

// The update of a step, P_k <- P_k J_k for every pair k, with the round-robin move of the
// panels folded in: thread = one row of a 64-column pair (64 inputs in registers, J_k broadcast
// from LDS, 16 outputs at a time), written straight to where the two panels go next.  The
// products are 64 x 64 x rows - far too thin for the 128 x 128 MFMA tiles, which spent their
// time in prologues - and this way the panel matrix is read once and written once per step
// instead of three times each (GEMM out, permutation in and out).
template <class T>
__global__ __launch_bounds__(kBlock) void PanelUpdateKernel(T* __restrict__ dst,
                                                            const T* __restrict__ src, int64_t rows,
                                                            const T* __restrict__ J,
                                                            const int32_t* __restrict__ dst_block) {
  __shared__ T Js[kJN][kJN];  // [k][c]
  const int t = threadIdx.x;
  const int64_t pair = blockIdx.y;
  const T* j = J + pair * kJN * kJN;  // column-major: J[k + c * kJN]
  for (int idx = t; idx < kJN * kJN; idx += kBlock) Js[idx % kJN][idx / kJN] = j[idx];
  __syncthreads();
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + t;
  if (r >= rows) return;
  const T* s = src + r + pair * kJN * rows;
  T x[kJN];
#pragma unroll
  for (int k = 0; k < kJN; ++k) x[k] = s[static_cast<int64_t>(k) * rows];
  T* d0 = dst + r + static_cast<int64_t>(dst_block[2 * pair]) * kJB * rows;
  T* d1 = dst + r + static_cast<int64_t>(dst_block[2 * pair + 1]) * kJB * rows;
#pragma unroll 1
  for (int c0 = 0; c0 < kJN; c0 += 16) {
    T y[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) y[c] = T(0);
#pragma unroll
    for (int k = 0; k < kJN; ++k) {
#pragma unroll
      for (int c = 0; c < 16; ++c) y[c] += x[k] * Js[k][c0 + c];
    }
    T* d = (c0 < kJB ? d0 : d1) + static_cast<int64_t>(c0 % kJB) * rows;
#pragma unroll
    for (int c = 0; c < 16; ++c) d[static_cast<int64_t>(c) * rows] = y[c];
  }
}

// ---- the three kernels of a block-Jacobi step on the matrix cores (fp32) ----------------------------
// The thread-per-row / 4x4-per-thread forms above are fp32 VALU code fed from LDS one scalar at a
// time: measured at n = 10^4 (157 pairs per step) 512 us for the Grams and 372 us per update,
// against ~65 us (Gram: 0.4 GB read) and ~130 us (update: 0.8 GB read + written) of HBM time.
// v_mfma_f32_32x32x2_f32 is exact fp32 at the vector peak rate, needs ONE register per operand
// and lane:
//   Gram   G = P^T P: the contraction runs over rows, so ANY assignment of rows to MFMA k-slots
//          is valid as long as both operands use the same one: a lane reads 16-byte pieces of
//          its own column straight into operand registers.
//   Update P J: operands swapped (D^T = J^T P^T) so that the accumulator holds 32 consecutive
//          rows of one output column on 32 consecutive lanes: loads and stores are 128-byte
//          runs of a column; the 64 x 64 rotation block lives in 64 registers per lane.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// partial Gram matrices: grid (nsplit, h); G[(split * h + pair)] = 64 x 64, both triangles.
// The contraction runs over rows, so ANY assignment of rows to MFMA k-slots is valid as long as
// both operands use the same one: lane (c = l & 31, kk = l >> 5) reads 16 consecutive bytes of
// its OWN column (rows 8j + 4kk .. +3 of a 32-row block); a block's eight loads per lane cover
// whole 128-byte lines of the 64 columns.  (Tried and dropped: staging 128-row tiles through LDS
// so that global loads run along rows in 512-byte runs - 158 us per launch against 135: two
// workgroups per CU and a barrier per stage leave the matrix pipe idle more than the better
// load pattern returns.)  The pairs are walked from the last to the first (the update of the
// previous step ran in ascending pair order: a 1-2 % Infinity Cache effect).
__global__ __launch_bounds__(kBlock, 2) void PanelGramMfmaKernel(const float* __restrict__ W,
                                                                 int64_t rows, int64_t chunk, int64_t h,
                                                                 float* __restrict__ G,
                                                                 const int32_t* __restrict__ tab) {
  __shared__ float red[4][3][16][64];  // [wave][tile 00, 10, 11][register][lane]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = lane & 31, kk = lane >> 5;
  const int64_t pair = h - 1 - static_cast<int64_t>(blockIdx.y);
  const int64_t split = blockIdx.x;
  // the pair's two 32-column panels, wherever they are stored (tab: panel ids of this step's pairs)
  const float* col0 = W + (static_cast<int64_t>(tab[2 * pair]) * kJB + c) * rows + split * chunk + 4 * kk;
  const float* col1 = W + (static_cast<int64_t>(tab[2 * pair + 1]) * kJB + c) * rows + split * chunk + 4 * kk;
  f32x16 a00 = {0}, a10 = {0}, a11 = {0};
  const int64_t nblk = chunk / 32;  // the chunk is a multiple of 32 rows (zero-padded)
  // two register buffers used alternately (no copies: a copy of the prefetched block would make
  // the wave wait for it before the matrix instructions it was meant to hide behind)
  float4 va0[4], va1[4], vb0[4], vb1[4];
  auto load = [&](float4* x0, float4* x1, int64_t blk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x0[j] = *reinterpret_cast<const float4*>(col0 + blk * 32 + 8 * j);
      x1[j] = *reinterpret_cast<const float4*>(col1 + blk * 32 + 8 * j);
    }
  };
  auto compute = [&](const float4* x0v, const float4* x1v, float on) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0[4] = {on * x0v[j].x, on * x0v[j].y, on * x0v[j].z, on * x0v[j].w};
      const float x1[4] = {on * x1v[j].x, on * x1v[j].y, on * x1v[j].z, on * x1v[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[e], x0[e], a00, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[e], x0[e], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[e], x1[e], a11, 0, 0, 0);
      }
    }
  };
  // straight-line loop body: loads AND products are unconditional (block index clamped to the
  // last block, a surplus block enters scaled by zero).  A load or a product under a branch
  // lets the compiler sink the load next to its use / merge the wait counts of the two paths,
  // and the wave then waits for the block it was meant to prefetch.
  const int64_t last = nblk - 1;
  auto clamp = [&](int64_t x) { return x < last ? x : last; };
  int64_t b = wave;
  load(va0, va1, clamp(b));
  while (b < nblk) {
    load(vb0, vb1, clamp(b + 4));
    __builtin_amdgcn_sched_barrier(0);  // the scheduler would otherwise sink the loads to their uses
    compute(va0, va1, 1.0f);
    __builtin_amdgcn_sched_barrier(0);
    b += 4;
    load(va0, va1, clamp(b + 4));
    __builtin_amdgcn_sched_barrier(0);
    compute(vb0, vb1, b < nblk ? 1.0f : 0.0f);
    __builtin_amdgcn_sched_barrier(0);
    b += 4;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[wave][0][r][lane] = a00[r];
    red[wave][1][r][lane] = a10[r];
    red[wave][2][r][lane] = a11[r];
  }
  __syncthreads();
  // D[row][col] of a tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* g = G + (split * h + pair) * kJN * kJN;  // column-major 64 x 64
  for (int idx = t; idx < 3 * 16 * 64; idx += kBlock) {
    const int tl = idx / 1024, r = (idx >> 6) & 15, l = idx & 63;
    const float v = red[0][tl][r][l] + red[1][tl][r][l] + red[2][tl][r][l] + red[3][tl][r][l];
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) + (tl >= 1 ? 32 : 0);
    const int col = (l & 31) + (tl == 2 ? 32 : 0);
    g[col + row * kJN] = v;             // (row, col) through the symmetric position: lanes contiguous
    if (tl == 1) g[row + col * kJN] = v;  // the mirror of the off-diagonal tile
  }
}

// dst panels <- src pair * J (64 x 64), 32-row blocks; grid (ceil(rows / (32 * 4 * RB)), h)
// (Tried: non-temporal loads / stores for the update of V, so that its 0.8 GB per step do not
// push the panels of W out of the Infinity Cache before the next Gram launch re-reads them:
// Gram 133 -> 122 us, but the non-temporal update itself 162 -> 172 us; dropped.)
constexpr int kUpdRB = 4;  // row blocks per wave: the 64 registers of J are set up once per wave
// IN PLACE: the panels never move - the round-robin tournament is a table of panel ids per step
// (`tab`), so a pair whose rotation block is the identity (`skip`, set by the rotation solve when
// the pair is already orthogonal to working accuracy: most pairs of the last sweeps) costs no
// memory traffic at all, and no second copy of W and V is needed.
__global__ __launch_bounds__(kBlock, 2) void PanelUpdateMfmaKernel(float* P, int64_t rows,
                                                                   const float* __restrict__ J,
                                                                   const int32_t* __restrict__ tab,
                                                                   const int32_t* __restrict__ skip) {
  __shared__ float Js[kJN * kJLd];  // [c][k], stride 65
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, kk = lane >> 5;
  const int64_t pair = blockIdx.y;
  if (skip[pair]) return;  // J = I
  const float* jg = J + pair * kJN * kJN;  // column-major: J[k + 64 c]
  for (int idx = t; idx < kJN * kJN; idx += kBlock) Js[(idx >> 6) * kJLd + (idx & 63)] = jg[idx];
  __syncthreads();
  const int64_t nblk = rows / 32;
  int64_t b = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * kUpdRB;
  if (b >= nblk) return;
  float jf[2][32];  // A-slot operand: J[2 s + kk][32 t + r]
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int s = 0; s < 32; ++s) jf[tt][s] = Js[(32 * tt + r) * kJLd + 2 * s + kk];
  // uniform bases + one 32-bit lane offset: every access is "scalar base + lane offset", which
  // keeps the 64 addresses of a block out of the vector registers (a pair is 64 * rows floats,
  // far below 2^32 bytes)
  float* base0 = P + static_cast<int64_t>(tab[2 * pair]) * kJB * rows;      // columns 0..31 of the pair
  float* base1 = P + static_cast<int64_t>(tab[2 * pair + 1]) * kJB * rows;  // columns 32..63
  const unsigned urows = static_cast<unsigned>(rows);
  const unsigned lin = static_cast<unsigned>(r) + static_cast<unsigned>(kk) * urows;       // loads
  const unsigned lout = static_cast<unsigned>(r) + static_cast<unsigned>(4 * kk) * urows;  // stores
  const int64_t bend = b + kUpdRB < nblk ? b + kUpdRB : nblk;
  float pa[32], pb[32];
  auto load = [&](float* x, int64_t blk) {
    const unsigned off = lin + static_cast<unsigned>(blk) * 32u;
#pragma unroll
    for (int s = 0; s < 32; ++s) x[s] = ((s < 16 ? base0 : base1) + static_cast<int64_t>(2 * (s & 15)) * rows)[off];
  };
  auto compute = [&](const float* x, int64_t blk, bool store) {
    f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(jf[0][s], x[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(jf[1][s], x[s], acc1, 0, 0, 0);
    }
    // acc[reg] on lane l = out[row 32 blk + (l & 31)][column (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)]
    if (!store) return;
    const unsigned off = lout + static_cast<unsigned>(blk) * 32u;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int64_t cbase = static_cast<int64_t>((g & 3) + 8 * (g >> 2)) * rows;
      (base0 + cbase)[off] = acc0[g];
      (base1 + cbase)[off] = acc1[g];
    }
  };
  const int64_t last = nblk - 1;  // unconditional, clamped loads (see PanelGramMfmaKernel)
  auto clamp = [&](int64_t x) { return x < last ? x : last; };
  load(pa, b);
  while (b < bend) {
    load(pb, clamp(b + 1));
    __builtin_amdgcn_sched_barrier(0);
    compute(pa, b, true);
    __builtin_amdgcn_sched_barrier(0);
    ++b;
    load(pa, clamp(b + 1));
    __builtin_amdgcn_sched_barrier(0);
    compute(pb, b, b < bend);  // the products run either way, only the stores are conditional
    __builtin_amdgcn_sched_barrier(0);
    ++b;
  }
}

// dst (rows_out x ncols, ld rows_out) column j = src (ld lds) column col[j], first rows_out rows
template <class T>
__global__ __launch_bounds__(kBlock) void GatherColsKernel(T* __restrict__ dst, int64_t rows_out,
                                                           const T* __restrict__ src, int64_t lds,
                                                           const int32_t* __restrict__ col) {
  const int64_t j = blockIdx.x;
  const T* s = src + static_cast<int64_t>(col[j]) * lds;
  T* d = dst + j * rows_out;
  for (int64_t i = threadIdx.x; i < rows_out; i += kBlock) d[i] = s[i];
}

// keep_v = false (matrix-core path only): the rotations are applied to W alone - V is neither read
// nor written - for callers that rebuild the right factor from W afterwards (prox_more.cc): a
// step then streams W once more instead of W and V, 1.2 GB instead of 2.0 GB at n = 1e4.
template <class T, bool kMfma = std::is_same<T, float>::value>
int BlockJacobiImpl(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps, bool warm,
                    bool row_sharded, bool keep_v = true) {
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  const DType dt = W.dt;
  // fp32: Gram, rotation solve and update on the matrix-core / fast-rotation kernels
  static const bool mfma_off = [] {
    const char* e = std::getenv("EPSILON_HIP_SVD_MFMA");
    return e && e[0] == '0';
  }();
  if constexpr (kMfma) {
    if (mfma_off) return BlockJacobiImpl<T, false>(W, m, n, V, max_sweeps, warm, row_sharded);
  }
  if (!kMfma) keep_v = true;
  int64_t nb = (n + kJB - 1) / kJB;
  if (nb & 1) ++nb;
  const int64_t h = nb / 2, npad = nb * kJB;
  const double tol = dt == F32 ? 1e-7 : 1e-15;       // rotation threshold
  const double done_tol = dt == F32 ? 3e-6 : 1e-13;  // pairs below this are left alone (no rotations)
  // max |cos| is measured by the rotation solves at the START of a sweep (on the Gram matrices
  // they are handed), so the value read after sweep k describes the columns BEFORE it.  Cyclic
  // Jacobi converges quadratically once that value is small: a sweep that started below
  // `last_below` ends below last_below^2 / (relative gaps) - with 3e-4 that is 1e-7 against the
  // 3e-6 floor of the fp32 Gram products - and the sweep that would only confirm it (one of the
  // 4-5 sweeps of a warm-started decomposition) is not run.  EPSILON_HIP_SVD_CONFIRM=1 runs it.
  static const bool confirm = [] {
    const char* e = std::getenv("EPSILON_HIP_SVD_CONFIRM");
    return e && e[0] == '1';
  }();
  const double stop_tol = done_tol;
  const double last_below = (dt == F32 && !confirm) ? 3e-4 : 0.0;
  // split-K for the Gram products so that more than h workgroups run; rows are padded with
  // zeros to nsplit equal chunks of a multiple of 32 rows (16-byte aligned slices)
  // Row-sharded (each rank holds a block of rows of W, V replicated): the panel Grams are sums
  // over ranks - one all-reduce per step - after which every rank solves the same small
  // eigenproblems and applies the same rotations to its rows and to its copy of V.  The split
  // count must agree across ranks, so it is derived from the largest row block.
  Comm* comm = row_sharded ? rt.comm() : nullptr;
  const int64_t m_ref = comm ? static_cast<int64_t>(comm->AllReduceMaxHost(static_cast<double>(m)) + 0.5) : m;
  int64_t nsplit = 1;
  if constexpr (kMfma) {
    // matrix-core Gram: a workgroup is four waves that each take every fourth 32-row block of the
    // chunk; enough workgroups (~5 per CU) that the dispatcher evens out the CUs
    static const char* ns_env = std::getenv("EPSILON_HIP_SVD_NSPLIT");  // tuning knob
    const int64_t want = ns_env && std::atoi(ns_env) > 0 ? std::atoi(ns_env) : 512;
    while (nsplit < 16 && h * nsplit < want && m_ref / (nsplit * 2) >= 256) nsplit *= 2;
  } else {
    while (nsplit < 8 && h * nsplit < 512 && m_ref / (nsplit * 2) >= 256) nsplit *= 2;
  }
  const int64_t kchunk = ((m_ref + nsplit - 1) / nsplit + 31) / 32 * 32;
  const int64_t mp = nsplit * kchunk;
  // padded working copies
  // (the matrix-core path updates in place: no second copy)
  DVec Wp = DVec::Zeros(mp * npad, dt), Wt = kMfma ? DVec() : DVec::Empty(mp * npad, dt);
  DVec Vp = keep_v ? DVec::Zeros(npad * npad, dt) : DVec(), Vt = kMfma ? DVec() : DVec::Empty(npad * npad, dt);
  if (m > 0)
    EPS_HIP(hipMemcpy2DAsync(Wp.data(), mp * sizeof(T), W.data(), m * sizeof(T), m * sizeof(T), n,
                             hipMemcpyDeviceToDevice, s));
  if (!keep_v) {
    // nothing: the caller applied its start to W already
  } else if (warm) {  // V holds the orthogonal start; identity on the padding columns
    EPS_HIP(hipMemcpy2DAsync(Vp.data(), npad * sizeof(T), V.data(), n * sizeof(T), n * sizeof(T), n,
                             hipMemcpyDeviceToDevice, s));
    if (npad > n) AddDiag(Vp.Slice(n + n * npad, Vp.n - (n + n * npad)), npad - n, npad, 1.0, nullptr);
  } else {
    AddDiag(Vp, npad, npad, 1.0, nullptr);
  }
  DVec G = DVec::Empty(nsplit * h * kJN * kJN, dt), J = DVec::Empty(h * kJN * kJN, dt);
  // round-robin: position layout [top_0 bot_0 top_1 bot_1 ...]; after a step
  //   new_top[0] = top[0], new_top[1] = bot[0], new_top[k] = top[k-1] (k >= 2),
  //   new_bot[k] = bot[k+1] (k < h-1), new_bot[h-1] = top[h-1]
  std::vector<int32_t> dst(nb);
  if (h == 1) {
    dst[0] = 0;
    dst[1] = 1;
  } else {
    dst[0] = 0;                                           // top[0] stays
    dst[1] = 2 * 1;                                       // bot[0] -> top[1]
    for (int64_t k = 1; k < h - 1; ++k) dst[2 * k] = static_cast<int32_t>(2 * (k + 1));  // top[k] -> top[k+1]
    dst[2 * (h - 1)] = static_cast<int32_t>(2 * (h - 1) + 1);                            // top[h-1] -> bot[h-1]
    for (int64_t k = 1; k < h; ++k) dst[2 * k + 1] = static_cast<int32_t>(2 * (k - 1) + 1);  // bot[k] -> bot[k-1]
  }
  auto dst_buf = rt.Alloc(nb * sizeof(int32_t));
  EPS_HIP(hipMemcpyAsync(dst_buf->p, dst.data(), nb * sizeof(int32_t), hipMemcpyHostToDevice, s));
  EPS_HIP(hipStreamSynchronize(s));
  const int32_t* dst_dev = static_cast<const int32_t*>(dst_buf->p);
  auto flag_buf = rt.Alloc(sizeof(unsigned int));
  unsigned int* offmax = static_cast<unsigned int*>(flag_buf->p);
  std::vector<int32_t> block_at(nb);  // original block id at each position
  for (int64_t i = 0; i < nb; ++i) block_at[i] = static_cast<int32_t>(i);
  const int64_t steps = nb - 1;
  // matrix-core path: the panels stay where they are; the tournament is a table of the panel ids
  // at the positions [top_0 bot_0 top_1 bot_1 ...] of every step (after nb - 1 steps the
  // arrangement is the initial one again, so one table serves every sweep)
  std::shared_ptr<Buffer> tab_buf, skip_buf;
  const int32_t* tab_dev = nullptr;
  int32_t* skip_dev = nullptr;
  if constexpr (kMfma) {
    std::vector<int32_t> tab(static_cast<size_t>(steps * nb)), at(nb), next(nb);
    for (int64_t i = 0; i < nb; ++i) at[i] = static_cast<int32_t>(i);
    for (int64_t st = 0; st < steps; ++st) {
      std::copy(at.begin(), at.end(), tab.begin() + st * nb);
      for (int64_t i = 0; i < nb; ++i) next[dst[i]] = at[i];
      at.swap(next);
    }
    for (int64_t i = 0; i < nb; ++i) EPS_CHECK(at[i] == static_cast<int32_t>(i));
    tab_buf = rt.Alloc(tab.size() * sizeof(int32_t));
    EPS_HIP(hipMemcpyAsync(tab_buf->p, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    EPS_HIP(hipStreamSynchronize(s));
    tab_dev = static_cast<const int32_t*>(tab_buf->p);
    skip_buf = rt.Alloc(static_cast<size_t>(h) * sizeof(int32_t));
    skip_dev = static_cast<int32_t*>(skip_buf->p);
  }
  static const char* inner_env = std::getenv("EPSILON_HIP_SVD_INNER");  // tuning knob
  // one inner sweep per step: measured 7-18 % faster than two, cold and warm-started, at n = 2048
  // and 4096 (three is slower still); the outer iteration absorbs what an inner sweep leaves
  const int inner = inner_env && std::atoi(inner_env) > 0 ? std::atoi(inner_env) : 1;
  int sweeps = 0;
  float prev_mx = 1e30f;
  for (; sweeps < max_sweeps; ++sweeps) {
    EPS_HIP(hipMemsetAsync(offmax, 0, sizeof(unsigned int), s));
    for (int64_t step = 0; step < steps; ++step) {
      // partial Grams in one launch: batch index = split * h + pair, split sp covers rows
      // [sp * kchunk, (sp + 1) * kchunk) of the panels (the last chunk is padded with zero rows)
      if constexpr (kMfma) {
        const dim3 gg(static_cast<unsigned>(nsplit), static_cast<unsigned>(h));
        hipLaunchKernelGGL(PanelGramMfmaKernel, gg, dim3(kBlock), 0, s, Wp.as<float>(), mp, kchunk, h,
                           G.as<float>(), tab_dev + step * nb);
      } else if (mp >= 3072) {
        const dim3 gg(static_cast<unsigned>(nsplit), static_cast<unsigned>(h));
        hipLaunchKernelGGL(PanelGramKernel<T>, gg, dim3(kBlock), 0, s, Wp.as<T>(), mp, kchunk, h, G.as<T>());
      } else {
        GemmBatched(true, false, kJN, kJN, kchunk, 1.0, Wp, mp, kJN * mp, Wp, mp, kJN * mp, 0.0, G,
                    kJN, kJN * kJN, h, false, nsplit, kchunk, kchunk);
      }
      if (comm) comm->AllReduceSum(G);
      if constexpr (kMfma) {
        hipLaunchKernelGGL(PairEigFastKernel, dim3(static_cast<unsigned>(h)), dim3(kBlock), 0, s,
                           G.as<float>(), static_cast<int>(nsplit), h * kJN * kJN, J.as<float>(), inner,
                           static_cast<float>(tol), static_cast<float>(done_tol), offmax, skip_dev);
      } else {
        hipLaunchKernelGGL(PairEigKernel<T>, dim3(static_cast<unsigned>(h)), dim3(kBlock), 0, s,
                           G.as<T>(), static_cast<int>(nsplit), h * kJN * kJN, J.as<T>(), inner, tol,
                           done_tol, offmax);
      }
      if constexpr (kMfma) {
        const dim3 gw(static_cast<unsigned>((mp / 32 + 4 * kUpdRB - 1) / (4 * kUpdRB)), static_cast<unsigned>(h));
        hipLaunchKernelGGL(PanelUpdateMfmaKernel, gw, dim3(kBlock), 0, s, Wp.as<float>(), mp, J.as<float>(),
                           tab_dev + step * nb, skip_dev);
        if (keep_v) {
          const dim3 gv(static_cast<unsigned>((npad / 32 + 4 * kUpdRB - 1) / (4 * kUpdRB)), static_cast<unsigned>(h));
          hipLaunchKernelGGL(PanelUpdateMfmaKernel, gv, dim3(kBlock), 0, s, Vp.as<float>(), npad, J.as<float>(),
                             tab_dev + step * nb, skip_dev);
        }
      } else if (mp >= 3072) {
        // P <- P J with the panels' move folded in (Wt / Vt receive the new layout, then swap);
        // a thread per row needs thousands of rows to fill the chip (n = 2048: slower than the
        // batched GEMMs, n = 4096: 25 % faster per sweep)
        const dim3 gw(static_cast<unsigned>((mp + kBlock - 1) / kBlock), static_cast<unsigned>(h));
        hipLaunchKernelGGL(PanelUpdateKernel<T>, gw, dim3(kBlock), 0, s, Wt.as<T>(), Wp.as<T>(), mp,
                           J.as<T>(), dst_dev);
        const dim3 gv(static_cast<unsigned>((npad + kBlock - 1) / kBlock), static_cast<unsigned>(h));
        hipLaunchKernelGGL(PanelUpdateKernel<T>, gv, dim3(kBlock), 0, s, Vt.as<T>(), Vp.as<T>(), npad,
                           J.as<T>(), dst_dev);
        std::swap(Wp, Wt);
        std::swap(Vp, Vt);
      } else {
        GemmBatched(false, false, mp, kJN, kJN, 1.0, Wp, mp, kJN * mp, J, kJN, kJN * kJN, 0.0, Wt,
                    mp, kJN * mp, h);
        GemmBatched(false, false, npad, kJN, kJN, 1.0, Vp, npad, kJN * npad, J, kJN, kJN * kJN, 0.0,
                    Vt, npad, kJN * npad, h);
        const dim3 gw(static_cast<unsigned>(std::min<int64_t>(64, (mp * kJB + kBlock - 1) / kBlock)),
                      static_cast<unsigned>(nb));
        hipLaunchKernelGGL(PermuteBlocksKernel<T>, gw, dim3(kBlock), 0, s, Wp.as<T>(), Wt.as<T>(), mp,
                           dst_dev);
        const dim3 gv(static_cast<unsigned>(std::min<int64_t>(64, (npad * kJB + kBlock - 1) / kBlock)),
                      static_cast<unsigned>(nb));
        hipLaunchKernelGGL(PermuteBlocksKernel<T>, gv, dim3(kBlock), 0, s, Vp.as<T>(), Vt.as<T>(),
                           npad, dst_dev);
      }
      if constexpr (!kMfma) {  // (the panels of the matrix-core path never move)
        std::vector<int32_t> next(nb);
        for (int64_t i = 0; i < nb; ++i) next[dst[i]] = block_at[i];
        block_at.swap(next);
      }
    }
    unsigned int bits = 0;
    EPS_HIP(hipMemcpyAsync(&bits, offmax, sizeof(bits), hipMemcpyDeviceToHost, s));
    EPS_HIP(hipStreamSynchronize(s));
    float mx;
    std::memcpy(&mx, &bits, sizeof(mx));
    if (std::getenv("EPSILON_HIP_SVD_VERBOSE"))
      std::fprintf(stderr, "block jacobi sweep %d: max |cos| %.3e\n", sweeps, static_cast<double>(mx));
    // converged, or stagnating at the noise floor of the Gram products (fp32: ~1e-5)
    if (mx <= stop_tol || mx <= last_below || (sweeps >= 3 && mx < 1e-3f && mx >= 0.5f * prev_mx)) {
      ++sweeps;
      break;
    }
    prev_mx = mx;
  }
  // the n real columns, wherever they ended up (padding columns are exactly zero and never mix)
  std::vector<int32_t> cols;
  cols.reserve(n);
  for (int64_t pos = 0; pos < nb; ++pos)
    for (int64_t c = 0; c < kJB; ++c)
      if (static_cast<int64_t>(block_at[pos]) * kJB + c < n) cols.push_back(static_cast<int32_t>(pos * kJB + c));
  EPS_CHECK(static_cast<int64_t>(cols.size()) == n);
  auto col_buf = rt.Alloc(n * sizeof(int32_t));
  EPS_HIP(hipMemcpyAsync(col_buf->p, cols.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  EPS_HIP(hipStreamSynchronize(s));
  const int32_t* col_dev = static_cast<const int32_t*>(col_buf->p);
  hipLaunchKernelGGL(GatherColsKernel<T>, dim3(static_cast<unsigned>(n)), dim3(kBlock), 0, s,
                     W.as<T>(), m, Wp.as<T>(), mp, col_dev);
  if (keep_v)
    hipLaunchKernelGGL(GatherColsKernel<T>, dim3(static_cast<unsigned>(n)), dim3(kBlock), 0, s,
                       V.as<T>(), n, Vp.as<T>(), npad, col_dev);
  EPS_HIP(hipGetLastError());
  return sweeps;
}

}  // namespace

int BlockJacobiSvd(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps, bool warm,
                   bool row_sharded) {
  EPS_CHECK(W.n >= m * n && V.n >= n * n && W.dt == V.dt);
  if (n == 0 || (m == 0 && !row_sharded)) return 0;  // a rank without rows still joins the collectives
  ProfScope prof("block_jacobi_svd", m, n);
  return W.dt == F32 ? BlockJacobiImpl<float>(W, m, n, V, max_sweeps, warm, row_sharded)
                     : BlockJacobiImpl<double>(W, m, n, V, max_sweeps, warm, row_sharded);
}

// Can the decomposition of an m x n matrix of this type run without accumulating V?  (The fp32
// block form on the matrix cores, where JacobiSvd would route it; not with the scalar / on-chip
// kernels, for which V costs next to nothing.)
bool JacobiSvdCanSkipV(int64_t m, int64_t n, DType dt) {
  if (dt != F32 || m == 0) return false;
  // the one-launch on-chip kernel (robust PCA at the reference's published size, n = 100)
  const char* form = std::getenv("EPSILON_HIP_SVD");  // "scalar" | "block": a forced form keeps its old routes
  const bool on_chip = m <= 128 && n <= 128 && n >= 2 && SmallFastEnabled() && !(form && form[0] != 0);
  if (!on_chip && n < 512) return false;
  if (const char* env = std::getenv("EPSILON_HIP_SVD")) {
    if (env[0] == 's') return false;
  }
  if (const char* e = std::getenv("EPSILON_HIP_SVD_MFMA")) {
    if (e[0] == '0') return false;
  }
  if (const char* e = std::getenv("EPSILON_HIP_SVD_NO_V")) {
    if (e[0] == '0') return false;
  }
  return true;
}

// One-sided Jacobi on the columns of W with the rotations applied to W only (see BlockJacobiImpl,
// keep_v): on return the columns of W are orthogonal - the left singular vectors scaled by the
// singular values.  Requires JacobiSvdCanSkipV(m, n, W.dt).
int JacobiSvdNoV(const DVec& W, int64_t m, int64_t n, int max_sweeps) {
  EPS_CHECK(W.n >= m * n && JacobiSvdCanSkipV(m, n, W.dt));
  if (m <= 128 && n <= 128) {
    ProfScope prof("jacobi_svd_no_v_on_chip", m, n);
    static const bool verbose = std::getenv("EPSILON_HIP_SVD_VERBOSE") != nullptr;
    int done = -1;  // unknown unless asked for: fetching the count would stall the host once per prox
    EPS_CHECK(SmallJacobiSvdFast(W, m, n, DVec(), max_sweeps, 2e-7, false, verbose ? &done : nullptr, false));
    if (verbose)
      std::fprintf(stderr, "[svd] on-chip (no V) %lld x %lld: %d sweeps\n", static_cast<long long>(m),
                   static_cast<long long>(n), done);
    return done;
  }
  ProfScope prof("block_jacobi_svd_no_v", m, n);
  // de Rijk's ordering - columns by decreasing norm before the sweeps (the caller does not care in
  // which order the orthogonal columns come back) - measured at n = 1e4 on the reference's
  // robust-PCA matrix: 18 sweeps with and without (gpurun_out r3k), so it is off;
  // EPSILON_HIP_SVD_SORT=1 switches it on.
  static const bool sort_cols = [] {
    const char* e = std::getenv("EPSILON_HIP_SVD_SORT");
    return e && e[0] == '1';
  }();
  if (sort_cols && n > 1) {
    Runtime& rt = Runtime::Get();
    hipStream_t s = rt.stream();
    DVec sig = DVec::Empty(n, W.dt);
    ColNorms(W, m, n, sig, false);
    const std::vector<double> h = sig.ToHost();
    std::vector<int32_t> perm(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; ++i) perm[static_cast<size_t>(i)] = static_cast<int32_t>(i);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return h[a] > h[b]; });
    auto pb = rt.Alloc(static_cast<size_t>(n) * sizeof(int32_t));
    EPS_HIP(hipMemcpyAsync(pb->p, perm.data(), static_cast<size_t>(n) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DVec Ws = DVec::Empty(m * n, W.dt);
    hipLaunchKernelGGL(GatherColsKernel<float>, dim3(static_cast<unsigned>(n)), dim3(kBlock), 0, s, Ws.as<float>(), m,
                       W.as<float>(), m, static_cast<const int32_t*>(pb->p));
    EPS_HIP(hipStreamSynchronize(s));  // (perm is a host vector)
    Copy(W.Slice(0, m * n), Ws);
  }
  return BlockJacobiImpl<float>(W, m, n, DVec(), max_sweeps, true, false, false);
}

int JacobiSvd(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps, bool warm,
              bool row_sharded) {
  EPS_CHECK(W.n >= m * n && V.n >= n * n && W.dt == V.dt);
  if (n == 0) return 0;
  if (row_sharded) return BlockJacobiSvd(W, m, n, V, max_sweeps, warm, true);  // the sharded form
  if (m == 0) return 0;
  {
    const char* env = std::getenv("EPSILON_HIP_SVD");  // "scalar" | "block" (read per call)
    const bool force_scalar = env && env[0] == 's', force_block = env && env[0] == 'b';
    if (!force_scalar && (force_block || n >= 1536))
      return BlockJacobiSvd(W, m, n, V, max_sweeps, warm, false);
    // fp32: with its kernels on the matrix cores the block form also wins below 1536 columns,
    // wherever the one-launch on-chip kernel does not apply (measured cold, scalar steps -> block:
    // n = 200 9.8 -> 6.7 ms, 384 22 -> 12 ms, 768 76 -> 29 ms, 1400 253 -> 59 ms)
    if (!force_scalar && W.dt == F32 &&
        static_cast<size_t>(m * n + n * n) * sizeof(float) > kSmallLdsBytes)
      return BlockJacobiSvd(W, m, n, V, max_sweeps, warm, false);
  }
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("jacobi_svd", m, n);
  const bool f32 = W.dt == F32;
  const double tol = f32 ? 2e-7 : 1e-15;
  {
    const char* env = std::getenv("EPSILON_HIP_SVD");
    int done = 0;
    if (!(env && env[0] == 's') && n >= 2 &&
        (f32 ? SmallJacobiSvd<float>(W, m, n, V, max_sweeps, tol, warm, &done)
             : SmallJacobiSvd<double>(W, m, n, V, max_sweeps, tol, warm, &done))) {
      if (std::getenv("EPSILON_HIP_SVD_VERBOSE"))
        std::fprintf(stderr, "[svd] on-chip %lld x %lld: %d sweeps\n", static_cast<long long>(m),
                     static_cast<long long>(n), done);
      return done;
    }
  }
  const int64_t total = n * n;
  if (warm) {
    // V already holds the orthogonal start
  } else if (f32) {
    hipLaunchKernelGGL(EyeKernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, V.as<float>(), n);
  } else {
    hipLaunchKernelGGL(EyeKernel<double>, dim3((total + 255) / 256), dim3(256), 0, s, V.as<double>(), n);
  }
  if (n == 1) return 0;
  const int64_t npad = n + (n & 1);
  auto flag_buf = rt.Alloc(sizeof(int));
  int* flag = static_cast<int*>(flag_buf->p);
  int sweeps = 0;
  for (; sweeps < max_sweeps; ++sweeps) {
    EPS_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
    for (int64_t step = 0; step < npad - 1; ++step) {
      if (f32)
        hipLaunchKernelGGL(JacobiStepKernel<float>, dim3(npad / 2), dim3(kBlock), 0, s,
                           W.as<float>(), m, n, V.as<float>(), npad, step, tol, flag);
      else
        hipLaunchKernelGGL(JacobiStepKernel<double>, dim3(npad / 2), dim3(kBlock), 0, s,
                           W.as<double>(), m, n, V.as<double>(), npad, step, tol, flag);
    }
    int h = 0;
    EPS_HIP(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    EPS_HIP(hipStreamSynchronize(s));
    if (h == 0) break;
  }
  if (std::getenv("EPSILON_HIP_SVD_VERBOSE"))
    std::fprintf(stderr, "[svd] step launches %lld x %lld: %d sweeps\n", static_cast<long long>(m),
                 static_cast<long long>(n), sweeps);
  return sweeps;
}

void ColNorms(const DVec& W, int64_t m, int64_t n, const DVec& sigma, bool row_sharded) {
  EPS_CHECK(W.n >= m * n && sigma.n == n && W.dt == sigma.dt);
  if (n == 0) return;
  hipStream_t s = Runtime::Get().stream();
  const int sq = row_sharded ? 1 : 0;  // rows on other ranks: sum the squares first
  if (W.dt == F32)
    hipLaunchKernelGGL(ColNormKernel<float>, dim3(n), dim3(kBlock), 0, s, W.as<float>(), m, n, sigma.as<float>(), sq);
  else
    hipLaunchKernelGGL(ColNormKernel<double>, dim3(n), dim3(kBlock), 0, s, W.as<double>(), m, n, sigma.as<double>(), sq);
  if (row_sharded) {
    Runtime::Get().comm()->AllReduceSum(sigma);
    const unsigned grid = static_cast<unsigned>((n + 255) / 256);
    if (W.dt == F32) hipLaunchKernelGGL(SqrtInPlaceKernel<float>, dim3(grid), dim3(256), 0, s, sigma.as<float>(), n);
    else hipLaunchKernelGGL(SqrtInPlaceKernel<double>, dim3(grid), dim3(256), 0, s, sigma.as<double>(), n);
  }
}

void ColScaleByRatio(const DVec& W, int64_t m, int64_t n, const DVec& sigma, const DVec& xt) {
  EPS_CHECK(W.n >= m * n && sigma.n == n && xt.n == n && W.dt == sigma.dt && W.dt == xt.dt);
  if (n == 0) return;
  hipStream_t s = Runtime::Get().stream();
  if (W.dt == F32)
    hipLaunchKernelGGL(ColScaleKernel<float>, dim3(n), dim3(kBlock), 0, s, W.as<float>(), m, n, sigma.as<float>(), xt.as<float>());
  else
    hipLaunchKernelGGL(ColScaleKernel<double>, dim3(n), dim3(kBlock), 0, s, W.as<double>(), m, n, sigma.as<double>(), xt.as<double>());
}

}  // namespace k
}  // namespace eps

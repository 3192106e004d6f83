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

// sigma[j] = ||W[:, j]||_2
template <class T>
__global__ __launch_bounds__(kBlock) void ColNormKernel(const T* W, int64_t m, int64_t n,
                                                        T* sigma) {
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
    sigma[j] = static_cast<T>(sqrt(t));
  }
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

}  // namespace

int JacobiSvd(const DVec& W, int64_t m, int64_t n, const DVec& V, int max_sweeps) {
  EPS_CHECK(W.n >= m * n && V.n >= n * n && W.dt == V.dt);
  if (m == 0 || n == 0) return 0;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("jacobi_svd", m, n);
  const bool f32 = W.dt == F32;
  const double tol = f32 ? 2e-7 : 1e-15;
  const int64_t total = n * n;
  if (f32) hipLaunchKernelGGL(EyeKernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, V.as<float>(), n);
  else hipLaunchKernelGGL(EyeKernel<double>, dim3((total + 255) / 256), dim3(256), 0, s, V.as<double>(), n);
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
  return sweeps;
}

void ColNorms(const DVec& W, int64_t m, int64_t n, const DVec& sigma) {
  EPS_CHECK(W.n >= m * n && sigma.n == n && W.dt == sigma.dt);
  if (n == 0) return;
  hipStream_t s = Runtime::Get().stream();
  if (W.dt == F32)
    hipLaunchKernelGGL(ColNormKernel<float>, dim3(n), dim3(kBlock), 0, s, W.as<float>(), m, n, sigma.as<float>());
  else
    hipLaunchKernelGGL(ColNormKernel<double>, dim3(n), dim3(kBlock), 0, s, W.as<double>(), m, n, sigma.as<double>());
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

// K6 / K9 / K12: proximal-operator kernels.
//
//  ScaledZone  - two-sided soft threshold with dead zone, the one kernel behind NORM_1,
//                SUM_DEADZONE, SUM_HINGE and SUM_QUANTILE
//                (reference src/epsilon/prox/scaled_zone.cc:78-104).  Branch order is kept
//                exactly as in the reference so that the selected branch - and therefore the
//                result for a given input - is bit-identical.
//  MaxZero     - projection onto R+ (reference prox/non_negative.cc:8), bit-exact.
//  Norm2Shrink - group soft threshold (reference prox/norm_2.cc:11-16); the norm is read from
//                a device slot so no host sync sits inside the ADMM sweep.
//  Tv1d        - exact 1-D total-variation prox (reference prox/total_variation_1d.cc:21 ->
//                glmgen tf_dp, Johnson's dynamic program).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;

inline int GridFor(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return static_cast<int>(g);
}

// 16 bytes per lane and access: V = 4 floats / 2 doubles.
template <class T> struct VecOf;
template <> struct VecOf<float> {
  static constexpr int V = 4;
  typedef float4 type;
  __device__ static void unpack(const float4& p, float (&e)[4]) { e[0] = p.x; e[1] = p.y; e[2] = p.z; e[3] = p.w; }
  __device__ static float4 pack(const float (&e)[4]) { return make_float4(e[0], e[1], e[2], e[3]); }
};
template <> struct VecOf<double> {
  static constexpr int V = 2;
  typedef double2 type;
  __device__ static void unpack(const double2& p, double (&e)[2]) { e[0] = p.x; e[1] = p.y; }
  __device__ static double2 pack(const double (&e)[2]) { return make_double2(e[0], e[1]); }
};

// reference prox/scaled_zone.cc:90-101, one element: the branch order is the reference's, so
// the branch taken - and with it the result - is the same for the same operands
template <class T> __device__ inline T ScaledZoneElem(T vi, T C, T M, T lam, T alpha, T beta) {
  const T xi = vi - C;
  if (fabs(xi) <= M) return xi;
  if (xi > M + lam * alpha) return xi - lam * alpha;
  if (xi < -M - lam * beta) return xi + lam * beta;
  if (xi > T(0)) return M;
  return -M;
}

// Vector form: every array is read / written 16 bytes per lane (the caller guarantees the
// alignment and period == 0); UNROLL independent accesses per thread are in flight.  PV says
// which of lam / alpha / beta are per-element vectors (bit 0 / 1 / 2).
template <class T, int PV>
__global__ __launch_bounds__(kBlock) void ScaledZoneVecKernel(
    T* __restrict__ x, const T* __restrict__ v, int64_t nvec, T lam_s, T alpha_s, T beta_s, T M, T C,
    const T* __restrict__ lam_v, const T* __restrict__ alpha_v, const T* __restrict__ beta_v) {
  using VT = typename VecOf<T>::type;
  constexpr int V = VecOf<T>::V;
  constexpr int UNROLL = 4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t q0 = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x; q0 < nvec; q0 += stride * UNROLL) {
    VT pv[UNROLL], pl[UNROLL], pa[UNROLL], pb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t q = q0 + u * stride;
      if (q < nvec) {
        pv[u] = reinterpret_cast<const VT*>(v)[q];
        if (PV & 1) pl[u] = reinterpret_cast<const VT*>(lam_v)[q];
        if (PV & 2) pa[u] = reinterpret_cast<const VT*>(alpha_v)[q];
        if (PV & 4) pb[u] = reinterpret_cast<const VT*>(beta_v)[q];
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t q = q0 + u * stride;
      if (q < nvec) {
        T ev[V], el[V], ea[V], eb[V], eo[V];
        VecOf<T>::unpack(pv[u], ev);
        if (PV & 1) VecOf<T>::unpack(pl[u], el);
        if (PV & 2) VecOf<T>::unpack(pa[u], ea);
        if (PV & 4) VecOf<T>::unpack(pb[u], eb);
#pragma unroll
        for (int e = 0; e < V; ++e)
          eo[e] = ScaledZoneElem<T>(ev[e], C, M, (PV & 1) ? el[e] : lam_s, (PV & 2) ? ea[e] : alpha_s,
                                    (PV & 4) ? eb[e] : beta_s);
        reinterpret_cast<VT*>(x)[q] = VecOf<T>::pack(eo);
      }
    }
  }
}

// General form (tails, unaligned buffers, periodic parameter vectors).  The periodic index is
// advanced by the stride modulo the period instead of a 64-bit division per element.
template <class T>
__global__ __launch_bounds__(kBlock) void ScaledZoneKernel(
    T* x, const T* v, int64_t i0, int64_t n, T lam_s, T alpha_s, T beta_s, T M, T C, const T* lam_v,
    const T* alpha_v, const T* beta_v, int64_t period) {
  const int64_t tid = i0 + blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  int64_t pi = period > 0 ? tid % period : 0;
  const int64_t pstep = period > 0 ? stride % period : 0;
  for (int64_t i = tid; i < n; i += stride) {
    const int64_t k = period > 0 ? pi : i;
    const T lam = lam_v ? lam_v[k] : lam_s;
    const T alpha = alpha_v ? alpha_v[k] : alpha_s;
    const T beta = beta_v ? beta_v[k] : beta_s;
    x[i] = ScaledZoneElem<T>(v[i], C, M, lam, alpha, beta);
    if (period > 0) {
      pi += pstep;
      if (pi >= period) pi -= period;
    }
  }
}

template <class T>
__global__ __launch_bounds__(kBlock) void MaxZeroKernel(T* __restrict__ x, const T* __restrict__ v,
                                                        int64_t n, bool vec) {
  using VT = typename VecOf<T>::type;
  constexpr int V = VecOf<T>::V;
  const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t nvec = vec ? n / V : 0;
  constexpr int UNROLL = 4;
  for (int64_t q0 = tid; q0 < nvec; q0 += stride * UNROLL) {
    VT pv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (q0 + u * stride < nvec) pv[u] = reinterpret_cast<const VT*>(v)[q0 + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (q0 + u * stride < nvec) {
        T e[V];
        VecOf<T>::unpack(pv[u], e);
#pragma unroll
        for (int k = 0; k < V; ++k) e[k] = e[k] > T(0) ? e[k] : T(0);  // non_negative.cc:8
        reinterpret_cast<VT*>(x)[q0 + u * stride] = VecOf<T>::pack(e);
      }
    }
  }
  for (int64_t i = nvec * V + tid; i < n; i += stride) {
    const T vi = v[i];
    x[i] = vi > T(0) ? vi : T(0);
  }
}

template <class T>
__global__ __launch_bounds__(kBlock) void Norm2ShrinkKernel(T* __restrict__ x, const T* __restrict__ v,
                                                            int64_t n, double lam,
                                                            const double* normsq, bool vec) {
  using VT = typename VecOf<T>::type;
  constexpr int V = VecOf<T>::V;
  const double nv = sqrt(*normsq);
  const T scale = (nv >= lam) ? static_cast<T>(1.0 - lam / nv) : T(0);
  const bool zero = !(nv >= lam);
  const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t nvec = vec ? n / V : 0;
  for (int64_t q = tid; q < nvec; q += stride) {
    T e[V];
    VecOf<T>::unpack(reinterpret_cast<const VT*>(v)[q], e);
#pragma unroll
    for (int k = 0; k < V; ++k) e[k] = zero ? T(0) : scale * e[k];
    reinterpret_cast<VT*>(x)[q] = VecOf<T>::pack(e);
  }
  for (int64_t i = nvec * V + tid; i < n; i += stride) x[i] = zero ? T(0) : scale * v[i];
}

// ---- exact 1-D TV prox ---------------------------------------------------------------------
// Sequential dynamic program (one lane); correctness path for small n.  The knot buffers live
// in a global workspace `ws` of 8n doubles laid out as x|a|b (2n each) and tm|tp (n each).
template <class T>
__global__ void Tv1dSerialKernel(T* beta, const T* y, int64_t n, double lam, double* ws) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (n == 0) return;
  if (n == 1 || lam == 0) {
    for (int64_t i = 0; i < n; ++i) beta[i] = y[i];
    return;
  }
  double* x = ws;
  double* a = ws + 2 * n;
  double* b = ws + 4 * n;
  double* tm = ws + 6 * n;
  double* tp = ws + 7 * n;
  double y0 = static_cast<double>(y[0]), y1 = static_cast<double>(y[1]);
  tm[0] = -lam + y0;
  tp[0] = lam + y0;
  int64_t lo_i = n - 1, hi_i = n;
  x[lo_i] = tm[0];
  x[hi_i] = tp[0];
  a[lo_i] = 1;
  b[lo_i] = -y0 + lam;
  a[hi_i] = -1;
  b[hi_i] = y0 + lam;
  double afirst = 1, bfirst = -lam - y1, alast = -1, blast = -lam + y1;
  for (int64_t k = 1; k < n - 1; ++k) {
    double alo = afirst, blo = bfirst;
    int64_t lo = lo_i;
    for (; lo <= hi_i; ++lo) {
      if (alo * x[lo] + blo > -lam) break;
      alo += a[lo];
      blo += b[lo];
    }
    tm[k] = (-lam - blo) / alo;
    lo_i = lo - 1;
    x[lo_i] = tm[k];
    double ahi = alast, bhi = blast;
    int64_t hi = hi_i;
    for (; hi >= lo_i; --hi) {
      if (-ahi * x[hi] - bhi < lam) break;
      ahi += a[hi];
      bhi += b[hi];
    }
    tp[k] = (lam + bhi) / (-ahi);
    hi_i = hi + 1;
    x[hi_i] = tp[k];
    a[lo_i] = alo;
    b[lo_i] = blo + lam;
    a[hi_i] = ahi;
    b[hi_i] = bhi + lam;
    const double yk1 = static_cast<double>(y[k + 1]);
    afirst = 1;
    bfirst = -lam - yk1;
    alast = -1;
    blast = -lam + yk1;
  }
  double alo = afirst, blo = bfirst;
  for (int64_t lo = lo_i; lo <= hi_i; ++lo) {
    if (alo * x[lo] + blo > 0) break;
    alo += a[lo];
    blo += b[lo];
  }
  double bn = -blo / alo;
  beta[n - 1] = static_cast<T>(bn);
  for (int64_t k = n - 2; k >= 0; --k) {
    if (bn > tp[k]) bn = tp[k];
    else if (bn < tm[k]) bn = tm[k];
    beta[k] = static_cast<T>(bn);
  }
}

}  // namespace

namespace {

inline bool Aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

template <class T> void LaunchScaledZone(const DVec& x, const DVec& v, const ScaledZoneArgs& g) {
  using VTr = VecOf<T>;
  const int64_t n = x.n;
  hipStream_t s = Runtime::Get().stream();
  const T* lv = g.lam_vec ? g.lam_vec->as<T>() : nullptr;
  const T* av = g.alpha_vec ? g.alpha_vec->as<T>() : nullptr;
  const T* bv = g.beta_vec ? g.beta_vec->as<T>() : nullptr;
  int64_t done = 0;
  const bool vec_ok = g.period == 0 && Aligned16(x.data()) && Aligned16(v.data()) &&
                      (!lv || Aligned16(lv)) && (!av || Aligned16(av)) && (!bv || Aligned16(bv));
  const int64_t nvec = vec_ok ? n / VTr::V : 0;
  if (nvec > 0) {
    const int pvm = (lv ? 1 : 0) | (av ? 2 : 0) | (bv ? 4 : 0);
    int64_t grid = (nvec + kBlock * 4 - 1) / (kBlock * 4);
    if (grid > 4096) grid = 4096;
#define EPS_SZ_CASE(PVV)                                                                          \
  case PVV:                                                                                       \
    hipLaunchKernelGGL((ScaledZoneVecKernel<T, PVV>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, s, \
                       x.as<T>(), v.as<T>(), nvec, T(g.lam), T(g.alpha), T(g.beta), T(g.M), T(g.C), lv, av, bv); \
    break;
    switch (pvm) {
      EPS_SZ_CASE(0) EPS_SZ_CASE(1) EPS_SZ_CASE(2) EPS_SZ_CASE(3) EPS_SZ_CASE(4) EPS_SZ_CASE(5)
      EPS_SZ_CASE(6) EPS_SZ_CASE(7)
    }
#undef EPS_SZ_CASE
    done = nvec * VTr::V;
  }
  if (done < n)
    hipLaunchKernelGGL(ScaledZoneKernel<T>, dim3(GridFor(n - done)), dim3(kBlock), 0, s, x.as<T>(),
                       v.as<T>(), done, n, T(g.lam), T(g.alpha), T(g.beta), T(g.M), T(g.C), lv, av, bv,
                       g.period);
}

}  // namespace

void ScaledZone(const DVec& x, const DVec& v, const ScaledZoneArgs& g) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  const int64_t n = x.n;
  if (n == 0) return;
  const int64_t need = g.period > 0 ? g.period : n;
  if (g.lam_vec) EPS_CHECK(g.lam_vec->n >= need && g.lam_vec->dt == x.dt);
  if (g.alpha_vec) EPS_CHECK(g.alpha_vec->n >= need && g.alpha_vec->dt == x.dt);
  if (g.beta_vec) EPS_CHECK(g.beta_vec->n >= need && g.beta_vec->dt == x.dt);
  ProfScope prof("scaled_zone", n);
  if (x.dt == F32) LaunchScaledZone<float>(x, v, g);
  else LaunchScaledZone<double>(x, v, g);
}

void MaxZero(const DVec& x, const DVec& v) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  if (x.n == 0) return;
  hipStream_t s = Runtime::Get().stream();
  ProfScope prof("max_zero", x.n);
  const bool vec = Aligned16(x.data()) && Aligned16(v.data());
  const int64_t per = static_cast<int64_t>(kBlock) * 16;
  int64_t grid = (x.n + per - 1) / per;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  if (x.dt == F32)
    hipLaunchKernelGGL(MaxZeroKernel<float>, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, s,
                       x.as<float>(), v.as<float>(), x.n, vec);
  else
    hipLaunchKernelGGL(MaxZeroKernel<double>, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, s,
                       x.as<double>(), v.as<double>(), x.n, vec);
}

void Norm2Shrink(const DVec& x, const DVec& v, double lam, const double* normsq) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  if (x.n == 0) return;
  hipStream_t s = Runtime::Get().stream();
  const bool vec = Aligned16(x.data()) && Aligned16(v.data());
  if (x.dt == F32)
    hipLaunchKernelGGL(Norm2ShrinkKernel<float>, dim3(GridFor(x.n)), dim3(kBlock), 0, s,
                       x.as<float>(), v.as<float>(), x.n, lam, normsq, vec);
  else
    hipLaunchKernelGGL(Norm2ShrinkKernel<double>, dim3(GridFor(x.n)), dim3(kBlock), 0, s,
                       x.as<double>(), v.as<double>(), x.n, lam, normsq, vec);
}

void Tv1dSerial(const DVec& x, const DVec& v, double lam) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  const int64_t n = x.n;
  if (n == 0) return;
  Runtime& rt = Runtime::Get();
  auto ws = rt.Alloc(static_cast<size_t>(8 * n) * sizeof(double));
  if (x.dt == F32)
    hipLaunchKernelGGL(Tv1dSerialKernel<float>, dim3(1), dim3(64), 0, rt.stream(), x.as<float>(),
                       v.as<float>(), n, lam, static_cast<double*>(ws->p));
  else
    hipLaunchKernelGGL(Tv1dSerialKernel<double>, dim3(1), dim3(64), 0, rt.stream(),
                       x.as<double>(), v.as<double>(), n, lam, static_cast<double*>(ws->p));
}

// ---- epigraph projections -----------------------------------------------------------------------

namespace {

struct Cplx {
  double re, im;
};
__device__ inline Cplx cmul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ inline Cplx csub(Cplx a, Cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ inline Cplx cadd(Cplx a, Cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ inline Cplx cdiv(Cplx a, Cplx b) {
  const double d = b.re * b.re + b.im * b.im;
  return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__device__ inline double cabs(Cplx a) { return sqrt(a.re * a.re + a.im * a.im); }
__device__ inline Cplx cubic(Cplx z, double b, double c, double d) {
  // z^3 + b z^2 + c z + d
  Cplx z2 = cmul(z, z), z3 = cmul(z2, z);
  return {z3.re + b * z2.re + c * z.re + d, z3.im + b * z2.im + c * z.im};
}

// reference prox/newton.cc:293-323 (Durand-Kerner, same start values and tolerances)
__device__ double LargestRealCubicRootDev(double b, double c, double d) {
  const double eps = 1e-12;
  Cplx p{0.4, 0.9};
  Cplx q = cmul(p, p), r = cmul(q, p);
  for (int it = 0; it < 100; ++it) {
    Cplx fp = cubic(p, b, c, d), fq = cubic(q, b, c, d), fr = cubic(r, b, c, d);
    if (cabs(fp) < eps && cabs(fq) < eps && cabs(fr) < eps) break;
    Cplx np = csub(p, cdiv(fp, cmul(csub(p, q), csub(p, r))));
    Cplx nq = csub(q, cdiv(fq, cmul(csub(q, p), csub(q, r))));
    Cplx nr = csub(r, cdiv(fr, cmul(csub(r, p), csub(r, q))));
    p = np;
    q = nq;
    r = nr;
  }
  double m = -1e41;
  if (fabs(p.im) < eps && p.re > m) m = p.re;
  if (fabs(q.im) < eps && q.re > m) m = q.re;
  if (fabs(r.im) < eps && r.re > m) m = r.re;
  return m;
}

template <class T>
__global__ void SumSquareEpiScalarKernel(const T* s, const double* normsq, double* lam_out, T* t) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const double sv = static_cast<double>(*s);
  double lam = LargestRealCubicRootDev(1 + sv, 0.25 + sv, (sv - *normsq) / 4);
  if (lam < 0) lam = 0;
  *lam_out = lam;
  *t = static_cast<T>(sv + lam);
}

template <class T>
__global__ __launch_bounds__(kBlock) void ScaleByLamKernel(T* x, const T* u, int64_t n,
                                                           const double* lam) {
  const T f = static_cast<T>(1.0 / (1.0 + 2.0 * *lam));
  const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  // the reference divides: x = u / (1 + 2 lam)
  const double denom = 1.0 + 2.0 * *lam;
  (void)f;
  for (int64_t i = tid; i < n; i += stride) x[i] = static_cast<T>(static_cast<double>(u[i]) / denom);
}

__device__ inline double WaveSumP(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <class T>
__global__ __launch_bounds__(kBlock) void ZoneKeysKernel(const T* v, int64_t n, double alpha_s,
                                                         double beta_s, const T* alpha_v,
                                                         const T* beta_v, double M, double C,
                                                         double* key, double* w2, double* fval) {
  __shared__ double red[kBlock / 64];
  const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double f = 0;
  for (int64_t i = tid; i < n; i += stride) {
    const double y = static_cast<double>(v[i]) - C;
    const double a = alpha_v ? static_cast<double>(alpha_v[i]) : alpha_s;
    const double b = beta_v ? static_cast<double>(beta_v[i]) : beta_s;
    const double w = y > 0 ? a : b;
    double kk = 0, ww = 0;
    if (fabs(y) > M && w != 0) {  // reference filter, scaled_zone.cc:185-188
      const double ex = fabs(y) - M;
      f += w * ex;
      kk = ex / w;
      ww = w * w;
    }
    key[i] = kk;
    w2[i] = ww;
  }
  f = WaveSumP(f);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < kBlock / 64; ++w) t += red[w];
    atomicAdd(fval, t);
  }
}

__global__ __launch_bounds__(kBlock) void ZoneSumsKernel(int64_t n, const double* key,
                                                         const double* w2, double lam,
                                                         double* sums) {
  __shared__ double red[kBlock / 64][3];
  const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double a = 0, b = 0, c = 0;
  for (int64_t i = tid; i < n; i += stride) {
    const double ww = w2[i], kk = key[i];
    if (ww > 0 && kk > lam) {
      a += ww * kk;
      b += ww;
      c += 1;
    }
  }
  a = WaveSumP(a);
  b = WaveSumP(b);
  c = WaveSumP(c);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][0] = a;
    red[threadIdx.x >> 6][1] = b;
    red[threadIdx.x >> 6][2] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0, tc = 0;
    for (int w = 0; w < kBlock / 64; ++w) {
      ta += red[w][0];
      tb += red[w][1];
      tc += red[w][2];
    }
    atomicAdd(&sums[0], ta);
    atomicAdd(&sums[1], tb);
    atomicAdd(&sums[2], tc);
  }
}

}  // namespace

void SumSquareEpigraph(const DVec& x, const DVec& t, const DVec& u, const DVec& s,
                       const double* normsq, double* lam_scratch) {
  EPS_CHECK(x.n == u.n && x.dt == u.dt && t.n == 1 && s.n == 1 && t.dt == x.dt && s.dt == x.dt);
  hipStream_t st = Runtime::Get().stream();
  if (x.dt == F32) {
    hipLaunchKernelGGL(SumSquareEpiScalarKernel<float>, dim3(1), dim3(64), 0, st, s.as<float>(),
                       normsq, lam_scratch, t.as<float>());
    if (x.n) hipLaunchKernelGGL(ScaleByLamKernel<float>, dim3(GridFor(x.n)), dim3(kBlock), 0, st,
                                x.as<float>(), u.as<float>(), x.n, lam_scratch);
  } else {
    hipLaunchKernelGGL(SumSquareEpiScalarKernel<double>, dim3(1), dim3(64), 0, st, s.as<double>(),
                       normsq, lam_scratch, t.as<double>());
    if (x.n) hipLaunchKernelGGL(ScaleByLamKernel<double>, dim3(GridFor(x.n)), dim3(kBlock), 0, st,
                                x.as<double>(), u.as<double>(), x.n, lam_scratch);
  }
}

void ZoneEpigraphKeys(const DVec& v, double alpha, double beta, const DVec* alpha_vec,
                      const DVec* beta_vec, double M, double C, double* key, double* w2,
                      double* fval) {
  if (v.n == 0) return;
  hipStream_t st = Runtime::Get().stream();
  if (v.dt == F32)
    hipLaunchKernelGGL(ZoneKeysKernel<float>, dim3(GridFor(v.n)), dim3(kBlock), 0, st, v.as<float>(),
                       v.n, alpha, beta, alpha_vec ? alpha_vec->as<float>() : nullptr,
                       beta_vec ? beta_vec->as<float>() : nullptr, M, C, key, w2, fval);
  else
    hipLaunchKernelGGL(ZoneKeysKernel<double>, dim3(GridFor(v.n)), dim3(kBlock), 0, st,
                       v.as<double>(), v.n, alpha, beta,
                       alpha_vec ? alpha_vec->as<double>() : nullptr,
                       beta_vec ? beta_vec->as<double>() : nullptr, M, C, key, w2, fval);
}

void ZoneEpigraphSums(int64_t n, const double* key, const double* w2, double lam, double* sums) {
  if (n == 0) return;
  hipLaunchKernelGGL(ZoneSumsKernel, dim3(GridFor(n)), dim3(kBlock), 0, Runtime::Get().stream(), n,
                     key, w2, lam, sums);
}

}  // namespace k
}  // namespace eps

// Batched ("segmented") and Newton-family proximal kernels for gfx950.
//
// The reference applies these operators one vector at a time on the host, looping over the
// rows / columns of a matrix argument when the function carries an axis
// (reference src/epsilon/prox/vector_prox.cc:150-177), and finds thresholds by std::sort
// (prox/max.cc:18, prox/sum_largest.cc:27) or by damped Newton iterations with a global line
// search (prox/newton.cc:49-237).  Here every slice ("segment") of the argument is solved by a
// group of G lanes of one launch - G = 1 for the short strided rows of a tall matrix (coalesced
// across lanes), up to a whole 256-lane workgroup for long contiguous segments - and the
// sequential algorithms are replaced by ones that need only reductions:
//
//   * thresholds of piecewise-linear equations (MAX, its epigraph, SUM_LARGEST): Newton on the
//     piecewise-linear function, finite and exact, instead of sorting;
//   * separable smooth functions (SUM_EXP, SUM_LOGISTIC, SUM_NEG_ENTR, SUM_INV_POS,
//     SUM_KL_DIV, EXP): the reference's damped Newton run per element (the systems are
//     decoupled; only its line search and stopping test couple them);
//   * LOG_SUM_EXP: x = v - lam*w with w = softmax(x) reduces to ONE scalar unknown c = log Z:
//     w_i = W(lam e^{v_i - c}) / lam (Lambert W), sum_i w_i = 1, convex and decreasing in c;
//   * epigraph projections: the multiplier lam >= 0 of f(prox_{lam f}(v)) = s + lam is found by
//     a safeguarded scalar Newton (the reference's ImplicitNewtonEpigraph form, newton.cc:
//     196-237; its joint Newton for the other functions converges to the same KKT point).
//
// All iterations run in fp64 whatever the storage type; group reductions are butterflies, so
// every lane of a group holds identical bits and control flow stays group-uniform.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;

// ---- group reductions ----------------------------------------------------------------------------

template <int G> __device__ inline double GroupSum(double x) {
  if constexpr (G <= 64) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
  } else {
    __shared__ double red[kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) s += red[w];
    return s;
  }
}

template <int G> __device__ inline double GroupMax(double x) {
  if constexpr (G <= 64) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off, 64));
    return x;
  } else {
    __shared__ double red[kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) s = fmax(s, red[w]);
    return s;
  }
}

struct SegCtx {
  int64_t seg, base, stride, len;
  int lane;
};

template <int G> __device__ inline bool SegInit(const Segs& S, SegCtx* c) {
  constexpr int kGroups = kBlock / G;
  c->lane = threadIdx.x % G;
  c->seg = static_cast<int64_t>(blockIdx.x) * kGroups + threadIdx.x / G;
  if (c->seg >= S.count) return false;
  c->base = c->seg * S.seg_stride;
  c->stride = S.elem_stride;
  c->len = S.len;
  return true;
}

#define SEG_FOR(p) for (int64_t p = c.lane; p < c.len; p += G)
#define SEG_AT(ptr, p) (ptr)[c.base + (p)*c.stride]

// ---- NORM_2 per segment (reference prox/norm_2.cc:11-16 under the axis loop) ---------------------

template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegNorm2Kernel(T* x, const T* v, double lam, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  double ss = 0;
  SEG_FOR(p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    ss += d * d;
  }
  const double nv = sqrt(GroupSum<G>(ss));
  const double scale = (nv >= lam && nv > 0) ? 1.0 - lam / nv : 0.0;
  SEG_FOR(p) SEG_AT(x, p) = static_cast<T>(scale * static_cast<double>(SEG_AT(v, p)));
}

// ---- MAX prox and epigraph (reference prox/max.cc) --------------------------------------------------
// prox: x = min(v, t) with sum_i (v_i - t)_+ = lam.  g(t) = sum (v_i - t)_+ - lam is convex,
// piecewise linear and decreasing; Newton from t0 = max(v) - lam (g(t0) >= 0) increases
// monotonically and stops on the exact root when the active set {v_i > t} repeats.

template <class T, int G>
__device__ inline double MaxThreshold(const T* v, const SegCtx& c, double t, double num0,
                                      double den0) {
  double cprev = -1;
  for (int it = 0; it < 256; ++it) {
    double sum = 0, cnt = 0;
    SEG_FOR(p) {
      const double d = static_cast<double>(SEG_AT(v, p));
      if (d > t) {
        sum += d;
        cnt += 1;
      }
    }
    sum = GroupSum<G>(sum);
    cnt = GroupSum<G>(cnt);
    if (cnt == cprev || cnt + den0 == 0) break;
    cprev = cnt;
    t = (sum + num0) / (cnt + den0);
  }
  return t;
}

template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegMaxProxKernel(T* x, const T* v, double lam, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  double mx = -INFINITY;
  SEG_FOR(p) mx = fmax(mx, static_cast<double>(SEG_AT(v, p)));
  mx = GroupMax<G>(mx);
  const double t = MaxThreshold<T, G>(v, c, mx - lam, -lam, 0.0);
  SEG_FOR(p) SEG_AT(x, p) = static_cast<T>(fmin(static_cast<double>(SEG_AT(v, p)), t));
}

// epigraph: t - s = sum (v_i - t)_+ ; Newton from t0 = s (max.cc:46-87)
template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegMaxEpiKernel(T* x, T* tout, const T* v, const T* sin,
                                                          Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  const double s = static_cast<double>(sin[c.seg]);
  double mx = -INFINITY;
  SEG_FOR(p) mx = fmax(mx, static_cast<double>(SEG_AT(v, p)));
  mx = GroupMax<G>(mx);
  double t = s;
  if (!(s >= mx)) t = MaxThreshold<T, G>(v, c, s, s, 1.0);
  SEG_FOR(p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    SEG_AT(x, p) = static_cast<T>(d - fmax(0.0, d - t));
  }
  if (c.lane == 0) tout[c.seg] = static_cast<T>(t);
}

// ---- SUM_LARGEST prox (reference prox/sum_largest.cc:17-62) ----------------------------------------
// x = v - clip(v - q, 0, lam) with h(q) = sum_i clip(v_i - q, 0, lam) - k lam = 0; h is piecewise
// linear and decreasing.  Bracketed Newton on the linear pieces: on the piece around q,
// h(q') = a lam + sI - cI q' - k lam with a = #{v >= q + lam}, I = {q <= v < q + lam}.

template <class T, int G>
__device__ inline double SumLargestThreshold(const T* v, const SegCtx& c, double lam, double k) {
  double mn = INFINITY, mx = -INFINITY;
  SEG_FOR(p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    mn = fmin(mn, d);
    mx = fmax(mx, d);
  }
  mx = GroupMax<G>(mx);
  mn = -GroupMax<G>(-mn);
  double lo = mn - lam, hi = mx;
  if (k >= static_cast<double>(c.len) || !(lam > 0)) return lo;  // every entry gives up lam
  double q = 0.5 * (lo + hi);
  for (int it = 0; it < 200; ++it) {
    double a = 0, cI = 0, sI = 0;
    SEG_FOR(p) {
      const double d = static_cast<double>(SEG_AT(v, p));
      if (d >= q + lam) {
        a += 1;
      } else if (d >= q) {
        cI += 1;
        sI += d;
      }
    }
    a = GroupSum<G>(a);
    cI = GroupSum<G>(cI);
    sI = GroupSum<G>(sI);
    const double h = a * lam + sI - cI * q - k * lam;
    if (h == 0) break;
    if (h > 0) lo = q;
    else hi = q;
    double qn = cI > 0 ? (a * lam + sI - k * lam) / cI : 0.5 * (lo + hi);
    if (cI > 0 && fabs(qn - q) <= 1e-15 * fmax(1.0, fabs(q))) {
      q = qn;
      break;  // q is the root of its own linear piece
    }
    if (!(qn > lo && qn < hi)) qn = 0.5 * (lo + hi);
    if (qn == q || !(hi > lo)) break;
    q = qn;
  }
  return q;
}

template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegSumLargestKernel(T* x, const T* v, double lam,
                                                              double k, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  const double q = SumLargestThreshold<T, G>(v, c, lam, k);
  SEG_FOR(p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    SEG_AT(x, p) = static_cast<T>(d - fmax(0.0, fmin(lam, d - q)));
  }
}

// sum of the k largest entries of a segment through its variational form
//   sum_largest(x, k) = min_tau  k tau + sum_i (x_i - tau)_+   (minimiser: the k-th largest entry),
// tau found by bisection on the count #{x_i > tau}.  `shifted` evaluates it on the prox point
// x = v - clip(v - q, 0, lam) without materialising it.
template <class T, int G>
__device__ inline double SumLargestEval(const T* v, const SegCtx& c, double k, double q, double lam,
                                        bool shifted) {
  auto val = [&](int64_t p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    return shifted ? d - fmax(0.0, fmin(lam, d - q)) : d;
  };
  double mn = INFINITY, mx = -INFINITY, total = 0;
  SEG_FOR(p) {
    const double d = val(p);
    mn = fmin(mn, d);
    mx = fmax(mx, d);
    total += d;
  }
  mx = GroupMax<G>(mx);
  mn = -GroupMax<G>(-mn);
  total = GroupSum<G>(total);
  if (k >= static_cast<double>(c.len)) return total;
  double lo = mn - 1, hi = mx;  // #{> lo} = len > k ; #{> hi} = 0 <= k
  for (int it = 0; it < 200; ++it) {
    const double mid = lo + 0.5 * (hi - lo);
    if (!(mid > lo && mid < hi)) break;
    double cnt = 0;
    SEG_FOR(p) cnt += val(p) > mid ? 1.0 : 0.0;
    cnt = GroupSum<G>(cnt);
    if (cnt > k) lo = mid;
    else hi = mid;
  }
  double above = 0;
  SEG_FOR(p) above += fmax(val(p) - hi, 0.0);
  return k * hi + GroupSum<G>(above);
}

// SUM_LARGEST epigraph: the reference's bisection on lam (newton.cc:239-288) with the prox and
// the function value evaluated on the device.
template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegSumLargestEpiKernel(T* x, T* tout, const T* v,
                                                                 const T* sin, double k, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  const double s = static_cast<double>(sin[c.seg]);
  double fval = SumLargestEval<T, G>(v, c, k, 0, 0, false);
  if (fval <= s) {
    SEG_FOR(p) SEG_AT(x, p) = SEG_AT(v, p);
    if (c.lane == 0) tout[c.seg] = static_cast<T>(s);
    return;
  }
  double lam = 1, upper = 1, lower = 0, q = 0;
  bool upper_fixed = false, converged = false;
  const double eps = 1e-5;
  for (int it = 0; it < 100; ++it) {
    q = SumLargestThreshold<T, G>(v, c, lam, k);
    fval = SumLargestEval<T, G>(v, c, k, q, lam, true);
    const double g = fval - (lam + s);
    if (fabs(g) <= eps) {
      converged = true;
      break;
    }
    if (g > 0 && !upper_fixed) {
      lam *= 2;
      upper = lam;
    } else if (g > 0) {
      lower = lam;
      lam = (lam + upper) / 2;
    } else {
      upper = lam;
      lam = (lam + lower) / 2;
      upper_fixed = true;
    }
  }
  if (!converged) q = SumLargestThreshold<T, G>(v, c, lam, k);
  SEG_FOR(p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    SEG_AT(x, p) = static_cast<T>(d - fmax(0.0, fmin(lam, d - q)));
  }
  if (c.lane == 0) tout[c.seg] = static_cast<T>(lam + s);
}

// ---- scaled-zone epigraph per segment (reference prox/scaled_zone.cc:123-279 under the axis loop)
// keys k_i = (|y_i| - M)/w_i, weights w_i^2; lam is the root of sum w^2 max(k - lam, 0) = s + lam,
// found by Newton from lam = 0 on the convex piecewise-linear function (Michelot), on chip.

template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegZoneEpiKernel(T* x, T* tout, const T* v, const T* sin,
                                                           double alpha_s, double beta_s,
                                                           const T* alpha_v, const T* beta_v,
                                                           double M, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  const double s = static_cast<double>(sin[c.seg]);
  auto weight = [&](int64_t p, double y) {
    const double a = alpha_v ? static_cast<double>(alpha_v[p]) : alpha_s;
    const double b = beta_v ? static_cast<double>(beta_v[p]) : beta_s;
    return y > 0 ? a : b;
  };
  double fval = 0;
  SEG_FOR(p) {
    const double y = static_cast<double>(SEG_AT(v, p));
    const double w = weight(p, y);
    if (fabs(y) > M && w != 0) fval += w * (fabs(y) - M);
  }
  fval = GroupSum<G>(fval);
  double lam = 0;
  if (!(fval <= s)) {
    double cprev = -1;
    for (int it = 0; it < 256; ++it) {
      double swk = 0, sw2 = 0, cnt = 0;
      SEG_FOR(p) {
        const double y = static_cast<double>(SEG_AT(v, p));
        const double w = weight(p, y);
        if (fabs(y) > M && w != 0) {
          const double ex = fabs(y) - M;
          if (ex / w > lam) {
            swk += w * ex;
            sw2 += w * w;
            cnt += 1;
          }
        }
      }
      swk = GroupSum<G>(swk);
      sw2 = GroupSum<G>(sw2);
      cnt = GroupSum<G>(cnt);
      if (cnt == cprev) break;
      cprev = cnt;
      lam = (swk - s) / (sw2 + 1);
    }
  }
  SEG_FOR(p) {
    const double y = static_cast<double>(SEG_AT(v, p));
    const double a = alpha_v ? static_cast<double>(alpha_v[p]) : alpha_s;
    const double b = beta_v ? static_cast<double>(beta_v[p]) : beta_s;
    double out;  // ApplyScaledZone (scaled_zone.cc:78-104), same branch order
    if (fabs(y) <= M) out = y;
    else if (y > M + lam * a) out = y - lam * a;
    else if (y < -M - lam * b) out = y + lam * b;
    else if (y > 0) out = M;
    else out = -M;
    SEG_AT(x, p) = static_cast<T>(out);
  }
  if (c.lane == 0) tout[c.seg] = static_cast<T>(s + lam);
}

// ---- second-order cone, one cone per segment (reference prox/second_order_cone.cc:58-79) -----------

template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegSocKernel(T* x, T* tout, const T* v, const T* tin,
                                                       double beta, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  double ss = 0;
  SEG_FOR(p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    ss += d * d;
  }
  const double vnorm = sqrt(GroupSum<G>(ss));
  double t = static_cast<double>(tin[c.seg]);
  const double beta2 = beta * beta;
  double alpha = (1 / (beta2 + 1)) * (beta2 + beta * t / vnorm);
  if (isnan(alpha) || alpha > 1) {
    alpha = 1;
  } else if (alpha < 0) {
    alpha = 0;
    t = 0;
  } else {
    t = (1 / beta) * alpha * vnorm;
  }
  SEG_FOR(p) SEG_AT(x, p) = static_cast<T>(alpha * static_cast<double>(SEG_AT(v, p)));
  if (c.lane == 0) tout[c.seg] = static_cast<T>(t);
}

// ---- smooth separable functions (reference prox/sum_exp.cc, sum_logistic.cc, sum_neg_entr.cc,
//      sum_inv_pos.cc, sum_neg_log.cc) ------------------------------------------------------------------

struct FnExp {
  static constexpr bool kImplicit = false, kClosedForm = false, kNoEasy = false;
  __device__ static double f(double x) { return exp(x); }
  __device__ static double g(double x) { return exp(x); }
  __device__ static double h(double x) { return exp(x); }
  __device__ static double proj(double x) { return x; }
};
struct FnLogistic {
  static constexpr bool kImplicit = false, kClosedForm = false, kNoEasy = false;
  __device__ static double f(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
  __device__ static double g(double x) { return 1 / (1 + exp(-x)); }
  __device__ static double h(double x) {
    const double s = 1 / (1 + exp(-x));
    return s * (1 - s);
  }
  __device__ static double proj(double x) { return x; }
};
struct FnNegEntr {
  static constexpr bool kImplicit = true, kClosedForm = false, kNoEasy = false;
  __device__ static double f(double x) { return x <= 0 ? 0.0 : x * log(x); }
  __device__ static double g(double x) { return 1 + log(x); }
  __device__ static double h(double x) { return 1 / x; }
  __device__ static double proj(double x) { return fmax(x, 1e-6); }
};
struct FnInvPos {
  static constexpr bool kImplicit = false, kClosedForm = false, kNoEasy = false;
  __device__ static double f(double x) { return 1 / x; }
  __device__ static double g(double x) { return -1 / (x * x); }
  __device__ static double h(double x) { return 2 / (x * x * x); }
  __device__ static double proj(double x) { return fmax(x, 1e-6); }
};
struct FnNegLog {  // closed-form prox (sum_neg_log.cc:9-24); epigraph without the easy case
  static constexpr bool kImplicit = true, kClosedForm = true, kNoEasy = true;
  __device__ static double f(double x) { return -log(x); }
  __device__ static double g(double x) { return -1 / x; }
  __device__ static double h(double x) { return 1 / (x * x); }
  __device__ static double proj(double x) { return x; }
};

// argmin_x lam f(x) + 1/2 (x - v)^2 for one element: the damped Newton of newton.cc:49-103
// specialised to n = 1 (same step, same Armijo test on |x - v + lam f'(x)|).
template <class Fn> __device__ inline double ProxElem(double v, double lam) {
  if constexpr (Fn::kClosedForm) {
    const double z = sqrt(v * v + 4 * lam);
    return v >= 0 ? (v + z) / 2 : 2 * lam / (-v + z);
  } else {
    const double eps = 1e-14;
    double x = Fn::proj(v);
    double res = x - v + lam * Fn::g(x);
    for (int it = 0; it < 100; ++it) {
      if (fabs(res) < eps * (1 + fabs(v))) break;
      const double dx = res / (1 + lam * Fn::h(x));
      double theta = 1;
      bool moved = false;
      while (theta > 1e-12) {
        const double nx = Fn::proj(x - theta * dx);
        const double nres = nx - v + lam * Fn::g(nx);
        if (fabs(nres) <= (1 - 0.001 * theta) * fabs(res)) {
          x = nx;
          res = nres;
          moved = true;
          break;
        }
        theta *= 0.5;
      }
      if (!moved) break;
    }
    return x;
  }
}

template <class T, class Fn>
__global__ __launch_bounds__(kBlock) void SmoothProxKernel(T* x, const T* v, int64_t n, double lam,
                                                           const T* lam_vec) {
  const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = tid; i < n; i += stride) {
    const double l = lam_vec ? static_cast<double>(lam_vec[i]) : lam;
    x[i] = static_cast<T>(ProxElem<Fn>(static_cast<double>(v[i]), l));
  }
}

// Epigraph of a separable smooth function: (x, t) = (prox_{lam f}(v), s + lam) with lam >= 0 the
// root of phi(lam) = f(prox_{lam f}(v)) - lam - s, phi' = -sum g_i^2 / (1 + lam h_i) - 1.
template <class T, class Fn, int G>
__global__ __launch_bounds__(kBlock) void SegSmoothEpiKernel(T* x, T* tout, const T* v,
                                                             const T* sin, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  const double s = static_cast<double>(sin[c.seg]);
  if constexpr (!Fn::kNoEasy) {
    double fv = 0, dist2 = 0;
    SEG_FOR(p) {
      const double d = static_cast<double>(SEG_AT(v, p));
      const double pd = Fn::proj(d);
      fv += Fn::f(pd);
      dist2 += (d - pd) * (d - pd);
    }
    fv = GroupSum<G>(fv);
    dist2 = GroupSum<G>(dist2);
    const double eps = fmax(1e-12, 1e-10 / static_cast<double>(c.len));
    // newton.cc:127-133 (explicit form: only a feasible v is left alone) / :205-212 (implicit)
    if (fv <= s && (Fn::kImplicit || sqrt(dist2) < eps)) {
      SEG_FOR(p) {
        const double d = static_cast<double>(SEG_AT(v, p));
        SEG_AT(x, p) = static_cast<T>(Fn::kImplicit ? Fn::proj(d) : d);
      }
      if (c.lane == 0) tout[c.seg] = static_cast<T>(s);
      return;
    }
  }
  const double lam_min = Fn::kNoEasy ? 1e-10 : 0.0;
  double lam = 1, lo = lam_min, hi = INFINITY;
  for (int it = 0; it < 200; ++it) {
    double F = 0, Hs = 0;
    SEG_FOR(p) {
      const double xi = ProxElem<Fn>(static_cast<double>(SEG_AT(v, p)), lam);
      const double g = Fn::g(xi);
      F += Fn::f(xi);
      Hs += g * g / (1 + lam * Fn::h(xi));
    }
    F = GroupSum<G>(F);
    Hs = GroupSum<G>(Hs);
    const double phi = F - lam - s;
    if (fabs(phi) <= 1e-12 * fmax(1.0, fmax(fabs(F), fabs(s)))) break;
    if (phi > 0) lo = lam;
    else hi = lam;
    double ln = lam - phi / (-Hs - 1);
    if (!(ln > lo && ln < hi)) ln = isinf(hi) ? 2 * lam : 0.5 * (lo + hi);
    if (ln == lam) break;
    lam = ln;
  }
  SEG_FOR(p) SEG_AT(x, p) = static_cast<T>(ProxElem<Fn>(static_cast<double>(SEG_AT(v, p)), lam));
  if (c.lane == 0) tout[c.seg] = static_cast<T>(s + lam);
}

// ---- SUM_KL_DIV (reference prox/sum_kl_div.cc) --------------------------------------------------------

__device__ inline void KlProxElem(double lam, double u, double v, double* x, double* y) {
  const double eps = 1e-13;
  double yhat = fmax((0.5 + lam - v) / lam, eps);
  if (fabs(u) < eps * eps && fabs(v) < eps * eps) {
    *x = u;
    *y = v;
    return;
  }
  for (int it = 0; it < 1000; ++it) {
    const double f = lam * yhat * yhat + (v - lam) * yhat - u + lam * log(yhat);
    const double F = 2 * lam * yhat + (v - lam) + lam / yhat;
    const double res = f / F;
    if (fabs(res) < eps || (yhat <= eps * 2 && res > 0) ||
        (lam * yhat + v - lam <= eps * 2 && res > 0))
      break;
    yhat = yhat - res;
    if (yhat < eps) yhat = eps;
    if (lam * yhat + v - lam < eps) yhat = (eps + lam - v) / lam;
  }
  *y = lam * yhat + v - lam;
  *x = *y * yhat;
}

template <class T>
__global__ __launch_bounds__(kBlock) void KlDivProxKernel(T* x, T* y, const T* u, const T* v,
                                                          int64_t n, double lam, const T* lam_vec) {
  const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = tid; i < n; i += stride) {
    double xi, yi;
    KlProxElem(lam_vec ? static_cast<double>(lam_vec[i]) : lam, static_cast<double>(u[i]),
               static_cast<double>(v[i]), &xi, &yi);
    x[i] = static_cast<T>(xi);
    y[i] = static_cast<T>(yi);
  }
}

// epigraph (sum_kl_div.cc:71-126): Newton on lam with lam >= 1e-10, no easy case
template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegKlDivEpiKernel(T* x, T* y, T* tout, const T* u,
                                                            const T* v, const T* sin, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  const double s = static_cast<double>(sin[c.seg]);
  const double eps = 1e-10;
  double lam = 1;
  for (int it = 0; it < 100; ++it) {
    double glam = 0, hs = 0;
    SEG_FOR(p) {
      double xi, yi;
      KlProxElem(lam, static_cast<double>(SEG_AT(u, p)), static_cast<double>(SEG_AT(v, p)), &xi,
                 &yi);
      glam += xi * log(xi / yi) - xi + yi;
      const double g0 = log(xi / yi), g1 = -xi / yi + 1;
      // (I + lam H)^{-1} g with H = [[1/x, -1/y], [-1/y, x/y^2]]
      const double a = 1 + lam / xi, b = -lam / yi, d = 1 + lam * xi / (yi * yi);
      const double det = a * d - b * b;
      hs += (g0 * (d * g0 - b * g1) + g1 * (-b * g0 + a * g1)) / det;
    }
    glam = GroupSum<G>(glam) - s - lam;
    const double hlam = -1 - GroupSum<G>(hs);
    if (fabs(glam) < eps || (lam <= eps * 2 && glam / hlam > 0)) break;
    lam = lam - glam / hlam;
    if (lam < eps) lam = eps;
  }
  SEG_FOR(p) {
    double xi, yi;
    KlProxElem(lam, static_cast<double>(SEG_AT(u, p)), static_cast<double>(SEG_AT(v, p)), &xi, &yi);
    SEG_AT(x, p) = static_cast<T>(xi);
    SEG_AT(y, p) = static_cast<T>(yi);
  }
  if (c.lane == 0) tout[c.seg] = static_cast<T>(s + lam);
}

// ---- EXP epigraph, elementwise (reference prox/exp.cc:12-77) ------------------------------------------

template <class T>
__global__ __launch_bounds__(kBlock) void ExpEpiKernel(T* xo, T* to, const T* vin, const T* sin,
                                                       int64_t n) {
  const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = tid; i < n; i += stride) {
    const double v = static_cast<double>(vin[i]), s = static_cast<double>(sin[i]);
    double x = v, t = s, l = 1;
    if (exp(v) <= s) {  // already in the epigraph (:67-73)
      xo[i] = vin[i];
      to[i] = sin[i];
      continue;
    }
    for (int it = 0; it < 100; ++it) {
      const double ex = exp(x);
      const double r0 = (x - v) + l * ex, r1 = t - s - l, r2 = ex - t;
      if (fabs(r0) < 1e-12 && fabs(r1) < 1e-12 && fabs(r2) < 1e-12) break;
      const double h = 1 + l * ex, d = ex;
      const double dl = (-d * r0 + h * (r1 + r2)) / (d * d + h);
      const double dx = -(d * dl + r0) / h;
      const double dt = dl - r1;
      x += dx;
      t += dt;
      l += dl;
    }
    xo[i] = static_cast<T>(x);
    to[i] = static_cast<T>(t);
  }
}

// ---- LOG_SUM_EXP (reference prox/log_sum_exp.cc + newton.cc) -------------------------------------------

// omega with omega * e^omega = e^L, i.e. omega + log(omega) = L (Lambert W in the log domain)
__device__ inline double LambertWExp(double L) {
  if (L < 0) {
    // y = log(omega): e^y + y = L, convex and increasing, Newton from the right (y0 = L)
    double y = L;
    for (int it = 0; it < 50; ++it) {
      const double ey = exp(y);
      const double d = (ey + y - L) / (ey + 1);
      y -= d;
      if (fabs(d) <= 1e-16 * fmax(1.0, fabs(y))) break;
    }
    return exp(y);
  }
  // omega + log(omega) - L is concave and increasing: Newton from the left converges monotonically
  double w = L >= 1 ? L - log(L) : 0.5;
  for (int it = 0; it < 50; ++it) {
    const double d = (w + log(w) - L) / (1 + 1 / w);
    w -= d;
    if (fabs(d) <= 1e-16 * w) break;
  }
  return w;
}

// c = log Z of the prox of lam*lse at v: sum_i W(lam e^{v_i - c}) = lam.  On return *tsum =
// sum_i w_i^2 / (1 + lam w_i) (the curvature term the epigraph Newton needs).
template <class T, int G>
__device__ inline double LseProxLogZ(const T* v, const SegCtx& c, double lam, double* tsum) {
  double mx = -INFINITY;
  SEG_FOR(p) mx = fmax(mx, static_cast<double>(SEG_AT(v, p)));
  mx = GroupMax<G>(mx);
  double se = 0;
  SEG_FOR(p) se += exp(static_cast<double>(SEG_AT(v, p)) - mx);
  const double lse = mx + log(GroupSum<G>(se));
  const double loglam = log(lam);
  double cz = lse - lam;  // Z in [e^-lam sum e^v, sum e^v]; the residual below is >= 0 here
  double t = 0;
  for (int it = 0; it < 100; ++it) {
    double sw = 0, dsw = 0, tt = 0;
    SEG_FOR(p) {
      const double om = LambertWExp(loglam + static_cast<double>(SEG_AT(v, p)) - cz);
      sw += om;                // lam * w_i
      dsw += om / (1 + om);    // -d(om)/dc
      const double w = om / lam;
      tt += w * w / (1 + om);
    }
    sw = GroupSum<G>(sw);
    dsw = GroupSum<G>(dsw);
    t = GroupSum<G>(tt);
    const double r = sw - lam;  // decreasing, convex in c
    const double step = r / dsw;
    if (!(step > 1e-16 * fmax(1.0, fabs(cz)))) break;  // monotone from the left: done
    cz += step;
  }
  *tsum = t;
  return cz;
}

template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegLseProxKernel(T* x, const T* v, double lam, Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  if (!(lam > 0)) {
    SEG_FOR(p) SEG_AT(x, p) = SEG_AT(v, p);
    return;
  }
  double t;
  const double cz = LseProxLogZ<T, G>(v, c, lam, &t);
  const double loglam = log(lam);
  SEG_FOR(p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    SEG_AT(x, p) = static_cast<T>(d - LambertWExp(loglam + d - cz));
  }
}

// epigraph: lam with lse(prox_{lam lse}(v)) = s + lam; lse of the prox point is c itself.
// phi(lam) = c(lam) - lam - s, phi' = -t/(1 - lam t) - 1 with t = sum w^2/(1 + lam w)
// (Sherman-Morrison on I + lam (diag(w) - w w'), log_sum_exp.cc:11-18).
template <class T, int G>
__global__ __launch_bounds__(kBlock) void SegLseEpiKernel(T* x, T* tout, const T* v, const T* sin,
                                                          Segs S) {
  SegCtx c;
  if (!SegInit<G>(S, &c)) return;
  const double s = static_cast<double>(sin[c.seg]);
  double mx = -INFINITY;
  SEG_FOR(p) mx = fmax(mx, static_cast<double>(SEG_AT(v, p)));
  mx = GroupMax<G>(mx);
  double se = 0;
  SEG_FOR(p) se += exp(static_cast<double>(SEG_AT(v, p)) - mx);
  const double lse = mx + log(GroupSum<G>(se));
  if (lse <= s) {
    SEG_FOR(p) SEG_AT(x, p) = SEG_AT(v, p);
    if (c.lane == 0) tout[c.seg] = static_cast<T>(s);
    return;
  }
  double lam = 1, lo = 0, hi = INFINITY, cz = lse;
  for (int it = 0; it < 200; ++it) {
    double t;
    cz = LseProxLogZ<T, G>(v, c, lam, &t);
    const double phi = cz - lam - s;
    if (fabs(phi) <= 1e-13 * fmax(1.0, fmax(fabs(cz), fabs(s)))) break;
    if (phi > 0) lo = lam;
    else hi = lam;
    double ln = lam - phi / (-t / (1 - lam * t) - 1);
    if (!(ln > lo && ln < hi)) ln = isinf(hi) ? 2 * lam : 0.5 * (lo + hi);
    if (ln == lam) break;
    lam = ln;
  }
  const double loglam = log(lam);
  SEG_FOR(p) {
    const double d = static_cast<double>(SEG_AT(v, p));
    SEG_AT(x, p) = static_cast<T>(d - LambertWExp(loglam + d - cz));
  }
  if (c.lane == 0) tout[c.seg] = static_cast<T>(s + lam);
}

// ---- one LONG slice on the whole chip ------------------------------------------------------------------
// The kernels above give a slice to at most one workgroup; a single slice of 1e7 entries (the max
// of a long vector, log-sum-exp over all samples) then runs on one CU - measured 99 ms (MAX) and
// 272 ms (LOG_SUM_EXP) at n = 1e7.  For long slices the scalar iteration stays the same but every
// reduction of it becomes ONE launch of the whole grid: each workgroup reduces its share, stores
// its partial results, takes a ticket, and the workgroup that arrives last adds the partials in a
// fixed order and advances the scalar state (a few doubles in device memory).  The host enqueues
// iterations in batches and looks at the `done` word between batches; an iteration launched after
// convergence returns at once.  Same algorithms, same fp64 scalars, same stopping tests as the
// per-segment device functions above.
struct GridState {
  double s[24];
  int done;
  int iters;
  unsigned ticket;
  int pad;
};
constexpr int kGridMaxBlocks = 1024;
constexpr int kGridK = 3;  // partial results per workgroup
constexpr int kGridS = 24;  // doubles of scalar state

__device__ inline double WaveReduceK(double v, bool is_max) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = is_max ? fmax(v, o) : v + o;
  }
  return v;
}

// Op: Identity(k, s) / Acc(acc, d, s) / IsMax(k, s) / Update(s, tot, &done) / Apply(d, s)
template <class T, class Op>
__global__ __launch_bounds__(kBlock) void GridIterKernel(Op op, const T* __restrict__ v, int64_t n,
                                                         int64_t stride, GridState* st,
                                                         double* partial) {
  __shared__ double red[kBlock / 64][kGridK];
  __shared__ double tot[kGridK];
  __shared__ bool last;
  if (st->done) return;  // (written by a previous launch: visible at the kernel boundary)
  double s[kGridS];
#pragma unroll
  for (int i = 0; i < kGridS; ++i) s[i] = st->s[i];
  double acc[kGridK];
#pragma unroll
  for (int k = 0; k < kGridK; ++k) acc[k] = op.Identity(k, s);
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) op.Acc(acc, static_cast<double>(v[i * stride]), s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kGridK; ++k) {
    const double t = WaveReduceK(acc[k], op.IsMax(k, s));
    if (lane == 0) red[wave][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < kGridK) {
    const int k = threadIdx.x;
    double t = red[0][k];
    for (int w = 1; w < kBlock / 64; ++w) t = op.IsMax(k, s) ? fmax(t, red[w][k]) : t + red[w][k];
    __hip_atomic_store(partial + blockIdx.x * kGridK + k, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = prev == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x < kGridK) {
    const int k = threadIdx.x;
    const bool mx = op.IsMax(k, s);
    double t = op.Identity(k, s);
    for (unsigned b = 0; b < gridDim.x; ++b) {
      const double p = __hip_atomic_load(partial + b * kGridK + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = mx ? fmax(t, p) : t + p;
    }
    tot[k] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int done = 0;
    double t3[kGridK];
#pragma unroll
    for (int k = 0; k < kGridK; ++k) t3[k] = tot[k];
    op.Update(s, t3, &done);
#pragma unroll
    for (int i = 0; i < kGridS; ++i) st->s[i] = s[i];
    st->done = done;
    st->iters += 1;
    __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <class T, class Op>
__global__ __launch_bounds__(kBlock) void GridApplyKernel(Op op, T* x, const T* v, int64_t n, int64_t stride,
                                                          const GridState* st, T* tout) {
  double s[kGridS];
#pragma unroll
  for (int i = 0; i < kGridS; ++i) s[i] = st->s[i];
  for (int64_t i = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    x[i * stride] = static_cast<T>(op.Apply(static_cast<double>(v[i * stride]), s));
  if (tout != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *tout = static_cast<T>(op.OutT(s));  // epigraphs
}

// MAX prox and epigraph (MaxThreshold above): s = {t, previous count, num0, den0, phase, s_in, mode}
// mode 0: prox, t0 = max(v) + num0 with num0 = -lam, den0 = 0; mode 1: epigraph, t0 = s_in,
// num0 = s_in, den0 = 1, nothing to do when s_in >= max(v)
struct GridMaxOp {
  __device__ double Identity(int k, const double* s) const { return (s[4] == 0 && k == 0) ? -INFINITY : 0.0; }
  __device__ bool IsMax(int k, const double* s) const { return s[4] == 0 && k == 0; }
  __device__ void Acc(double* acc, double d, const double* s) const {
    if (s[4] == 0) {
      acc[0] = fmax(acc[0], d);
    } else if (d > s[0]) {
      acc[1] += d;
      acc[2] += 1;
    }
  }
  __device__ void Update(double* s, const double* tot, int* done) const {
    if (s[4] == 0) {
      s[1] = -1;
      s[4] = 1;
      if (s[6] == 0) {
        s[0] = tot[0] + s[2];
      } else {
        s[0] = s[5];
        if (s[5] >= tot[0]) *done = 1;
      }
      return;
    }
    const double cnt = tot[2];
    if (cnt == s[1] || cnt + s[3] == 0) {
      *done = 1;
      return;
    }
    s[1] = cnt;
    s[0] = (tot[1] + s[2]) / (cnt + s[3]);
  }
  __device__ double Apply(double d, const double* s) const { return fmin(d, s[0]); }
  __device__ double OutT(const double* s) const { return s[0]; }
};

// SUM_LARGEST prox (SumLargestThreshold above): s = {q, lo, hi, lam, k, phase, len}
struct GridSumLargestProx {
  __device__ double Identity(int k, const double* s) const { return (s[5] == 0 && k < 2) ? -INFINITY : 0.0; }
  __device__ bool IsMax(int k, const double* s) const { return s[5] == 0 && k < 2; }
  __device__ void Acc(double* acc, double d, const double* s) const {
    if (s[5] == 0) {
      acc[0] = fmax(acc[0], d);
      acc[1] = fmax(acc[1], -d);
    } else if (d >= s[0] + s[3]) {
      acc[0] += 1;
    } else if (d >= s[0]) {
      acc[1] += 1;
      acc[2] += d;
    }
  }
  __device__ void Update(double* s, const double* tot, int* done) const {
    const double lam = s[3], k = s[4];
    if (s[5] == 0) {
      const double mx = tot[0], mn = -tot[1];
      s[1] = mn - lam;
      s[2] = mx;
      s[5] = 1;
      if (k >= s[6] || !(lam > 0)) {  // every entry gives up lam
        s[0] = s[1];
        *done = 1;
        return;
      }
      s[0] = 0.5 * (s[1] + s[2]);
      return;
    }
    double q = s[0], lo = s[1], hi = s[2];
    const double a = tot[0], cI = tot[1], sI = tot[2];
    const double h = a * lam + sI - cI * q - k * lam;
    if (h == 0) {
      *done = 1;
      return;
    }
    if (h > 0) lo = q;
    else hi = q;
    double qn = cI > 0 ? (a * lam + sI - k * lam) / cI : 0.5 * (lo + hi);
    s[1] = lo;
    s[2] = hi;
    if (cI > 0 && fabs(qn - q) <= 1e-15 * fmax(1.0, fabs(q))) {
      s[0] = qn;
      *done = 1;  // q is the root of its own linear piece
      return;
    }
    if (!(qn > lo && qn < hi)) qn = 0.5 * (lo + hi);
    if (qn == q || !(hi > lo)) {
      *done = 1;
      return;
    }
    s[0] = qn;
  }
  __device__ double Apply(double d, const double* s) const { return d - fmax(0.0, fmin(s[3], d - s[0])); }
  __device__ double OutT(const double*) const { return 0.0; }
};

// LOG_SUM_EXP prox (LseProxLogZ above): s = {c = log Z, max, lam, log lam, phase}
struct GridLseProx {
  __device__ double Identity(int k, const double* s) const { return (s[4] == 0 && k == 0) ? -INFINITY : 0.0; }
  __device__ bool IsMax(int k, const double* s) const { return s[4] == 0 && k == 0; }
  __device__ void Acc(double* acc, double d, const double* s) const {
    if (s[4] == 0) {
      acc[0] = fmax(acc[0], d);
    } else if (s[4] == 1) {
      acc[0] += exp(d - s[1]);
    } else {
      const double om = LambertWExp(s[3] + d - s[0]);
      acc[0] += om;              // lam * w_i
      acc[1] += om / (1 + om);   // -d(om)/dc
    }
  }
  __device__ void Update(double* s, const double* tot, int* done) const {
    if (s[4] == 0) {
      s[1] = tot[0];
      s[4] = 1;
      return;
    }
    if (s[4] == 1) {
      s[0] = s[1] + log(tot[0]) - s[2];  // c0 = lse(v) - lam: the residual below is >= 0 there
      s[4] = 2;
      return;
    }
    const double r = tot[0] - s[2];
    const double step = r / tot[1];
    if (!(step > 1e-16 * fmax(1.0, fabs(s[0])))) {
      *done = 1;  // monotone from the left: done
      return;
    }
    s[0] += step;
  }
  __device__ double Apply(double d, const double* s) const { return d - LambertWExp(s[3] + d - s[0]); }
  __device__ double OutT(const double*) const { return 0.0; }
};

// Epigraph of a separable smooth function (SegSmoothEpiKernel above):
// s = {lam, lo, hi, s_in, phase, len}; phase 0 the easy case (only when the function has one),
// 1 the safeguarded Newton on lam, 9 "v itself (projected) is the answer"
template <class Fn> struct GridSmoothEpi {
  __device__ double Identity(int, const double*) const { return 0.0; }
  __device__ bool IsMax(int, const double*) const { return false; }
  __device__ void Acc(double* acc, double d, const double* s) const {
    if (s[4] == 0) {
      const double pd = Fn::proj(d);
      acc[0] += Fn::f(pd);
      acc[1] += (d - pd) * (d - pd);
    } else {
      const double xi = ProxElem<Fn>(d, s[0]);
      const double g = Fn::g(xi);
      acc[0] += Fn::f(xi);
      acc[1] += g * g / (1 + s[0] * Fn::h(xi));
    }
  }
  __device__ void Update(double* s, const double* tot, int* done) const {
    const double sin = s[3];
    if (s[4] == 0) {
      const double eps = fmax(1e-12, 1e-10 / s[5]);
      if (tot[0] <= sin && (Fn::kImplicit || sqrt(tot[1]) < eps)) {
        s[4] = 9;
        *done = 1;
        return;
      }
      s[4] = 1;
      return;
    }
    double lam = s[0], lo = s[1], hi = s[2];
    const double F = tot[0], Hs = tot[1];
    const double phi = F - lam - sin;
    if (fabs(phi) <= 1e-12 * fmax(1.0, fmax(fabs(F), fabs(sin)))) {
      *done = 1;
      return;
    }
    if (phi > 0) lo = lam;
    else hi = lam;
    double ln = lam - phi / (-Hs - 1);
    if (!(ln > lo && ln < hi)) ln = isinf(hi) ? 2 * lam : 0.5 * (lo + hi);
    s[1] = lo;
    s[2] = hi;
    if (ln == lam) {
      *done = 1;
      return;
    }
    s[0] = ln;
  }
  __device__ double Apply(double d, const double* s) const {
    if (s[4] == 9) return Fn::kImplicit ? Fn::proj(d) : d;
    return ProxElem<Fn>(d, s[0]);
  }
  __device__ double OutT(const double* s) const { return s[4] == 9 ? s[3] : s[3] + s[0]; }
};

// LOG_SUM_EXP epigraph (SegLseEpiKernel above): s = {c, max -> lse, lam, log lam, phase, s_in, lo, hi}
// phase 0 max, 1 sum of exponentials, 2 the inner Newton on c = log Z at the current lam (its
// convergence triggers the outer safeguarded Newton step on lam), 9 "v is in the epigraph"
struct GridLseEpi {
  __device__ double Identity(int k, const double* s) const { return (s[4] == 0 && k == 0) ? -INFINITY : 0.0; }
  __device__ bool IsMax(int k, const double* s) const { return s[4] == 0 && k == 0; }
  __device__ void Acc(double* acc, double d, const double* s) const {
    if (s[4] == 0) {
      acc[0] = fmax(acc[0], d);
    } else if (s[4] == 1) {
      acc[0] += exp(d - s[1]);
    } else {
      const double om = LambertWExp(s[3] + d - s[0]);
      acc[0] += om;
      acc[1] += om / (1 + om);
      const double w = om / s[2];
      acc[2] += w * w / (1 + om);
    }
  }
  __device__ void Update(double* s, const double* tot, int* done) const {
    const double sin = s[5];
    if (s[4] == 0) {
      s[1] = tot[0];
      s[4] = 1;
      return;
    }
    if (s[4] == 1) {
      const double lse = s[1] + log(tot[0]);
      s[1] = lse;
      if (lse <= sin) {
        s[4] = 9;
        *done = 1;
        return;
      }
      s[2] = 1;            // lam
      s[3] = 0;            // log lam
      s[6] = 0;            // lo
      s[7] = INFINITY;     // hi
      s[0] = lse - s[2];   // the inner iteration starts from c = lse - lam
      s[4] = 2;
      return;
    }
    double lam = s[2];
    const double r = tot[0] - lam;
    const double step = r / tot[1];
    if (step > 1e-16 * fmax(1.0, fabs(s[0]))) {  // inner Newton continues
      s[0] += step;
      return;
    }
    // c(lam) is known: the outer step
    const double cz = s[0], t = tot[2];
    const double phi = cz - lam - sin;
    if (fabs(phi) <= 1e-13 * fmax(1.0, fmax(fabs(cz), fabs(sin)))) {
      *done = 1;
      return;
    }
    double lo = s[6], hi = s[7];
    if (phi > 0) lo = lam;
    else hi = lam;
    double ln = lam - phi / (-t / (1 - lam * t) - 1);
    if (!(ln > lo && ln < hi)) ln = isinf(hi) ? 2 * lam : 0.5 * (lo + hi);
    s[6] = lo;
    s[7] = hi;
    if (ln == lam) {
      *done = 1;
      return;
    }
    s[2] = ln;
    s[3] = log(ln);
    s[0] = s[1] - ln;
  }
  __device__ double Apply(double d, const double* s) const {
    if (s[4] == 9) return d;
    return d - LambertWExp(s[3] + d - s[0]);
  }
  __device__ double OutT(const double* s) const { return s[4] == 9 ? s[5] : s[5] + s[2]; }
};

// SUM_LARGEST epigraph (SegSumLargestEpiKernel above): the bisection on lam with, inside it, the
// threshold q(lam) (SumLargestThreshold) and the value of sum_largest at the prox point
// (SumLargestEval: a bisection on the count above tau, then one pass).  Indices into s:
enum {
  SL_LAM = 0, SL_UPPER, SL_LOWER, SL_UPFIXED, SL_OUTER, SL_Q, SL_QLO, SL_QHI, SL_MN, SL_MX, SL_K, SL_LEN,
  SL_SIN, SL_ELO, SL_EHI, SL_MID, SL_TOTAL, SL_PHASE, SL_SHIFTED, SL_EIT, SL_TIT, SL_FINAL, SL_EASY
};
// phases: 0 range + total of the values being evaluated, 1 one step of the count bisection,
// 2 the pass above tau, 3 (no pass of its own) start of the threshold, 4 one threshold step
struct GridSumLargestEpi {
  __device__ double Val(double d, const double* s) const {
    return s[SL_SHIFTED] != 0 ? d - fmax(0.0, fmin(s[SL_LAM], d - s[SL_Q])) : d;
  }
  __device__ double Identity(int k, const double* s) const { return (s[SL_PHASE] == 0 && k < 2) ? -INFINITY : 0.0; }
  __device__ bool IsMax(int k, const double* s) const { return s[SL_PHASE] == 0 && k < 2; }
  __device__ void Acc(double* acc, double d, const double* s) const {
    const int ph = static_cast<int>(s[SL_PHASE]);
    if (ph == 0) {
      const double x = Val(d, s);
      acc[0] = fmax(acc[0], x);
      acc[1] = fmax(acc[1], -x);
      acc[2] += x;
    } else if (ph == 1) {
      acc[0] += Val(d, s) > s[SL_MID] ? 1.0 : 0.0;
    } else if (ph == 2) {
      acc[0] += fmax(Val(d, s) - s[SL_EHI], 0.0);
    } else {  // threshold step on the raw values
      if (d >= s[SL_Q] + s[SL_LAM]) {
        acc[0] += 1;
      } else if (d >= s[SL_Q]) {
        acc[1] += 1;
        acc[2] += d;
      }
    }
  }
  // the count bisection's next midpoint, or on to the last pass of the evaluation
  __device__ void NextMid(double* s) const {
    const double mid = s[SL_ELO] + 0.5 * (s[SL_EHI] - s[SL_ELO]);
    if (s[SL_EIT] < 200 && mid > s[SL_ELO] && mid < s[SL_EHI]) {
      s[SL_MID] = mid;
      s[SL_PHASE] = 1;
    } else {
      s[SL_PHASE] = 2;
    }
  }
  __device__ void StartThreshold(double* s, int* done) const {
    const double lam = s[SL_LAM];
    s[SL_QLO] = s[SL_MN] - lam;
    s[SL_QHI] = s[SL_MX];
    if (s[SL_K] >= s[SL_LEN] || !(lam > 0)) {
      s[SL_Q] = s[SL_QLO];
      ThresholdDone(s, done);
      return;
    }
    s[SL_Q] = 0.5 * (s[SL_QLO] + s[SL_QHI]);
    s[SL_TIT] = 0;
    s[SL_PHASE] = 4;
  }
  __device__ void ThresholdDone(double* s, int* done) const {
    if (s[SL_FINAL] != 0) {
      *done = 1;
      return;
    }
    s[SL_SHIFTED] = 1;
    s[SL_PHASE] = 0;  // evaluate sum_largest at the prox point
  }
  // f = sum_largest of the evaluated values is known
  __device__ void ValueDone(double* s, double fval, int* done) const {
    const double sin = s[SL_SIN];
    if (s[SL_SHIFTED] == 0) {
      if (fval <= sin) {
        s[SL_EASY] = 1;
        *done = 1;
        return;
      }
      s[SL_LAM] = 1;
      s[SL_UPPER] = 1;
      s[SL_LOWER] = 0;
      s[SL_UPFIXED] = 0;
      s[SL_OUTER] = 0;
      StartThreshold(s, done);
      return;
    }
    double lam = s[SL_LAM];
    const double g = fval - (lam + sin);
    if (fabs(g) <= 1e-5) {
      *done = 1;  // converged: q belongs to this lam
      return;
    }
    if (g > 0 && s[SL_UPFIXED] == 0) {
      lam *= 2;
      s[SL_UPPER] = lam;
    } else if (g > 0) {
      s[SL_LOWER] = lam;
      lam = (lam + s[SL_UPPER]) / 2;
    } else {
      s[SL_UPPER] = lam;
      lam = (lam + s[SL_LOWER]) / 2;
      s[SL_UPFIXED] = 1;
    }
    s[SL_LAM] = lam;
    s[SL_OUTER] += 1;
    if (s[SL_OUTER] >= 100) s[SL_FINAL] = 1;  // not converged: the threshold of the last lam, then stop
    StartThreshold(s, done);
  }
  __device__ void Update(double* s, const double* tot, int* done) const {
    const int ph = static_cast<int>(s[SL_PHASE]);
    const double k = s[SL_K];
    if (ph == 0) {
      const double mx = tot[0], mn = -tot[1], total = tot[2];
      if (s[SL_SHIFTED] == 0) {  // the range of the raw values also serves every threshold
        s[SL_MN] = mn;
        s[SL_MX] = mx;
      }
      if (k >= s[SL_LEN]) {
        ValueDone(s, total, done);
        return;
      }
      s[SL_ELO] = mn - 1;
      s[SL_EHI] = mx;
      s[SL_EIT] = 0;
      NextMid(s);
    } else if (ph == 1) {
      if (tot[0] > k) s[SL_ELO] = s[SL_MID];
      else s[SL_EHI] = s[SL_MID];
      s[SL_EIT] += 1;
      NextMid(s);
    } else if (ph == 2) {
      ValueDone(s, k * s[SL_EHI] + tot[0], done);
    } else {
      const double lam = s[SL_LAM];
      double q = s[SL_Q], lo = s[SL_QLO], hi = s[SL_QHI];
      const double a = tot[0], cI = tot[1], sI = tot[2];
      const double h = a * lam + sI - cI * q - k * lam;
      if (h == 0) {
        ThresholdDone(s, done);
        return;
      }
      if (h > 0) lo = q;
      else hi = q;
      double qn = cI > 0 ? (a * lam + sI - k * lam) / cI : 0.5 * (lo + hi);
      s[SL_QLO] = lo;
      s[SL_QHI] = hi;
      if (cI > 0 && fabs(qn - q) <= 1e-15 * fmax(1.0, fabs(q))) {
        s[SL_Q] = qn;
        ThresholdDone(s, done);
        return;
      }
      if (!(qn > lo && qn < hi)) qn = 0.5 * (lo + hi);
      s[SL_TIT] += 1;
      if (qn == q || !(hi > lo) || s[SL_TIT] >= 200) {
        ThresholdDone(s, done);
        return;
      }
      s[SL_Q] = qn;
    }
  }
  __device__ double Apply(double d, const double* s) const {
    if (s[SL_EASY] != 0) return d;
    return d - fmax(0.0, fmin(s[SL_LAM], d - s[SL_Q]));
  }
  __device__ double OutT(const double* s) const { return s[SL_EASY] != 0 ? s[SL_SIN] : s[SL_LAM] + s[SL_SIN]; }
};
static_assert(SL_EASY < kGridS, "GridState too small");

// NORM_2 shrinkage of one slice (SegNorm2Kernel above): s = {scale, lam, phase}
struct GridNorm2 {
  __device__ double Identity(int, const double*) const { return 0.0; }
  __device__ bool IsMax(int, const double*) const { return false; }
  __device__ void Acc(double* acc, double d, const double*) const { acc[0] += d * d; }
  __device__ void Update(double* s, const double* tot, int* done) const {
    const double nv = sqrt(tot[0]), lam = s[1];
    s[0] = (nv >= lam && nv > 0) ? 1.0 - lam / nv : 0.0;
    *done = 1;
  }
  __device__ double Apply(double d, const double* s) const { return s[0] * d; }
  __device__ double OutT(const double*) const { return 0.0; }
};

// second-order cone, one cone per slice (SegSocKernel above): s = {alpha, t, beta}
struct GridSoc {
  __device__ double Identity(int, const double*) const { return 0.0; }
  __device__ bool IsMax(int, const double*) const { return false; }
  __device__ void Acc(double* acc, double d, const double*) const { acc[0] += d * d; }
  __device__ void Update(double* s, const double* tot, int* done) const {
    const double vnorm = sqrt(tot[0]), beta = s[2], beta2 = beta * beta;
    double t = s[1];
    double alpha = (1 / (beta2 + 1)) * (beta2 + beta * t / vnorm);
    if (isnan(alpha) || alpha > 1) {
      alpha = 1;
    } else if (alpha < 0) {
      alpha = 0;
      t = 0;
    } else {
      t = (1 / beta) * alpha * vnorm;
    }
    s[0] = alpha;
    s[1] = t;
    *done = 1;
  }
  __device__ double Apply(double d, const double* s) const { return s[0] * d; }
  __device__ double OutT(const double* s) const { return s[1]; }
};

// scaled-zone epigraph of one slice with scalar alpha / beta (SegZoneEpiKernel above):
// s = {lam, previous count, s_in, M, alpha, beta, phase}
struct GridZoneEpi {
  __device__ double Identity(int, const double*) const { return 0.0; }
  __device__ bool IsMax(int, const double*) const { return false; }
  __device__ void Acc(double* acc, double y, const double* s) const {
    const double M = s[3], w = y > 0 ? s[4] : s[5];
    if (!(fabs(y) > M && w != 0)) return;
    const double ex = fabs(y) - M;
    if (s[6] == 0) {
      acc[0] += w * ex;
    } else if (ex / w > s[0]) {
      acc[0] += w * ex;
      acc[1] += w * w;
      acc[2] += 1;
    }
  }
  __device__ void Update(double* s, const double* tot, int* done) const {
    if (s[6] == 0) {
      s[0] = 0;
      s[1] = -1;
      s[6] = 1;
      if (tot[0] <= s[2]) *done = 1;
      return;
    }
    if (tot[2] == s[1]) {
      *done = 1;
      return;
    }
    s[1] = tot[2];
    s[0] = (tot[0] - s[2]) / (tot[1] + 1);
  }
  __device__ double Apply(double y, const double* s) const {  // ApplyScaledZone (scaled_zone.cc:78-104)
    const double lam = s[0], M = s[3], a = s[4], b = s[5];
    if (fabs(y) <= M) return y;
    if (y > M + lam * a) return y - lam * a;
    if (y < -M - lam * b) return y + lam * b;
    return y > 0 ? M : -M;
  }
  __device__ double OutT(const double* s) const { return s[2] + s[0]; }
};

// slices at least this long, and few of them, take the grid-wide route
constexpr int64_t kGridMinLen = int64_t(1) << 17;
inline bool UseGridRoute(const Segs& S) {
  const char* e = std::getenv("EPSILON_HIP_SEG_GRID");  // "0": one workgroup per slice whatever its length
  if (e && e[0] == '0') return false;                  // (read per call: the tests compare the two routes)
  return S.count >= 1 && S.count <= 8 && S.len >= kGridMinLen;
}

template <class T, class Op>
void RunGridSlice(const Op& op, T* x, const T* v, int64_t n, int64_t stride, const double* init, int ninit,
                  int max_iters, T* tout = nullptr) {
  Runtime& rt = Runtime::Get();
  hipStream_t st = rt.stream();
  auto sbuf = rt.Alloc(sizeof(GridState));
  auto pbuf = rt.Alloc(sizeof(double) * kGridMaxBlocks * kGridK);
  GridState h{};
  for (int i = 0; i < ninit && i < kGridS; ++i) h.s[i] = init[i];
  EPS_HIP(hipMemcpyAsync(sbuf->p, &h, sizeof(h), hipMemcpyHostToDevice, st));
  EPS_HIP(hipStreamSynchronize(st));  // (h is a stack object)
  GridState* gs = static_cast<GridState*>(sbuf->p);
  double* partial = static_cast<double*>(pbuf->p);
  int64_t grid = (n + 8 * kBlock - 1) / (8 * kBlock);
  grid = std::min<int64_t>(std::max<int64_t>(grid, 1), kGridMaxBlocks);
  int launched = 0, batch = 8;
  while (launched < max_iters) {
    const int nb = std::min(batch, max_iters - launched);
    for (int i = 0; i < nb; ++i)
      hipLaunchKernelGGL((GridIterKernel<T, Op>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, st, op, v, n,
                         stride, gs, partial);
    launched += nb;
    int done = 0;
    EPS_HIP(hipMemcpyAsync(&done, &gs->done, sizeof(int), hipMemcpyDeviceToHost, st));
    EPS_HIP(hipStreamSynchronize(st));
    if (done) break;
    if (batch < 64) batch *= 2;  // long iterations (nested loops): fewer host round trips
  }
  hipLaunchKernelGGL((GridApplyKernel<T, Op>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, st, op, x, v, n,
                     stride, gs, tout);
  EPS_HIP(hipGetLastError());
}

// the scalar inputs of the (few) slices of an epigraph, on the host
inline std::vector<double> HostScalars(const DVec& s) { return s.ToHost(); }

// ---- launch helpers ---------------------------------------------------------------------------------------

int GroupFor(const Segs& S) {
  if (S.elem_stride != 1 && S.count >= 1024) return 1;  // strided slices: one lane each, coalesced
  if (S.len <= 1) return 1;
  if (S.len <= 8) return 8;
  if (S.len <= 64 * 4) return 64;
  return 256;
}

void CheckSegs(const Segs& S, int64_t n) {
  EPS_CHECK_MSG(S.count >= 0 && S.len >= 0 && S.count * S.len == n,
                "segment layout " << S.count << " x " << S.len << " does not cover " << n);
  if (S.count > 0 && S.len > 0) {
    const int64_t last = (S.count - 1) * S.seg_stride + (S.len - 1) * S.elem_stride;
    EPS_CHECK_MSG(last < n && S.seg_stride >= 1 && S.elem_stride >= 1, "segment layout out of range");
  }
}

inline int GridElem(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return static_cast<int>(g);
}

}  // namespace

#define EPS_DISPATCH_T(dt, ...) \
  do {                          \
    if ((dt) == F32) {          \
      using T = float;          \
      __VA_ARGS__;              \
    } else {                    \
      using T = double;         \
      __VA_ARGS__;              \
    }                           \
  } while (0)

// launches KERNEL<T, G>(args...) with G chosen from the segment shape
#define EPS_LAUNCH_SEG(S, KERNEL, ...)                                                          \
  do {                                                                                          \
    const int g_ = GroupFor(S);                                                                 \
    const int64_t per_block_ = kBlock / g_;                                                     \
    const dim3 grid_(static_cast<unsigned>((S.count + per_block_ - 1) / per_block_));           \
    hipStream_t st_ = Runtime::Get().stream();                                                  \
    switch (g_) {                                                                               \
      case 1: hipLaunchKernelGGL((KERNEL<T, 1>), grid_, dim3(kBlock), 0, st_, __VA_ARGS__); break;   \
      case 8: hipLaunchKernelGGL((KERNEL<T, 8>), grid_, dim3(kBlock), 0, st_, __VA_ARGS__); break;   \
      case 64: hipLaunchKernelGGL((KERNEL<T, 64>), grid_, dim3(kBlock), 0, st_, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL((KERNEL<T, 256>), grid_, dim3(kBlock), 0, st_, __VA_ARGS__);       \
    }                                                                                           \
    EPS_HIP(hipGetLastError());                                                                 \
  } while (0)

void SegNorm2Shrink(const DVec& x, const DVec& v, double lam, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  CheckSegs(S, v.n);
  if (v.n == 0) return;
  ProfScope prof("seg_norm2", S.count, S.len);
  if (UseGridRoute(S)) {
    for (int64_t g = 0; g < S.count; ++g) {
      const double init[3] = {0, lam, 0};
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridNorm2{}, x.as<T>() + g * S.seg_stride, v.as<T>() + g * S.seg_stride,
                                           S.len, S.elem_stride, init, 3, 1));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegNorm2Kernel, x.as<T>(), v.as<T>(), lam, S));
}

void SegMaxProx(const DVec& x, const DVec& v, double lam, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  CheckSegs(S, v.n);
  if (v.n == 0) return;
  ProfScope prof("seg_max", S.count, S.len);
  if (UseGridRoute(S)) {  // long slices: every reduction on the whole grid
    for (int64_t g = 0; g < S.count; ++g) {
      const double init[7] = {0, -1, -lam, 0, 0, 0, 0};
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridMaxOp{}, x.as<T>() + g * S.seg_stride, v.as<T>() + g * S.seg_stride,
                                           S.len, S.elem_stride, init, 7, 256 + 1));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegMaxProxKernel, x.as<T>(), v.as<T>(), lam, S));
}

void SegMaxEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt && t.n == S.count && s.n == S.count && t.dt == v.dt &&
            s.dt == v.dt);
  CheckSegs(S, v.n);
  if (S.count == 0) return;
  ProfScope prof("seg_max_epi", S.count, S.len);
  if (UseGridRoute(S)) {
    const std::vector<double> sh = HostScalars(s);
    for (int64_t g = 0; g < S.count; ++g) {
      const double init[7] = {0, -1, sh[g], 1, 0, sh[g], 1};
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridMaxOp{}, x.as<T>() + g * S.seg_stride, v.as<T>() + g * S.seg_stride,
                                           S.len, S.elem_stride, init, 7, 256 + 1, t.as<T>() + g));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegMaxEpiKernel, x.as<T>(), t.as<T>(), v.as<T>(),
                                      s.as<T>(), S));
}

void SegSumLargestProx(const DVec& x, const DVec& v, double lam, int k, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  CheckSegs(S, v.n);
  if (v.n == 0) return;
  ProfScope prof("seg_sum_largest", S.count, S.len);
  if (UseGridRoute(S)) {
    for (int64_t g = 0; g < S.count; ++g) {
      const double init[8] = {0, 0, 0, lam, static_cast<double>(k), 0, static_cast<double>(S.len), 0};
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridSumLargestProx{}, x.as<T>() + g * S.seg_stride,
                                           v.as<T>() + g * S.seg_stride, S.len, S.elem_stride, init, 8, 200 + 1));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegSumLargestKernel, x.as<T>(), v.as<T>(), lam,
                                      static_cast<double>(k), S));
}

void SegSumLargestEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s, int k,
                           const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt && t.n == S.count && s.n == S.count && t.dt == v.dt &&
            s.dt == v.dt);
  CheckSegs(S, v.n);
  if (S.count == 0) return;
  ProfScope prof("seg_sum_largest_epi", S.count, S.len);
  if (UseGridRoute(S)) {
    const std::vector<double> sh = HostScalars(s);
    for (int64_t g = 0; g < S.count; ++g) {
      double init[kGridS] = {};
      init[SL_K] = static_cast<double>(k);
      init[SL_LEN] = static_cast<double>(S.len);
      init[SL_SIN] = sh[g];
      // outer bisection x (threshold steps + count bisection + 3 passes): a generous bound
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridSumLargestEpi{}, x.as<T>() + g * S.seg_stride,
                                           v.as<T>() + g * S.seg_stride, S.len, S.elem_stride, init, kGridS,
                                           101 * 410, t.as<T>() + g));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegSumLargestEpiKernel, x.as<T>(), t.as<T>(), v.as<T>(),
                                      s.as<T>(), static_cast<double>(k), S));
}

void SegZoneEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s, double alpha,
                     double beta, const DVec* alpha_vec, const DVec* beta_vec, double M,
                     const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt && t.n == S.count && s.n == S.count && t.dt == v.dt &&
            s.dt == v.dt);
  if (alpha_vec) EPS_CHECK(alpha_vec->n == S.len && alpha_vec->dt == v.dt);
  if (beta_vec) EPS_CHECK(beta_vec->n == S.len && beta_vec->dt == v.dt);
  CheckSegs(S, v.n);
  if (S.count == 0) return;
  ProfScope prof("seg_zone_epi", S.count, S.len);
  if (UseGridRoute(S) && alpha_vec == nullptr && beta_vec == nullptr) {
    const std::vector<double> sh = HostScalars(s);
    for (int64_t g = 0; g < S.count; ++g) {
      const double init[7] = {0, -1, sh[g], M, alpha, beta, 0};
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridZoneEpi{}, x.as<T>() + g * S.seg_stride, v.as<T>() + g * S.seg_stride,
                                           S.len, S.elem_stride, init, 7, 256 + 1, t.as<T>() + g));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegZoneEpiKernel, x.as<T>(), t.as<T>(), v.as<T>(),
                                      s.as<T>(), alpha, beta,
                                      alpha_vec ? alpha_vec->as<T>() : nullptr,
                                      beta_vec ? beta_vec->as<T>() : nullptr, M, S));
}

void SegSocProject(const DVec& x, const DVec& t, const DVec& v, const DVec& tin, double beta,
                   const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt && t.n == S.count && tin.n == S.count && t.dt == v.dt &&
            tin.dt == v.dt);
  CheckSegs(S, v.n);
  if (S.count == 0) return;
  ProfScope prof("seg_soc", S.count, S.len);
  if (UseGridRoute(S)) {
    const std::vector<double> th = HostScalars(tin);
    for (int64_t g = 0; g < S.count; ++g) {
      const double init[3] = {0, th[g], beta};
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridSoc{}, x.as<T>() + g * S.seg_stride, v.as<T>() + g * S.seg_stride,
                                           S.len, S.elem_stride, init, 3, 1, t.as<T>() + g));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegSocKernel, x.as<T>(), t.as<T>(), v.as<T>(),
                                      tin.as<T>(), beta, S));
}

void SegLogSumExpProx(const DVec& x, const DVec& v, double lam, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  CheckSegs(S, v.n);
  if (v.n == 0) return;
  ProfScope prof("seg_lse", S.count, S.len);
  if (UseGridRoute(S) && lam > 0) {
    for (int64_t g = 0; g < S.count; ++g) {
      const double init[8] = {0, 0, lam, std::log(lam), 0, 0, 0, 0};
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridLseProx{}, x.as<T>() + g * S.seg_stride, v.as<T>() + g * S.seg_stride,
                                           S.len, S.elem_stride, init, 8, 100 + 2));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegLseProxKernel, x.as<T>(), v.as<T>(), lam, S));
}

void SegLogSumExpEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s,
                          const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt && t.n == S.count && s.n == S.count && t.dt == v.dt &&
            s.dt == v.dt);
  CheckSegs(S, v.n);
  if (S.count == 0) return;
  ProfScope prof("seg_lse_epi", S.count, S.len);
  if (UseGridRoute(S)) {
    const std::vector<double> sh = HostScalars(s);
    for (int64_t g = 0; g < S.count; ++g) {
      double init[8] = {0, 0, 0, 0, 0, sh[g], 0, 0};
      EPS_DISPATCH_T(v.dt, RunGridSlice<T>(GridLseEpi{}, x.as<T>() + g * S.seg_stride, v.as<T>() + g * S.seg_stride,
                                           S.len, S.elem_stride, init, 8, 200 * 101 + 2, t.as<T>() + g));
    }
    return;
  }
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegLseEpiKernel, x.as<T>(), t.as<T>(), v.as<T>(),
                                      s.as<T>(), S));
}

#define EPS_SMOOTH_SWITCH(fn, ...)                        \
  switch (fn) {                                           \
    case SMOOTH_EXP: { using Fn = FnExp; __VA_ARGS__; break; }          \
    case SMOOTH_LOGISTIC: { using Fn = FnLogistic; __VA_ARGS__; break; } \
    case SMOOTH_NEG_ENTR: { using Fn = FnNegEntr; __VA_ARGS__; break; }  \
    case SMOOTH_INV_POS: { using Fn = FnInvPos; __VA_ARGS__; break; }    \
    case SMOOTH_NEG_LOG: { using Fn = FnNegLog; __VA_ARGS__; break; }    \
    default: EPS_FATAL("unknown smooth function " << fn);  \
  }

void SmoothProx(SmoothFn fn, const DVec& x, const DVec& v, double lam, const DVec* lam_vec) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  if (lam_vec) EPS_CHECK(lam_vec->n == v.n && lam_vec->dt == v.dt);
  if (v.n == 0) return;
  ProfScope prof("smooth_prox", v.n, fn);
  hipStream_t st = Runtime::Get().stream();
  EPS_DISPATCH_T(v.dt, EPS_SMOOTH_SWITCH(fn, hipLaunchKernelGGL(
                                                 (SmoothProxKernel<T, Fn>), dim3(GridElem(v.n)),
                                                 dim3(kBlock), 0, st, x.as<T>(), v.as<T>(), v.n, lam,
                                                 lam_vec ? lam_vec->as<T>() : nullptr)));
  EPS_HIP(hipGetLastError());
}

namespace {
template <class T, int G> struct SmoothEpiLaunch {
  template <class Fn>
  static void Run(const dim3& grid, T* x, T* t, const T* v, const T* s, const Segs& S) {
    hipLaunchKernelGGL((SegSmoothEpiKernel<T, Fn, G>), grid, dim3(kBlock), 0,
                       Runtime::Get().stream(), x, t, v, s, S);
  }
};
}  // namespace

void SegSmoothEpigraph(SmoothFn fn, const DVec& x, const DVec& t, const DVec& v, const DVec& s,
                       const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt && t.n == S.count && s.n == S.count && t.dt == v.dt &&
            s.dt == v.dt);
  CheckSegs(S, v.n);
  if (S.count == 0) return;
  ProfScope prof("seg_smooth_epi", S.count, S.len);
  if (UseGridRoute(S)) {
    const std::vector<double> sh = HostScalars(s);
    for (int64_t sg = 0; sg < S.count; ++sg) {
      EPS_DISPATCH_T(v.dt, EPS_SMOOTH_SWITCH(fn, {
                       const double lam_min = Fn::kNoEasy ? 1e-10 : 0.0;
                       const double init[6] = {1, lam_min, INFINITY, sh[sg], Fn::kNoEasy ? 1.0 : 0.0,
                                               static_cast<double>(S.len)};
                       RunGridSlice<T>(GridSmoothEpi<Fn>{}, x.as<T>() + sg * S.seg_stride,
                                       v.as<T>() + sg * S.seg_stride, S.len, S.elem_stride, init, 6, 200 + 1,
                                       t.as<T>() + sg);
                     }));
    }
    return;
  }
  const int g = GroupFor(S);
  const int64_t per_block = kBlock / g;
  const dim3 grid(static_cast<unsigned>((S.count + per_block - 1) / per_block));
  EPS_DISPATCH_T(v.dt, EPS_SMOOTH_SWITCH(fn, {
                   switch (g) {
                     case 1: SmoothEpiLaunch<T, 1>::template Run<Fn>(grid, x.as<T>(), t.as<T>(), v.as<T>(), s.as<T>(), S); break;
                     case 8: SmoothEpiLaunch<T, 8>::template Run<Fn>(grid, x.as<T>(), t.as<T>(), v.as<T>(), s.as<T>(), S); break;
                     case 64: SmoothEpiLaunch<T, 64>::template Run<Fn>(grid, x.as<T>(), t.as<T>(), v.as<T>(), s.as<T>(), S); break;
                     default: SmoothEpiLaunch<T, 256>::template Run<Fn>(grid, x.as<T>(), t.as<T>(), v.as<T>(), s.as<T>(), S);
                   }
                 }));
  EPS_HIP(hipGetLastError());
}

void KlDivProx(const DVec& x, const DVec& y, const DVec& u, const DVec& v, double lam,
               const DVec* lam_vec) {
  EPS_CHECK(x.n == u.n && y.n == u.n && v.n == u.n && x.dt == u.dt && y.dt == u.dt && v.dt == u.dt);
  if (lam_vec) EPS_CHECK(lam_vec->n == u.n && lam_vec->dt == u.dt);
  if (u.n == 0) return;
  ProfScope prof("kl_div_prox", u.n);
  hipStream_t st = Runtime::Get().stream();
  EPS_DISPATCH_T(u.dt, hipLaunchKernelGGL((KlDivProxKernel<T>), dim3(GridElem(u.n)), dim3(kBlock),
                                          0, st, x.as<T>(), y.as<T>(), u.as<T>(), v.as<T>(), u.n,
                                          lam, lam_vec ? lam_vec->as<T>() : nullptr));
  EPS_HIP(hipGetLastError());
}

void SegKlDivEpigraph(const DVec& x, const DVec& y, const DVec& t, const DVec& u, const DVec& v,
                      const DVec& s, const Segs& S) {
  EPS_CHECK(x.n == u.n && y.n == u.n && v.n == u.n && x.dt == u.dt && y.dt == u.dt &&
            v.dt == u.dt && t.n == S.count && s.n == S.count && t.dt == u.dt && s.dt == u.dt);
  CheckSegs(S, u.n);
  if (S.count == 0) return;
  ProfScope prof("seg_kl_epi", S.count, S.len);
  EPS_DISPATCH_T(u.dt, EPS_LAUNCH_SEG(S, SegKlDivEpiKernel, x.as<T>(), y.as<T>(), t.as<T>(),
                                      u.as<T>(), v.as<T>(), s.as<T>(), S));
}

void ExpEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s) {
  EPS_CHECK(x.n == v.n && t.n == v.n && s.n == v.n && x.dt == v.dt && t.dt == v.dt && s.dt == v.dt);
  if (v.n == 0) return;
  ProfScope prof("exp_epi", v.n);
  hipStream_t st = Runtime::Get().stream();
  EPS_DISPATCH_T(v.dt, hipLaunchKernelGGL((ExpEpiKernel<T>), dim3(GridElem(v.n)), dim3(kBlock), 0,
                                          st, x.as<T>(), t.as<T>(), v.as<T>(), s.as<T>(), v.n));
  EPS_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace eps

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
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegNorm2Kernel, x.as<T>(), v.as<T>(), lam, S));
}

void SegMaxProx(const DVec& x, const DVec& v, double lam, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  CheckSegs(S, v.n);
  if (v.n == 0) return;
  ProfScope prof("seg_max", S.count, S.len);
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegMaxProxKernel, x.as<T>(), v.as<T>(), lam, S));
}

void SegMaxEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt && t.n == S.count && s.n == S.count && t.dt == v.dt &&
            s.dt == v.dt);
  CheckSegs(S, v.n);
  if (S.count == 0) return;
  ProfScope prof("seg_max_epi", S.count, S.len);
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegMaxEpiKernel, x.as<T>(), t.as<T>(), v.as<T>(),
                                      s.as<T>(), S));
}

void SegSumLargestProx(const DVec& x, const DVec& v, double lam, int k, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  CheckSegs(S, v.n);
  if (v.n == 0) return;
  ProfScope prof("seg_sum_largest", S.count, S.len);
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
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegSocKernel, x.as<T>(), t.as<T>(), v.as<T>(),
                                      tin.as<T>(), beta, S));
}

void SegLogSumExpProx(const DVec& x, const DVec& v, double lam, const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  CheckSegs(S, v.n);
  if (v.n == 0) return;
  ProfScope prof("seg_lse", S.count, S.len);
  EPS_DISPATCH_T(v.dt, EPS_LAUNCH_SEG(S, SegLseProxKernel, x.as<T>(), v.as<T>(), lam, S));
}

void SegLogSumExpEpigraph(const DVec& x, const DVec& t, const DVec& v, const DVec& s,
                          const Segs& S) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt && t.n == S.count && s.n == S.count && t.dt == v.dt &&
            s.dt == v.dt);
  CheckSegs(S, v.n);
  if (S.count == 0) return;
  ProfScope prof("seg_lse_epi", S.count, S.len);
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

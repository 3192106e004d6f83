// Fused sweep kernel for "least squares + separable threshold" problems (the compiled lasso,
// SURVEY.md 3.3): ONE pass over the data matrix per ADMM sweep instead of two.
//
// In the reference a sweep touches A twice - `A v` in the forward substitution and `A^T w` in
// the back substitution of the least-squares prox (reference vector/block_cholesky.cc:86-117 via
// linear/dense_matrix_impl.cc:63) - with the elementwise NORM_1 prox and the u/y bookkeeping of
// prox_admm.cc:135-147 in between.  Everything between `A^T w` of sweep k and `A v` of sweep k+1
// is elementwise in the column index j.  So for each column j, while it sits in registers:
//
//     d_j   = A[:,j] . w                         (back substitution of sweep k)
//     x0_j  = v0_j + kappa d_j ;  y0, u, prox-1 (two-sided threshold), y1, u  ... -> v0'_j
//     t'   += A[:,j] * v0'_j                     (forward substitution of sweep k+1)
//
// The elementwise chain repeats the reference's operations one by one, in order, one rounding
// each (this file is compiled with -ffp-contract=off), so given d_j the state is bit-identical
// to the unfused path.  HBM traffic per sweep drops from (2mn + m^2)s to (mn + m^2)s.
//
// Geometry: 256 threads own all m rows (thread t: rows 4(t + 256q), q < NR - 16 B loads,
// coalesced 4 KiB per q); w and the partial t' stay in registers for the whole launch; the dot
// products are reduced with wave shuffles + LDS in a fixed order (deterministic); per-workgroup
// partial t' vectors are summed by ReducePartials.  Two forms: LassoFusedKernel takes columns in
// pairs (2*NR loads, then reduce / chain / update with nothing in flight);
// LassoFusedStreamKernel (the default) takes one column per step with the next column's NR
// loads already issued, so the memory pipe never drains.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;

__device__ inline float WaveSumF(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ inline float ScaledZoneOne(float v, float lam, float alpha, float beta, float M) {
  // reference prox/scaled_zone.cc:90-101 with C = 0
  float xi = v;
  if (fabsf(xi) <= M) return xi;
  if (xi > M + lam * alpha) return xi - lam * alpha;
  if (xi < -M - lam * beta) return xi + lam * beta;
  if (xi > 0.0f) return M;
  return -M;
}

struct FusedScalars {
  float kappa;  // x0 = v0 + kappa * d
  float Bs, Cs, a1;
  float lam, alpha, beta, M;
  float a0, inv_aa;  // two-block form: constraint a0 x0 + a1 x1 = 0, 1 / (a0^2 + a1^2)
  const float* alpha_v;  // per-column alpha / beta of the scaled zone (nullptr: the uniform values)
  const float* beta_v;
};

// One column of the TWO-BLOCK driver's sweep (reference algorithms/prox_admm_two_block.cc:97-112):
//   zu = z - u ;  x0 = prox_0(zu)_0 = v0 + kappa d ;  x1 = prox_1(zu)_1 (scaled zone) ;
//   z = projection of x + u onto {a0 z0 + a1 z1 = 0} ;  u += x - z.
// Returns the next sweep's prox-0 input z0' - u0'.
__device__ inline float ChainTwoBlock(float d, const FusedScalars& c, float z0p, float z1p, float u0p,
                                      float u1p, float* x0o, float* x1o, float* z0o, float* z1o,
                                      float* u0o, float* u1o);

// ---- the same pass in either precision (the f64 form serves the fp64 mode: the reference's own
// arithmetic type, linear/linear_map.h:35) ---------------------------------------------------------
template <class T> struct FusedScalarsT {
  T kappa, Bs, Cs, a1, lam, alpha, beta, M;
  T a0, inv_aa;        // two-block form: constraint a0 x0 + a1 x1 = 0, 1 / (a0^2 + a1^2)
  const T* alpha_v;    // per-column alpha / beta of the scaled zone (nullptr: the uniform values)
  const T* beta_v;
};

template <class T> __device__ inline T WaveSumT(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <class T> __device__ inline T ScaledZoneOneT(T xi, T lam, T alpha, T beta, T M) {
  // reference prox/scaled_zone.cc:90-101 with C = 0
  if (fabs(xi) <= M) return xi;
  if (xi > M + lam * alpha) return xi - lam * alpha;
  if (xi < -M - lam * beta) return xi + lam * beta;
  if (xi > T(0)) return M;
  return -M;
}

// One column's elementwise chain (see ChainOne below for the line-by-line correspondence).
template <class T>
__device__ inline T ChainOneT(T d, const FusedScalarsT<T>& c, T u, T y0p, T y1p, T* x0o, T* x1o,
                              T* y0o, T* y1o, T* uo) {
  const T v0 = ((u - y0p) - y1p) + y0p;
  const T x0 = c.kappa * d + v0;
  const T y0 = x0;
  const T u1 = v0 - y0;
  const T u2 = u1 + y1p;
  const T vin = c.Bs * u2;
  const T xz = ScaledZoneOneT<T>(vin, c.lam, c.alpha, c.beta, c.M);
  const T x1 = c.Cs * xz;
  const T y1 = c.a1 * x1;
  const T u3 = u2 - y1;
  *x0o = x0;
  *x1o = x1;
  *y0o = y0;
  *y1o = y1;
  *uo = u3;
  return ((u3 - y0) - y1) + y0;
}

// The two-block driver's column (see ChainTwoBlock below for the line-by-line correspondence).
template <class T>
__device__ inline T ChainTwoBlockT(T d, const FusedScalarsT<T>& c, T z0p, T z1p, T u0p, T u1p, T* x0o,
                                   T* x1o, T* z0o, T* z1o, T* u0o, T* u1o) {
  const T v0 = z0p - u0p;
  const T v1 = z1p - u1p;
  const T x0 = c.kappa * d + v0;
  const T x1 = c.Cs * ScaledZoneOneT<T>(c.Bs * v1, c.lam, c.alpha, c.beta, c.M);
  const T w0 = x0 + u0p;
  const T w1 = x1 + u1p;
  const T t = (c.a0 * w0 + c.a1 * w1) * c.inv_aa;
  const T z0 = w0 - c.a0 * t;
  const T z1 = w1 - c.a1 * t;
  const T u0 = u0p + (x0 - z0);
  const T u1 = u1p + (x1 - z1);
  *x0o = x0;
  *x1o = x1;
  *z0o = z0;
  *z1o = z1;
  *u0o = u0;
  *u1o = u1;
  return z0 - u0;
}

// f64 pass: 16-byte loads hold two rows, so 512 threads x 10 chunks own up to 10240 rows.
// MODE 0 / 1 as in the f32 kernel below: the multi-block driver's chain, the two-block driver's.
template <int NR, int BS, int MODE>
__global__ __launch_bounds__(BS, 2) void LassoFusedStreamKernelF64(
    int64_t m, int64_t n, const double* __restrict__ A, int64_t lda, const double* __restrict__ w,
    FusedScalarsT<double> c, double* u, double* x0, double* x1, double* y0, double* y1,
    double* y1prev, double* __restrict__ tpart, unsigned* epoch, double* e0, double* e1) {
  __shared__ double red[2][BS / 64];
  // the exchange kernels behind this pass tag their granules with the sweep number
  if (epoch != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *epoch += 1u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double2 wv[NR], tp[NR];
  int64_t row[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    row[q] = (static_cast<int64_t>(q) * BS + tid) * 2;
    wv[q] = row[q] < m ? *reinterpret_cast<const double2*>(w + row[q]) : make_double2(0, 0);
    tp[q] = make_double2(0, 0);
  }
  const int64_t npairs = (n + 1) / 2;
  auto column = [&](int64_t step) -> int64_t {
    const int64_t jp = blockIdx.x + (step >> 1) * gridDim.x;
    const int64_t j = 2 * jp + (step & 1);
    return (jp < npairs && j < n) ? j : -1;
  };
  auto load = [&](double2 (&a)[NR], int64_t j) {
    const double* cp = A + j * lda;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      if (row[q] < m) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(cp + row[q]));
        a[q] = make_double2(v.x, v.y);
      } else {
        a[q] = make_double2(0, 0);
      }
    }
  };
  double2 cur[NR], nxt[NR];
  int64_t step = 0;
  int64_t j = column(0);
  if (j >= 0) load(cur, j);
  int par = 0;
  while (j >= 0) {
    int64_t jn = column(step + 1);
    if (jn < 0 && ((step + 1) & 1)) jn = column(step + 2);
    const int64_t step_n = (jn >= 0 && column(step + 1) < 0) ? step + 2 : step + 1;
    if (jn >= 0) load(nxt, jn);
    const double uj = u[j], y0j = y0[j], y1j = y1[j];
    double u1j = 0.0;
    if (MODE == 1) u1j = e0[j];
    double d = 0.0;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      d += cur[q].x * wv[q].x;
      d += cur[q].y * wv[q].y;
    }
    d = WaveSumT<double>(d);
    if (lane == 0) red[par][wave] = d;
    __syncthreads();
    d = red[par][0];
#pragma unroll
    for (int w2 = 1; w2 < BS / 64; ++w2) d += red[par][w2];
    par ^= 1;
    FusedScalarsT<double> cj = c;
    if (c.alpha_v != nullptr) cj.alpha = c.alpha_v[j];
    if (c.beta_v != nullptr) cj.beta = c.beta_v[j];
    double v0n;
    if (MODE == 0) {
      double nx0, nx1, ny0, ny1, nu;
      v0n = ChainOneT<double>(d, cj, uj, y0j, y1j, &nx0, &nx1, &ny0, &ny1, &nu);
      if (tid == 0) {
        y1prev[j] = y1j;
        x0[j] = nx0;
        x1[j] = nx1;
        y0[j] = ny0;
        y1[j] = ny1;
        u[j] = nu;
      }
    } else {
      double nx0, nx1, nz0, nz1, nu0, nu1;
      v0n = ChainTwoBlockT<double>(d, cj, y0j, y1j, uj, u1j, &nx0, &nx1, &nz0, &nz1, &nu0, &nu1);
      if (tid == 0) {
        y1prev[j] = y0j;  // z_prev
        e1[j] = y1j;
        x0[j] = nx0;
        x1[j] = nx1;
        y0[j] = nz0;
        y1[j] = nz1;
        u[j] = nu0;
        e0[j] = nu1;
      }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      tp[q].x += cur[q].x * v0n;
      tp[q].y += cur[q].y * v0n;
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) cur[q] = nxt[q];
    j = jn;
    step = step_n;
  }
  double* out = tpart + static_cast<int64_t>(blockIdx.x) * m;
#pragma unroll
  for (int q = 0; q < NR; ++q)
    if (row[q] < m) *reinterpret_cast<double2*>(out + row[q]) = tp[q];
}

// One column's elementwise chain.  Returns v0' (input of the next sweep's forward pass).
__device__ inline float ChainOne(float d, const FusedScalars& c, float u, float y0p, float y1p,
                                 float* x0o, float* x1o, float* y0o, float* y1o, float* uo) {
  // sweep start: u -= y0; u -= y1; then term 0: u += y0       (prox_admm.cc:137-142)
  float v0 = ((u - y0p) - y1p) + y0p;
  float x0 = c.kappa * d + v0;        // back substitution epilogue: alpha*acc + 1*y
  float y0 = x0;                       // y_0 = A_ x_0 with A_(c0,x) = I
  float u1 = v0 - y0;                  // u -= y_0
  float u2 = u1 + y1p;                 // term 1: u += y_1
  float vin = c.Bs * u2;               // VectorProx: B v (+ g = 0)      (vector_prox.cc:141)
  float xz = ScaledZoneOne(vin, c.lam, c.alpha, c.beta, c.M);
  float x1 = c.Cs * xz;                // C (x - g)                       (vector_prox.cc:145)
  float y1 = c.a1 * x1;                // y_1 = A_ x_1
  float u3 = u2 - y1;                  // u -= y_1
  *x0o = x0;
  *x1o = x1;
  *y0o = y0;
  *y1o = y1;
  *uo = u3;
  // next sweep's prox-0 input
  return ((u3 - y0) - y1) + y0;
}

__device__ inline float ChainTwoBlock(float d, const FusedScalars& c, float z0p, float z1p, float u0p,
                                      float u1p, float* x0o, float* x1o, float* z0o, float* z1o,
                                      float* u0o, float* u1o) {
  const float v0 = z0p - u0p;
  const float v1 = z1p - u1p;
  const float x0 = c.kappa * d + v0;
  const float x1 = c.Cs * ScaledZoneOne(c.Bs * v1, c.lam, c.alpha, c.beta, c.M);
  const float w0 = x0 + u0p;
  const float w1 = x1 + u1p;
  const float t = (c.a0 * w0 + c.a1 * w1) * c.inv_aa;
  const float z0 = w0 - c.a0 * t;
  const float z1 = w1 - c.a1 * t;
  const float u0 = u0p + (x0 - z0);
  const float u1 = u1p + (x1 - z1);
  *x0o = x0;
  *x1o = x1;
  *z0o = z0;
  *z1o = z1;
  *u0o = u0;
  *u1o = u1;
  return z0 - u0;
}

template <int NR>
__global__ __launch_bounds__(kBlock, 2) void LassoFusedKernel(
    int64_t m, int64_t n, const float* __restrict__ A, int64_t lda, const float* __restrict__ w,
    FusedScalars c, float* u, float* x0, float* x1, float* y0, float* y1, float* y1prev,
    float* __restrict__ tpart, unsigned* epoch) {
  __shared__ float red[2][kBlock / 64][2];
  if (epoch != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *epoch += 1u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4 wv[NR], tp[NR];
  int64_t row[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    row[q] = (static_cast<int64_t>(q) * kBlock + tid) * 4;
    wv[q] = row[q] < m ? *reinterpret_cast<const float4*>(w + row[q]) : make_float4(0, 0, 0, 0);
    tp[q] = make_float4(0, 0, 0, 0);
  }
  const int64_t npairs = (n + 1) / 2;
  int par = 0;
  for (int64_t jp = blockIdx.x; jp < npairs; jp += gridDim.x, par ^= 1) {
    const int64_t j = 2 * jp;
    const bool has2 = j + 1 < n;
    const float* c0p = A + j * lda;
    const float* c1p = c0p + lda;
    float4 a0[NR], a1[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q)
      a0[q] = row[q] < m ? *reinterpret_cast<const float4*>(c0p + row[q]) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NR; ++q)
      a1[q] = (has2 && row[q] < m) ? *reinterpret_cast<const float4*>(c1p + row[q])
                                   : make_float4(0, 0, 0, 0);
    // per-column state (same address in every lane: broadcast loads)
    const float uj0 = u[j], y0j0 = y0[j], y1j0 = y1[j];
    const int64_t j1 = has2 ? j + 1 : j;
    const float uj1 = u[j1], y0j1 = y0[j1], y1j1 = y1[j1];

    float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      d0 += a0[q].x * wv[q].x;
      d0 += a0[q].y * wv[q].y;
      d0 += a0[q].z * wv[q].z;
      d0 += a0[q].w * wv[q].w;
      d1 += a1[q].x * wv[q].x;
      d1 += a1[q].y * wv[q].y;
      d1 += a1[q].z * wv[q].z;
      d1 += a1[q].w * wv[q].w;
    }
    d0 = WaveSumF(d0);
    d1 = WaveSumF(d1);
    if (lane == 0) {
      red[par][wave][0] = d0;
      red[par][wave][1] = d1;
    }
    __syncthreads();
    d0 = ((red[par][0][0] + red[par][1][0]) + red[par][2][0]) + red[par][3][0];
    d1 = ((red[par][0][1] + red[par][1][1]) + red[par][2][1]) + red[par][3][1];

    float nx0, nx1, ny0, ny1, nu;
    const float v0n0 = ChainOne(d0, c, uj0, y0j0, y1j0, &nx0, &nx1, &ny0, &ny1, &nu);
    if (tid == 0) {
      y1prev[j] = y1j0;
      x0[j] = nx0;
      x1[j] = nx1;
      y0[j] = ny0;
      y1[j] = ny1;
      u[j] = nu;
    }
    float v0n1 = 0.0f;
    if (has2) {
      v0n1 = ChainOne(d1, c, uj1, y0j1, y1j1, &nx0, &nx1, &ny0, &ny1, &nu);
      if (tid == 0) {
        y1prev[j + 1] = y1j1;
        x0[j + 1] = nx0;
        x1[j + 1] = nx1;
        y0[j + 1] = ny0;
        y1[j + 1] = ny1;
        u[j + 1] = nu;
      }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      tp[q].x += a0[q].x * v0n0;
      tp[q].y += a0[q].y * v0n0;
      tp[q].z += a0[q].z * v0n0;
      tp[q].w += a0[q].w * v0n0;
      tp[q].x += a1[q].x * v0n1;
      tp[q].y += a1[q].y * v0n1;
      tp[q].z += a1[q].z * v0n1;
      tp[q].w += a1[q].w * v0n1;
    }
  }
  float* out = tpart + static_cast<int64_t>(blockIdx.x) * m;
#pragma unroll
  for (int q = 0; q < NR; ++q)
    if (row[q] < m) *reinterpret_cast<float4*>(out + row[q]) = tp[q];
}

// Streaming variant: ONE column per step, the next column already in flight.  The pair kernel
// above issues 2*NR loads, waits for all of them and has nothing in flight while it reduces,
// synchronises and runs the chain; here the loads of column j+1 are issued before the dot
// product of column j is reduced, so every workgroup keeps NR 16-byte loads per lane outstanding
// at all times.  Same arithmetic per column, bit-identical state.
// BS threads own the m rows: 256 (two workgroups per CU) up to m = 10240; 512 (one workgroup of 8
// waves per CU) up to m = 20480 (6.37 TB/s on 2e4 x 5e4).
// MODE 0: the multi-block driver's chain (ChainOne).  MODE 1: the two-block driver's (ChainTwoBlock);
// the state arrays then mean u -> u0, y0 -> z0, y1 -> z1, y1prev -> z0_prev, e0 -> u1, e1 -> z1_prev.
template <int NR, int BS, int MODE>
__global__ __launch_bounds__(BS, 2) void LassoFusedStreamKernel(
    int64_t m, int64_t n, const float* __restrict__ A, int64_t lda, const float* __restrict__ w,
    FusedScalars c, float* u, float* x0, float* x1, float* y0, float* y1, float* y1prev,
    float* __restrict__ tpart, unsigned* epoch, float* e0, float* e1) {
  constexpr int kBlock = BS;  // (shadows the file-level constant inside this kernel)
  __shared__ float red[2][kBlock / 64];
  // sweep counter of the peer exchange (kernels_peer.hip): the two exchange kernels that follow
  // this pass in stream order tag their granules with it
  if (epoch != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *epoch += 1u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4 wv[NR], tp[NR];
  int64_t row[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    row[q] = (static_cast<int64_t>(q) * kBlock + tid) * 4;
    wv[q] = row[q] < m ? *reinterpret_cast<const float4*>(w + row[q]) : make_float4(0, 0, 0, 0);
    tp[q] = make_float4(0, 0, 0, 0);
  }
  // this workgroup's columns: the same pairs the pair kernel would take, one column at a time
  const int64_t npairs = (n + 1) / 2;
  auto column = [&](int64_t step) -> int64_t {  // step -> column index, or -1 past the end
    const int64_t jp = blockIdx.x + (step >> 1) * gridDim.x;
    const int64_t j = 2 * jp + (step & 1);
    return (jp < npairs && j < n) ? j : -1;
  };
  // The matrix is read once per sweep with no reuse: non-temporal loads keep it from evicting
  // the cached inverse (K3 operand, 200 MB of tiles) from the 256 MB Infinity Cache.
  static const bool kNT = true;
  auto load = [&](float4 (&a)[NR], int64_t j) {
    const float* cp = A + j * lda;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      if (row[q] < m) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        if (kNT) {
          const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(cp + row[q]));
          a[q] = make_float4(v.x, v.y, v.z, v.w);
        } else {
          a[q] = *reinterpret_cast<const float4*>(cp + row[q]);
        }
      } else {
        a[q] = make_float4(0, 0, 0, 0);
      }
    }
  };
  float4 cur[NR], nxt[NR];
  int64_t step = 0;
  int64_t j = column(0);
  if (j >= 0) load(cur, j);
  int par = 0;
  while (j >= 0) {
    // skip the odd slot of a trailing unpaired column without breaking the sequence
    int64_t jn = column(step + 1);
    if (jn < 0 && ((step + 1) & 1)) jn = column(step + 2);
    const int64_t step_n = (jn >= 0 && column(step + 1) < 0) ? step + 2 : step + 1;
    if (jn >= 0) load(nxt, jn);
    const float uj = u[j], y0j = y0[j], y1j = y1[j];
    float u1j = 0.0f;
    if (MODE == 1) u1j = e0[j];
    float d = 0.0f;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      d += cur[q].x * wv[q].x;
      d += cur[q].y * wv[q].y;
      d += cur[q].z * wv[q].z;
      d += cur[q].w * wv[q].w;
    }
    d = WaveSumF(d);
    if (lane == 0) red[par][wave] = d;
    __syncthreads();
    d = ((red[par][0] + red[par][1]) + red[par][2]) + red[par][3];
#pragma unroll
    for (int wv2 = 4; wv2 < kBlock / 64; ++wv2) d += red[par][wv2];
    par ^= 1;
    FusedScalars cj = c;
    if (c.alpha_v != nullptr) cj.alpha = c.alpha_v[j];
    if (c.beta_v != nullptr) cj.beta = c.beta_v[j];
    float v0n;
    if (MODE == 0) {
      float nx0, nx1, ny0, ny1, nu;
      v0n = ChainOne(d, cj, uj, y0j, y1j, &nx0, &nx1, &ny0, &ny1, &nu);
      if (tid == 0) {
        y1prev[j] = y1j;
        x0[j] = nx0;
        x1[j] = nx1;
        y0[j] = ny0;
        y1[j] = ny1;
        u[j] = nu;
      }
    } else {
      float nx0, nx1, nz0, nz1, nu0, nu1;
      v0n = ChainTwoBlock(d, cj, y0j, y1j, uj, u1j, &nx0, &nx1, &nz0, &nz1, &nu0, &nu1);
      if (tid == 0) {
        y1prev[j] = y0j;  // z_prev
        e1[j] = y1j;
        x0[j] = nx0;
        x1[j] = nx1;
        y0[j] = nz0;
        y1[j] = nz1;
        u[j] = nu0;
        e0[j] = nu1;
      }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      tp[q].x += cur[q].x * v0n;
      tp[q].y += cur[q].y * v0n;
      tp[q].z += cur[q].z * v0n;
      tp[q].w += cur[q].w * v0n;
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) cur[q] = nxt[q];
    j = jn;
    step = step_n;
  }
  float* out = tpart + static_cast<int64_t>(blockIdx.x) * m;
#pragma unroll
  for (int q = 0; q < NR; ++q)
    if (row[q] < m) *reinterpret_cast<float4*>(out + row[q]) = tp[q];
}

template <int NR, int BS>
void LaunchFused(int grid, int64_t m, int64_t n, const float* A, int64_t lda, const float* w,
                 const FusedScalars& c, float* u, float* x0, float* x1, float* y0, float* y1,
                 float* y1prev, float* tpart, unsigned* epoch, int chain, float* e0, float* e1) {
  if (chain == 1) {
    hipLaunchKernelGGL((LassoFusedStreamKernel<NR, BS, 1>), dim3(grid), dim3(BS), 0,
                       Runtime::Get().stream(), m, n, A, lda, w, c, u, x0, x1, y0, y1, y1prev, tpart,
                       epoch, e0, e1);
    return;
  }
  // default: the streaming kernel (6.0 vs 5.75 TB/s on the 1e4 x 5e4 matrix); "pair" selects the
  // two-column form (256-thread workgroups only)
  static const char* env = std::getenv("EPSILON_HIP_FUSED_KERNEL");
  const bool stream = !(env && env[0] == 'p') || BS != 256 || c.alpha_v != nullptr || c.beta_v != nullptr;
  if (stream) {
    hipLaunchKernelGGL((LassoFusedStreamKernel<NR, BS, 0>), dim3(grid), dim3(BS), 0,
                       Runtime::Get().stream(), m, n, A, lda, w, c, u, x0, x1, y0, y1, y1prev,
                       tpart, epoch, e0, e1);
    return;
  }
  if constexpr (BS == 256)
    hipLaunchKernelGGL(LassoFusedKernel<NR>, dim3(grid), dim3(kBlock), 0, Runtime::Get().stream(), m,
                       n, A, lda, w, c, u, x0, x1, y0, y1, y1prev, tpart, epoch);
}

}  // namespace

namespace {

// The five sums of squares of a residual check of the fused structure in ONE launch (the generic
// path takes ~10 launches and three temporaries for them): ||y0||^2, ||y1||^2, ||y0 + y1||^2,
// ||y1 - y1prev||^2, ||u||^2, accumulated in double.  Every workgroup reduces a contiguous part
// and publishes five partials; the workgroup whose ticket comes last adds the partials in
// workgroup order - deterministic, one launch.  (Hand-off: sc1 stores, every storing wave drained,
// one agent-scope ticket per workgroup; the last arriver reads with sc1 loads - MI355X_MICROARCH.md,
// Valid forms, first row.)
constexpr int kNormBlocks = 64;

__device__ inline double WaveSumD(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <class T>
__global__ __launch_bounds__(kBlock) void LassoFusedNormsKernel(
    int64_t n, const T* __restrict__ u, const T* __restrict__ y0, const T* __restrict__ y1,
    const T* __restrict__ y1prev, double* partial, unsigned* ticket, double* out,
    const unsigned* peer_err) {
  __shared__ double red[kBlock / 64][5];
  __shared__ bool last;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per;
  int64_t hi = lo + per;
  if (hi > n) hi = n;
  double s[5] = {0, 0, 0, 0, 0};
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
    const double a = y0[i], b = y1[i], c = y1prev[i], d = u[i];
    s[0] += a * a;
    s[1] += b * b;
    s[2] += (a + b) * (a + b);
    s[3] += (b - c) * (b - c);
    s[4] += d * d;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double t = WaveSumD(s[k]);
    if (lane == 0) red[wave][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    const double t = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    __hip_atomic_store(partial + blockIdx.x * 5 + k, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = prev == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    double t = 0;
    for (unsigned b = 0; b < gridDim.x; ++b)
      t += __hip_atomic_load(partial + b * 5 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[k] = t;
  }
  // sixth value: "an exchange kernel of this rank timed out" - it travels with the norms through
  // the all-reduce of the check, so every rank learns of a failure on ANY rank at the same check
  if (threadIdx.x == 5)
    out[5] = (peer_err != nullptr &&
              __hip_atomic_load(peer_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1.0 : 0.0;
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

void LassoFusedNorms(const DVec& u, const DVec& y0, const DVec& y1, const DVec& y1prev, double* out6,
                     const DVec& work, const unsigned* peer_err) {
  const int64_t n = u.n;
  EPS_CHECK(y0.n == n && y1.n == n && y1prev.n == n && y0.dt == u.dt && y1.dt == u.dt && y1prev.dt == u.dt);
  EPS_CHECK(work.dt == F64 && work.n >= kNormBlocks * 5 + 1);
  int64_t grid = (n + 4 * kBlock - 1) / (4 * kBlock);
  if (grid > kNormBlocks) grid = kNormBlocks;
  if (grid < 1) grid = 1;
  double* partial = work.as<double>();
  unsigned* ticket = reinterpret_cast<unsigned*>(partial + kNormBlocks * 5);  // zero between launches
  ProfScope prof("lasso_fused_norms", n);
  if (u.dt == F32)
    hipLaunchKernelGGL(LassoFusedNormsKernel<float>, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                       Runtime::Get().stream(), n, u.as<float>(), y0.as<float>(), y1.as<float>(),
                       y1prev.as<float>(), partial, ticket, out6, peer_err);
  else
    hipLaunchKernelGGL(LassoFusedNormsKernel<double>, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                       Runtime::Get().stream(), n, u.as<double>(), y0.as<double>(), y1.as<double>(),
                       y1prev.as<double>(), partial, ticket, out6, peer_err);
  EPS_HIP(hipGetLastError());
}

bool LassoFusedSupported(int64_t m, int64_t n, const DVec& A, int64_t lda) {
  if (n < 1 || reinterpret_cast<uintptr_t>(A.data()) % 16 != 0) return false;
  if (A.dt == F32) return m >= 4 && m % 4 == 0 && lda % 4 == 0 && m <= 20 * 1024;
  return m >= 2 && m % 2 == 0 && lda % 2 == 0 && m <= 10 * 1024;  // f64: two rows per 16-byte load
}

// Threads per workgroup of the pass: 512 when the rows do not fit 256 threads (m > 10240).
// Measured on MI355X for m = 1e4 (EPSILON_HIP_FUSED_BLOCK=512 forces the wide form): one 512-thread
// workgroup per CU is SLOWER than two of 256 (61.9 vs 48.4 us on a 6272-column slab, 413 vs 310 us
// on 5e4 columns - one column in flight per CU instead of two), two per CU are equal; the halved
// number of partial vectors does not pay either, the reduce-exchange kernel behind the pass is
// latency-bound (11.5 us at 242 partials, 12.5 at 448).
int LassoFusedBlock(int64_t m, int64_t n, DType dt) {
  (void)n;
  if (dt == F64) return m > 5 * 1024 ? 512 : 256;
  if (m > 10 * 1024) return 512;
  static const char* env = std::getenv("EPSILON_HIP_FUSED_BLOCK");
  if (env && std::atoi(env) == 512 && m >= 2048) return 512;
  return 256;
}

int LassoFusedGrid(int64_t m, int64_t n, DType dt) {
  int64_t npairs = (n + 1) / 2;
  // two 256-thread workgroups per CU; the 512-thread form with up to 5 row chunks per thread
  // (m <= 10240 in f32) also fits twice, above that once
  const int64_t rows_per_chunk = dt == F32 ? 4 : 2;
  const bool one_per_cu = LassoFusedBlock(m, n, dt) == 512 && m > 512 * 5 * rows_per_chunk;
  int64_t g = one_per_cu ? 256 : 512;
  static const char* env = std::getenv("EPSILON_HIP_FUSED_GRID");  // tuning knob
  if (env && std::atoi(env) > 0) g = std::atoi(env);
  if (g > npairs) g = npairs;
  if (g < 1) g = 1;
  // equal shares: with `per` column pairs per workgroup, ceil(npairs / per) workgroups leave no
  // workgroup a pair short (a 6272-column slab on 512 workgroups is 6 or 7 pairs each - the
  // launch then lasts as long as the 7s; on 448 workgroups every one has 7)
  const int64_t per = (npairs + g - 1) / g;
  g = (npairs + per - 1) / per;
  return static_cast<int>(g < 1 ? 1 : g);
}

namespace {
void LassoFusedPassF64(const LassoFusedArgs& a, int grid, int block) {
  FusedScalarsT<double> c{a.kappa, a.Bs, a.Cs, a.a1, a.lam, a.sz_alpha, a.sz_beta, a.sz_M,
                          a.a0, 1.0 / (a.a0 * a.a0 + a.a1 * a.a1), nullptr, nullptr};
  if (a.sz_alpha_vec.n > 0) {
    EPS_CHECK(a.sz_alpha_vec.n == a.n && a.sz_alpha_vec.dt == F64);
    c.alpha_v = a.sz_alpha_vec.as<double>();
  }
  if (a.sz_beta_vec.n > 0) {
    EPS_CHECK(a.sz_beta_vec.n == a.n && a.sz_beta_vec.dt == F64);
    c.beta_v = a.sz_beta_vec.as<double>();
  }
  if (a.chain == 1) EPS_CHECK(a.e0.n == a.n && a.e1.n == a.n && a.e0.dt == F64 && a.e1.dt == F64);
  double* e0 = a.chain == 1 ? a.e0.as<double>() : nullptr;
  double* e1 = a.chain == 1 ? a.e1.as<double>() : nullptr;
  ProfScope prof("lasso_fused", a.m, a.n);
  const int64_t need = (a.m + 2 * block - 1) / (2 * block);  // double2 row chunks per thread
  hipStream_t s = Runtime::Get().stream();
#define EPS_FUSED_CASE64(NRV, BSV)                                                                        \
  do {                                                                                                    \
    if (a.chain == 1)                                                                                     \
      hipLaunchKernelGGL((LassoFusedStreamKernelF64<NRV, BSV, 1>), dim3(grid), dim3(BSV), 0, s, a.m, a.n, \
                         a.A.as<double>(), a.lda, a.w.as<double>(), c, a.u.as<double>(),                  \
                         a.x0.as<double>(), a.x1.as<double>(), a.y0.as<double>(), a.y1.as<double>(),      \
                         a.y1prev.as<double>(), a.tpart.as<double>(), a.epoch, e0, e1);                   \
    else                                                                                                  \
      hipLaunchKernelGGL((LassoFusedStreamKernelF64<NRV, BSV, 0>), dim3(grid), dim3(BSV), 0, s, a.m, a.n, \
                         a.A.as<double>(), a.lda, a.w.as<double>(), c, a.u.as<double>(),                  \
                         a.x0.as<double>(), a.x1.as<double>(), a.y0.as<double>(), a.y1.as<double>(),      \
                         a.y1prev.as<double>(), a.tpart.as<double>(), a.epoch, e0, e1);                   \
  } while (0)
  if (block == 256) {
    if (need <= 1) EPS_FUSED_CASE64(1, 256);
    else if (need <= 2) EPS_FUSED_CASE64(2, 256);
    else if (need <= 4) EPS_FUSED_CASE64(4, 256);
    else if (need <= 8) EPS_FUSED_CASE64(8, 256);
    else EPS_FUSED_CASE64(10, 256);
  } else {
    if (need <= 8) EPS_FUSED_CASE64(8, 512);
    else EPS_FUSED_CASE64(10, 512);
  }
#undef EPS_FUSED_CASE64
  EPS_HIP(hipGetLastError());
}
}  // namespace

void LassoFusedPass(const LassoFusedArgs& a) {
  EPS_CHECK(LassoFusedSupported(a.m, a.n, a.A, a.lda));
  const DType dt = a.A.dt;
  EPS_CHECK(a.w.n == a.m && a.w.dt == dt);
  for (const DVec* v : {&a.u, &a.x0, &a.x1, &a.y0, &a.y1, &a.y1prev})
    EPS_CHECK(v->n == a.n && v->dt == dt);
  const int grid = LassoFusedGrid(a.m, a.n, dt);
  const int block = LassoFusedBlock(a.m, a.n, dt);
  EPS_CHECK(a.tpart.n >= static_cast<int64_t>(grid) * a.m && a.tpart.dt == dt);
  if (dt == F64) {
    LassoFusedPassF64(a, grid, block);
    return;
  }
  EPS_CHECK(reinterpret_cast<uintptr_t>(a.w.data()) % 16 == 0 &&
            reinterpret_cast<uintptr_t>(a.tpart.data()) % 16 == 0);
  FusedScalars c;
  c.kappa = static_cast<float>(a.kappa);
  c.Bs = static_cast<float>(a.Bs);
  c.Cs = static_cast<float>(a.Cs);
  c.a1 = static_cast<float>(a.a1);
  c.lam = static_cast<float>(a.lam);
  c.alpha = static_cast<float>(a.sz_alpha);
  c.beta = static_cast<float>(a.sz_beta);
  c.M = static_cast<float>(a.sz_M);
  c.a0 = static_cast<float>(a.a0);
  c.inv_aa = static_cast<float>(1.0 / (a.a0 * a.a0 + a.a1 * a.a1));
  c.alpha_v = c.beta_v = nullptr;
  if (a.sz_alpha_vec.n > 0) {
    EPS_CHECK(a.sz_alpha_vec.n == a.n && a.sz_alpha_vec.dt == F32);
    c.alpha_v = a.sz_alpha_vec.as<float>();
  }
  if (a.sz_beta_vec.n > 0) {
    EPS_CHECK(a.sz_beta_vec.n == a.n && a.sz_beta_vec.dt == F32);
    c.beta_v = a.sz_beta_vec.as<float>();
  }
  // (the pair kernel has no per-column parameters: the streaming kernel is forced below)
  if (a.chain == 1) EPS_CHECK(a.e0.n == a.n && a.e1.n == a.n && a.e0.dt == F32 && a.e1.dt == F32);
  ProfScope prof("lasso_fused", a.m, a.n);
  const int64_t need = (a.m + 4 * block - 1) / (4 * block);  // float4 row chunks per thread
#define EPS_FUSED_CASE(NRV, BSV)                                                                      \
  LaunchFused<NRV, BSV>(grid, a.m, a.n, a.A.as<float>(), a.lda, a.w.as<float>(), c, a.u.as<float>(), \
                        a.x0.as<float>(), a.x1.as<float>(), a.y0.as<float>(), a.y1.as<float>(),      \
                        a.y1prev.as<float>(), a.tpart.as<float>(), a.epoch, a.chain,         \
                        a.chain == 1 ? a.e0.as<float>() : nullptr, a.chain == 1 ? a.e1.as<float>() : nullptr)
  if (block == 256) {
    if (need <= 1) EPS_FUSED_CASE(1, 256);
    else if (need <= 2) EPS_FUSED_CASE(2, 256);
    else if (need <= 4) EPS_FUSED_CASE(4, 256);
    else if (need <= 8) EPS_FUSED_CASE(8, 256);
    else EPS_FUSED_CASE(10, 256);
  } else {
    if (need <= 2) EPS_FUSED_CASE(2, 512);
    else if (need <= 5) EPS_FUSED_CASE(5, 512);
    else if (need <= 8) EPS_FUSED_CASE(8, 512);
    else EPS_FUSED_CASE(10, 512);
  }
#undef EPS_FUSED_CASE
}

}  // namespace k
}  // namespace eps

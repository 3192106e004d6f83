// K1 / K2 / K3: dense matrix-vector products on a column-major matrix, HBM-bound.
//
// The reference does both through one `dgemv_` with a trans flag on a shared buffer
// (reference src/epsilon/linear/dense_matrix_impl.cc:55-67, dense_matrix_impl.h:41-43).
// Here the two directions are separate kernels because their parallel structure differs:
//
//  GemvN  y = alpha*A x + beta*y : a thread owns 4 (f32) / 2 (f64) consecutive rows and walks
//         a slab of columns; every load is 16 B per lane, fully coalesced down a column.
//         x for the slab sits in LDS and is read as a broadcast.  The column range is split
//         over blockIdx.y so the grid fills 256 CUs; the per-slab partial vectors are summed
//         in a fixed order by a second tiny kernel (deterministic, no float atomics).
//
//  GemvT  y = alpha*A^T x + beta*y : a wavefront (64 lanes) owns 4 columns at a time; lanes
//         stride down the column 16 B each, x comes from LDS (staged once per workgroup when
//         it fits), the 4 dot products are finished with wave shuffles.  No partials needed.
//
// Algorithmic bytes: rows*cols*sizeof(T) per call (A read exactly once); vectors are noise.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;

template <class T> struct VT;
template <> struct VT<float> {
  using type = float4;
  static constexpr int V = 4;
};
template <> struct VT<double> {
  using type = double2;
  static constexpr int V = 2;
};

template <class T, int V> struct Ld {
  __device__ static inline void load(T (&r)[V], const T* p) {
    using P = typename VT<T>::type;
    *reinterpret_cast<P*>(r) = *reinterpret_cast<const P*>(p);
  }
};
template <class T> struct Ld<T, 1> {
  __device__ static inline void load(T (&r)[1], const T* p) { r[0] = *p; }
};

// Matrix loads.  `nt` (workgroup-uniform): the matrix is larger than the last-level cache and is
// streamed once per call, so it is read with non-temporal loads and does not evict operands that
// ARE reused between sweeps (cached inverses, vectors).  Measured on the fused sweep: 5.75 ->
// 6.4+ TB/s for the stream itself and 66 -> 50 us for the inverse apply that follows.
template <class T, int V> struct LdA {
  __device__ static inline void load(T (&r)[V], const T* p, bool nt) {
    if (nt) {
      typedef T vec __attribute__((ext_vector_type(V)));
      const vec v = __builtin_nontemporal_load(reinterpret_cast<const vec*>(p));
#pragma unroll
      for (int i = 0; i < V; ++i) r[i] = v[i];
    } else {
      Ld<T, V>::load(r, p);
    }
  }
};
template <class T> struct LdA<T, 1> {
  __device__ static inline void load(T (&r)[1], const T* p, bool nt) {
    r[0] = nt ? __builtin_nontemporal_load(p) : *p;
  }
};
inline bool StreamedMatrix(int64_t rows, int64_t cols, size_t elem) {
  return static_cast<double>(rows) * static_cast<double>(cols) * elem > 512.0 * 1024 * 1024;
}

// ------------------------------------------------------------------------------------------
// GemvN
// ------------------------------------------------------------------------------------------
constexpr int kNChunk = 512;  // columns of x staged in LDS at a time
constexpr int kNUnroll = 8;   // column loads in flight per thread

template <class T, int V>
__global__ __launch_bounds__(kBlock) void GemvNKernel(int64_t rows, int64_t cols,
                                                      const T* __restrict__ A, int64_t lda,
                                                      const T* __restrict__ x, T alpha, T beta,
                                                      T* y, T* partial, int64_t cols_per_split, bool nt) {
  __shared__ T xs[kNChunk];
  const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * V;
  const int64_t c_begin = static_cast<int64_t>(blockIdx.y) * cols_per_split;
  int64_t c_end = c_begin + cols_per_split;
  if (c_end > cols) c_end = cols;
  const bool active = row0 < rows;  // V | rows is guaranteed by the launcher when V > 1

  T acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = T(0);

  for (int64_t c0 = c_begin; c0 < c_end; c0 += kNChunk) {
    const int nchunk = static_cast<int>((c_end - c0 < kNChunk) ? (c_end - c0) : kNChunk);
    __syncthreads();
    for (int j = threadIdx.x; j < nchunk; j += kBlock) xs[j] = x[c0 + j];
    __syncthreads();
    if (active) {
      const T* Ap = A + row0 + c0 * lda;
      int j = 0;
      for (; j + kNUnroll <= nchunk; j += kNUnroll) {
        T a[kNUnroll][V];
#pragma unroll
        for (int u = 0; u < kNUnroll; ++u) LdA<T, V>::load(a[u], Ap + (j + u) * lda, nt);
#pragma unroll
        for (int u = 0; u < kNUnroll; ++u) {
          const T xv = xs[j + u];
#pragma unroll
          for (int v = 0; v < V; ++v) acc[v] += a[u][v] * xv;
        }
      }
      for (; j < nchunk; ++j) {
        T a[V];
        LdA<T, V>::load(a, Ap + j * lda, nt);
        const T xv = xs[j];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] += a[v] * xv;
      }
    }
  }
  if (!active) return;
  if (gridDim.y == 1) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int64_t r = row0 + v;
      y[r] = (beta == T(0)) ? alpha * acc[v] : alpha * acc[v] + beta * y[r];
    }
  } else {
    T* p = partial + static_cast<int64_t>(blockIdx.y) * rows + row0;
#pragma unroll
    for (int v = 0; v < V; ++v) p[v] = acc[v];
  }
}

// y[r] = alpha * sum_k partial[k][r] + beta*y[r].  64 rows per workgroup, the splits are dealt
// to the 4 wavefronts and summed in a FIXED order (k ascending inside a wave, then wave 0..3),
// so the result is deterministic; loads are independent and unrolled so that many are in flight.
template <class T>
__global__ __launch_bounds__(kBlock) void GemvNReduceKernel(int64_t rows, int nsplit,
                                                            const T* __restrict__ partial,
                                                            T alpha, T beta, T* y,
                                                            const T* __restrict__ add) {
  __shared__ T part[kBlock / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 64 + lane;
  T s = T(0);
  if (r < rows) {
    const int per = (nsplit + 3) / 4;
    const int k0 = wave * per;
    int k1 = k0 + per;
    if (k1 > nsplit) k1 = nsplit;
    const T* p = partial + r;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[static_cast<int64_t>(k + u) * rows];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < k1; ++k) s += p[static_cast<int64_t>(k) * rows];
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && r < rows) {
    const T t = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    T out = (beta == T(0)) ? alpha * t : alpha * t + beta * y[r];
    if (add) out += add[r];
    y[r] = out;
  }
}

template <class T>
void LaunchGemvN(int64_t rows, int64_t cols, double alpha, const T* A, int64_t lda, const T* x,
                 double beta, T* y) {
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  constexpr int VV = VT<T>::V;
  const bool nt = StreamedMatrix(rows, cols, sizeof(T));
  const bool vec = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (lda % VV == 0) &&
                   (rows % VV == 0);
  const int V = vec ? VV : 1;
  const int64_t row_blocks = (rows + static_cast<int64_t>(kBlock) * V - 1) / (kBlock * V);
  // split columns so that the grid has >= ~1024 workgroups, slabs of >= 64 columns
  int64_t nsplit = (1024 + row_blocks - 1) / row_blocks;
  int64_t max_split = (cols + 63) / 64;
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  int64_t cps = (cols + nsplit - 1) / nsplit;
  nsplit = (cols + cps - 1) / cps;
  std::shared_ptr<Buffer> part;
  T* partial = nullptr;
  if (nsplit > 1) {
    part = rt.Alloc(static_cast<size_t>(nsplit) * rows * sizeof(T));
    partial = static_cast<T*>(part->p);
  }
  dim3 grid(static_cast<unsigned>(row_blocks), static_cast<unsigned>(nsplit));
  if (vec) {
    hipLaunchKernelGGL((GemvNKernel<T, VV>), grid, dim3(kBlock), 0, s, rows, cols, A, lda, x,
                       T(alpha), T(beta), y, partial, cps, nt);
  } else {
    hipLaunchKernelGGL((GemvNKernel<T, 1>), grid, dim3(kBlock), 0, s, rows, cols, A, lda, x,
                       T(alpha), T(beta), y, partial, cps, nt);
  }
  if (nsplit > 1) {
    hipLaunchKernelGGL(GemvNReduceKernel<T>, dim3(static_cast<unsigned>((rows + 63) / 64)),
                       dim3(kBlock), 0, s, rows, static_cast<int>(nsplit), partial, T(alpha),
                       T(beta), y, static_cast<const T*>(nullptr));
  }
}

// ------------------------------------------------------------------------------------------
// GemvT
// ------------------------------------------------------------------------------------------
constexpr int kTCols = 4;                 // columns per wave per pass
constexpr int kTLdsBytes = 48 * 1024;     // x staging per workgroup

template <class T> __device__ inline T WaveSumT(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <class T, int V>
__global__ __launch_bounds__(kBlock) void GemvTKernel(int64_t rows, int64_t cols,
                                                      const T* __restrict__ A, int64_t lda,
                                                      const T* __restrict__ x, T alpha, T beta,
                                                      T* y, bool nt) {
  constexpr int RC = kTLdsBytes / sizeof(T);  // rows of x held in LDS
  __shared__ __attribute__((aligned(16))) T xs[RC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kWaves = kBlock / 64;
  const int64_t cols_per_pass = kWaves * kTCols;
  const int64_t npass = (cols + cols_per_pass - 1) / cols_per_pass;
  const bool single_chunk = rows <= RC;

  if (single_chunk) {
    for (int64_t i = threadIdx.x; i < rows; i += kBlock) xs[i] = x[i];
    __syncthreads();
  }

  for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
    const int64_t j0 = pass * cols_per_pass + wave * kTCols;
    T acc[kTCols];
#pragma unroll
    for (int c = 0; c < kTCols; ++c) acc[c] = T(0);

    for (int64_t r0 = 0; r0 < rows; r0 += RC) {
      const int64_t nr = (rows - r0 < RC) ? (rows - r0) : RC;
      if (!single_chunk) {
        __syncthreads();
        for (int64_t i = threadIdx.x; i < nr; i += kBlock) xs[i] = x[r0 + i];
        __syncthreads();
      }
      if (j0 < cols) {
        const T* Ap = A + r0 + j0 * lda;
        const int64_t nvec = (V > 1) ? (nr / V) : 0;
#pragma unroll 2
        for (int64_t p = lane; p < nvec; p += 64) {
          T xv[V];
          Ld<T, V>::load(xv, xs + p * V);
#pragma unroll
          for (int c = 0; c < kTCols; ++c) {
            if (j0 + c < cols) {
              T a[V];
              LdA<T, V>::load(a, Ap + c * lda + p * V, nt);
#pragma unroll
              for (int v = 0; v < V; ++v) acc[c] += a[v] * xv[v];
            }
          }
        }
        for (int64_t i = nvec * V + lane; i < nr; i += 64) {
          const T xv = xs[i];
#pragma unroll
          for (int c = 0; c < kTCols; ++c)
            if (j0 + c < cols) acc[c] += Ap[c * lda + i] * xv;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kTCols; ++c) {
      T s = WaveSumT(acc[c]);
      if (lane == 0 && j0 + c < cols) {
        const int64_t j = j0 + c;
        y[j] = (beta == T(0)) ? alpha * s : alpha * s + beta * y[j];
      }
    }
  }
}

// GemvT, wide form (the hot one): the workgroup owns 8 columns per pass and ALL rows; a thread
// owns rows {V*tid + V*256*q}, keeps one partial dot product per column in registers and issues
// the 8 column loads (16 B each) back to back before any use, so 32 KiB per workgroup are in
// flight; x sits in LDS.  The 8 partials are reduced wave-wise by shuffles, across the 4
// wavefronts through LDS, in a fixed order.
constexpr int kT2Cols = 8;

template <class T, int V>
__global__ __launch_bounds__(kBlock) void GemvT2Kernel(int64_t rows, int64_t cols,
                                                       const T* __restrict__ A, int64_t lda,
                                                       const T* __restrict__ x, T alpha, T beta,
                                                       T* y, bool nt) {
  constexpr int RC = kTLdsBytes / sizeof(T);
  __shared__ __attribute__((aligned(16))) T xs[RC];
  __shared__ T red[kBlock / 64][kT2Cols];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t i = threadIdx.x; i < rows; i += kBlock) xs[i] = x[i];  // rows <= RC (launcher)
  __syncthreads();
  const int64_t npass = (cols + kT2Cols - 1) / kT2Cols;
  const int64_t nvec = rows / V;  // rows % V == 0 (launcher)
  for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
    const int64_t j0 = pass * kT2Cols;
    T acc[kT2Cols];
#pragma unroll
    for (int c = 0; c < kT2Cols; ++c) acc[c] = T(0);
    const T* Ap = A + j0 * lda;
    if (j0 + kT2Cols <= cols) {
      for (int64_t p = threadIdx.x; p < nvec; p += kBlock) {
        T a[kT2Cols][V];
#pragma unroll
        for (int c = 0; c < kT2Cols; ++c) LdA<T, V>::load(a[c], Ap + c * lda + p * V, nt);
        T xv[V];
        Ld<T, V>::load(xv, xs + p * V);
#pragma unroll
        for (int c = 0; c < kT2Cols; ++c)
#pragma unroll
          for (int v = 0; v < V; ++v) acc[c] += a[c][v] * xv[v];
      }
    } else {
      for (int64_t p = threadIdx.x; p < nvec; p += kBlock) {
        T xv[V];
        Ld<T, V>::load(xv, xs + p * V);
#pragma unroll
        for (int c = 0; c < kT2Cols; ++c) {
          if (j0 + c < cols) {
            T a[V];
            LdA<T, V>::load(a, Ap + c * lda + p * V, nt);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[c] += a[v] * xv[v];
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kT2Cols; ++c) {
      const T sum = WaveSumT(acc[c]);
      if (lane == 0) red[wave][c] = sum;
    }
    __syncthreads();
    if (threadIdx.x < kT2Cols && j0 + threadIdx.x < cols) {
      const int c = threadIdx.x;
      const T t = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
      const int64_t j = j0 + c;
      y[j] = (beta == T(0)) ? alpha * t : alpha * t + beta * y[j];
    }
    __syncthreads();
  }
}

template <class T>
void LaunchGemvT(int64_t rows, int64_t cols, double alpha, const T* A, int64_t lda, const T* x,
                 double beta, T* y) {
  hipStream_t s = Runtime::Get().stream();
  constexpr int VV = VT<T>::V;
  // measured: non-temporal loads slow the column-panel kernels down (0.343 -> 0.521 ms on
  // 1e4 x 5e4), unlike the row-streaming GemvN and the fused sweep; kept off here
  const bool nt = false;
  const bool vec = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (lda % VV == 0);
  constexpr int64_t RC = kTLdsBytes / sizeof(T);
  if (vec && rows % VV == 0 && rows <= RC && rows >= 256 * VV) {
    int64_t npass = (cols + kT2Cols - 1) / kT2Cols;
    int64_t grid = npass < 1024 ? npass : 1024;
    hipLaunchKernelGGL((GemvT2Kernel<T, VV>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                       s, rows, cols, A, lda, x, T(alpha), T(beta), y, nt);
    return;
  }
  const int64_t cols_per_pass = (kBlock / 64) * kTCols;
  int64_t npass = (cols + cols_per_pass - 1) / cols_per_pass;
  int64_t grid = npass < 1024 ? npass : 1024;
  if (grid < 1) grid = 1;
  if (vec) {
    hipLaunchKernelGGL((GemvTKernel<T, VV>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                       s, rows, cols, A, lda, x, T(alpha), T(beta), y, nt);
  } else {
    hipLaunchKernelGGL((GemvTKernel<T, 1>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, s,
                       rows, cols, A, lda, x, T(alpha), T(beta), y, nt);
  }
}

// ------------------------------------------------------------------------------------------
// Symv: y = alpha * S x + beta * y for a SYMMETRIC S held in full storage, reading only the
// tiles on and below the diagonal - half the bytes of a GEMV.  The cached explicit inverse of a
// least-squares prox (K3) is such a matrix and its apply is ~1/6 of a lasso sweep.
//
// One workgroup per 128 x 128 tile (I >= J).  A thread owns 16 rows (four 16-byte loads per
// column) of 4 columns.  Off-diagonal tiles contribute twice: y_I += T x_J (row sums, reduced
// over the 32 column groups through LDS) and y_J += T^T x_I (column sums, reduced over the 8
// row groups with three shuffle steps).  Both land in per-tile partial
// vectors that SymvReduceKernel adds in a fixed order, so the result is deterministic.
// ------------------------------------------------------------------------------------------
constexpr int kSB = 128;

// PACKED: S is the tile-packed copy made by SymvPack - tile `lin` is 128 x 128 values, column-major,
// contiguous (64 KB in f32), zero-padded at the matrix edge: a workgroup's 64 KB then come out of
// a few DRAM pages instead of 128 column pieces of 512 bytes each 4 lds bytes apart.
template <class T, bool PACKED>
__global__ __launch_bounds__(kBlock) void SymvTileKernel(int64_t n, const T* __restrict__ S,
                                                         int64_t lds, const T* __restrict__ x,
                                                         T* __restrict__ prow,
                                                         T* __restrict__ pcol) {
  __shared__ T xI[kSB], xJ[kSB];
  __shared__ T red[32][kSB];
  // tile (I, J) from the linear index over the lower triangle
  const int64_t lin = blockIdx.x;
  int64_t I = static_cast<int64_t>((sqrt(8.0 * static_cast<double>(lin) + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= lin) ++I;
  while (I * (I + 1) / 2 > lin) --I;
  const int64_t J = lin - I * (I + 1) / 2;
  const int64_t i0 = I * kSB, j0 = J * kSB;
  const int t = threadIdx.x;
  if (t < kSB) {
    xI[t] = (i0 + t < n) ? x[i0 + t] : T(0);
  } else {
    const int c = t - kSB;
    xJ[c] = (j0 + c < n) ? x[j0 + c] : T(0);
  }
  __syncthreads();
  // thread (rg, cg): rows 4*(rg + 8q) + v (q, v < 4) of columns cg + 32k (k < 4).  Column sums
  // then need only 3 shuffle steps over the 8 row groups; row sums go through LDS.
  const int rg = t & 7, cg = t >> 3;
  const bool full = PACKED || ((i0 + kSB <= n) && (j0 + kSB <= n) && (lds % 4 == 0) &&
                               (reinterpret_cast<uintptr_t>(S) % 16 == 0));
  T a[4][4][4];  // [k][q][v]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cg + 32 * k;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * (rg + 8 * q);
      const T* src = PACKED ? S + lin * (kSB * kSB) + r + c * kSB : S + (i0 + r) + (j0 + c) * lds;
      if (full) {
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(a[k][q]) = *reinterpret_cast<const float4*>(src);
        } else {
          *reinterpret_cast<double2*>(&a[k][q][0]) = *reinterpret_cast<const double2*>(src);
          *reinterpret_cast<double2*>(&a[k][q][2]) = *reinterpret_cast<const double2*>(src + 2);
        }
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          a[k][q][v] = (i0 + r + v < n && j0 + c < n) ? src[v] : T(0);
      }
    }
  }
  // row sums over this thread's 4 columns, then over the 32 column groups through LDS
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      T rs = T(0);
#pragma unroll
      for (int k = 0; k < 4; ++k) rs += a[k][q][v] * xJ[cg + 32 * k];
      red[cg][4 * (rg + 8 * q) + v] = rs;
    }
  }
  // column sums (off-diagonal tiles only) over this thread's 16 rows, then over the 8 row groups
  if (I != J) {
    T xr[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int v = 0; v < 4; ++v) xr[q][v] = xI[4 * (rg + 8 * q) + v];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      T cs = T(0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int v = 0; v < 4; ++v) cs += a[k][q][v] * xr[q][v];
      cs += __shfl_xor(cs, 1, 64);
      cs += __shfl_xor(cs, 2, 64);
      cs += __shfl_xor(cs, 4, 64);
      if (rg == 0) pcol[lin * kSB + cg + 32 * k] = cs;
    }
  }
  __syncthreads();
  if (t < kSB) {
    T s = red[0][t];
#pragma unroll
    for (int g = 1; g < 32; ++g) s += red[g][t];
    prow[lin * kSB + t] = s;
  }
}

// y[r] = alpha * (sum_{J <= I} prow[(I,J)][r] + sum_{I' > I} pcol[(I',I)][r]) + beta * y[r].
// One workgroup per block of 128 rows; the nb partial vectors of a block are dealt to two
// half-workgroups (even / odd position in the list), loaded eight at a time, and the two halves
// are added in a fixed order.
template <class T>
__global__ __launch_bounds__(kBlock) void SymvReduceKernel(int64_t n, int64_t nb,
                                                           const T* __restrict__ prow,
                                                           const T* __restrict__ pcol, T alpha,
                                                           T beta, T* y) {
  __shared__ T half[kSB];
  const int64_t I = blockIdx.x;
  const int off = threadIdx.x & (kSB - 1), grp = threadIdx.x >> 7;  // kSB == 128
  const int64_t base = I * (I + 1) / 2;
  auto part = [&](int64_t p) -> const T* {  // p-th partial vector of this block, p < nb
    return p <= I ? prow + (base + p) * kSB : pcol + (p * (p + 1) / 2 + I) * kSB;
  };
  T s = T(0);
  int64_t p = grp;
  for (; p + 14 < nb; p += 16) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part(p + 2 * u)[off];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; p < nb; p += 2) s += part(p)[off];
  if (grp == 1) half[off] = s;
  __syncthreads();
  const int64_t r = I * kSB + off;
  if (grp == 0 && r < n) {
    const T tot = s + half[off];
    y[r] = (beta == T(0)) ? alpha * tot : alpha * tot + beta * y[r];
  }
}

// tile-packed copy of the lower tiles of a symmetric matrix (see SymvTileKernel<T, true>)
template <class T>
__global__ __launch_bounds__(kBlock) void SymvPackKernel(int64_t n, const T* __restrict__ S, int64_t lds,
                                                         T* __restrict__ P) {
  const int64_t lin = blockIdx.x;
  int64_t I = static_cast<int64_t>((sqrt(8.0 * static_cast<double>(lin) + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= lin) ++I;
  while (I * (I + 1) / 2 > lin) --I;
  const int64_t J = lin - I * (I + 1) / 2, i0 = I * kSB, j0 = J * kSB;
  for (int e = threadIdx.x; e < kSB * kSB; e += kBlock) {
    const int r = e & (kSB - 1), c = e >> 7;  // kSB == 128
    P[lin * (kSB * kSB) + e] = (i0 + r < n && j0 + c < n) ? S[(i0 + r) + (j0 + c) * lds] : T(0);
  }
}

template <class T, bool PACKED = false>
void LaunchSymv(int64_t n, double alpha, const T* S, int64_t lds, const T* x, double beta, T* y,
                void* work) {
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  const int64_t nb = (n + kSB - 1) / kSB;
  const int64_t ntiles = nb * (nb + 1) / 2;
  std::shared_ptr<Buffer> buf;
  if (work == nullptr) {
    buf = rt.Alloc(static_cast<size_t>(2 * ntiles * kSB) * sizeof(T));
    work = buf->p;
  }
  T* prow = static_cast<T*>(work);
  T* pcol = prow + ntiles * kSB;
  hipLaunchKernelGGL((SymvTileKernel<T, PACKED>), dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, s,
                     n, S, lds, x, prow, pcol);
  static_assert(kSB == 128 && kBlock == 256, "SymvReduceKernel geometry");
  hipLaunchKernelGGL((SymvReduceKernel<T>), dim3(static_cast<unsigned>(nb)), dim3(kBlock), 0, s, n,
                     nb, prow, pcol, T(alpha), T(beta), y);
}

}  // namespace

int64_t SymvWorkspace(int64_t n) {
  const int64_t nb = (n + kSB - 1) / kSB;
  return 2 * (nb * (nb + 1) / 2) * kSB;
}

void Symv(int64_t n, double alpha, const DVec& S, int64_t lds, const DVec& x, double beta,
          const DVec& y, const DVec* work) {
  EPS_CHECK(S.dt == x.dt && S.dt == y.dt && x.n == n && y.n == n && lds >= n);
  EPS_CHECK_MSG(n == 0 || S.n >= (n - 1) * lds + n, "symv: matrix buffer too small");
  EPS_CHECK_MSG(x.data() != y.data(), "symv: x and y alias");
  if (n == 0) return;
  if (work) EPS_CHECK(work->dt == S.dt && work->n >= SymvWorkspace(n));
  ProfScope prof("symv", n);
  void* wp = work ? work->data() : nullptr;
  if (S.dt == F32) LaunchSymv<float>(n, alpha, S.as<float>(), lds, x.as<float>(), beta, y.as<float>(), wp);
  else LaunchSymv<double>(n, alpha, S.as<double>(), lds, x.as<double>(), beta, y.as<double>(), wp);
  EPS_HIP(hipGetLastError());
}

int64_t SymvPackedSize(int64_t n) {
  const int64_t nb = (n + kSB - 1) / kSB;
  return nb * (nb + 1) / 2 * kSB * kSB;
}

DVec SymvPack(int64_t n, const DVec& S, int64_t lds) {
  EPS_CHECK(lds >= n && (n == 0 || S.n >= (n - 1) * lds + n));
  DVec P = DVec::Empty(SymvPackedSize(n), S.dt);
  if (n == 0) return P;
  const int64_t nb = (n + kSB - 1) / kSB;
  const unsigned tiles = static_cast<unsigned>(nb * (nb + 1) / 2);
  hipStream_t s = Runtime::Get().stream();
  if (S.dt == F32)
    hipLaunchKernelGGL(SymvPackKernel<float>, dim3(tiles), dim3(kBlock), 0, s, n, S.as<float>(), lds, P.as<float>());
  else
    hipLaunchKernelGGL(SymvPackKernel<double>, dim3(tiles), dim3(kBlock), 0, s, n, S.as<double>(), lds, P.as<double>());
  EPS_HIP(hipGetLastError());
  return P;
}

void SymvPacked(int64_t n, double alpha, const DVec& P, const DVec& x, double beta, const DVec& y, const DVec* work) {
  EPS_CHECK(P.dt == x.dt && P.dt == y.dt && x.n == n && y.n == n && P.n >= SymvPackedSize(n));
  EPS_CHECK_MSG(x.data() != y.data(), "symv: x and y alias");
  if (n == 0) return;
  if (work) EPS_CHECK(work->dt == P.dt && work->n >= SymvWorkspace(n));
  ProfScope prof("symv_packed", n);
  void* wp = work ? work->data() : nullptr;
  if (P.dt == F32) LaunchSymv<float, true>(n, alpha, P.as<float>(), kSB, x.as<float>(), beta, y.as<float>(), wp);
  else LaunchSymv<double, true>(n, alpha, P.as<double>(), kSB, x.as<double>(), beta, y.as<double>(), wp);
  EPS_HIP(hipGetLastError());
}

namespace {
// The same reduction with 16-byte loads for the partial vectors of the fused sweep (hundreds of
// vectors of m floats, 20 MB at config 2): a workgroup owns 32 rows, thread (rq, pl) = (t & 7,
// t >> 3) sums the float4 of rows 4 rq .. 4 rq + 3 over the partials k = pl, pl + 32, ... (all of a
// thread's loads independent), then the 32 part lanes are combined by three shuffle steps inside a
// wave and the 4 waves in order - a fixed summation order.
constexpr int kRQ4 = 8, kPL4 = kBlock / kRQ4;

__global__ __launch_bounds__(kBlock) void ReducePartials4Kernel(int64_t rows, int nparts,
                                                                const float* __restrict__ partial,
                                                                float alpha, float beta, float* y,
                                                                const float* __restrict__ add) {
  __shared__ float4 part[kBlock / 64][kRQ4];
  const int t = threadIdx.x, rq = t & (kRQ4 - 1), pl = t >> 3, wave = t >> 6;
  const int64_t r0 = (static_cast<int64_t>(blockIdx.x) * kRQ4 + rq) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r0 < rows) {
    const float* p = partial + r0;
    int k = pl;
    for (; k + 7 * kPL4 < nparts; k += 8 * kPL4) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = *reinterpret_cast<const float4*>(p + static_cast<int64_t>(k + u * kPL4) * rows);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s.x += v[u].x;
        s.y += v[u].y;
        s.z += v[u].z;
        s.w += v[u].w;
      }
    }
    for (; k < nparts; k += kPL4) {
      const float4 v = *reinterpret_cast<const float4*>(p + static_cast<int64_t>(k) * rows);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  }
#pragma unroll
  for (int off = 32; off >= 8; off >>= 1) {
    s.x += __shfl_down(s.x, off, 64);
    s.y += __shfl_down(s.y, off, 64);
    s.z += __shfl_down(s.z, off, 64);
    s.w += __shfl_down(s.w, off, 64);
  }
  if ((t & 63) < kRQ4) part[wave][rq] = s;
  __syncthreads();
  if (t >= kRQ4 || r0 >= rows) return;
  const float4 a = part[0][rq], b = part[1][rq], c = part[2][rq], d = part[3][rq];
  float o[4] = {alpha * (((a.x + b.x) + c.x) + d.x), alpha * (((a.y + b.y) + c.y) + d.y),
                alpha * (((a.z + b.z) + c.z) + d.z), alpha * (((a.w + b.w) + c.w) + d.w)};
  if (beta != 0.f) {
    const float4 yo = *reinterpret_cast<const float4*>(y + r0);
    o[0] += beta * yo.x;
    o[1] += beta * yo.y;
    o[2] += beta * yo.z;
    o[3] += beta * yo.w;
  }
  if (add) {
    const float4 ad = *reinterpret_cast<const float4*>(add + r0);
    o[0] += ad.x;
    o[1] += ad.y;
    o[2] += ad.z;
    o[3] += ad.w;
  }
  *reinterpret_cast<float4*>(y + r0) = make_float4(o[0], o[1], o[2], o[3]);
}
}  // namespace

void ReducePartials(int64_t rows, int nparts, const DVec& partial, double alpha, double beta,
                    const DVec& y, const DVec* add) {
  EPS_CHECK(partial.dt == y.dt && y.n == rows && partial.n >= static_cast<int64_t>(nparts) * rows);
  if (add) EPS_CHECK(add->n == rows && add->dt == y.dt);
  if (rows == 0) return;
  hipStream_t s = Runtime::Get().stream();
  ProfScope prof("reduce_partials", rows, nparts);
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  if (y.dt == F32 && rows % 4 == 0 && nparts >= 64 && al16(partial.data()) && al16(y.data()) &&
      (!add || al16(add->data()))) {
    hipLaunchKernelGGL(ReducePartials4Kernel, dim3(static_cast<unsigned>((rows / 4 + kRQ4 - 1) / kRQ4)),
                       dim3(kBlock), 0, s, rows, nparts, partial.as<float>(), static_cast<float>(alpha),
                       static_cast<float>(beta), y.as<float>(), add ? add->as<float>() : nullptr);
    return;
  }
  const unsigned grid = static_cast<unsigned>((rows + 63) / 64);
  if (y.dt == F32)
    hipLaunchKernelGGL(GemvNReduceKernel<float>, dim3(grid), dim3(kBlock), 0, s, rows, nparts,
                       partial.as<float>(), static_cast<float>(alpha), static_cast<float>(beta),
                       y.as<float>(), add ? add->as<float>() : nullptr);
  else
    hipLaunchKernelGGL(GemvNReduceKernel<double>, dim3(grid), dim3(kBlock), 0, s, rows, nparts,
                       partial.as<double>(), alpha, beta, y.as<double>(),
                       add ? add->as<double>() : nullptr);
}

void Gemv(bool trans, int64_t rows, int64_t cols, double alpha, const DVec& A, int64_t lda,
          const DVec& x, double beta, const DVec& y) {
  EPS_CHECK(A.dt == x.dt && A.dt == y.dt);
  EPS_CHECK_MSG(lda >= rows, "gemv: lda < rows");
  EPS_CHECK_MSG(cols == 0 || A.n >= (cols - 1) * lda + rows, "gemv: matrix buffer too small");
  EPS_CHECK_MSG(x.n == (trans ? rows : cols), "gemv: x has " << x.n << " entries");
  EPS_CHECK_MSG(y.n == (trans ? cols : rows), "gemv: y has " << y.n << " entries");
  EPS_CHECK_MSG(x.data() != y.data(), "gemv: x and y alias");
  if (y.n == 0) return;
  if (rows == 0 || cols == 0) {
    if (beta == 0) Fill(y, 0.0);
    else Axpby(y, beta, y, 0.0);  // y = beta*y
    return;
  }
  ProfScope prof(trans ? "gemv_t" : "gemv_n", rows, cols);
  if (A.dt == F32) {
    if (trans) LaunchGemvT<float>(rows, cols, alpha, A.as<float>(), lda, x.as<float>(), beta, y.as<float>());
    else LaunchGemvN<float>(rows, cols, alpha, A.as<float>(), lda, x.as<float>(), beta, y.as<float>());
  } else {
    if (trans) LaunchGemvT<double>(rows, cols, alpha, A.as<double>(), lda, x.as<double>(), beta, y.as<double>());
    else LaunchGemvN<double>(rows, cols, alpha, A.as<double>(), lda, x.as<double>(), beta, y.as<double>());
  }
}

}  // namespace k
}  // namespace eps

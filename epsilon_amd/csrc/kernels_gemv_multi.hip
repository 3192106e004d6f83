// Dense matrix times a SKINNY matrix (up to 16 right-hand sides), fp32: C = alpha op(A) B + beta C.
//
// The Kronecker maps of the multiclass problems apply kron(I_k, X) and its transpose to vec(W):
// X W (60000 x 784 times 784 x 10) and X^T R (784 x 60000 times 60000 x 10) every sweep
// (reference linear/kronecker_product_impl.cc:27-42 does them as `(A (B X)^T)^T` through its
// multiply table).  On the 128 x 128 MFMA tiles a 10-column result wastes 118 of 128 columns and a
// 784 x 10 result has 7 tiles; both products are really mat-vecs with a few right-hand sides:
// HBM-bound, the matrix read ONCE, the k vectors riding along ("Kron/mnist
// sweep": 2 * samples * features * s bytes per pass).
//
//   MultiGemvN: A is M x K column-major (contiguous along the output index).  A thread owns four
//               consecutive rows and NR accumulators per row; a workgroup covers 1024 rows and a
//               chunk of columns, whose slice of B sits in LDS (broadcast reads).  Column chunks
//               give the grid its width; their partial results are added in a fixed order.
//   MultiGemvT: op(A) = A^T with A stored K x M (contiguous along the contraction index).  A
//               workgroup takes four columns of A and a slice of K: 16-byte loads of A and of the
//               NR columns of B, 4 x NR dot products per thread, reduced through LDS; K slices are
//               added in a fixed order.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;
constexpr int kRowsPerWg = 4 * kBlock;  // MultiGemvN
constexpr int kColChunk = 64;           // columns of A per workgroup (MultiGemvN)
constexpr int kTCols = 4;               // columns of A per workgroup (MultiGemvT)

template <int NR>
__global__ __launch_bounds__(kBlock) void MultiGemvNKernel(int64_t M, int64_t K, int nrhs,
                                                           const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb,
                                                           int64_t kslice, float* __restrict__ P) {
  __shared__ float xs[kColChunk][NR];
  const int t = threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kRowsPerWg + 4 * t;
  const int64_t jlo = static_cast<int64_t>(blockIdx.y) * kslice;
  const int64_t jhi = jlo + kslice < K ? jlo + kslice : K;
  float4 acc[NR];
#pragma unroll
  for (int c = 0; c < NR; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool full = r0 + 3 < M;
  for (int64_t j0 = jlo; j0 < jhi; j0 += kColChunk) {
    const int jn = static_cast<int>(jhi - j0 < kColChunk ? jhi - j0 : kColChunk);
    __syncthreads();
    for (int idx = t; idx < kColChunk * NR; idx += kBlock) {
      const int jj = idx / NR, c = idx % NR;
      xs[jj][c] = (jj < jn && c < nrhs) ? B[(j0 + jj) + static_cast<int64_t>(c) * ldb] : 0.f;
    }
    __syncthreads();
    if (r0 >= M) continue;
    const float* a = A + r0 + j0 * lda;
#pragma unroll 4
    for (int jj = 0; jj < jn; ++jj) {
      float4 v;
      if (full) {
        v = *reinterpret_cast<const float4*>(a + static_cast<int64_t>(jj) * lda);
      } else {
        const float* q = a + static_cast<int64_t>(jj) * lda;
        v.x = q[0];
        v.y = r0 + 1 < M ? q[1] : 0.f;
        v.z = r0 + 2 < M ? q[2] : 0.f;
        v.w = 0.f;
      }
#pragma unroll
      for (int c = 0; c < NR; ++c) {
        const float x = xs[jj][c];
        acc[c].x += v.x * x;
        acc[c].y += v.y * x;
        acc[c].z += v.z * x;
        acc[c].w += v.w * x;
      }
    }
  }
  if (r0 >= M) return;
  // partial block `blockIdx.y`: column-major M x nrhs
  float* p = P + static_cast<int64_t>(blockIdx.y) * M * nrhs + r0;
#pragma unroll
  for (int c = 0; c < NR; ++c) {  // no early exit: a rolled loop would index acc dynamically (scratch)
    if (c < nrhs) {
      float* q = p + static_cast<int64_t>(c) * M;
      if (full && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {
        *reinterpret_cast<float4*>(q) = acc[c];
      } else {
        q[0] = acc[c].x;
        if (r0 + 1 < M) q[1] = acc[c].y;
        if (r0 + 2 < M) q[2] = acc[c].z;
        if (r0 + 3 < M) q[3] = acc[c].w;
      }
    }
  }
}

template <int NR>
__global__ __launch_bounds__(kBlock) void MultiGemvTKernel(int64_t M, int64_t K, int nrhs,
                                                           const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb,
                                                           int64_t kslice, float* __restrict__ P) {
  __shared__ float red[kBlock][NR + 1];
  const int t = threadIdx.x;
  const int64_t j0 = static_cast<int64_t>(blockIdx.x) * kTCols;
  const int64_t k0 = static_cast<int64_t>(blockIdx.y) * kslice;
  const int64_t k1 = k0 + kslice < K ? k0 + kslice : K;
  float acc[kTCols][NR];
#pragma unroll
  for (int g = 0; g < kTCols; ++g)
#pragma unroll
    for (int c = 0; c < NR; ++c) acc[g][c] = 0.f;
  // kslice is a multiple of 4 and lda, ldb are multiples of 4 (checked by the launcher), so the
  // 16-byte loads are aligned; the tail of K is handled element-wise
  const int64_t kv = k0 + ((k1 - k0) / 4) * 4;
  for (int64_t kk = k0 + 4 * t; kk < kv; kk += 4 * kBlock) {
    float4 a[kTCols];
#pragma unroll
    for (int g = 0; g < kTCols; ++g)
      a[g] = (j0 + g < M) ? *reinterpret_cast<const float4*>(A + kk + (j0 + g) * lda)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      if (c < nrhs) {
        const float4 b = *reinterpret_cast<const float4*>(B + kk + static_cast<int64_t>(c) * ldb);
#pragma unroll
        for (int g = 0; g < kTCols; ++g)
          acc[g][c] += (a[g].x * b.x + a[g].y * b.y) + (a[g].z * b.z + a[g].w * b.w);
      }
    }
  }
  for (int64_t kk = kv + t; kk < k1; kk += kBlock) {
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      if (c < nrhs) {
        const float b = B[kk + static_cast<int64_t>(c) * ldb];
#pragma unroll
        for (int g = 0; g < kTCols; ++g)
          if (j0 + g < M) acc[g][c] += A[kk + (j0 + g) * lda] * b;
      }
    }
  }
  // reduce the 256 per-thread partials of each (column, rhs) pair in a fixed tree
#pragma unroll
  for (int g = 0; g < kTCols; ++g) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NR; ++c) red[t][c] = acc[g][c];
    __syncthreads();
    for (int stride = kBlock / 2; stride > 0; stride >>= 1) {
      if (t < stride) {
#pragma unroll
        for (int c = 0; c < NR; ++c) red[t][c] += red[t + stride][c];
      }
      __syncthreads();
    }
    if (t < nrhs && j0 + g < M)
      P[static_cast<int64_t>(blockIdx.y) * M * nrhs + (j0 + g) + static_cast<int64_t>(t) * M] = red[0][t];
  }
}

template <int NR>
void LaunchMulti(bool transA, int64_t M, int64_t K, int nrhs, const float* A, int64_t lda,
                 const float* B, int64_t ldb, float* P, int64_t parts, int64_t kslice) {
  hipStream_t s = Runtime::Get().stream();
  if (!transA) {
    dim3 grid(static_cast<unsigned>((M + kRowsPerWg - 1) / kRowsPerWg), static_cast<unsigned>(parts));
    hipLaunchKernelGGL(MultiGemvNKernel<NR>, grid, dim3(kBlock), 0, s, M, K, nrhs, A, lda, B, ldb, kslice, P);
  } else {
    dim3 grid(static_cast<unsigned>((M + kTCols - 1) / kTCols), static_cast<unsigned>(parts));
    hipLaunchKernelGGL(MultiGemvTKernel<NR>, grid, dim3(kBlock), 0, s, M, K, nrhs, A, lda, B, ldb, kslice, P);
  }
}

}  // namespace

bool MultiGemv(bool transA, int64_t M, int64_t N, int64_t K, double alpha, const DVec& A, int64_t lda,
               const DVec& B, int64_t ldb, double beta, const DVec& C, int64_t ldc) {
  if (A.dt != F32 || N < 1 || N > 16 || M < 1 || K < 1) return false;
  if (ldc != M) return false;
  if (reinterpret_cast<uintptr_t>(A.data()) % 16 != 0 || lda % 4 != 0) return false;
  if (transA && (reinterpret_cast<uintptr_t>(B.data()) % 16 != 0 || ldb % 4 != 0)) return false;
  ProfScope prof(transA ? "multi_gemv_t" : "multi_gemv_n", M * N, K);
  int64_t parts, kslice = 0;
  if (!transA) {
    // column slices so that ~1024 workgroups run: (M / 1024) row blocks x K slices
    const int64_t rowblocks = (M + kRowsPerWg - 1) / kRowsPerWg;
    parts = std::max<int64_t>(1, std::min<int64_t>((1024 + rowblocks - 1) / rowblocks,
                                                   (K + kColChunk - 1) / kColChunk));
    kslice = ((K + parts - 1) / parts + kColChunk - 1) / kColChunk * kColChunk;
    parts = (K + kslice - 1) / kslice;
  } else {
    // enough workgroups to fill the chip: (M / 4) column groups x K slices
    const int64_t groups = (M + kTCols - 1) / kTCols;
    parts = std::max<int64_t>(1, std::min<int64_t>((1024 + groups - 1) / groups, K / 4096));
    kslice = ((K + parts - 1) / parts + 3) / 4 * 4;
    parts = (K + kslice - 1) / kslice;
  }
  EPS_CHECK_MSG(parts <= 65535, "multi_gemv: too many partial blocks");
  DVec P = DVec::Empty(parts * M * N, F32);
  const int nrhs = static_cast<int>(N);
  const float* a = A.as<float>();
  const float* b = B.as<float>();
  float* p = P.as<float>();
  if (nrhs <= 4) LaunchMulti<4>(transA, M, K, nrhs, a, lda, b, ldb, p, parts, kslice);
  else if (nrhs <= 8) LaunchMulti<8>(transA, M, K, nrhs, a, lda, b, ldb, p, parts, kslice);
  else if (nrhs <= 12) LaunchMulti<12>(transA, M, K, nrhs, a, lda, b, ldb, p, parts, kslice);
  else LaunchMulti<16>(transA, M, K, nrhs, a, lda, b, ldb, p, parts, kslice);
  EPS_HIP(hipGetLastError());
  ReducePartials(M * N, static_cast<int>(parts), P, alpha, beta, C.Slice(0, M * N));
  return true;
}

}  // namespace k
}  // namespace eps

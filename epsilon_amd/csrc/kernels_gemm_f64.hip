// K4 in the fp64 mode: matrix products on v_mfma_f64_16x16x4_f64, software-pipelined.
//
// The reference's arithmetic type is double (reference src/epsilon/linear/linear_map.h:35, its
// products are `dgemm_`: linear/linear_map_multiply.cc:14-37), so the fp64 mode is the one that
// reproduces it digit for digit; this kernel is what its Gram product and the GEMMs of its
// Cholesky inverse run on.
//
// MI355X's f64 matrix rate equals its f64 vector rate (78.6 TFLOP/s: 16 multiply-adds per cycle
// and SIMD), so the matrix cores buy no arithmetic - they buy operand reuse: one MFMA consumes
// 16 bytes of LDS per lane for 1024 multiply-adds, where the 4 x 4 register tile of the VALU
// kernel reads 64 bytes per lane for 16 and is bound by the LDS port at about half the peak.
//
// Geometry: 128 x 128 output tile per 256-thread workgroup, 4 waves as 2 x 2, each wave 4 x 4
// blocks of 16 x 16 (128 accumulator registers).  k slabs of 8 staged through LDS as [k][row]
// with row stride 144 doubles (consecutive k rows land 128 bytes apart modulo the 256-byte bank
// period: a half-wave's ds_read_b64 of two k rows touches every bank once).  Two LDS stages and
// a register stage: the global loads of slab k+2 are issued before the MFMAs of slab k+1, one
// barrier per slab.  The MFMA "A" operand takes the B tile, so a result register's 16 lanes hold
// 16 consecutive rows of column-major C (128-byte stores).
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;
constexpr int MT = 128;  // output tile
constexpr int MK = 8;    // k slab
constexpr int LD = 144;  // LDS row stride (doubles)

using f64x4 = __attribute__((ext_vector_type(4))) double;

// CR: op(X) contiguous along the output index.  Each thread brings two double2.
template <bool CR, bool INTERIOR>
__device__ inline void LoadSlab(double2 (&r)[2], const double* __restrict__ X, int64_t ld, int64_t r0,
                                int64_t k0, int64_t R, int64_t K) {
  const int t = threadIdx.x;
  if (CR) {
    const int rr2 = (t & 63) * 2;
    const int kb = t >> 6;  // 0..3
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int64_t gk = k0 + kb + 4 * p, gr = r0 + rr2;
      const double* src = X + gr + gk * ld;
      if (INTERIOR) {
        r[p] = *reinterpret_cast<const double2*>(src);
      } else {
        double2 v = make_double2(0, 0);
        if (gk < K) {
          if (gr + 0 < R) v.x = src[0];
          if (gr + 1 < R) v.y = src[1];
        }
        r[p] = v;
      }
    }
  } else {
    const int kk2 = (t & 3) * 2;
    const int rb = t >> 2;  // 0..63
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int64_t gk = k0 + kk2, gr = r0 + rb + 64 * p;
      const double* src = X + gk + gr * ld;
      if (INTERIOR) {
        r[p] = *reinterpret_cast<const double2*>(src);
      } else {
        double2 v = make_double2(0, 0);
        if (gr < R) {
          if (gk + 0 < K) v.x = src[0];
          if (gk + 1 < K) v.y = src[1];
        }
        r[p] = v;
      }
    }
  }
}

template <bool CR>
__device__ inline void StoreSlab(double* __restrict__ S, const double2 (&r)[2]) {
  const int t = threadIdx.x;
  if (CR) {
    const int rr2 = (t & 63) * 2;
    const int kb = t >> 6;
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<double2*>(S + (kb + 4 * p) * LD + rr2) = r[p];
  } else {
    const int kk2 = (t & 3) * 2;
    const int rb = t >> 2;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int rr = rb + 64 * p;
      S[(kk2 + 0) * LD + rr] = r[p].x;
      S[(kk2 + 1) * LD + rr] = r[p].y;
    }
  }
}

// CA: op(A) contiguous along i (A not transposed); CB: op(B) contiguous along j (B transposed)
template <bool CA, bool CB>
__global__ __launch_bounds__(kBlock, 2) void GemmMfmaF64PipeKernel(
    int64_t M, int64_t N, int64_t K, double alpha, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, double beta, double* C, int64_t ldc, int lower_only,
    int64_t sA, int64_t sB, int64_t sC, int64_t n1, int64_t sA2, int64_t sB2) {
  {
    const int64_t z = blockIdx.z, z1 = z % n1, z2 = z / n1;
    A += z1 * sA + z2 * sA2;
    B += z1 * sB + z2 * sB2;
    C += z * sC;
  }
  __shared__ __attribute__((aligned(16))) double As[2][MK * LD];
  __shared__ __attribute__((aligned(16))) double Bs[2][MK * LD];
  int64_t i0 = static_cast<int64_t>(blockIdx.x) * MT;
  int64_t j0 = static_cast<int64_t>(blockIdx.y) * MT;
  if (lower_only == 2) {
    // compact 1-D grid over the tiles on and below the diagonal
    const int64_t lin = blockIdx.x;
    int64_t I = static_cast<int64_t>((sqrt(8.0 * static_cast<double>(lin) + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= lin) ++I;
    while (I * (I + 1) / 2 > lin) --I;
    i0 = I * MT;
    j0 = (lin - I * (I + 1) / 2) * MT;
  } else if (lower_only && i0 + MT <= j0) {
    return;
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = wave & 1, wj = wave >> 1;
  const int l15 = lane & 15, lk = lane >> 4;
  // an edge tile loads the full 128-wide window that ends at the matrix edge and stores only
  // its own rows / columns, so all its k slabs but the last take the unchecked loads
  int64_t li0 = i0, lj0 = j0;
  bool in_i = i0 + MT <= M, in_j = j0 + MT <= N;
  if (!in_i && M >= MT && (!CA || M % 2 == 0)) {
    li0 = M - MT;
    in_i = true;
  }
  if (!in_j && N >= MT && (!CB || N % 2 == 0)) {
    lj0 = N - MT;
    in_j = true;
  }

  f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0;

  double2 ra[2], rb[2];
  const int64_t nk = (K + MK - 1) / MK;
  const int64_t nfull = (in_i && in_j) ? K / MK : 0;  // slabs that need no bounds checks
  auto load = [&](int64_t kt) {
    if (kt < nfull) {
      LoadSlab<CA, true>(ra, A, lda, li0, kt * MK, M, K);
      LoadSlab<CB, true>(rb, B, ldb, lj0, kt * MK, N, K);
    } else if (kt < nk) {
      LoadSlab<CA, false>(ra, A, lda, li0, kt * MK, M, K);
      LoadSlab<CB, false>(rb, B, ldb, lj0, kt * MK, N, K);
    }
  };
  auto slab = [&](auto fast_tag, int64_t kt) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const double* as = As[kt & 1] + wi * 64 + l15;
    const double* bs = Bs[kt & 1] + wj * 64 + l15;
    double av[2][4], bv[2][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) av[0][a] = as[lk * LD + a * 16];
#pragma unroll
    for (int b = 0; b < 4; ++b) bv[0][b] = bs[lk * LD + b * 16];
#pragma unroll
    for (int kk = 0; kk < MK; kk += 4) {
      const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
      if (kk + 4 < MK) {
#pragma unroll
        for (int a = 0; a < 4; ++a) av[nxt][a] = as[(kk + 4 + lk) * LD + a * 16];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[nxt][b] = bs[(kk + 4 + lk) * LD + b * 16];
      }
      // D'[j][i] += Bop[k][j] * Aop[i][k]: MFMA "A" operand = B values, "B" operand = A values
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[cur][b], av[cur][a], acc[a][b], 0, 0, 0);
      if (FAST) __builtin_amdgcn_sched_barrier(0);  // keep the read-ahead order
    }
    if (FAST || kt + 1 < nk) {
      StoreSlab<CA>(As[(kt + 1) & 1], ra);
      StoreSlab<CB>(Bs[(kt + 1) & 1], rb);
    }
    __syncthreads();
    if (FAST) {
      LoadSlab<CA, true>(ra, A, lda, li0, (kt + 2) * MK, M, K);
      LoadSlab<CB, true>(rb, B, ldb, lj0, (kt + 2) * MK, N, K);
      __builtin_amdgcn_sched_barrier(0);  // do not let the scheduler sink these loads
    } else {
      load(kt + 2);
    }
  };
  load(0);
  StoreSlab<CA>(As[0], ra);
  StoreSlab<CB>(Bs[0], rb);
  __syncthreads();
  load(1);
  int64_t kt = 0;
  for (; kt + 2 < nfull; ++kt) slab(std::true_type(), kt);
  for (; kt < nk; ++kt) slab(std::false_type(), kt);

  // acc[a][b][reg]: i_local = l15 (the MFMA's column), j_local = lk + 4 * reg (its row)
  // (beta != 0: the old values of a row of blocks are loaded as one batch before its stores -
  // see the f32 kernel)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = li0 + wi * 64 + a * 16 + l15;
    const bool i_ok = i >= i0 && i < M;
    const int64_t ic = i < M ? i : M - 1;
    double old[4][4];
    if (beta != 0.0) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t j = lj0 + wj * 64 + b * 16 + lk + 4 * r;
          old[b][r] = C[ic + (j < N ? j : N - 1) * ldc];
        }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = lj0 + wj * 64 + b * 16 + lk + 4 * r;
        if (!i_ok || j < j0 || j >= N) continue;
        const double v = alpha * acc[a][b][r];
        C[i + j * ldc] = (beta == 0.0) ? v : v + beta * old[b][r];
      }
    }
  }
}

}  // namespace

bool GemmF64Pipe(bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha, const DVec& A,
                 int64_t lda, int64_t sA, const DVec& B, int64_t ldb, int64_t sB, double beta,
                 const DVec& C, int64_t ldc, int64_t sC, int64_t n1, int64_t batch, bool lower_only,
                 int64_t sA2, int64_t sB2) {
  if (A.dt != F64 || B.dt != F64 || C.dt != F64) return false;
  // 16-byte operand loads
  const bool aligned = lda % 2 == 0 && ldb % 2 == 0 && sA % 2 == 0 && sB % 2 == 0 && sA2 % 2 == 0 &&
                       sB2 % 2 == 0 && reinterpret_cast<uintptr_t>(A.data()) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(B.data()) % 16 == 0;
  if (!aligned) return false;
  hipStream_t s = Runtime::Get().stream();
  dim3 grid(static_cast<unsigned>((M + MT - 1) / MT), static_cast<unsigned>((N + MT - 1) / MT),
            static_cast<unsigned>(batch));
  int lo = lower_only ? 1 : 0;
  if (lower_only && batch == 1) {
    const int64_t T = (M + MT - 1) / MT;
    grid = dim3(static_cast<unsigned>(T * (T + 1) / 2), 1, 1);
    lo = 2;
  }
#define EPS_PIPE(CA, CB)                                                                              \
  hipLaunchKernelGGL((GemmMfmaF64PipeKernel<CA, CB>), grid, dim3(kBlock), 0, s, M, N, K, alpha,        \
                     A.as<double>(), lda, B.as<double>(), ldb, beta, C.as<double>(), ldc, lo, sA, sB, \
                     sC, n1, sA2, sB2)
  if (!transA && transB) EPS_PIPE(true, true);
  else if (!transA && !transB) EPS_PIPE(true, false);
  else if (transA && transB) EPS_PIPE(false, true);
  else EPS_PIPE(false, false);
#undef EPS_PIPE
  return true;
}

}  // namespace k
}  // namespace eps

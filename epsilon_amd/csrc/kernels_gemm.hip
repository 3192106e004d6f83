// K4: dense matrix-matrix products (the Gram contraction A A^T / A^T A of the least-squares
// prox, the blocked Cholesky updates, Kronecker applies).
//
// The reference calls `dgemm_` (reference src/epsilon/linear/linear_map_multiply.cc:14-37).
// Two kernels:
//
//  * GemmMfmaF32: 128x128 output tile per 256-thread workgroup (4 wavefronts as 2x2, each
//    wavefront 2x2 tiles of v_mfma_f32_32x32x2_f32), BK = 32 slabs staged through LDS as
//    [k][row] so one ds_read_b32 per lane feeds an MFMA operand directly.  f32 in / f32
//    accumulate MFMA is bit-for-bit an fmaf chain on gfx950 (no TF32), so this is exact fp32.
//    The accumulator is produced transposed (MFMA "A" operand <- B tile) so that a register's
//    32 lanes hit 32 consecutive rows of column-major C: 128-byte coalesced stores.
//  * GemmGeneric<T>: 64x64 LDS-tiled VALU kernel for f64 and for small shapes.
//  * GemmMfmaF64: the f32 tile on v_mfma_f64_16x16x4_f64, unpipelined (EPSILON_HIP_GEMM=mfma_simple);
//    the fp64 mode's products run on the pipelined kernel of kernels_gemm_f64.hip.
//
// `lower_only` skips tiles strictly above the diagonal (SYRK-style, half the flops).
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;

// element (r, c) of op(X) where X is column-major with leading dimension ld
template <class T>
__device__ inline T OpAt(const T* X, int64_t ld, bool trans, int64_t r, int64_t c) {
  return trans ? X[c + r * ld] : X[r + c * ld];
}

// ------------------------------------------------------------------------------------------
// generic tiled kernel
// ------------------------------------------------------------------------------------------
constexpr int GT = 64;   // tile
constexpr int GK = 16;   // k slab

template <class T>
__global__ __launch_bounds__(kBlock) void GemmGenericKernel(
    int transA, int transB, int64_t M, int64_t N, int64_t K, T alpha, const T* __restrict__ A,
    int64_t lda, const T* __restrict__ B, int64_t ldb, T beta, T* C, int64_t ldc,
    int lower_only, int64_t sA, int64_t sB, int64_t sC, int64_t n1, int64_t sA2, int64_t sB2) {
  {
    const int64_t z = blockIdx.z, z1 = z % n1, z2 = z / n1;
    A += z1 * sA + z2 * sA2;
    B += z1 * sB + z2 * sB2;
    C += z * sC;
  }
  __shared__ T As[GK][GT + 1];
  __shared__ T Bs[GK][GT + 1];
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * GT;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * GT;
  if (lower_only && i0 + GT <= j0) return;  // tile entirely above the diagonal
  const int t = threadIdx.x;
  const int tx = t & 15, ty = t >> 4;
  T acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = T(0);

  for (int64_t k0 = 0; k0 < K; k0 += GK) {
    // stage op(A)[i0:i0+64, k0:k0+16] as As[k][i], op(B)[k0:k0+16, j0:j0+64] as Bs[k][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int ii, kk;
      if (!transA) { ii = t & 63; kk = (t >> 6) + 4 * r; }   // contiguous along i
      else { kk = t & 15; ii = (t >> 4) + 16 * r; }          // contiguous along k
      const int64_t gi = i0 + ii, gk = k0 + kk;
      As[kk][ii] = (gi < M && gk < K) ? OpAt(A, lda, transA != 0, gi, gk) : T(0);
      int jj, kb;
      if (transB) { jj = t & 63; kb = (t >> 6) + 4 * r; }    // op(B)(k,j) = B[j + k*ldb]
      else { kb = t & 15; jj = (t >> 4) + 16 * r; }          // op(B)(k,j) = B[k + j*ldb]
      const int64_t gj = j0 + jj, gkb = k0 + kb;
      Bs[kb][jj] = (gj < N && gkb < K) ? OpAt(B, ldb, transB != 0, gkb, gj) : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; ++kk) {
      T av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = As[kk][tx + 16 * a];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = Bs[kk][ty + 16 * b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
    }
    __syncthreads();
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int64_t j = j0 + ty + 16 * b;
    if (j >= N) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t i = i0 + tx + 16 * a;
      if (i >= M) continue;
      T* c = C + i + j * ldc;
      *c = (beta == T(0)) ? alpha * acc[a][b] : alpha * acc[a][b] + beta * (*c);
    }
  }
}

// ------------------------------------------------------------------------------------------
// f32 MFMA kernel
// ------------------------------------------------------------------------------------------
constexpr int MT = 128;        // output tile (rows and cols)
constexpr int MK = 32;         // k slab
constexpr int MLD = MT + 1;    // LDS row stride (floats): conflict-free for both loaders

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Stage op(X)[r0 : r0+128, k0 : k0+32] (r = the non-contracted index) into S[k][r].
//   contiguous_r == true : memory is contiguous along r   (X[r + k*ld])
//   contiguous_r == false: memory is contiguous along k   (X[k + r*ld])
__device__ inline void StageTile(float (*S)[MLD], const float* __restrict__ X, int64_t ld,
                                 bool contiguous_r, int64_t r0, int64_t k0, int64_t R,
                                 int64_t K) {
  const int t = threadIdx.x;
  if (contiguous_r) {
    const int rr = t & 127;
    const int kb = t >> 7;  // 0..1
    const int64_t gr = r0 + rr;
#pragma unroll
    for (int p = 0; p < MK / 2; ++p) {
      const int kk = kb + 2 * p;
      const int64_t gk = k0 + kk;
      S[kk][rr] = (gr < R && gk < K) ? X[gr + gk * ld] : 0.0f;
    }
  } else {
    const int kk = t & 31;
    const int rb = t >> 5;  // 0..7
    const int64_t gk = k0 + kk;
#pragma unroll
    for (int p = 0; p < MT / 8; ++p) {
      const int rr = rb + 8 * p;
      const int64_t gr = r0 + rr;
      S[kk][rr] = (gr < R && gk < K) ? X[gk + gr * ld] : 0.0f;
    }
  }
}

__global__ __launch_bounds__(kBlock) void GemmMfmaF32Kernel(
    int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    float beta, float* C, int64_t ldc, int lower_only, int64_t sA, int64_t sB, int64_t sC, int64_t n1, int64_t sA2, int64_t sB2) {
  {
    const int64_t z = blockIdx.z, z1 = z % n1, z2 = z / n1;
    A += z1 * sA + z2 * sA2;
    B += z1 * sB + z2 * sB2;
    C += z * sC;
  }
  __shared__ float As[MK][MLD];
  __shared__ float Bs[MK][MLD];
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * MT;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * MT;
  if (lower_only && i0 + MT <= j0) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = wave & 1, wj = wave >> 1;  // wavefront's 64x64 quadrant
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  for (int64_t k0 = 0; k0 < K; k0 += MK) {
    // op(A)(i,k): contiguous along i when A is not transposed; op(B)(k,j): contiguous along j
    // when B IS transposed (B stored N x K).
    StageTile(As, A, lda, transA == 0, i0, k0, M, K);
    StageTile(Bs, B, ldb, transB != 0, j0, k0, N, K);
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < MK; kk += 2) {
      float av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) av[a] = As[kk + lh][wi * 64 + a * 32 + l31];
#pragma unroll
      for (int b = 0; b < 2; ++b) bv[b] = Bs[kk + lh][wj * 64 + b * 32 + l31];
      // D'[j][i] += Bop[k][j] * Aop[i][k]: MFMA "A" operand = B values, "B" operand = A values
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[b], av[a], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }

  // acc[a][b][reg]: j_local = (reg&3) + 8*(reg>>2) + 4*lh , i_local = l31
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int64_t i = i0 + wi * 64 + a * 32 + l31;
    if (i >= M) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t j = j0 + wj * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (j >= N) continue;
        float* c = C + i + j * ldc;
        const float v = alpha * acc[a][b][r];
        *c = (beta == 0.0f) ? v : v + beta * (*c);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// f32 MFMA kernel, pipelined form (16-byte aligned operands): same tile / wave geometry, but
//   * every global load is 16 B per lane (4 per thread per operand per k-slab) and is issued
//     for slab k+1 BEFORE the MFMAs of slab k, so HBM/L2 latency hides under the matrix work;
//   * operands contiguous along the output index land in LDS with ds_write_b128 (row stride
//     132 floats), operands contiguous along k are transposed through scalar LDS writes with
//     row stride 129 (conflict-free in both cases);
//   * bounds checks only on edge tiles / the last k-slab.
// ------------------------------------------------------------------------------------------
template <bool CR> struct StageCfg { static constexpr int LD = CR ? 132 : 129; };

template <bool CR, bool INTERIOR>
__device__ inline void LoadSlab(float4 (&r)[4], const float* __restrict__ X, int64_t ld,
                                int64_t r0, int64_t k0, int64_t R, int64_t K) {
  constexpr bool interior = INTERIOR;
  const int t = threadIdx.x;
  if (CR) {
    const int rr4 = (t & 31) * 4;
    const int kb = t >> 5;  // 0..7
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t gk = k0 + kb + 8 * p, gr = r0 + rr4;
      const float* src = X + gr + gk * ld;
      if (interior) {
        r[p] = *reinterpret_cast<const float4*>(src);
      } else {
        float4 v = make_float4(0, 0, 0, 0);
        if (gk < K) {
          if (gr + 0 < R) v.x = src[0];
          if (gr + 1 < R) v.y = src[1];
          if (gr + 2 < R) v.z = src[2];
          if (gr + 3 < R) v.w = src[3];
        }
        r[p] = v;
      }
    }
  } else {
    const int kk4 = (t & 7) * 4;
    const int rb = t >> 3;  // 0..31
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t gk = k0 + kk4, gr = r0 + rb + 32 * p;
      const float* src = X + gk + gr * ld;
      if (interior) {
        r[p] = *reinterpret_cast<const float4*>(src);
      } else {
        float4 v = make_float4(0, 0, 0, 0);
        if (gr < R) {
          if (gk + 0 < K) v.x = src[0];
          if (gk + 1 < K) v.y = src[1];
          if (gk + 2 < K) v.z = src[2];
          if (gk + 3 < K) v.w = src[3];
        }
        r[p] = v;
      }
    }
  }
}

template <bool CR>
__device__ inline void StoreSlab(float* __restrict__ S, const float4 (&r)[4]) {
  const int t = threadIdx.x;
  constexpr int LD = StageCfg<CR>::LD;
  if (CR) {
    const int rr4 = (t & 31) * 4;
    const int kb = t >> 5;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(S + (kb + 8 * p) * LD + rr4) = r[p];
  } else {
    const int kk4 = (t & 7) * 4;
    const int rb = t >> 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rr = rb + 32 * p;
      S[(kk4 + 0) * LD + rr] = r[p].x;
      S[(kk4 + 1) * LD + rr] = r[p].y;
      S[(kk4 + 2) * LD + rr] = r[p].z;
      S[(kk4 + 3) * LD + rr] = r[p].w;
    }
  }
}

// CA: op(A) contiguous along i (A not transposed); CB: op(B) contiguous along j (B transposed)
template <bool CA, bool CB>
__global__ __launch_bounds__(kBlock) void GemmMfmaF32PipeKernel(
    int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A, int64_t lda,
    const float* __restrict__ B, int64_t ldb, float beta, float* C, int64_t ldc,
    int lower_only, int64_t sA, int64_t sB, int64_t sC, int64_t n1, int64_t sA2, int64_t sB2,
    int64_t lin0, int64_t kchunk, float* __restrict__ P) {
  {
    const int64_t z = blockIdx.z, z1 = z % n1, z2 = z / n1;
    A += z1 * sA + z2 * sA2;
    B += z1 * sB + z2 * sB2;
    C += z * sC;
  }
  // split-K form of the compact triangle (the tail tiles of a SYRK, see GemmBatched): blockIdx.y
  // selects a chunk of K, the raw partial tile goes to P instead of C
  if (kchunk > 0) {
    const int64_t k_begin = static_cast<int64_t>(blockIdx.y) * kchunk;
    A += CA ? k_begin * lda : k_begin;
    B += CB ? k_begin * ldb : k_begin;
    K = K - k_begin < kchunk ? K - k_begin : kchunk;
  }
  constexpr int LDA = StageCfg<CA>::LD, LDB = StageCfg<CB>::LD;
  // two LDS stages: slab k+1 is written while slab k is still being read, one barrier per slab
  __shared__ __attribute__((aligned(16))) float As[2][MK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][MK * LDB];
  int64_t i0 = static_cast<int64_t>(blockIdx.x) * MT;
  int64_t j0 = static_cast<int64_t>(blockIdx.y) * MT;
  if (lower_only == 2) {
    // compact 1-D grid over the tiles on and below the diagonal (no empty workgroups, and the
    // round-robin of workgroups over the 8 XCDs splits the real tiles evenly)
    const int64_t lin = lin0 + blockIdx.x;
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
  const int l31 = lane & 31, lh = lane >> 5;
  // Edge tiles load a full 128-wide window that ends at the matrix edge (origin shifted back)
  // and store only their own rows / columns: every k-slab but the last then takes the
  // unchecked load path, where the bounds-checked one runs several times slower - and an edge
  // tile that starts late is the tail of the whole launch.
  int64_t li0 = i0, lj0 = j0;
  bool in_i = i0 + MT <= M, in_j = j0 + MT <= N;
  if (!in_i && M >= MT && (!CA || M % 4 == 0)) {
    li0 = M - MT;
    in_i = true;
  }
  if (!in_j && N >= MT && (!CB || N % 4 == 0)) {
    lj0 = N - MT;
    in_j = true;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  float4 ra[4], rb[4];
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
  // One k-slab: MFMAs on stage kt&1 (LDS operands read one k-step ahead), then the prefetched
  // registers of slab kt+1 go to the other stage, barrier, and the global loads of slab kt+2
  // are issued - a whole slab of MFMAs ahead of their use.  FAST: slab kt+2 lies fully inside
  // the operands, so its loads carry no bounds checks (the checked form compiles to a chain of
  // conditional loads with vmcnt(0) waits).
  auto slab = [&](auto fast_tag, int64_t kt) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const float* as = As[kt & 1];
    const float* bs = Bs[kt & 1];
    float av[2][2], bv[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) av[0][a] = as[lh * LDA + wi * 64 + a * 32 + l31];
#pragma unroll
    for (int b = 0; b < 2; ++b) bv[0][b] = bs[lh * LDB + wj * 64 + b * 32 + l31];
#pragma unroll
    for (int kk = 0; kk < MK; kk += 2) {
      const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
      if (kk + 2 < MK) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
          av[nxt][a] = as[(kk + 2 + lh) * LDA + wi * 64 + a * 32 + l31];
#pragma unroll
        for (int b = 0; b < 2; ++b)
          bv[nxt][b] = bs[(kk + 2 + lh) * LDB + wj * 64 + b * 32 + l31];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] =
              __builtin_amdgcn_mfma_f32_32x32x2f32(bv[cur][b], av[cur][a], acc[a][b], 0, 0, 0);
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

  // beta != 0: the 16 old values of a block are loaded as one independent batch (clamped
  // addresses) before the block's stores - interleaved with the stores through a pointer that
  // may alias them, every load waited for the store before it, and a short contraction (the
  // rank-256 updates of the Cholesky) spent most of its time in this epilogue.
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int64_t i = li0 + wi * 64 + a * 32 + l31;
    const bool i_ok = i >= i0 && i < M;
    const int64_t ic = i < M ? i : M - 1;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float old[16];
      if (P == nullptr && beta != 0.0f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t j = lj0 + wj * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          old[r] = C[ic + (j < N ? j : N - 1) * ldc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t j = lj0 + wj * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (!i_ok || j < j0 || j >= N) continue;
        if (P != nullptr) {
          P[(static_cast<int64_t>(blockIdx.x) * gridDim.y + blockIdx.y) * (MT * MT) + (i - i0) +
            (j - j0) * MT] = acc[a][b][r];
          continue;
        }
        const float v = alpha * acc[a][b][r];
        C[i + j * ldc] = (beta == 0.0f) ? v : v + beta * old[r];
      }
    }
  }
}

// C tile = alpha * (sum of the S partial tiles, in order) + beta * C for the tail tiles of the
// compact triangle (lin0 + blockIdx.x).
__global__ __launch_bounds__(kBlock) void SyrkTailFixupKernel(int64_t M, int64_t lin0, int S,
                                                              const float* __restrict__ P, float alpha,
                                                              float beta, float* C, int64_t ldc) {
  const int64_t lin = lin0 + blockIdx.x;
  int64_t I = static_cast<int64_t>((sqrt(8.0 * static_cast<double>(lin) + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= lin) ++I;
  while (I * (I + 1) / 2 > lin) --I;
  const int64_t i0 = I * MT, j0 = (lin - I * (I + 1) / 2) * MT;
  const float* p0 = P + static_cast<int64_t>(blockIdx.x) * S * (MT * MT);
  const int per = MT * MT / static_cast<int>(gridDim.y);  // gridDim.y workgroups share a tile
  const int e_begin = static_cast<int>(blockIdx.y) * per;
  for (int e = e_begin + threadIdx.x; e < e_begin + per; e += kBlock) {
    const int64_t i = i0 + (e & (MT - 1)), j = j0 + (e >> 7);
    if (i >= M || j >= M) continue;
    float sum = p0[e];
    for (int c = 1; c < S; ++c) sum += p0[static_cast<int64_t>(c) * (MT * MT) + e];
    float* dst = C + i + j * ldc;
    const float v = alpha * sum;
    *dst = (beta == 0.0f) ? v : v + beta * (*dst);
  }
}

// ------------------------------------------------------------------------------------------
// f64 MFMA kernel (the fp64 mode's GEMMs: the reference arithmetic is fp64, linear/linear_map.h:35)
//   v_mfma_f64_16x16x4_f64: A / B one f64 per lane (A[l&15][k = l>>4], B[k = l>>4][l&15]),
//   C / D four f64 per lane with  col = l & 15, row = (l >> 4) + 4 * reg  (NOT the f32 map).
// Same geometry as the f32 kernel: 128 x 128 tile, 4 waves as 2 x 2, each wave 4 x 4 blocks of
// 16 x 16; the MFMA "A" operand takes the B tile, so a register's lanes 0..15 hit 16 consecutive
// rows of column-major C (128-byte stores).  k slabs of 16, staged through LDS as [k][row].
// ------------------------------------------------------------------------------------------
constexpr int DK = 16;           // k slab
constexpr int DLD = MT + 2;      // LDS row stride (doubles)

using f64x4 = __attribute__((ext_vector_type(4))) double;

__device__ inline void StageTileF64(double (*S)[DLD], const double* __restrict__ X, int64_t ld,
                                    bool contiguous_r, int64_t r0, int64_t k0, int64_t R, int64_t K) {
  const int t = threadIdx.x;
  if (contiguous_r) {
    const int rr = t & 127;
    const int kb = t >> 7;  // 0..1
    const int64_t gr = r0 + rr;
#pragma unroll
    for (int p = 0; p < DK / 2; ++p) {
      const int kk = kb + 2 * p;
      const int64_t gk = k0 + kk;
      S[kk][rr] = (gr < R && gk < K) ? X[gr + gk * ld] : 0.0;
    }
  } else {
    const int kk = t & 15;
    const int rb = t >> 4;  // 0..15
    const int64_t gk = k0 + kk;
#pragma unroll
    for (int p = 0; p < MT / 16; ++p) {
      const int rr = rb + 16 * p;
      const int64_t gr = r0 + rr;
      S[kk][rr] = (gr < R && gk < K) ? X[gk + gr * ld] : 0.0;
    }
  }
}

__global__ __launch_bounds__(kBlock) void GemmMfmaF64Kernel(
    int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha,
    const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
    double beta, double* C, int64_t ldc, int lower_only, int64_t sA, int64_t sB, int64_t sC,
    int64_t n1, int64_t sA2, int64_t sB2) {
  {
    const int64_t z = blockIdx.z, z1 = z % n1, z2 = z / n1;
    A += z1 * sA + z2 * sA2;
    B += z1 * sB + z2 * sB2;
    C += z * sC;
  }
  __shared__ double As[DK][DLD];
  __shared__ double Bs[DK][DLD];
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * MT;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * MT;
  if (lower_only && i0 + MT <= j0) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = wave & 1, wj = wave >> 1;  // wavefront's 64 x 64 quadrant
  const int l15 = lane & 15, lk = lane >> 4;

  f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0;

  for (int64_t k0 = 0; k0 < K; k0 += DK) {
    StageTileF64(As, A, lda, transA == 0, i0, k0, M, K);
    StageTileF64(Bs, B, ldb, transB != 0, j0, k0, N, K);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < DK; kk += 4) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = As[kk + lk][wi * 64 + a * 16 + l15];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = Bs[kk + lk][wj * 64 + b * 16 + l15];
      // D'[j][i] += Bop[k][j] * Aop[i][k]: MFMA "A" operand = B values, "B" operand = A values
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b], av[a], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }

  // acc[a][b][reg]: i_local = l15 (the MFMA's column), j_local = lk + 4 * reg (its row)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = i0 + wi * 64 + a * 16 + l15;
    if (i >= M) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + wj * 64 + b * 16 + lk + 4 * r;
        if (j >= N) continue;
        double* c = C + i + j * ldc;
        const double v = alpha * acc[a][b][r];
        *c = (beta == 0.0) ? v : v + beta * (*c);
      }
    }
  }
}

// ---- small products ---------------------------------------------------------------------------------
// A product whose result fits a handful of 128 x 128 tiles (the 100 x 100 factors of the
// reference's robust-PCA benchmark, the panels of a small SVD) ran on ONE workgroup of the
// matrix-core kernel: 22 us for 100-cubed, prologue and latency, not arithmetic.  Here every
// 32 x 32 block of the result is a workgroup (16 of them at 100 x 100), k in chunks of 32 through
// LDS, a 2 x 2 block per thread, the sums in plain ascending-k order.
template <class T>
__global__ __launch_bounds__(256) void GemmSmallKernel(int tA, int tB, int64_t M, int64_t N, int64_t K, T alpha,
                                                       const T* __restrict__ A, int64_t lda,
                                                       const T* __restrict__ B, int64_t ldb, T beta, T* C,
                                                       int64_t ldc) {
  __shared__ T As[32][33], Bs[32][33];  // [k][i], [k][j]
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * 32, j0 = static_cast<int64_t>(blockIdx.y) * 32;
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  T acc[2][2] = {{T(0), T(0)}, {T(0), T(0)}};
  for (int64_t k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = t + 256 * e;
      {  // op(A)(i, k): A[i + k lda] (lanes along i) or A[k + i lda] (lanes along k)
        const int ii = tA ? idx >> 5 : idx & 31, kk = tA ? idx & 31 : idx >> 5;
        const int64_t i = i0 + ii, k = k0 + kk;
        As[kk][ii] = (i < M && k < K) ? (tA ? A[k + i * lda] : A[i + k * lda]) : T(0);
      }
      {  // op(B)(k, j): B[k + j ldb] (lanes along k) or B[j + k ldb] (lanes along j)
        const int kk = tB ? idx >> 5 : idx & 31, jj = tB ? idx & 31 : idx >> 5;
        const int64_t k = k0 + kk, j = j0 + jj;
        Bs[kk][jj] = (k < K && j < N) ? (tB ? B[j + k * ldb] : B[k + j * ldb]) : T(0);
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      const T a0 = As[kk][2 * tx], a1 = As[kk][2 * tx + 1], b0 = Bs[kk][2 * ty], b1 = Bs[kk][2 * ty + 1];
      acc[0][0] += a0 * b0;
      acc[1][0] += a1 * b0;
      acc[0][1] += a0 * b1;
      acc[1][1] += a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int64_t i = i0 + 2 * tx + a, j = j0 + 2 * ty + b;
      if (i >= M || j >= N) continue;
      T* c = C + i + j * ldc;
      const T v = alpha * acc[a][b];
      *c = (beta == T(0)) ? v : v + beta * (*c);
    }
}

bool SmallGemmWanted(int64_t M, int64_t N, int64_t K) {
  static const bool off = [] {
    const char* e = std::getenv("EPSILON_HIP_GEMM_SMALL");
    return e && e[0] == '0';
  }();
  return !off && M >= 1 && N >= 1 && M <= 256 && N <= 256 && K <= 4096;
}

int GemmMode() {  // 0 auto, 1 generic, 2 mfma, 3 mfma without the pipelined kernel
  // (read on every call: eps_set_option("gemm", ...) switches it between products in the tests)
  const char* e = std::getenv("EPSILON_HIP_GEMM");
  int mode = 0;
  if (e && std::strcmp(e, "generic") == 0) mode = 1;
  if (e && std::strcmp(e, "mfma") == 0) mode = 2;
  if (e && std::strcmp(e, "mfma_simple") == 0) mode = 3;
  return mode;
}

}  // namespace

void Gemm(bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha,
          const DVec& A, int64_t lda, const DVec& B, int64_t ldb, double beta, const DVec& C,
          int64_t ldc, bool lower_only) {
  // large f32 products (the Gram product, the GEMMs of the Cholesky inverse): f16 matrix cores
  // with split operands
  if (A.dt == F32 && GemmMode() == 0 &&
      GemmSplitF16(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower_only))
    return;
  // A long contraction into a small result (X^T R of the multiclass hinge: 784 x 10 out of
  // K = 60000) has a handful of output tiles, each looping over all of K on one workgroup - 7
  // workgroups on 256 CUs, 7.3 ms.  Split K over batches into partial results and add them in
  // a fixed order (deterministic): 29 x 7 workgroups, ~0.1 ms.
  // a few right-hand sides: a mat-vec that reads the matrix once, not a matrix product
  if (!transB && !lower_only && N <= 16 && M * K >= (int64_t(1) << 18) &&
      MultiGemv(transA, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc))
    return;
  if (!lower_only && GemmMode() == 0 && SmallGemmWanted(M, N, K)) {
    EPS_CHECK(A.dt == B.dt && A.dt == C.dt);
    const int64_t a_rows = transA ? K : M, a_cols = transA ? M : K, b_rows = transB ? N : K, b_cols = transB ? K : N;
    EPS_CHECK_MSG(lda >= a_rows && ldb >= b_rows && ldc >= M, "gemm: bad leading dimension");
    EPS_CHECK_MSG(K == 0 || (A.n >= (a_cols - 1) * lda + a_rows && B.n >= (b_cols - 1) * ldb + b_rows),
                  "gemm: operand buffer too small");
    EPS_CHECK_MSG(C.n >= (N - 1) * ldc + M, "gemm: C buffer too small");
    ProfScope prof("gemm_small", M * N, K);
    const dim3 grid(static_cast<unsigned>((M + 31) / 32), static_cast<unsigned>((N + 31) / 32));
    hipStream_t st = Runtime::Get().stream();
    if (A.dt == F32)
      hipLaunchKernelGGL(GemmSmallKernel<float>, grid, dim3(256), 0, st, transA ? 1 : 0, transB ? 1 : 0, M, N, K,
                         static_cast<float>(alpha), A.as<float>(), lda, B.as<float>(), ldb, static_cast<float>(beta),
                         C.as<float>(), ldc);
    else
      hipLaunchKernelGGL(GemmSmallKernel<double>, grid, dim3(256), 0, st, transA ? 1 : 0, transB ? 1 : 0, M, N, K,
                         alpha, A.as<double>(), lda, B.as<double>(), ldb, beta, C.as<double>(), ldc);
    EPS_HIP(hipGetLastError());
    return;
  }
  const int64_t tiles = ((M + MT - 1) / MT) * ((N + MT - 1) / MT);
  if (!lower_only && tiles <= 64 && K >= 8192 && M > 0 && N > 0) {
    int64_t nsplit = std::min<int64_t>(K / 2048, 512 / tiles);
    if (nsplit >= 2) {
      const int64_t kc = ((K + nsplit - 1) / nsplit + 63) / 64 * 64;
      const int64_t nfull = K / kc, rem = K - nfull * kc;
      const int64_t parts = nfull + (rem > 0 ? 1 : 0);
      DVec P = DVec::Empty(parts * M * N, C.dt);
      const int64_t sA = transA ? kc : kc * lda, sB = transB ? kc * ldb : kc;
      GemmBatched(transA, transB, M, N, kc, alpha, A, lda, sA, B, ldb, sB, 0.0, P, M, M * N, nfull,
                  false, 1, 0, 0);
      if (rem > 0) {
        const int64_t oA = nfull * sA, oB = nfull * sB;
        GemmBatched(transA, transB, M, N, rem, alpha, A.Slice(oA, A.n - oA), lda, 0,
                    B.Slice(oB, B.n - oB), ldb, 0, 0.0, P.Slice(nfull * M * N, M * N), M, 0, 1, false, 1,
                    0, 0);
      }
      if (ldc == M) {
        ReducePartials(M * N, static_cast<int>(parts), P, 1.0, beta, C.Slice(0, M * N));
      } else {
        DVec S = DVec::Empty(M * N, C.dt);
        ReducePartials(M * N, static_cast<int>(parts), P, 1.0, 0.0, S);
        for (int64_t j = 0; j < N; ++j)  // strided C: column by column (N is small here)
          Axpby(C.Slice(j * ldc, M), 1.0, S.Slice(j * M, M), beta);
      }
      return;
    }
  }
  GemmBatched(transA, transB, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1,
              lower_only, 1, 0, 0);
}

void GemmBatched(bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha,
                 const DVec& A, int64_t lda, int64_t sA, const DVec& B, int64_t ldb, int64_t sB,
                 double beta, const DVec& C, int64_t ldc, int64_t sC, int64_t batch,
                 bool lower_only, int64_t outer, int64_t sA2, int64_t sB2) {
  EPS_CHECK(A.dt == B.dt && A.dt == C.dt);
  if (M == 0 || N == 0 || batch == 0 || outer == 0) return;
  const int64_t n1 = batch;  // inner batch count; C is indexed by the flat index
  batch = batch * outer;
  EPS_CHECK_MSG(batch >= 1 && batch <= 65535, "gemm: batch count out of range");
  const int64_t a_rows = transA ? K : M, a_cols = transA ? M : K;
  const int64_t b_rows = transB ? N : K, b_cols = transB ? K : N;
  EPS_CHECK_MSG(lda >= a_rows && ldb >= b_rows && ldc >= M, "gemm: bad leading dimension");
  EPS_CHECK_MSG(K == 0 || A.n >= (n1 - 1) * sA + (outer - 1) * sA2 + (a_cols - 1) * lda + a_rows,
                "gemm: A buffer too small");
  EPS_CHECK_MSG(K == 0 || B.n >= (n1 - 1) * sB + (outer - 1) * sB2 + (b_cols - 1) * ldb + b_rows,
                "gemm: B buffer too small");
  EPS_CHECK_MSG(C.n >= (batch - 1) * sC + (N - 1) * ldc + M, "gemm: C buffer too small");
  if (lower_only) EPS_CHECK(M == N);
  hipStream_t s = Runtime::Get().stream();
  ProfScope prof(lower_only ? "syrk" : "gemm", M * N, K);
  const int mode = GemmMode();
  const bool use_mfma = A.dt == F32 && mode != 1 &&
                        (mode >= 2 || (M >= 64 && N >= 64 && K >= 32));
  if (use_mfma) {
    dim3 grid(static_cast<unsigned>((M + MT - 1) / MT), static_cast<unsigned>((N + MT - 1) / MT),
              static_cast<unsigned>(batch));
    const bool aligned = lda % 4 == 0 && ldb % 4 == 0 && sA % 4 == 0 && sB % 4 == 0 &&
                         sA2 % 4 == 0 && sB2 % 4 == 0 &&
                         reinterpret_cast<uintptr_t>(A.data()) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(B.data()) % 16 == 0;
    if (aligned && mode != 3) {
      const float al = static_cast<float>(alpha), be = static_cast<float>(beta);
      int lo = lower_only ? 1 : 0;
      static const bool compact = std::getenv("EPSILON_HIP_SYRK_2D") == nullptr;
      int64_t lin0 = 0, kchunk = 0, tail = 0;
      int S = 1;
      float* P = nullptr;
      std::shared_ptr<Buffer> pbuf;
      if (lower_only && batch == 1 && compact) {
        const int64_t T = (M + MT - 1) / MT;
        const int64_t total = T * (T + 1) / 2;
        grid = dim3(static_cast<unsigned>(total), 1, 1);
        lo = 2;
        // A long contraction (the Gram product: 3160 tiles of ~6 ms each at config 2) runs in
        // rounds of 512 resident workgroups and ends in a ragged one - 88 tiles on a chip that
        // holds 512, a seventh round as long as the six full ones.  The tail tiles are split over
        // K instead (S chunks each, S * tail <= 512: one short round), their partial tiles summed
        // in a fixed order by a fix-up launch.
        static const bool split_tail = [] {
          const char* e = std::getenv("EPSILON_HIP_SYRK_TAIL");
          return !(e && e[0] == '0');
        }();
        const int64_t slots = 512;
        tail = total % slots;
        if (split_tail && K >= 8192 && total > slots && tail > 0 && tail <= slots / 2) {
          S = static_cast<int>(std::min<int64_t>(8, slots / tail));
          kchunk = ((K + S - 1) / S + MK - 1) / MK * MK;
          S = static_cast<int>((K + kchunk - 1) / kchunk);
          pbuf = Runtime::Get().Alloc(static_cast<size_t>(tail) * S * MT * MT * sizeof(float));
          P = static_cast<float*>(pbuf->p);
          grid = dim3(static_cast<unsigned>(total - tail), 1, 1);
        } else {
          tail = 0;
        }
      }
#define EPS_PIPE(CA, CB, GRID, LIN0, KCH, PP)                                                     \
  hipLaunchKernelGGL((GemmMfmaF32PipeKernel<CA, CB>), GRID, dim3(kBlock), 0, s, M, N, K, al,       \
                     A.as<float>(), lda, B.as<float>(), ldb, be, C.as<float>(), ldc, lo, sA, sB, sC, n1, sA2, sB2, \
                     static_cast<int64_t>(LIN0), static_cast<int64_t>(KCH), PP)
#define EPS_PIPE_ALL(GRID, LIN0, KCH, PP)                          \
  do {                                                             \
    if (!transA && transB) EPS_PIPE(true, true, GRID, LIN0, KCH, PP);        \
    else if (!transA && !transB) EPS_PIPE(true, false, GRID, LIN0, KCH, PP); \
    else if (transA && transB) EPS_PIPE(false, true, GRID, LIN0, KCH, PP);   \
    else EPS_PIPE(false, false, GRID, LIN0, KCH, PP);                        \
  } while (0)
      EPS_PIPE_ALL(grid, lin0, 0, static_cast<float*>(nullptr));
      if (tail > 0) {
        const int64_t first = static_cast<int64_t>(grid.x);
        dim3 tgrid(static_cast<unsigned>(tail), static_cast<unsigned>(S), 1);
        EPS_PIPE_ALL(tgrid, first, kchunk, P);
        hipLaunchKernelGGL(SyrkTailFixupKernel, dim3(static_cast<unsigned>(tail), 8), dim3(kBlock), 0, s, M,
                           first, S, P, al, be, C.as<float>(), ldc);
      }
#undef EPS_PIPE_ALL
#undef EPS_PIPE
      return;
    }
    hipLaunchKernelGGL(GemmMfmaF32Kernel, grid, dim3(kBlock), 0, s, transA ? 1 : 0,
                       transB ? 1 : 0, M, N, K, static_cast<float>(alpha), A.as<float>(), lda,
                       B.as<float>(), ldb, static_cast<float>(beta), C.as<float>(), ldc,
                       lower_only ? 1 : 0, sA, sB, sC, n1, sA2, sB2);
    return;
  }
  // fp64: the software-pipelined MFMA kernel (kernels_gemm_f64.hip) for everything with at least
  // half a tile of output; the plain MFMA kernel below it (19.4 TFLOP/s on 4096^3, against 28.9
  // for the VALU tile kernel) is kept as EPSILON_HIP_GEMM=mfma_simple and for unaligned operands
  // of EPSILON_HIP_GEMM=mfma.
  if (A.dt == F64 && mode != 1 && mode != 3 && (mode == 2 || (M >= 64 && N >= 64 && K >= 32)) &&
      GemmF64Pipe(transA, transB, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, n1, batch,
                  lower_only, sA2, sB2))
    return;
  if (A.dt == F64 && mode >= 2) {
    dim3 grid(static_cast<unsigned>((M + MT - 1) / MT), static_cast<unsigned>((N + MT - 1) / MT),
              static_cast<unsigned>(batch));
    hipLaunchKernelGGL(GemmMfmaF64Kernel, grid, dim3(kBlock), 0, s, transA ? 1 : 0, transB ? 1 : 0, M, N,
                       K, alpha, A.as<double>(), lda, B.as<double>(), ldb, beta, C.as<double>(), ldc,
                       lower_only ? 1 : 0, sA, sB, sC, n1, sA2, sB2);
    return;
  }
  dim3 grid(static_cast<unsigned>((M + GT - 1) / GT), static_cast<unsigned>((N + GT - 1) / GT),
            static_cast<unsigned>(batch));
  if (A.dt == F32) {
    hipLaunchKernelGGL(GemmGenericKernel<float>, grid, dim3(kBlock), 0, s, transA ? 1 : 0,
                       transB ? 1 : 0, M, N, K, static_cast<float>(alpha), A.as<float>(), lda,
                       B.as<float>(), ldb, static_cast<float>(beta), C.as<float>(), ldc,
                       lower_only ? 1 : 0, sA, sB, sC, n1, sA2, sB2);
  } else {
    hipLaunchKernelGGL(GemmGenericKernel<double>, grid, dim3(kBlock), 0, s, transA ? 1 : 0,
                       transB ? 1 : 0, M, N, K, alpha, A.as<double>(), lda, B.as<double>(), ldb,
                       beta, C.as<double>(), ldc, lower_only ? 1 : 0, sA, sB, sC, n1, sA2, sB2);
  }
}

}  // namespace k
}  // namespace eps

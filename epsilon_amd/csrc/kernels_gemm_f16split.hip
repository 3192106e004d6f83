// K4, long contractions: the Gram product A A^T on the f16 matrix cores with a two-term operand
// split - fp32 accuracy at several times the fp32 MFMA rate.
//
// gfx950 runs v_mfma_f32_32x32x16_f16 at 16x the rate of the exact-f32 MFMA (2.5 PFLOP/s against
// 157 TFLOP/s) and accumulates in f32.  An f32 value is split into two f16 terms,
//       a * s_i = h + l ,   h = f16(a s_i),  l = f16(a s_i - h) ,
// with one power-of-two scale s_i per ROW (max_k |a_ik| s_i in [1024, 2048): every row uses the
// top of the f16 range whatever its norm).  h carries 11 bits of a s_i, l the next 11 (fewer only
// for entries below 2^-13 of their row's maximum, whose l is subnormal: an absolute error of
// 2^-25, i.e. 1e-11 of the row's scale): the pair represents a s_i to 2^-22 relative - two ulps
// of f32 - and
//       (a s_i)(b s_j) = h_a h_b + h_a l_b + l_a h_b + O(2^-22) ,
// each of the three products exact in f32 (11 x 11 bits), all three into ONE f32 accumulator.
// The error against exact arithmetic is that of an f32 GEMM whose inputs were perturbed by two
// ulps - the order of the f32 MFMA's own accumulation rounding over K = 5e4 terms - at three
// f16 MFMAs per f32 one: 3/16 of the matrix-core time.  The epilogue divides by s_i s_j (exact).
//
// The reference forms this product with dgemm_ (linear/linear_map_multiply.cc:14-37); it is the
// one place the north star puts on the matrix cores, and 60 % of the time to OPTIMAL at config 2.
//
// Layout.  A (m x K, column-major f32) is converted once into two f16 arrays stored slab-major,
// [K / 32][m_pad][32]: the 32 k-values of a row sit in 64 contiguous bytes, the rows of a tile in
// 16 KB contiguous per slab - every global load of the product kernel is a full 16-byte lane load
// of exactly what an MFMA fragment needs (8 consecutive k of one row).  Tile 256 x 256 per
// 512-thread workgroup (8 waves as 2 x 4, each 128 x 64 = 4 x 2 blocks of 32 x 32), k slabs of 32,
// operands staged through LDS (80-byte rows: conflict-free 16-byte fragment reads), the next
// slab's global loads in flight while the current one multiplies.  Only tiles on and below the
// diagonal are computed (compact 1-D grid); the caller mirrors.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TS = 256;       // tile rows / columns
constexpr int SK = 32;        // k slab
constexpr int LROW = 40;      // LDS row stride in halfs (80 bytes)
constexpr int kThreads = 512;

// ---- max_k |a_ik| per row (bit pattern of a non-negative float: integer order = float order) -------
__global__ __launch_bounds__(256) void RowAbsMaxKernel(const float* __restrict__ A, int64_t M, int64_t K,
                                                       int64_t lda, unsigned* __restrict__ rowmax_bits) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= M) return;
  const int64_t per = (K + gridDim.y - 1) / gridDim.y;
  const int64_t k0 = blockIdx.y * per;
  int64_t k1 = k0 + per;
  if (k1 > K) k1 = K;
  float mx = 0.0f;
  const float* p = A + i;
  int64_t k = k0;
  for (; k + 4 <= k1; k += 4) {  // lanes run along the rows: every load is a coalesced 256-byte line
    const float a = p[k * lda], b = p[(k + 1) * lda], c = p[(k + 2) * lda], d = p[(k + 3) * lda];
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(d)));
  }
  for (; k < k1; ++k) mx = fmaxf(mx, fabsf(p[k * lda]));
  atomicMax(rowmax_bits + i, __float_as_uint(mx));
}

// 1 / SplitScale (a power of two: exact), without the division
__device__ inline float SplitScaleInv(unsigned amax_bits) {
  const float amax = __uint_as_float(amax_bits);
  if (!(amax > 0.0f) || !isfinite(amax)) return 1.0f;
  int e;
  frexpf(amax, &e);
  int se = 11 - e;
  if (se > 120) se = 120;
  return ldexpf(1.0f, -se);
}

__device__ inline float SplitScale(unsigned amax_bits) {
  const float amax = __uint_as_float(amax_bits);
  if (!(amax > 0.0f) || !isfinite(amax)) return 1.0f;
  int e;
  frexpf(amax, &e);  // amax = f * 2^e, f in [0.5, 1)
  int se = 11 - e;   // amax * 2^se in [1024, 2048)
  if (se > 120) se = 120;  // (rows of subnormal size: keep the scale itself representable)
  return ldexpf(1.0f, se);
}

// ---- f32 column-major -> two f16 arrays, slab-major [K/32][m_pad][32] ------------------------------
// One workgroup converts a 64-row x 32-k block through an LDS transpose: reads are contiguous
// along the rows (the matrix' storage order), writes are 16 bytes per lane along k.
__global__ __launch_bounds__(256) void SplitConvertKernel(const float* __restrict__ A, int64_t M, int64_t K,
                                                          int64_t lda, int64_t m_pad,
                                                          const unsigned* __restrict__ rowmax_bits,
                                                          _Float16* __restrict__ H, _Float16* __restrict__ L) {
  __shared__ float tile[SK][65];
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * 64;
  const int64_t ks = blockIdx.y;
  const int t = threadIdx.x;
  {
    const int r = t & 63, kq = t >> 6;  // 4 k's per pass
    const float s = i0 + r < M ? SplitScale(rowmax_bits[i0 + r]) : 1.0f;
#pragma unroll
    for (int p = 0; p < SK / 4; ++p) {
      const int kk = kq + 4 * p;
      const int64_t i = i0 + r, k = ks * SK + kk;
      tile[kk][r] = (i < M && k < K) ? A[i + k * lda] * s : 0.0f;
    }
  }
  __syncthreads();
  {
    const int r = t >> 2, seg = t & 3;  // 64 rows x 4 segments of 8 k
    half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = tile[seg * 8 + j][r];
      const _Float16 hv = static_cast<_Float16>(v);
      h[j] = hv;
      l[j] = static_cast<_Float16>(v - static_cast<float>(hv));
    }
    const int64_t off = ((ks * m_pad + i0 + r) * SK) + seg * 8;
    if (i0 + r < m_pad) {
      *reinterpret_cast<half8*>(H + off) = h;
      *reinterpret_cast<half8*>(L + off) = l;
    }
  }
}

// The same for an operand stored contiguously along k (X[k + r * ld]): one wave per row for the
// maximum, 16-byte lane writes for the conversion.
__global__ __launch_bounds__(256) void RowAbsMaxKKernel(const float* __restrict__ X, int64_t R, int64_t K,
                                                        int64_t ld, unsigned* __restrict__ rowmax_bits) {
  const int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const float* p = X + r * ld;
  float mx = 0.0f;
  for (int64_t k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(p[k]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
  if (lane == 0) rowmax_bits[r] = __float_as_uint(mx);
}

__global__ __launch_bounds__(256) void SplitConvertKKernel(const float* __restrict__ X, int64_t R, int64_t K,
                                                           int64_t ld, int64_t r_pad,
                                                           const unsigned* __restrict__ rowmax_bits,
                                                           _Float16* __restrict__ H, _Float16* __restrict__ L) {
  const int64_t r = blockIdx.x * 64ll + (threadIdx.x >> 2);
  const int seg = threadIdx.x & 3;
  const int64_t ks = blockIdx.y;
  if (r >= r_pad) return;
  half8 h, l;
  const float s = r < R ? SplitScale(rowmax_bits[r]) : 1.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t k = ks * SK + seg * 8 + j;
    const float v = (r < R && k < K) ? X[k + r * ld] * s : 0.0f;
    const _Float16 hv = static_cast<_Float16>(v);
    h[j] = hv;
    l[j] = static_cast<_Float16>(v - static_cast<float>(hv));
  }
  const int64_t off = ((ks * r_pad + r) * SK) + seg * 8;
  *reinterpret_cast<half8*>(H + off) = h;
  *reinterpret_cast<half8*>(L + off) = l;
}

// One converted operand: rows = the output index it contributes (i for op(A), j for op(B)).
struct SplitOperand {
  const _Float16* H;
  const _Float16* L;
  const unsigned* rowmax;  // per-row maxima (bit patterns) -> the row's power-of-two scale
  int64_t rows_pad;
};

// ---- the product -----------------------------------------------------------------------------------
// C (M x N) = alpha * X_a X_b^T + beta * C with X_a = op(A) (rows i), X_b = op(B)^T (rows j).
// tri != 0: M == N and only the tiles on and below the diagonal (compact 1-D grid, lin0 offset).
// `order` (optional): the tiles as packed (I << 16 | J) words in PATCH-MAJOR order (8 x 8 patches,
// row-major inside), and workgroup -> tile through the XCD-aware permutation below.  Workgroups
// are dealt round-robin over the 8 XCDs (b and b + 8 share one L2) and a CU holds one of them,
// so the 32 that an XCD runs together are slots [32 r, 32 r + 32) of its own sequence: with the
// permutation those are 4 x 8 neighbouring tiles - 12 operand panels for 32 tiles in that L2
// instead of ~40 (every panel of the matrix).  Worth 1-4 % (measured, round 3).
// Staging variants (EPSILON_HIP_GEMM_STAGE; measured on the Gram product of config 2, random
// operands, counters in profiles/r03_gemm_split_pmc.txt):
//   V = 3 "ring" (default): LDS-DMA into a ring of four 16-deep steps, the barrier in the middle
//          of a step's matrix instructions: matrix pipe busy 87 % (the chip then holds 1.41 GHz:
//          the product sits at the power limit, not at a stall), 11.8 ms;
//   V = 1 "lds": LDS-DMA (global_load_lds: no staging registers, no LDS write pass), two
//          32-deep buffers, one barrier per slab: busy 66 % at 1.7 GHz, 12.5 ms;
//   V = 0 "reg": round 2 - register-staged slabs, padded LDS image (it was NOT conflict free:
//          SQ_LDS_BANK_CONFLICT 6e8 against 0 for the swizzled images), two barriers per slab:
//          busy 54 % at 1.8 GHz, 14.8 ms.
template <int V>
__global__ __launch_bounds__(kThreads, 2) void GemmSplitF16Kernel(
    int64_t M, int64_t N, int64_t nslab, SplitOperand oa, SplitOperand ob, float alpha, float beta, float* C,
    int64_t ldc, int tri, int64_t lin0, int64_t slab0, int64_t slab_count, float* __restrict__ P,
    const int* __restrict__ order, int64_t perm_limit) {
  const int64_t lin = lin0 + blockIdx.x;
  int64_t I, J;
  if (order != nullptr) {
    int64_t tpos = lin;
    if (lin < perm_limit) {  // perm_limit is a multiple of 256: a bijection of [0, perm_limit)
      const int64_t xcd = lin & 7, slot = lin >> 3;
      tpos = ((((slot >> 5) << 3) + xcd) << 5) + (slot & 31);
    }
    const int w = order[tpos];
    I = w >> 16;
    J = w & 0xffff;
  } else if (tri == 1 || tri == 2) {  // tile (I, J), I >= J, from the linear index over the lower triangle
    I = static_cast<int64_t>((sqrt(8.0 * static_cast<double>(lin) + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= lin) ++I;
    while (I * (I + 1) / 2 > lin) --I;
    J = lin - I * (I + 1) / 2;
  } else {
    const int64_t TI = (M + TS - 1) / TS;
    I = lin % TI;
    J = lin / TI;
  }
  const int64_t i0 = I * TS, j0 = J * TS;
  // split-K form (tail tiles): blockIdx.y selects a run of slabs
  int64_t ks0 = slab0, ks1 = slab0 + slab_count;
  // tri == 2: X^T X of a LOWER-TRIANGULAR X (rows of the operands = columns of X, k = its rows):
  // column i of X is zero above row i, so tile (I, J), I >= J, only has terms from k >= i0 on
  if (tri == 2 && i0 / SK > ks0) ks0 = i0 / SK;
  // full grid, one operand lower triangular (the doubling levels of the triangular inverse):
  // tri == 3: B(k, j) = 0 for k < j - terms from k >= j0 on; tri == 4: A(i, k) = 0 for k > i -
  // terms up to the tile's last row
  if (tri == 3 && j0 / SK > ks0) ks0 = j0 / SK;
  if (tri == 4 && (i0 + TS + SK - 1) / SK < ks1) ks1 = (i0 + TS + SK - 1) / SK;
  if (P != nullptr) {
    ks0 = slab0 + static_cast<int64_t>(blockIdx.y) * slab_count;
    ks1 = ks0 + slab_count;
  }
  if (ks1 > nslab) ks1 = nslab;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wi = wave & 1, wj = wave >> 1;  // 2 x 4 waves: rows wi*128, columns wj*64
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  if constexpr (V == 3) {
    // V == 3: a RING of four 16-deep steps (4 x 4 arrays x 256 rows x 32 bytes = 128 KB), the
    // barrier in the MIDDLE of a step's matrix instructions and the next step's fragments read
    // into a second register set - nothing at a step boundary waits.  Step s computes from buffer
    // s & 3.  B(s), the barrier inside step s, is preceded by `vmcnt(4)`: this wave's pieces of
    // step s + 1 have landed (the 4 of step s + 2 may still fly), so behind B(s) every wave may
    // read step s + 1 - and every wave has left step s - 1, whose buffer is the one step s + 3
    // goes to: its 4 loads are issued right behind B(s) and have two whole steps to land.
    //   issue stage(u) after B(u - 3);  wait for it before B(u - 1);  read it after B(u - 1).
    // Past the last step the stream stays branch-free: the last step is staged again (into
    // buffers nobody reads any more) and drained before the epilogue.
    // LDS image of an array and step: the 32-byte row pieces in row order (what a wave's DMA of
    // 32 rows x 2 x 16 bytes writes), the two 16-byte halves of a row swapped where (row >> 3) & 1:
    // the 16 lanes of a fragment read (16 rows, one half) cover the 16 slots of a bank row.
    __shared__ __attribute__((aligned(16))) _Float16 ring[4][4][TS * 16];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int arr_w = wave_u & 3, pg = wave_u >> 2;  // this wave stages rows pg*128 .. +128 of array arr_w
    int src_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = (pg * 4 + q) * 64 + lane;
      const int row = slot >> 1, seg = (slot & 1) ^ ((row >> 3) & 1);
      src_off[q] = row * SK + seg * 8;
    }
    const _Float16* arr_base = arr_w == 0   ? oa.H + i0 * SK
                               : arr_w == 1 ? oa.L + i0 * SK
                               : arr_w == 2 ? ob.H + j0 * SK
                                            : ob.L + j0 * SK;
    const int64_t arr_stride = (arr_w < 2 ? oa.rows_pad : ob.rows_pad) * SK;  // halfs per slab
    const int64_t nsteps = ks1 > ks0 ? 2 * (ks1 - ks0) : 0;
    auto stage_src = [&](int64_t step) {  // scalar arithmetic: placed IN FRONT of the barrier
      const int64_t sc = step < nsteps ? step : nsteps - 1;
      return arr_base + (ks0 + (sc >> 1)) * arr_stride + (sc & 1) * 16;
    };
    auto stage_issue = [&](int64_t step, const _Float16* src) {
      _Float16* dst = &ring[step & 3][arr_w][pg * 4 * 512];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + src_off[q]),
                                         (__attribute__((address_space(3))) void*)(dst + q * 512), 16, 0, 0);
    };
    auto stage = [&](int64_t step) { stage_issue(step, stage_src(step)); };
    const int sw = (l31 >> 3) & 1;
    const int fragI = (wi * 128 + l31) * 16 + ((lh ^ sw) << 3);  // + a * 512
    const int fragJ = (wj * 64 + l31) * 16 + ((lh ^ sw) << 3);   // + b * 512
    auto readf = [&](int64_t step, half8(&ih)[4], half8(&il)[4], half8(&jh)[2], half8(&jl)[2]) {
      const _Float16* B = &ring[step & 3][0][0];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        jh[b] = *reinterpret_cast<const half8*>(B + 2 * TS * 16 + fragJ + b * 512);
        jl[b] = *reinterpret_cast<const half8*>(B + 3 * TS * 16 + fragJ + b * 512);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        ih[a] = *reinterpret_cast<const half8*>(B + fragI + a * 512);
        il[a] = *reinterpret_cast<const half8*>(B + TS * 16 + fragI + a * 512);
      }
    };
    // 12 matrix instructions: the three product terms on the four accumulators of row blocks ap, ap + 1
    auto half_step = [&](int ap, const half8(&ih)[4], const half8(&il)[4], const half8(&jh)[2],
                         const half8(&jl)[2]) {
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int a = ap; a < ap + 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? jl[b] : jh[b], term == 1 ? il[a] : ih[a],
                                                               acc[a][b], 0, 0, 0);
    };
    if (nsteps > 0) {
      half8 aih[4], ail[4], ajh[2], ajl[2], bih[4], bil[4], bjh[2], bjl[2];
      stage(0);
      stage(1);
      stage(2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      readf(0, aih, ail, ajh, ajl);
      auto body = [&](int64_t st, const half8(&cih)[4], const half8(&cil)[4], const half8(&cjh)[2],
                      const half8(&cjl)[2], half8(&nih)[4], half8(&nil)[4], half8(&njh)[2], half8(&njl)[2]) {
        const _Float16* nsrc = stage_src(st + 3);
        half_step(0, cih, cil, cjh, cjl);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stage_issue(st + 3, nsrc);
        readf(st + 1, nih, nil, njh, njl);
        half_step(2, cih, cil, cjh, cjl);
        // issue order behind the barrier: the 4 staging loads and the 12 fragment reads between
        // the matrix instructions (whose pipe time hides their issue), not in front of them
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int64_t st = 0; st < nsteps; st += 2) {
        body(st, aih, ail, ajh, ajl, bih, bil, bjh, bjl);
        body(st + 1, bih, bil, bjh, bjl, aih, ail, ajh, ajl);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the workgroup's LDS
    }
    // Epilogue of an interior tile through LDS: the accumulator layout gives a store instruction
    // 32 consecutive rows of two columns (128-byte pieces, 4 bytes per lane, 128 instructions per
    // wave) - with a rank-256 update of the Cholesky trailing matrix, where the old C is read as
    // well, the epilogue IS the kernel.  Staged through the (now idle) ring, half a tile at a time
    // ([column][row], 128 KB), a wave instead moves whole 1 KB column pieces, 16 bytes per lane: a
    // quarter of the memory instructions, and 8 consecutive lines of a DRAM page per piece.  The
    // value written is the same expression as below, bit for bit.
    const bool interior = P == nullptr && i0 + TS <= M && j0 + TS <= N && ldc % 4 == 0 &&
                          reinterpret_cast<uintptr_t>(C) % 16 == 0;
    if (interior) {
      float* stage = reinterpret_cast<float*>(&ring[0][0][0]);  // 128 columns x 256 rows
      const int row4 = lane * 4;
      float ui[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) ui[e] = SplitScaleInv(oa.rowmax[i0 + row4 + e]);
      __syncthreads();  // every wave has left the main loop: the ring is free
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        if ((wj >> 1) == h) {
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int col = (wj & 1) * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                stage[col * TS + wi * 128 + a * 32 + l31] = acc[a][b][r];
              }
        }
        __syncthreads();
        // wave w: columns w, w + 8, ... of this half; lane: rows 4 lane .. 4 lane + 3.  The old
        // values in batches of 8 independent loads ahead of their stores (C may alias nothing
        // here, but the compiler cannot know: a load behind a store would wait for it)
#pragma unroll 1
        for (int qb = 0; qb < 16; qb += 8) {
          float4 old[8];
          if (beta != 0.0f) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int64_t j = j0 + h * 128 + wave + 8 * (qb + q);
              old[q] = *reinterpret_cast<const float4*>(C + i0 + row4 + j * ldc);
            }
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int c = wave + 8 * (qb + q);
            const int64_t j = j0 + h * 128 + c;
            const float uj = SplitScaleInv(ob.rowmax[j]);
            const float4 a4 = *reinterpret_cast<const float4*>(stage + c * TS + row4);
            float4 v;
            v.x = alpha * ((ui[0] * a4.x) * uj);
            v.y = alpha * ((ui[1] * a4.y) * uj);
            v.z = alpha * ((ui[2] * a4.z) * uj);
            v.w = alpha * ((ui[3] * a4.w) * uj);
            if (beta != 0.0f) {
              v.x = v.x + beta * old[q].x;
              v.y = v.y + beta * old[q].y;
              v.z = v.z + beta * old[q].z;
              v.w = v.w + beta * old[q].w;
            }
            *reinterpret_cast<float4*>(C + i0 + row4 + j * ldc) = v;
          }
        }
        __syncthreads();  // the next half overwrites the stage
      }
      return;
    }
  } else if constexpr (V >= 1) {
    // two buffers x (H_I, L_I, H_J, L_J) x 256 rows x 64 bytes = 128 KB.  A wave-instruction of
    // the LDS-DMA writes 64 x 16 bytes contiguously (16 rows), so the image is the global slab's
    // own order; the conflict-free form comes from a permutation of the four 16-byte segments of
    // a row, applied to the SOURCE address of the load and to the fragment read alike:
    // slot(row, seg) = 4 row + (seg ^ ((row >> 2) & 3)) - the 16 lanes of a ds_read_b128 group
    // (16 consecutive rows, one segment) then cover all 16 slots of the 256-byte bank row.
    __shared__ __attribute__((aligned(16))) _Float16 sm2[2][4][TS * SK];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // this thread's two pieces of an array: LDS slots wi * 64 + lane, wi = 2 wave + q
    int src_off[2];  // in halfs, inside the 256-row slab of one array
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int slot = (wave_u * 2 + q) * 64 + lane;
      const int row = slot >> 2, seg = (slot & 3) ^ ((row >> 2) & 3);
      src_off[q] = row * SK + seg * 8;
    }
    auto stage = [&](int buf, int64_t ks) {
      const _Float16* base[4] = {oa.H + (ks * oa.rows_pad + i0) * SK, oa.L + (ks * oa.rows_pad + i0) * SK,
                                 ob.H + (ks * ob.rows_pad + j0) * SK, ob.L + (ks * ob.rows_pad + j0) * SK};
#pragma unroll
      for (int arr = 0; arr < 4; ++arr)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(base[arr] + src_off[q]),
              (__attribute__((address_space(3))) void*)(&sm2[buf][arr][(wave_u * 2 + q) * 512]), 16, 0, 0);
    };
    // fragment offsets (halfs) of this lane inside an array: row-dependent part and the swizzle
    int offI[4], offJ[2], swI[4], swJ[2];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = wi * 128 + a * 32 + l31;
      offI[a] = row * SK;
      swI[a] = (row >> 2) & 3;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int row = wj * 64 + b * 32 + l31;
      offJ[b] = row * SK;
      swJ[b] = (row >> 2) & 3;
    }
    int cur = 0;
    if (ks0 < ks1) {
      stage(0, ks0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    {
      for (int64_t ks = ks0; ks < ks1; ++ks) {
        if (ks + 1 < ks1) stage(cur ^ 1, ks + 1);  // lands under this slab's MFMAs
#pragma unroll
        for (int kk = 0; kk < SK; kk += 16) {
          half8 ih[4], il[4], jh[2], jl[2];
          const int seg = kk / 8 + lh;
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const int off = offI[a] + ((seg ^ swI[a]) << 3);
            ih[a] = *reinterpret_cast<const half8*>(&sm2[cur][0][off]);
            il[a] = *reinterpret_cast<const half8*>(&sm2[cur][1][off]);
          }
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int off = offJ[b] + ((seg ^ swJ[b]) << 3);
            jh[b] = *reinterpret_cast<const half8*>(&sm2[cur][2][off]);
            jl[b] = *reinterpret_cast<const half8*>(&sm2[cur][3][off]);
          }
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(jh[b], ih[a], acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(jh[b], il[a], acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(jl[b], ih[a], acc[a][b], 0, 0, 0);
            }
        }
        // the matrix instructions are issued (their fragment reads have returned) before the
        // barrier: a wave past it overwrites this buffer
        __builtin_amdgcn_sched_barrier(0);
        // the next slab has landed (this wave's pieces), every wave is done reading this one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
      }
    }
  } else {
    // staging: a slab of one array and one side is 256 rows x 64 bytes = 1024 lane loads of 16 B,
    // two per thread; q = t + 512 p -> row q >> 2, segment q & 3
    __shared__ __attribute__((aligned(16))) _Float16 sm[4][TS * LROW];  // H_I, L_I, H_J, L_J: 80 KB
    half8 pre[4][2];
    auto gload = [&](int64_t ks) {
      const _Float16* hI = oa.H + (ks * oa.rows_pad + i0) * SK;
      const _Float16* lI = oa.L + (ks * oa.rows_pad + i0) * SK;
      const _Float16* hJ = ob.H + (ks * ob.rows_pad + j0) * SK;
      const _Float16* lJ = ob.L + (ks * ob.rows_pad + j0) * SK;
  #pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int q = t + kThreads * p;
        pre[0][p] = *reinterpret_cast<const half8*>(hI + q * 8);
        pre[1][p] = *reinterpret_cast<const half8*>(lI + q * 8);
        pre[2][p] = *reinterpret_cast<const half8*>(hJ + q * 8);
        pre[3][p] = *reinterpret_cast<const half8*>(lJ + q * 8);
      }
    };
    auto lstore = [&]() {
  #pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int q = t + kThreads * p;
        const int off = (q >> 2) * LROW + (q & 3) * 8;
  #pragma unroll
        for (int arr = 0; arr < 4; ++arr) *reinterpret_cast<half8*>(&sm[arr][off]) = pre[arr][p];
      }
    };
    if (ks0 < ks1) gload(ks0);
    for (int64_t ks = ks0; ks < ks1; ++ks) {
      __syncthreads();  // the previous slab's fragment reads are done
      lstore();
      __syncthreads();
      if (ks + 1 < ks1) gload(ks + 1);  // in flight under this slab's MFMAs
  #pragma unroll
      for (int kk = 0; kk < SK; kk += 16) {
        half8 ih[4], il[4], jh[2], jl[2];
  #pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int off = (wi * 128 + a * 32 + l31) * LROW + kk + 8 * lh;
          ih[a] = *reinterpret_cast<const half8*>(&sm[0][off]);
          il[a] = *reinterpret_cast<const half8*>(&sm[1][off]);
        }
  #pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int off = (wj * 64 + b * 32 + l31) * LROW + kk + 8 * lh;
          jh[b] = *reinterpret_cast<const half8*>(&sm[2][off]);
          jl[b] = *reinterpret_cast<const half8*>(&sm[3][off]);
        }
        // D'[j][i] += X_J[j][k] X_I[i][k]: the MFMA's "A" operand takes the J side, so a register's
        // lanes run along i - consecutive rows of column-major C
  #pragma unroll
        for (int a = 0; a < 4; ++a)
  #pragma unroll
          for (int b = 0; b < 2; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(jh[b], ih[a], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(jh[b], il[a], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(jl[b], ih[a], acc[a][b], 0, 0, 0);
          }
      }
    }

  }

  // raw partial tile of a split-K tail: scales and beta are the fix-up kernel's
  if (P != nullptr) {
    float* pt = P + (static_cast<int64_t>(blockIdx.x) * gridDim.y + blockIdx.y) * (TS * TS);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int il = wi * 128 + a * 32 + l31;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pt[il + (wj * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TS] = acc[a][b][r];
    }
    return;
  }
  // Undo the row scales: the accumulator holds s_i s_j (X_a X_b^T)_ij; 1 / s is a power of two.
  // A short contraction (the rank-256 updates of the Cholesky inverse) spends most of its time
  // here, so the epilogue is laid out for memory latency: the 16 column scales of a block and
  // the 16 old values of C (beta != 0) are loaded as independent batches from clamped addresses
  // BEFORE the block's stores - interleaved with the stores through a pointer that may alias,
  // every load waited for the store before it (~100 us per tile at K = 256).
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float uj[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t j = j0 + wj * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      uj[r] = SplitScaleInv(ob.rowmax[j < N ? j : N - 1]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t i = i0 + wi * 128 + a * 32 + l31;
      const int64_t ic = i < M ? i : M - 1;
      const float ui = SplitScaleInv(oa.rowmax[ic]);
      float old[16];
      if (beta != 0.0f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t j = j0 + wj * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          old[r] = C[ic + (j < N ? j : N - 1) * ldc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t j = j0 + wj * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (i >= M || j >= N) continue;
        const float v = alpha * ((ui * acc[a][b][r]) * uj[r]);
        C[i + j * ldc] = (beta == 0.0f) ? v : v + beta * old[r];
      }
    }
  }
}

// tail tiles: C tile = alpha * (sum of the S partial tiles, in order) / (s_i s_j) + beta * C
__global__ __launch_bounds__(256) void SyrkSplitTailFixupKernel(int64_t M, int64_t N, int64_t lin0, int S,
                                                                const float* __restrict__ P,
                                                                const unsigned* __restrict__ rowmax_i,
                                                                const unsigned* __restrict__ rowmax_j,
                                                                float alpha, float beta, float* C, int64_t ldc,
                                                                const int* __restrict__ order, int lower) {
  const int64_t lin = lin0 + blockIdx.x;
  int64_t I, Jt;
  if (order != nullptr) {  // the tail tiles keep their place in the order table
    I = order[lin] >> 16;
    Jt = order[lin] & 0xffff;
  } else if (lower) {
    I = static_cast<int64_t>((sqrt(8.0 * static_cast<double>(lin) + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= lin) ++I;
    while (I * (I + 1) / 2 > lin) --I;
    Jt = lin - I * (I + 1) / 2;
  } else {
    const int64_t TI = (M + TS - 1) / TS;
    I = lin % TI;
    Jt = lin / TI;
  }
  const int64_t i0 = I * TS, j0 = Jt * TS;
  const float* p0 = P + static_cast<int64_t>(blockIdx.x) * S * (TS * TS);
  // gridDim.y workgroups share a tile (one each left 50 workgroups on the chip with 256 dependent
  // iterations per thread: 236 us for the Gram's tail at config 2)
  const int per = TS * TS / static_cast<int>(gridDim.y);
  const int e_begin = static_cast<int>(blockIdx.y) * per;
  for (int e = e_begin + threadIdx.x; e < e_begin + per; e += 256) {
    const int64_t i = i0 + (e & (TS - 1)), j = j0 + (e >> 8);
    if (i >= M || j >= N) continue;
    float sum = p0[e];
    for (int c = 1; c < S; ++c) sum += p0[static_cast<int64_t>(c) * (TS * TS) + e];
    float* dst = C + i + j * ldc;
    const float v = alpha * ((SplitScaleInv(rowmax_i[i]) * sum) * SplitScaleInv(rowmax_j[j]));
    *dst = (beta == 0.0f) ? v : v + beta * (*dst);
  }
}

}  // namespace

namespace {

struct ConvertedOperand {
  std::shared_ptr<Buffer> h, l, mx;
  SplitOperand op;
};

// EPSILON_HIP_GEMM_STAGE = reg: the register-staged loop (V = 0); EPSILON_HIP_GEMM_ORDER = 0: tiles
// in plain linear order.  Read on every call (the microbenchmark switches them between launches).
int StageVariant() {
  const char* e = std::getenv("EPSILON_HIP_GEMM_STAGE");
  if (e == nullptr) return 3;
  if (std::strcmp(e, "reg") == 0) return 0;  // register-staged, padded image (round 2)
  if (std::strcmp(e, "lds") == 0) return 1;  // LDS-DMA, two 32-deep buffers, compiler-scheduled
  return 3;                                   // "ring": four 16-deep steps
}
bool PatchOrderEnabled() {
  const char* e = std::getenv("EPSILON_HIP_GEMM_ORDER");
  return !(e != nullptr && e[0] == '0');
}

// The tiles of a TI x TJ grid (lower: J <= I only) in patch-major order, as (I << 16 | J) words on
// the device.  Built once per shape and kept for the life of the process (a few KB each).
const int* TileOrder(int64_t TI, int64_t TJ, bool lower, int64_t* count) {
  struct Entry {
    std::shared_ptr<Buffer> buf;
    int64_t count;
  };
  static std::mutex mu;
  static auto* cache = new std::map<std::tuple<int64_t, int64_t, bool>, Entry>();  // never destroyed
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple(TI, TJ, lower);
  auto it = cache->find(key);
  if (it == cache->end()) {
    EPS_CHECK(TI < 65536 && TJ < 65536);
    std::vector<int> h;
    constexpr int64_t PS = 8;
    for (int64_t pr = 0; pr * PS < TI; ++pr)
      for (int64_t pc = 0; pc * PS < TJ; ++pc) {
        if (lower && pc > pr) continue;
        for (int64_t i = pr * PS; i < std::min(TI, (pr + 1) * PS); ++i)
          for (int64_t j = pc * PS; j < std::min(TJ, (pc + 1) * PS); ++j)
            if (!lower || j <= i) h.push_back(static_cast<int>((i << 16) | j));
      }
    Entry e;
    e.count = static_cast<int64_t>(h.size());
    e.buf = Runtime::Get().Alloc(h.size() * sizeof(int));
    // on the library's own stream (the legacy default stream may not exist yet in this process,
    // and creating it costs tens of milliseconds); once per shape, so the wait is affordable
    hipStream_t st = Runtime::Get().stream();
    EPS_HIP(hipMemcpyAsync(e.buf->p, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, st));
    EPS_HIP(hipStreamSynchronize(st));
    it = cache->emplace(key, std::move(e)).first;
  }
  *count = it->second.count;
  return static_cast<const int*>(it->second.buf->p);
}

void LaunchSplit(dim3 grid, int64_t M, int64_t N, int64_t nslab, const SplitOperand& oa, const SplitOperand& ob,
                 float al, float be, float* C, int64_t ldc, int tri, int64_t lin0, int64_t slab0, int64_t slab_count,
                 float* P, const int* order, int64_t perm_limit) {
  hipStream_t s = Runtime::Get().stream();
  const int v = StageVariant();
  if (v == 3)
    hipLaunchKernelGGL(GemmSplitF16Kernel<3>, grid, dim3(kThreads), 0, s, M, N, nslab, oa, ob, al, be, C, ldc, tri,
                       lin0, slab0, slab_count, P, order, perm_limit);
  else if (v == 1)
    hipLaunchKernelGGL(GemmSplitF16Kernel<1>, grid, dim3(kThreads), 0, s, M, N, nslab, oa, ob, al, be, C, ldc, tri,
                       lin0, slab0, slab_count, P, order, perm_limit);
  else
    hipLaunchKernelGGL(GemmSplitF16Kernel<0>, grid, dim3(kThreads), 0, s, M, N, nslab, oa, ob, al, be, C, ldc, tri,
                       lin0, slab0, slab_count, P, order, perm_limit);
}

// X holds `rows` rows of K entries: element (r, k) at X[r + k * ld] (contig_r) or X[k + r * ld].
ConvertedOperand ConvertOperand(const float* X, int64_t rows, int64_t K, int64_t ld, bool contig_r) {
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ConvertedOperand c;
  const int64_t r_pad = (rows + TS - 1) / TS * TS;
  const int64_t nslab = (K + SK - 1) / SK;
  c.h = rt.Alloc(static_cast<size_t>(nslab) * r_pad * SK * sizeof(_Float16));
  c.l = rt.Alloc(static_cast<size_t>(nslab) * r_pad * SK * sizeof(_Float16));
  c.mx = rt.Alloc(static_cast<size_t>(rows) * sizeof(unsigned));
  _Float16* H = static_cast<_Float16*>(c.h->p);
  _Float16* L = static_cast<_Float16*>(c.l->p);
  unsigned* amax = static_cast<unsigned*>(c.mx->p);
  if (contig_r) {
    EPS_HIP(hipMemsetAsync(amax, 0, static_cast<size_t>(rows) * sizeof(unsigned), s));
    const unsigned gx = static_cast<unsigned>((rows + 255) / 256);
    const unsigned gy = static_cast<unsigned>(
        std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(256, K), 2048 / gx)));
    hipLaunchKernelGGL(RowAbsMaxKernel, dim3(gx, gy), dim3(256), 0, s, X, rows, K, ld, amax);
    hipLaunchKernelGGL(SplitConvertKernel, dim3(static_cast<unsigned>(r_pad / 64), static_cast<unsigned>(nslab)),
                       dim3(256), 0, s, X, rows, K, ld, r_pad, amax, H, L);
  } else {
    hipLaunchKernelGGL(RowAbsMaxKKernel, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, s, X, rows, K,
                       ld, amax);
    hipLaunchKernelGGL(SplitConvertKKernel, dim3(static_cast<unsigned>(r_pad / 64), static_cast<unsigned>(nslab)),
                       dim3(256), 0, s, X, rows, K, ld, r_pad, amax, H, L);
  }
  c.op = SplitOperand{H, L, amax, r_pad};
  return c;
}

// EPSILON_HIP_GEMM = generic / mfma / mfma_simple forces another kernel family (read on every
// call: the tests switch it between products)
bool AutoKernelChoice() {
  const char* e = std::getenv("EPSILON_HIP_GEMM");
  return e == nullptr || e[0] == 0 || std::strcmp(e, "auto") == 0;
}

bool SplitEnabled() {
  static const bool off = [] {
    const char* e = std::getenv("EPSILON_HIP_GRAM_F16SPLIT");
    return e && e[0] == '0';
  }();
  return !off;
}

}  // namespace

// General product C = alpha op(A) op(B) + beta C on the split-f16 kernel (large f32 products:
// the GEMMs of the blocked Cholesky inverse).  false: not eligible, nothing done.
bool GemmSplitF16(bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha, const DVec& A,
                  int64_t lda, const DVec& B, int64_t ldb, double beta, const DVec& C, int64_t ldc,
                  bool lower_only) {
  if (!SplitEnabled() || A.dt != F32 || B.dt != F32 || C.dt != F32) return false;
  if (M < 1024 || N < 1024 || K < 256 || static_cast<double>(M) * N * K < 8.0e9) return false;
  if (lower_only && M != N) return false;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof(lower_only ? "syrk_f16split" : "gemm_f16split", M * N, K);
  const int64_t nslab = (K + SK - 1) / SK;
  // op(A)(i, k): A[i + k lda] (not transposed: contiguous along the rows) or A[k + i lda]
  ConvertedOperand ca = ConvertOperand(A.as<float>(), M, K, lda, !transA);
  const bool same = A.data() == B.data() && lda == ldb && transA != transB && M == N;
  ConvertedOperand cb;
  // op(B)^T(j, k) = op(B)(k, j): B[k + j ldb] (not transposed: contiguous along k) or B[j + k ldb]
  if (!same) cb = ConvertOperand(B.as<float>(), N, K, ldb, transB);
  const SplitOperand& oa = ca.op;
  const SplitOperand& ob = same ? ca.op : cb.op;
  const float al = static_cast<float>(alpha), be = static_cast<float>(beta);
  const int64_t TI = (M + TS - 1) / TS, TJ = (N + TS - 1) / TS;
  int64_t ocount = 0;
  const int* order = PatchOrderEnabled() ? TileOrder(TI, TJ, lower_only, &ocount) : nullptr;
  const int64_t total = lower_only ? TI * (TI + 1) / 2 : TI * TJ;
  const int tri = lower_only ? 1 : 0;
  EPS_CHECK(order == nullptr || ocount == total);
  // one 512-thread workgroup per CU: rounds of 256 tiles; a ragged last round of a LONG
  // contraction is split over K (raw partial tiles, summed in order by the fix-up kernel)
  const int64_t slots = 256;
  int64_t tail = total % slots;
  int S = 1;
  int64_t per = nslab;
  if (nslab >= 64 && total > slots && tail > 0 && tail <= slots / 2) {
    S = static_cast<int>(std::min<int64_t>(8, slots / tail));
    per = (nslab + S - 1) / S;
    S = static_cast<int>((nslab + per - 1) / per);
  } else {
    tail = 0;
  }
  const int64_t full = total - tail;
  if (full > 0)
    LaunchSplit(dim3(static_cast<unsigned>(full)), M, N, nslab, oa, ob, al, be, C.as<float>(), ldc, tri, 0, 0, nslab,
                nullptr, order, full / 256 * 256);
  if (tail > 0) {
    auto pbuf = rt.Alloc(static_cast<size_t>(tail) * S * TS * TS * sizeof(float));
    float* P = static_cast<float*>(pbuf->p);
    LaunchSplit(dim3(static_cast<unsigned>(tail), static_cast<unsigned>(S)), M, N, nslab, oa, ob, al, be,
                C.as<float>(), ldc, tri, full, 0, per, P, order, 0);
    hipLaunchKernelGGL(SyrkSplitTailFixupKernel, dim3(static_cast<unsigned>(tail), 16), dim3(256), 0, s, M, N, full,
                       S, P, oa.rowmax, ob.rowmax, al, be, C.as<float>(), ldc, order, tri);
  }
  EPS_HIP(hipGetLastError());
  return true;
}

// C (M x N) = alpha A B with ONE operand lower triangular (zeros stored): kmode 3: B (K x N, its
// rows k and columns j on the same index: B(k, j) = 0 for k < j); kmode 4: A (M x K, A(i, k) = 0
// for k > i).  Every tile runs over its own k range - half the flops of the dense product in one
// launch.  false: not eligible, nothing done.
bool GemmSplitF16KRange(int kmode, int64_t M, int64_t N, int64_t K, double alpha, const DVec& A, int64_t lda,
                        const DVec& B, int64_t ldb, const DVec& C, int64_t ldc) {
  if (!SplitEnabled() || !AutoKernelChoice() || A.dt != F32 || B.dt != F32 || C.dt != F32) return false;
  if (M < 1024 || N < 1024 || K < 256 || static_cast<double>(M) * N * K < 8.0e9) return false;
  if (kmode != 3 && kmode != 4) return false;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("gemm_f16split_krange", M * N, K);
  const int64_t nslab = (K + SK - 1) / SK;
  ConvertedOperand ca = ConvertOperand(A.as<float>(), M, K, lda, true);   // A(i, k) = A[i + k lda]
  ConvertedOperand cb = ConvertOperand(B.as<float>(), N, K, ldb, false);  // B(k, j) = B[k + j ldb]
  const int64_t TI = (M + TS - 1) / TS, TJ = (N + TS - 1) / TS;
  LaunchSplit(dim3(static_cast<unsigned>(TI * TJ)), M, N, nslab, ca.op, cb.op, static_cast<float>(alpha), 0.0f,
              C.as<float>(), ldc, kmode, 0, 0, nslab, nullptr, nullptr, 0);
  EPS_HIP(hipGetLastError());
  return true;
}

// C (lower tiles, n x n) = X^T X for a lower-triangular X (n x n, column-major, zeros above the
// diagonal stored): the last product of the explicit inverse, W^-1 = L^-T L^-1.  ONE launch over
// the lower tiles, each with its own k range (tri == 2 above) - a third of the dense flops, no
// accumulation passes over C.  Tiles are dispatched in order of decreasing k length.
bool SyrkSplitF16LowerTriangular(int64_t n, const DVec& X, int64_t ldx, const DVec& C, int64_t ldc) {
  if (!SplitEnabled() || !AutoKernelChoice() || X.dt != F32 || C.dt != F32 || n < 2048) return false;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("syrk_f16split_tri", n * n, n);
  const int64_t nslab = (n + SK - 1) / SK;
  ConvertedOperand cx = ConvertOperand(X.as<float>(), n, n, ldx, false);  // rows = columns of X
  const int64_t T = (n + TS - 1) / TS;
  LaunchSplit(dim3(static_cast<unsigned>(T * (T + 1) / 2)), n, n, nslab, cx.op, cx.op, 1.0f, 0.0f, C.as<float>(), ldc,
              2, 0, 0, nslab, nullptr, nullptr, 0);
  EPS_HIP(hipGetLastError());
  return true;
}

}  // namespace k
}  // namespace eps

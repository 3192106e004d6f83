// K9 (round 3): exact 1-D total-variation (fused-lasso) prox, parallel, FOUR-way level-set splits.
//
//   x = argmin 1/2 ||x - y||^2 + lam * sum_i |x[i+1] - x[i]|
//
// The reference calls glmgen's `tf_dp` (reference src/epsilon/prox/total_variation_1d.cc:8,21):
// Johnson's dynamic program, sequential.  kernels_tv.hip (round 2) computes the same unique
// minimiser by divide and conquer over LEVEL SETS with one threshold per region and level (depth
// ~ log2 of the number of constant pieces, every level three scans over all n).  This file keeps
// the mathematics and changes the economics:
//
//   * THREE thresholds per region and level.  For a region R whose neighbours are known to lie
//     strictly above / below it, the restricted problem is a plain TV problem on the chain R with
//     the neighbour terms folded into the end samples (y'), so {i : x*_i > t} is the minimal
//     minimiser of the binary chain problem with costs (t - y'_i) for ANY t - not only for the
//     region mean tau.  The forward clamp-shift scan d_i = a_i + clip(d_{i-1}, -lam, lam) runs for
//     t = tau - delta, tau, tau + delta at once (one read of y, three maps per sample); the three
//     decoded sets are nested, their sum is a label 0..3, runs of equal label are the new regions,
//     strictly ordered across every cut.  tau stays the region mean, so "the middle set is empty"
//     still means "the region is constant".  delta is a guess (half the distance from the new
//     mean to the nearest parent threshold that bounds the region): any value is valid, a good one
//     makes the split 4-way.  Depth on the reference's tv_1d generator: 8 levels where the binary
//     recursion takes 12 (n = 2e5), see tools_tv3_prototype.py (the CPU prototype of this file).
//   * Four passes over the data per level instead of six: the apply phase of the forward scan
//     also emits the tile aggregates of the backward decode scan, the decode's apply phase those
//     of the backward boundary scan (it knows the label right of every sample from its own scan).
//   * Tiles whose samples are all finished are skipped by every later kernel (their scan
//     aggregates are constants), and termination is decided on the device: every kernel of level
//     k returns at once when level k - 1 made no cut, so the host enqueues levels in batches and
//     synchronises once per batch.
//   * Up to kDirectTiles tiles no aggregate-scan launches at all: every workgroup folds the
//     aggregates of the tiles before it itself (n = 1e5: 4 launches per level).
//
// State: ONE byte per sample - bit 0 head of a region, bit 1 end, bits 2-3 / 4-5 / 6-7 the class
// of the sample for the three thresholds (0, 1, 2 = copy from the right; all three 3 = finished) -
// a 16-byte record per region head (tau, delta, sides of the neighbours, "did not split") and a
// 16-byte record per region end (the parent's tau / delta / right side, for the children's
// delta), plus the fp64 prefix sums of y that only region boundaries touch.  All decisions in
// fp64 (data may be f32), as in round 2.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;
// samples per thread.  8, not 16: the kernels are bound by the latency of a workgroup's serial
// chain (loads -> scan -> records -> compose -> scan -> classify -> store) times the workgroups a
// CU can hold, and the forward apply kernel needs 160 registers at 16 (3 waves per SIMD)
constexpr int kItems = 8;
constexpr int kWords = kItems / 4;          // state bytes of a chunk, as 32-bit words
constexpr int kTile = kBlock * kItems;      // samples per workgroup
constexpr unsigned kAllItems = (1u << kItems) - 1u;
constexpr int kDirectTiles = 256;           // up to here: no aggregate-scan launches
constexpr int32_t kInf = 0x7fffffff;
constexpr int kHead = 1, kEnd = 2, kDone = 0xFC;
constexpr int kMaxLevels = 4096;

// ---- scan algebras --------------------------------------------------------------------------------

struct ClipMap {
  double p, lo, hi;
};
__device__ inline ClipMap ClipCombine(const ClipMap& f, const ClipMap& g) {  // g after f
  ClipMap o;
  o.p = f.p + g.p;
  o.lo = fmin(fmax(f.lo + g.p, g.lo), g.hi);
  o.hi = fmin(fmax(f.hi + g.p, g.lo), g.hi);
  return o;
}
struct Clip3 {
  ClipMap m[3];
};
struct Clip3Alg {
  using S = Clip3;
  __device__ static S combine(const S& f, const S& g) {
    S o;
#pragma unroll
    for (int j = 0; j < 3; ++j) o.m[j] = ClipCombine(f.m[j], g.m[j]);
    return o;
  }
  __device__ static S identity() {
    S o;
#pragma unroll
    for (int j = 0; j < 3; ++j) o.m[j] = ClipMap{0.0, -INFINITY, INFINITY};
    return o;
  }
};
__device__ inline Clip3 Inert3() {  // finished samples: the constant-zero map
  Clip3 o;
#pragma unroll
  for (int j = 0; j < 3; ++j) o.m[j] = ClipMap{0.0, 0.0, 0.0};
  return o;
}
// three 2-bit classes at bits 0-5; first definite value in scan order wins
struct Dec3Alg {
  using S = unsigned;
  __device__ static S combine(S acc, S next) {
    const unsigned und = (next >> 1) & ~next & 0x15u;  // fields of `next` equal to 2
    const unsigned m = und | (und << 1);
    return (acc & m) | (next & ~m);
  }
  __device__ static S identity() { return 0x2Au; }
};
struct Int2 {
  int32_t a, b;
};
struct MinAlg {
  using S = Int2;
  __device__ static S combine(S x, S y) { return S{x.a < y.a ? x.a : y.a, x.b < y.b ? x.b : y.b}; }
  __device__ static S identity() { return S{kInf, kInf}; }
};
struct MaxAlg {
  using S = int32_t;
  __device__ static S combine(S x, S y) { return x > y ? x : y; }
  __device__ static S identity() { return -1; }
};
struct SumAlg {
  using S = double;
  __device__ static S combine(S x, S y) { return x + y; }
  __device__ static S identity() { return 0.0; }
};

template <class S> __device__ inline S ShflUp(const S& v, int off) {
  constexpr int W = (sizeof(S) + 3) / 4;
  int w[W] = {};
  memcpy(w, &v, sizeof(S));
#pragma unroll
  for (int k = 0; k < W; ++k) w[k] = __shfl_up(w[k], off, 64);
  S o;
  memcpy(&o, w, sizeof(S));
  return o;
}

// One step of a wave scan on the DPP path of the vector ALU: every 32-bit word of the value moves
// by the same lane pattern; lanes without a source (row_shr at the start of a row, rows outside
// ROW_MASK) receive `none` - the algebra's identity - so the caller combines unconditionally (a
// select per word and step was a quarter of the forward kernel's instructions).
template <int CTRL, int ROW_MASK, class S> __device__ inline S DppMove(const S& v, const S& none) {
  constexpr int W = (sizeof(S) + 3) / 4;
  int w[W] = {}, o[W] = {};
  memcpy(w, &v, sizeof(S));
  memcpy(o, &none, sizeof(S));
#pragma unroll
  for (int k = 0; k < W; ++k) w[k] = __builtin_amdgcn_update_dpp(o[k], w[k], CTRL, ROW_MASK, 0xF, false);
  S r;
  memcpy(&r, w, sizeof(S));
  return r;
}

// Inclusive scan over the 64 lanes of a wave, in lane order (Alg::combine(earlier, later)):
// Hillis-Steele inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of rows 0 / 2 into
// rows 1 / 3 (row_bcast15) and lane 31 into rows 2 and 3 (row_bcast31).
template <class Alg> __device__ inline typename Alg::S WaveInclusive(typename Alg::S v, int lane) {
  using S = typename Alg::S;
  (void)lane;
  const S id = Alg::identity();
  v = Alg::combine(DppMove<0x111, 0xF>(v, id), v);
  v = Alg::combine(DppMove<0x112, 0xF>(v, id), v);
  v = Alg::combine(DppMove<0x114, 0xF>(v, id), v);
  v = Alg::combine(DppMove<0x118, 0xF>(v, id), v);
  v = Alg::combine(DppMove<0x142, 0xA>(v, id), v);
  v = Alg::combine(DppMove<0x143, 0xC>(v, id), v);
  return v;
}

// Exclusive scan of one value per thread over the workgroup, in thread order.  *total = the
// workgroup aggregate (the same in every thread; nullptr: not wanted).  `lds`: kBlock / 64 values.
template <class Alg>
__device__ inline typename Alg::S BlockExclusive(typename Alg::S mine, typename Alg::S* lds,
                                                 typename Alg::S* total) {
  using S = typename Alg::S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const S incl = WaveInclusive<Alg>(mine, lane);
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  // what the waves before this one hold: wave-uniform, so the loop costs scalar control only
  S before = Alg::identity();
  for (int w = 0; w < wave; ++w) before = Alg::combine(before, lds[w]);
  if (total != nullptr) {
    S tot = lds[0];
    for (int w = 1; w < kBlock / 64; ++w) tot = Alg::combine(tot, lds[w]);
    *total = tot;
  }
  S excl = ShflUp(incl, 1);
  if (lane == 0) excl = Alg::identity();
  excl = Alg::combine(before, excl);
  __syncthreads();
  return excl;
}

// What the tiles before scan position `pos` contribute.  Scanned mode: the aggregate array was
// turned into exclusive prefixes by RunAggScan.  Direct mode (few tiles): every workgroup folds
// agg[0 .. pos) itself, in order - a contiguous chunk per thread, then the threads in order.
// In two halves, so that the loads are in flight before the kernel's own data arrives:
// TilePrefixLoad (no synchronisation) and TilePrefixFinish (a workgroup scan in direct mode).
template <class Alg>
__device__ inline typename Alg::S TilePrefixLoad(const typename Alg::S* agg, int64_t pos, bool direct) {
  using S = typename Alg::S;
  if (!direct) return agg[pos];
  const int64_t per = (pos + kBlock - 1) / kBlock;
  const int64_t b0 = static_cast<int64_t>(threadIdx.x) * per;
  int64_t b1 = b0 + per;
  if (b1 > pos) b1 = pos;
  S acc = Alg::identity();
  for (int64_t i = b0; i < b1; ++i) acc = Alg::combine(acc, agg[i]);
  return acc;
}
template <class Alg>
__device__ inline typename Alg::S TilePrefixFinish(const typename Alg::S& part, bool direct, typename Alg::S* lds) {
  using S = typename Alg::S;
  if (!direct) return part;
  S total;
  BlockExclusive<Alg>(part, lds, &total);
  return total;
}
template <class Alg>
__device__ inline typename Alg::S TilePrefix(const typename Alg::S* agg, int64_t pos, bool direct,
                                             typename Alg::S* lds) {
  return TilePrefixFinish<Alg>(TilePrefixLoad<Alg>(agg, pos, direct), direct, lds);
}

// ---- exclusive scan of an aggregate array, in place (many tiles) ---------------------------------

template <class Alg> struct AggGeom {
  static constexpr int kAggItems = sizeof(typename Alg::S) > 32 ? 2 : 8;
  static constexpr int kAggTile = kBlock * kAggItems;
};

template <class Alg>
__global__ __launch_bounds__(kBlock) void AggScanKernel(int64_t nb, typename Alg::S* agg, const unsigned long long* gate) {
  using S = typename Alg::S;
  if (gate != nullptr && *gate == 0ull) return;
  __shared__ S lds[kBlock / 64];
  const int64_t chunk = (nb + kBlock - 1) / kBlock;
  const int64_t b0 = static_cast<int64_t>(threadIdx.x) * chunk;
  int64_t b1 = b0 + chunk;
  if (b1 > nb) b1 = nb;
  S acc = Alg::identity();
  for (int64_t i = b0; i < b1; ++i) acc = Alg::combine(acc, agg[i]);
  S total;
  S run = BlockExclusive<Alg>(acc, lds, &total);
  for (int64_t i = b0; i < b1; ++i) {
    const S v = agg[i];
    agg[i] = run;
    run = Alg::combine(run, v);
  }
}

template <class Alg>
__global__ __launch_bounds__(kBlock) void AggReduceKernel(int64_t nb, const typename Alg::S* agg,
                                                          typename Alg::S* agg2, const unsigned long long* gate) {
  using S = typename Alg::S;
  if (gate != nullptr && *gate == 0ull) return;
  constexpr int kAggItems = AggGeom<Alg>::kAggItems, kAggTile = AggGeom<Alg>::kAggTile;
  __shared__ S lds[kBlock / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kAggTile + static_cast<int64_t>(threadIdx.x) * kAggItems;
  S acc = Alg::identity();
#pragma unroll
  for (int k = 0; k < kAggItems; ++k)
    if (base + k < nb) acc = Alg::combine(acc, agg[base + k]);
  S total;
  BlockExclusive<Alg>(acc, lds, &total);
  if (threadIdx.x == 0) agg2[blockIdx.x] = total;
}

// Second half of the two-level scan: exclusive prefixes of agg[] written to out[] (out == agg: in
// place).  The prefix of a workgroup's chunk is folded from the first-level totals agg2[0 ..
// blockIdx) by the workgroup itself (a few dozen values), so no launch scans agg2.
template <class Alg>
__global__ __launch_bounds__(kBlock) void AggApplyKernel(int64_t nb, const typename Alg::S* agg,
                                                         const typename Alg::S* agg2, typename Alg::S* out,
                                                         const unsigned long long* gate) {
  using S = typename Alg::S;
  if (gate != nullptr && *gate == 0ull) return;
  constexpr int kAggItems = AggGeom<Alg>::kAggItems, kAggTile = AggGeom<Alg>::kAggTile;
  __shared__ S lds[kBlock / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kAggTile + static_cast<int64_t>(threadIdx.x) * kAggItems;
  const S pre_part = TilePrefixLoad<Alg>(agg2, blockIdx.x, true);
  S item[kAggItems];
  S acc = Alg::identity();
#pragma unroll
  for (int k = 0; k < kAggItems; ++k) {
    item[k] = base + k < nb ? agg[base + k] : Alg::identity();
    acc = Alg::combine(acc, item[k]);
  }
  S total;
  S excl = BlockExclusive<Alg>(acc, lds, &total);
  const S pre = TilePrefixFinish<Alg>(pre_part, true, lds);
  S run = Alg::combine(pre, excl);
#pragma unroll
  for (int k = 0; k < kAggItems; ++k) {
    if (base + k < nb) {
      out[base + k] = run;
      run = Alg::combine(run, item[k]);
    }
  }
}

// `gate` (optional): a device counter; every kernel of the scan returns at once when it is zero
// (the recursion made no cut a level ago: nothing to scan).  out: where the exclusive prefixes go
// (nullptr: in place).
template <class Alg>
void RunAggScan(int64_t nb, typename Alg::S* agg, const unsigned long long* gate = nullptr,
                typename Alg::S* out = nullptr) {
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  constexpr int kAggTile = AggGeom<Alg>::kAggTile;
  if (out == nullptr) out = agg;
  if (nb > 2 * kAggTile) {
    const int64_t nb2 = (nb + kAggTile - 1) / kAggTile;
    auto buf = rt.Alloc(static_cast<size_t>(nb2) * sizeof(typename Alg::S));
    auto* agg2 = static_cast<typename Alg::S*>(buf->p);
    hipLaunchKernelGGL((AggReduceKernel<Alg>), dim3(static_cast<unsigned>(nb2)), dim3(kBlock), 0, s, nb, agg, agg2, gate);
    hipLaunchKernelGGL((AggApplyKernel<Alg>), dim3(static_cast<unsigned>(nb2)), dim3(kBlock), 0, s, nb, agg, agg2, out, gate);
  } else {
    if (out != agg)
      EPS_HIP(hipMemcpyAsync(out, agg, static_cast<size_t>(nb) * sizeof(typename Alg::S), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL((AggScanKernel<Alg>), dim3(1), dim3(kBlock), 0, s, nb, out, gate);
  }
}

// ---- prefix sums of y (fp64): Pp[i + 1] = sum_{k <= i} y_k, Pp[0] = 0 -----------------------------
// (Pp points one double behind a 16-byte aligned buffer, so the results of a chunk are aligned)

template <class T> __device__ inline void LoadSamples(const T* y, int64_t n, int64_t c0, bool aligned, double (&v)[kItems]) {
  if (aligned && c0 + kItems <= n) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int q = 0; q < kItems / 4; ++q) {
        const float4 a = reinterpret_cast<const float4*>(y + c0)[q];
        v[4 * q] = a.x;
        v[4 * q + 1] = a.y;
        v[4 * q + 2] = a.z;
        v[4 * q + 3] = a.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < kItems / 2; ++q) {
        const double2 a = reinterpret_cast<const double2*>(y + c0)[q];
        v[2 * q] = a.x;
        v[2 * q + 1] = a.y;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k) v[k] = c0 + k < n ? static_cast<double>(y[c0 + k]) : 0.0;
  }
}

template <class T>
__global__ __launch_bounds__(kBlock) void PrefixReduceKernel(const T* __restrict__ y, int64_t n, int aligned, double* agg) {
  __shared__ double lds[kBlock / 64];
  const int64_t c0 = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * kItems;
  double v[kItems];
  LoadSamples(y, n, c0, aligned != 0, v);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) acc += v[k];
  double total;
  BlockExclusive<SumAlg>(acc, lds, &total);
  if (threadIdx.x == 0) agg[blockIdx.x] = total;
}

// (Pp + 1 is 16-byte aligned and c0 is a multiple of 8: the chunk's results go out as 16-byte pairs)
template <class T>
__global__ __launch_bounds__(kBlock) void PrefixApplyKernel(const T* __restrict__ y, int64_t n, int aligned, const double* agg,
                                                            int direct, double* __restrict__ Pp) {
  __shared__ double lds[kBlock / 64];
  const int64_t c0 = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * kItems;
  double v[kItems];
  LoadSamples(y, n, c0, aligned != 0, v);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) acc += v[k];
  double total;
  const double excl = BlockExclusive<SumAlg>(acc, lds, &total);
  double run = TilePrefix<SumAlg>(agg, blockIdx.x, direct != 0, lds) + excl;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    run += v[k];
    v[k] = run;
  }
  if (c0 + kItems <= n) {
#pragma unroll
    for (int q = 0; q < kItems / 2; ++q) reinterpret_cast<double2*>(Pp + c0 + 1)[q] = make_double2(v[2 * q], v[2 * q + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k)
      if (c0 + k < n) Pp[c0 + k + 1] = v[k];
  }
}

// ---- records ---------------------------------------------------------------------------------------

struct __attribute__((aligned(16))) HeadRec {  // by region head
  double tau;
  float delta;
  unsigned flags;  // bits 0-1 side of the left neighbour, 2-3 of the right one, bit 4 "did not split"
};
struct __attribute__((aligned(16))) EndRec {   // by region end: what the region's children need of it
  double tau;
  float delta;
  unsigned cr;
};

__device__ inline int SideSign(int code) { return code == 1 ? 1 : (code == 2 ? -1 : 0); }
__device__ inline int Label(int b) { return ((b >> 2) & 1) + ((b >> 4) & 1) + ((b >> 6) & 1); }

template <class T> struct TvS {
  const T* y;
  T* x;
  const double* Pp;
  uint8_t* st;    // this level's state
  uint8_t* st2;   // next level's (written by the boundary pass)
  HeadRec* hrec;
  EndRec* erec;
  int32_t* tile_head;        // per tile: last head of an ACTIVE region inside it (-1 none)
  const int32_t* tile_l_in;  // scanned mode: exclusive running maximum of tile_head
  int32_t* tdone;            // per tile: every sample finished (both state buffers agree on it)
  int32_t* tend;             // per finished tile: first region end inside it (kInf none)
  Clip3* agg_clip;           // by tile
  unsigned* dagg;            // by scan position of the backward scans (nb - 1 - tile)
  Int2* bagg;                // by scan position
  unsigned long long* cuts;  // per level
  double lam;
  int64_t n;
  int64_t nb;
  int level;
  bool aligned;  // y and x are 16-byte aligned
  bool direct;   // nb <= kDirectTiles
};

template <class T> __device__ inline bool LevelIsDead(const TvS<T>& s) {
  return s.level > 0 && s.cuts[s.level - 1] == 0ull;
}

template <class T> __device__ inline void LoadY(const TvS<T>& s, int64_t c0, T (&yv)[kItems]) {
  if (s.aligned && c0 + kItems <= s.n) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int q = 0; q < kItems / 4; ++q) {
        const float4 a = reinterpret_cast<const float4*>(s.y + c0)[q];
        yv[4 * q] = a.x;
        yv[4 * q + 1] = a.y;
        yv[4 * q + 2] = a.z;
        yv[4 * q + 3] = a.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < kItems / 2; ++q) {
        const double2 a = reinterpret_cast<const double2*>(s.y + c0)[q];
        yv[2 * q] = a.x;
        yv[2 * q + 1] = a.y;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k) yv[k] = c0 + k < s.n ? s.y[c0 + k] : T(0);
  }
}

__device__ inline int ByteOf(const unsigned (&w)[kWords], int k) { return static_cast<int>((w[k >> 2] >> (8 * (k & 3))) & 0xffu); }
__device__ inline void LoadBytes(const uint8_t* p, unsigned (&w)[kWords]) {
  if constexpr (kWords == 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  } else {
    static_assert(kWords == 2 || kWords == 4, "8 or 16 samples per thread");
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    w[0] = v.x; w[1] = v.y;
  }
}
__device__ inline void StoreBytes(uint8_t* p, const unsigned (&w)[kWords]) {
  if constexpr (kWords == 4) *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  else *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
}

// ---- pass 1 / 2: forward clamp-shift scan for the three thresholds --------------------------------
// MODE 0: reduce (tile aggregates).  MODE 1: apply (classes into the state byte, x of regions that
// finished a level ago, parent records at region ends, and the decode scan's tile aggregate).
//
// The scan is bound by fp64 issue, not by bytes (three maps per sample): a thread composes the
// maps of its 16 samples incrementally - appending a sample with value a to (p, lo, hi) is
// p += a, lo = a + clip(lo), hi = a + clip(hi), 7 operations - and the apply phase, once the
// value entering the chunk is known, follows the SCALAR recursion d = a + clip(d) (3 operations)
// instead of composing maps again.
template <class T, int MODE>
__global__ __launch_bounds__(kBlock) void TvClipKernel(TvS<T> s) {
  __shared__ Clip3 lds3[kBlock / 64];
  __shared__ int32_t ldsi[kBlock / 64];
  __shared__ unsigned ldsu[kBlock / 64];
  __shared__ HeadRec ldsr[kBlock];
  __shared__ int ldsp[kBlock / 64];
  if (LevelIsDead(s)) return;
  const int64_t tile = blockIdx.x;
  const int t = threadIdx.x;
  if (s.tdone[tile]) {
    if (t == 0) {
      if (MODE == 0) s.agg_clip[tile] = Inert3();
      if (MODE == 1) s.dagg[s.nb - 1 - tile] = 0u;  // finished samples decode as definite zeros
    }
    return;
  }
  // everything that does not depend on the data is requested first: the tile's bytes and samples,
  // and this thread's share of the other tiles' aggregates (direct mode) or the scanned prefixes
  const int64_t c0 = (tile * kBlock + t) * kItems;
  unsigned w[kWords];
  LoadBytes(s.st + c0, w);
  T yv[kItems];
  LoadY(s, c0, yv);
  const int32_t lin_part = TilePrefixLoad<MaxAlg>(s.direct ? s.tile_head : s.tile_l_in, tile, s.direct);
  // The record of the region that reaches into the tile from the left: in scanned mode its head
  // is known before any data arrives, so the record is requested now (not behind the head scan)
  HeadRec rec_in = HeadRec{0.0, 0.f, 0u};
  if (!s.direct && lin_part >= 0) rec_in = s.hrec[lin_part];
  // region head of every sample = running maximum of head positions.  A thread whose chunk holds
  // a head fetches that region's record itself and leaves it in LDS for the threads after it, so
  // no thread gathers a record BEHIND the scan (a second dependent round trip per workgroup).
  int32_t last = -1;
#pragma unroll
  for (int k = 0; k < kItems; ++k)
    if ((ByteOf(w, k) & kHead) && c0 + k < s.n) last = static_cast<int32_t>(c0 + k);
  HeadRec rec_last = HeadRec{0.0, 0.f, 0u};
  if (last >= 0) rec_last = s.hrec[last];
  ldsr[t] = rec_last;
  // a state byte is zero for a sample that takes part and is neither head nor end of its region
  bool special = false;
#pragma unroll
  for (int q = 0; q < kWords; ++q) special = special || w[q] != 0u;
  {
    const bool wave_special = __any(special ? 1 : 0) != 0;
    if ((t & 63) == 0) ldsp[t >> 6] = wave_special ? 1 : 0;
  }
  int32_t tot_head;
  int32_t l = BlockExclusive<MaxAlg>(last, ldsi, &tot_head);  // (its barriers also publish ldsp)
  const int32_t lin = TilePrefixFinish<MaxAlg>(lin_part, s.direct, ldsi);
  if (s.direct && lin >= 0) rec_in = s.hrec[lin];
  // ---- the plain tile: every sample belongs to the one region that reaches in from the left and
  // takes part (the first levels of a long signal are all of this kind).  No per-sample state to
  // decode, no region switches, no x to write: a = tau - y, three recursions.
  bool plain = lin >= 0 && !(rec_in.flags & 16u) && (tile + 1) * kTile <= s.n;
#pragma unroll
  for (int wv = 0; wv < kBlock / 64; ++wv) plain = plain && ldsp[wv] == 0;
  if (plain) {
    const double lam = s.lam, tau = rec_in.tau, dd = static_cast<double>(rec_in.delta);
    ClipMap acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = ClipMap{0.0, -INFINITY, INFINITY};
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const double a0 = tau - static_cast<double>(yv[k]);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double aj = j == 0 ? a0 - dd : (j == 2 ? a0 + dd : a0);
        acc[j].p += aj;
        acc[j].lo = aj + fmin(fmax(acc[j].lo, -lam), lam);
        acc[j].hi = aj + fmin(fmax(acc[j].hi, -lam), lam);
      }
    }
    Clip3 acc3;
#pragma unroll
    for (int j = 0; j < 3; ++j) acc3.m[j] = acc[j];
    Clip3 pre_part = Clip3Alg::identity();
    if (MODE == 1) pre_part = TilePrefixLoad<Clip3Alg>(s.agg_clip, tile, s.direct);
    Clip3 total;
    const Clip3 excl = BlockExclusive<Clip3Alg>(acc3, lds3, MODE == 0 ? &total : nullptr);
    if (MODE == 0) {
      if (t == 0) s.agg_clip[tile] = total;
      return;
    }
    const Clip3 pre = TilePrefixFinish<Clip3Alg>(pre_part, s.direct, lds3);
    const Clip3 run = Clip3Alg::combine(pre, excl);
    double d[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) d[j] = fmin(fmax(run.m[j].p, run.m[j].lo), run.m[j].hi);
    unsigned o[kWords] = {};
    unsigned dec = Dec3Alg::identity();
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const double a0 = tau - static_cast<double>(yv[k]);
      unsigned f = 0u;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double aj = j == 0 ? a0 - dd : (j == 2 ? a0 + dd : a0);
        const double dj = aj + fmin(fmax(d[j], -lam), lam);
        d[j] = dj;
        const unsigned below = dj < -lam ? 1u : 0u, above = dj >= lam ? 1u : 0u;
        f |= (below | ((1u - below) & (1u - above)) << 1) << (2 * j);
      }
      dec = Dec3Alg::combine(f, dec);
      o[k >> 2] |= (f << 2) << (8 * (k & 3));
    }
    StoreBytes(s.st + c0, o);
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned up = __shfl_down(dec, off, 64);
      if ((lane & (2 * off - 1)) == 0) dec = Dec3Alg::combine(up, dec);
    }
    if (lane == 0) ldsu[wave] = dec;
    __syncthreads();
    if (t == 0) {
      unsigned r = Dec3Alg::identity();
      for (int wv = kBlock / 64 - 1; wv >= 0; --wv) r = Dec3Alg::combine(r, ldsu[wv]);
      s.dagg[s.nb - 1 - tile] = r;
    }
    return;
  }
  HeadRec rec_cur = rec_in;
  if (l > lin) rec_cur = ldsr[(l - static_cast<int32_t>(tile * kTile)) / kItems];
  l = l > lin ? l : lin;

  // Straight-line arithmetic from here on (the kernel is bound by instruction issue: ~300 VALU
  // instructions per sample in its first form).  A sample is described by three numbers:
  //   av  = tau - y'   (0 for samples that do not take part: finished regions, padding)
  //   dv  = delta      (0 for those)
  //   le  = lam, or 0 at a region head and for samples that do not take part:
  //         appending (av, le) to a map (p, lo, hi) is p += a, lo = a + clip(lo, -le, le), hi
  //         likewise; with le = 0 that is the constant map a - exactly the head's reset and the
  //         finished samples' zero map (p is irrelevant once lo = hi).
  const int nvalid = s.n - c0 >= kItems ? kItems : (s.n - c0 > 0 ? static_cast<int>(s.n - c0) : 0);
  const double lam = s.lam;
  // The region a sample belongs to changes at heads; both loops below walk the chunk with the
  // same few lines instead of keeping per-sample arrays in registers (the apply kernel held 197).
  struct Region {
    double tau, dd, sl, sr;
    bool fin;
    unsigned crf;
  };
  auto region_of = [&](const HeadRec& r) {
    Region g;
    g.tau = r.tau;
    g.dd = static_cast<double>(r.delta);
    g.sl = lam * static_cast<double>(SideSign(r.flags & 3));
    g.sr = lam * static_cast<double>(SideSign((r.flags >> 2) & 3));
    g.fin = (r.flags & 16u) != 0;
    g.crf = (r.flags >> 2) & 3u;
    return g;
  };
  // sample k: av = tau - y' and dk = delta (0 where the sample does not take part), lek = lam (0 at
  // a head and where it does not take part), act / fnow = takes part / its region finished a level ago
  auto sample = [&](int k, Region& g, double* av, double* dk, double* lek, bool* act, bool* fnow) {
    const int b = ByteOf(w, k);
    const bool valid = k < nvalid;
    if (valid && (b & kHead))  // (several heads in one chunk: only the last one's record was fetched ahead)
      g = region_of(static_cast<int32_t>(c0 + k) == last ? rec_last : s.hrec[c0 + k]);
    const bool live = valid && (b & kDone) != kDone;  // not finished before this level
    *act = live && !g.fin;
    *fnow = live && g.fin;
    double yp = static_cast<double>(yv[k]);
    yp -= (b & kHead) ? g.sl : 0.0;
    yp -= (b & kEnd) ? g.sr : 0.0;
    *av = *act ? g.tau - yp : 0.0;
    *dk = *act ? g.dd : 0.0;
    *lek = (*act && !(b & kHead)) ? lam : 0.0;
  };
  ClipMap acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) acc[j] = ClipMap{0.0, -INFINITY, INFINITY};
  {
    Region g = region_of(rec_cur);
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      // (finished samples and the padding behind the last sample act as the zero map)
      double a0, dk, lek;
      bool act, fnow;
      sample(k, g, &a0, &dk, &lek, &act, &fnow);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double aj = j == 0 ? a0 - dk : (j == 2 ? a0 + dk : a0);
        acc[j].p += aj;
        acc[j].lo = aj + fmin(fmax(acc[j].lo, -lek), lek);
        acc[j].hi = aj + fmin(fmax(acc[j].hi, -lek), lek);
      }
    }
  }
  Clip3 acc3;
#pragma unroll
  for (int j = 0; j < 3; ++j) acc3.m[j] = acc[j];
  // (requested here, not at the top: 18 registers that would be live across the loop above)
  Clip3 pre_part = Clip3Alg::identity();
  if (MODE == 1) pre_part = TilePrefixLoad<Clip3Alg>(s.agg_clip, tile, s.direct);
  Clip3 total;
  const Clip3 excl = BlockExclusive<Clip3Alg>(acc3, lds3, MODE == 0 ? &total : nullptr);
  if (MODE == 0) {
    if (t == 0) s.agg_clip[tile] = total;
    return;
  }
  const Clip3 pre = TilePrefixFinish<Clip3Alg>(pre_part, s.direct, lds3);
  const Clip3 run = Clip3Alg::combine(pre, excl);
  double d[3];  // the value entering this chunk, per threshold
#pragma unroll
  for (int j = 0; j < 3; ++j) d[j] = fmin(fmax(run.m[j].p, run.m[j].lo), run.m[j].hi);
  unsigned o[kWords] = {};
  unsigned dec = Dec3Alg::identity();  // decode aggregate of the chunk: its LEFTMOST definite classes
  unsigned finm = 0;
  T xv[kItems];  // x of the samples whose region finished a level ago
  {
    Region g = region_of(rec_cur);
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const unsigned b = static_cast<unsigned>(ByteOf(w, k));
      double a0, dk, lek;
      bool act, fnow;
      sample(k, g, &a0, &dk, &lek, &act, &fnow);
      if (act && (b & kEnd)) s.erec[c0 + k] = EndRec{g.tau, static_cast<float>(g.dd), g.crf};
      xv[k] = static_cast<T>(g.tau);
      finm |= (fnow ? 1u : 0u) << k;
      unsigned f = 0u;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double aj = j == 0 ? a0 - dk : (j == 2 ? a0 + dk : a0);
        const double dj = aj + fmin(fmax(d[j], -lek), lek);
        d[j] = dj;
        // region end: definite (d < 0); elsewhere 1 below -lam, 0 from lam on, 2 (copy) in between
        const unsigned below = dj < -lam ? 1u : 0u, above = dj >= lam ? 1u : 0u, neg = dj < 0.0 ? 1u : 0u;
        const unsigned cls = (b & kEnd) ? neg : (below | ((1u - below) & (1u - above)) << 1);
        f |= cls << (2 * j);
      }
      const unsigned nb8 = act ? ((b & 3u) | (f << 2)) : (fnow ? ((b & 3u) | static_cast<unsigned>(kDone)) : b);
      const unsigned fd = act ? f : (k < nvalid ? 0u : 0x2Au);  // finished samples decode as zeros
      dec = Dec3Alg::combine(fd, dec);  // keeps what is already definite: the leftmost wins
      o[k >> 2] |= nb8 << (8 * (k & 3));
    }
  }
  StoreBytes(s.st + c0, o);
  if (finm == kAllItems && s.aligned) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int q = 0; q < kItems / 4; ++q)
        reinterpret_cast<float4*>(s.x + c0)[q] = make_float4(xv[4 * q], xv[4 * q + 1], xv[4 * q + 2], xv[4 * q + 3]);
    } else {
#pragma unroll
      for (int q = 0; q < kItems / 2; ++q) reinterpret_cast<double2*>(s.x + c0)[q] = make_double2(xv[2 * q], xv[2 * q + 1]);
    }
  } else if (finm) {
#pragma unroll
    for (int k = 0; k < kItems; ++k)
      if (finm & (1u << k)) s.x[c0 + k] = xv[k];
  }
  // tile aggregate of the backward decode scan: the fold from the tile's right end to its left
  // end keeps, per threshold, the leftmost definite class
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_down(dec, off, 64);
    if ((lane & (2 * off - 1)) == 0) dec = Dec3Alg::combine(up, dec);
  }
  if (lane == 0) ldsu[wave] = dec;
  __syncthreads();
  if (t == 0) {
    unsigned r = Dec3Alg::identity();
    for (int wv = kBlock / 64 - 1; wv >= 0; --wv) r = Dec3Alg::combine(r, ldsu[wv]);
    s.dagg[s.nb - 1 - tile] = r;
  }
}

// ---- pass 3: backward decode scan (apply), emits the boundary scan's tile aggregates --------------
template <class T>
__global__ __launch_bounds__(kBlock) void TvDecodeKernel(TvS<T> s) {
  __shared__ unsigned ldsu[kBlock / 64];
  __shared__ Int2 lds2[kBlock / 64];
  __shared__ int ldsc[kBlock / 64];
  if (LevelIsDead(s)) return;
  const int64_t pos = blockIdx.x, tile = s.nb - 1 - pos;
  const int t = threadIdx.x, chunk = kBlock - 1 - t;
  if (s.tdone[tile]) {
    if (t == 0) s.bagg[pos] = Int2{s.tend[tile], s.tend[tile]};
    return;
  }
  if (t == 0) s.tile_head[tile] = -1;  // the boundary pass of this level rebuilds it
  const int64_t c0 = (tile * kBlock + chunk) * kItems;
  unsigned w[kWords];
  LoadBytes(s.st + c0, w);
  const unsigned pre_part = TilePrefixLoad<Dec3Alg>(s.dagg, pos, s.direct);
  // ---- the plain tile: no region boundary and no finished sample in it (a state byte then holds
  // its three classes and nothing else): no flags to test, no region ends, cuts only
  bool plain_t = (tile + 1) * kTile <= s.n;
#pragma unroll
  for (int q = 0; q < kWords; ++q) {
    const unsigned x = (w[q] & 0xFCFCFCFCu) ^ 0xFCFCFCFCu;  // a zero byte: a finished sample
    plain_t = plain_t && (w[q] & 0x03030303u) == 0u && ((x - 0x01010101u) & ~x & 0x80808080u) == 0u;
  }
  {
    const bool wave_plain = __all(plain_t ? 1 : 0) != 0;
    if ((t & 63) == 0) ldsc[t >> 6] = wave_plain ? 1 : 0;
  }
  unsigned acc = Dec3Alg::identity();
#pragma unroll
  for (int k = kItems - 1; k >= 0; --k) {
    const int b = ByteOf(w, k);
    const unsigned f = c0 + k >= s.n ? 0x2Au : ((b & kDone) == kDone ? 0u : static_cast<unsigned>(b >> 2));
    acc = Dec3Alg::combine(acc, f);
  }
  unsigned tot;
  const unsigned excl = BlockExclusive<Dec3Alg>(acc, ldsu, &tot);  // (its barriers also publish ldsc)
  const unsigned pre = TilePrefixFinish<Dec3Alg>(pre_part, s.direct, ldsu);
  unsigned run = Dec3Alg::combine(pre, excl);
  bool plain = true;
#pragma unroll
  for (int wv = 0; wv < kBlock / 64; ++wv) plain = plain && ldsc[wv] != 0;
  if (plain) {
    auto dec01 = [](unsigned r) {
      const unsigned und = (r >> 1) & ~r & 0x15u;
      return r & ~(und | (und << 1)) & 0x15u;
    };
    int lab_next = __popc(dec01(run));
    unsigned o[kWords] = {};
    int32_t fne = kInf;
    int ncut = 0;
#pragma unroll
    for (int k = kItems - 1; k >= 0; --k) {
      run = Dec3Alg::combine(run, static_cast<unsigned>(ByteOf(w, k) >> 2));
      const unsigned u = dec01(run);
      const int lab = __popc(u);
      o[k >> 2] |= (u << 2) << (8 * (k & 3));
      if (lab != lab_next) {
        fne = static_cast<int32_t>(c0 + k);
        ++ncut;
      }
      lab_next = lab;
    }
    StoreBytes(s.st + c0, o);
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      fne = min(fne, __shfl_down(fne, off, 64));
      ncut += __shfl_down(ncut, off, 64);
    }
    __syncthreads();  // (every thread has read ldsc)
    if (lane == 0) {
      lds2[wave] = Int2{fne, kInf};
      ldsc[wave] = ncut;
    }
    __syncthreads();
    if (t == 0) {
      Int2 r = lds2[0];
      int c = ldsc[0];
      for (int wv = 1; wv < kBlock / 64; ++wv) {
        r = MinAlg::combine(r, lds2[wv]);
        c += ldsc[wv];
      }
      s.bagg[pos] = r;
      if (c) atomicAdd(&s.cuts[s.level], static_cast<unsigned long long>(c));
    }
    return;
  }
  __syncthreads();  // (every thread has read ldsc before the general path reuses it)
  // `run` = the decoded classes of the sample right of this chunk (2 where nothing definite follows)
  auto decoded = [](unsigned r) {  // 2 -> 0
    const unsigned und = (r >> 1) & ~r & 0x15u;
    return r & ~(und | (und << 1)) & 0x15u;
  };
  int lab_next = __popc(decoded(run));
  unsigned o[kWords];
#pragma unroll
  for (int q = 0; q < kWords; ++q) o[q] = w[q];
  int32_t fne = kInf, foe = kInf;
  int ncut = 0;
#pragma unroll
  for (int k = kItems - 1; k >= 0; --k) {
    const int64_t i = c0 + k;
    if (i >= s.n) continue;
    const int b = ByteOf(w, k);
    const bool active = (b & kDone) != kDone;
    int lab = 0;
    if (!active) {
      run = 0u;
    } else {
      run = Dec3Alg::combine(run, static_cast<unsigned>(b >> 2));
      const unsigned u = decoded(run);
      lab = __popc(u);
      const unsigned nb8 = static_cast<unsigned>(b & 3) | (u << 2);
      o[k >> 2] = (o[k >> 2] & ~(0xffu << (8 * (k & 3)))) | (nb8 << (8 * (k & 3)));
    }
    const bool end = (b & kEnd) != 0;
    const bool cut_r = active && !end && lab != lab_next;
    if (end || cut_r) fne = static_cast<int32_t>(i);
    if (end) foe = static_cast<int32_t>(i);
    ncut += cut_r ? 1 : 0;
    lab_next = lab;
  }
  StoreBytes(s.st + c0, o);
  // tile aggregates of the boundary scan (a minimum: any order) and the cut count
  Int2 mn{fne, foe};
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn.a = min(mn.a, __shfl_down(mn.a, off, 64));
    mn.b = min(mn.b, __shfl_down(mn.b, off, 64));
    ncut += __shfl_down(ncut, off, 64);
  }
  if (lane == 0) {
    lds2[wave] = mn;
    ldsc[wave] = ncut;
  }
  __syncthreads();
  if (t == 0) {
    Int2 r = lds2[0];
    int c = ldsc[0];
    for (int wv = 1; wv < kBlock / 64; ++wv) {
      r = MinAlg::combine(r, lds2[wv]);
      c += ldsc[wv];
    }
    s.bagg[pos] = r;
    if (c) atomicAdd(&s.cuts[s.level], static_cast<unsigned long long>(c));
  }
}

// ---- pass 4: backward boundary scan (apply): next level's state bytes and region records ----------
template <class T>
__global__ __launch_bounds__(kBlock) void TvBoundKernel(TvS<T> s) {
  __shared__ Int2 lds2[kBlock / 64];
  __shared__ int ldsd[kBlock / 64];
  if (LevelIsDead(s)) return;
  const int64_t pos = blockIdx.x, tile = s.nb - 1 - pos;
  const int t = threadIdx.x, chunk = kBlock - 1 - t;
  if (s.tdone[tile]) return;
  const int64_t c0 = (tile * kBlock + chunk) * kItems;
  unsigned w[kWords];
  LoadBytes(s.st + c0, w);
  const Int2 pre_part = TilePrefixLoad<MinAlg>(s.bagg, pos, s.direct);
  // The bytes left and right of the chunk sit in the neighbouring lanes' registers (the scan runs
  // backwards: the chunk before this one belongs to lane + 1, the one after it to lane - 1); only
  // the lanes at a wave's edge read them from memory.
  const int lane = t & 63;
  const unsigned up = __shfl_down(w[kWords - 1], 1, 64), dn = __shfl_up(w[0], 1, 64);
  int bl = static_cast<int>(up >> 24), br = static_cast<int>(dn & 0xff);
  if (lane == 63) bl = c0 > 0 ? s.st[c0 - 1] : 0;
  if (lane == 0) br = c0 + kItems < s.n ? s.st[c0 + kItems] : 0;
  if (c0 == 0) bl = 0;
  if (c0 + kItems >= s.n) br = 0;
  // ---- the plain tile: one decoded value everywhere, no region boundary, the neighbours across
  // both tile edges carry it too - no cut, no record, nothing to scan: the next level's bytes are
  // zero (the first levels of a long signal: a few thousand cuts in 50 000 tiles)
  {
    const unsigned v = w[0] & 0xffu;
    bool same = (v & 3u) == 0u && (v & kDone) != static_cast<unsigned>(kDone) &&
                static_cast<unsigned>(bl) == v && static_cast<unsigned>(br) == v && c0 > 0 && c0 + kItems < s.n;
#pragma unroll
    for (int q = 0; q < kWords; ++q) same = same && w[q] == v * 0x01010101u;
    const bool wave_same = __all(same ? 1 : 0) != 0;
    if (lane == 0) ldsd[t >> 6] = wave_same ? 1 : 0;
    __syncthreads();
    bool plain = true;
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) plain = plain && ldsd[wv] != 0;
    __syncthreads();  // (ldsd is used again below)
    if (plain) {
      unsigned z[kWords] = {};
      StoreBytes(s.st2 + c0, z);
      return;
    }
  }
  unsigned new_head = 0, new_end = 0;
  bool all_done = true;
  int32_t foe = kInf;
  Int2 item[kItems];
  Int2 acc = MinAlg::identity();
#pragma unroll
  for (int k = kItems - 1; k >= 0; --k) {
    const int64_t i = c0 + k;
    item[k] = MinAlg::identity();
    if (i >= s.n) continue;
    const int b = ByteOf(w, k);
    const int bprev = k > 0 ? ByteOf(w, k - 1) : bl;
    const int bnext = k + 1 < kItems ? ByteOf(w, k + 1) : br;
    const bool active = (b & kDone) != kDone;
    const bool head = (b & kHead) != 0, end = (b & kEnd) != 0;
    all_done = all_done && !active;
    // inside a region (not across its old boundary) both neighbours are active too
    const bool cr = active && !end && Label(bnext) != Label(b);
    const bool cl = active && !head && Label(bprev) != Label(b);
    if (head || cl) new_head |= 1u << k;
    if (end || cr) new_end |= 1u << k;
    if (end) foe = static_cast<int32_t>(i);
    item[k] = Int2{(end || cr) ? static_cast<int32_t>(i) : kInf, end ? static_cast<int32_t>(i) : kInf};
    acc = MinAlg::combine(acc, item[k]);
  }
  Int2 tot;
  const Int2 excl = BlockExclusive<MinAlg>(acc, lds2, &tot);
  const Int2 pre = TilePrefixFinish<MinAlg>(pre_part, s.direct, lds2);
  Int2 run = MinAlg::combine(pre, excl);
  unsigned o[kWords] = {};
#pragma unroll
  for (int k = kItems - 1; k >= 0; --k) {
    const int64_t i = c0 + k;
    if (i >= s.n) continue;
    run = MinAlg::combine(run, item[k]);
    const int b = ByteOf(w, k);
    const bool active = (b & kDone) != kDone;
    const bool nh = new_head & (1u << k), ne = new_end & (1u << k);
    const unsigned nb8 = (nh ? kHead : 0) | (ne ? kEnd : 0) | (active ? 0 : kDone);
    o[k >> 2] |= nb8 << (8 * (k & 3));
    if (nh && active) {
      // record of the new region [i, r] inside the old region that ends at eo
      const int32_t r = run.a, eo = run.b;
      const int lab = Label(b);
      const int bprev = k > 0 ? ByteOf(w, k - 1) : bl;
      // across a cut the side with the larger label lies strictly above the other
      const unsigned clc = (b & kHead) ? (s.hrec[i].flags & 3u) : (lab > Label(bprev) ? 1u : 2u);
      const EndRec par = s.erec[eo];
      unsigned crc;
      if (r == eo) {
        crc = par.cr;
      } else {
        const int b1 = s.st[r], b2 = s.st[static_cast<int64_t>(r) + 1];
        crc = Label(b1) > Label(b2) ? 1u : 2u;
      }
      const double tot_y = s.Pp[static_cast<int64_t>(r) + 1] - s.Pp[i] -
                           s.lam * static_cast<double>(SideSign(clc) + SideSign(crc));
      const double tau = tot_y / static_cast<double>(r - static_cast<int32_t>(i) + 1);
      // delta: half the distance to the nearest parent threshold that bounds this region's values
      const double t1 = par.tau - static_cast<double>(par.delta), t3 = par.tau + static_cast<double>(par.delta);
      double dn;
      if (lab == 0) dn = t1 - tau;
      else if (lab == 3) dn = tau - t3;
      else if (lab == 1) dn = fmin(tau - t1, par.tau - tau);
      else dn = fmin(tau - par.tau, t3 - tau);
      dn = fmax(0.5 * dn, 0.0);
      const bool fin = (b & kHead) && r == eo;  // did not split: constant
      s.hrec[i] = HeadRec{tau, static_cast<float>(dn), clc | (crc << 2) | (fin ? 16u : 0u)};
      atomicMax(&s.tile_head[i / kTile], static_cast<int32_t>(i));
    }
  }
  StoreBytes(s.st2 + c0, o);
  // a tile whose samples are all finished drops out of every later pass
  const int wave = t >> 6;
  const bool wave_done = __all(all_done ? 1 : 0);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) foe = min(foe, __shfl_down(foe, off, 64));
  if (lane == 0) {
    ldsd[wave] = wave_done ? 1 : 0;
    lds2[wave] = Int2{foe, 0};
  }
  __syncthreads();
  if (t == 0) {
    bool d = true;
    int32_t fe = kInf;
    for (int wv = 0; wv < kBlock / 64; ++wv) {
      d = d && ldsd[wv] != 0;
      fe = min(fe, lds2[wv].a);
    }
    if (d) {
      s.tend[tile] = fe;
      s.tdone[tile] = 1;
    }
  }
}

// After the last level every remaining region is constant ("did not split"): write x = tau.
template <class T>
__global__ __launch_bounds__(kBlock) void TvFlushKernel(TvS<T> s) {
  __shared__ int32_t ldsi[kBlock / 64];
  const int64_t tile = blockIdx.x;
  const int t = threadIdx.x;
  if (s.tdone[tile]) return;
  const int64_t c0 = (tile * kBlock + t) * kItems;
  unsigned w[kWords];
  LoadBytes(s.st + c0, w);
  int32_t last = -1;
#pragma unroll
  for (int k = 0; k < kItems; ++k)
    if ((ByteOf(w, k) & kHead) && c0 + k < s.n) last = static_cast<int32_t>(c0 + k);
  int32_t tot_head;
  int32_t l = BlockExclusive<MaxAlg>(last, ldsi, &tot_head);
  const int32_t lin = s.direct ? TilePrefix<MaxAlg>(s.tile_head, tile, true, ldsi) : s.tile_l_in[tile];
  l = l > lin ? l : lin;
  double tau = 0.0;
  int32_t lrec = -2;
#pragma unroll 1
  for (int k = 0; k < kItems; ++k) {
    const int64_t i = c0 + k;
    if (i >= s.n) break;
    const int b = ByteOf(w, k);
    if (b & kHead) l = static_cast<int32_t>(i);
    if ((b & kDone) == kDone) continue;
    if (l != lrec) {
      tau = s.hrec[l].tau;
      lrec = l;
    }
    s.x[i] = static_cast<T>(tau);
  }
}

template <class T>
__global__ __launch_bounds__(64) void TvInitKernel(TvS<T> s) {
  // one region [0, n-1] without neighbours; delta from the spread of 64 block means (one lane each)
  const int lane = threadIdx.x;
  const int64_t nblk = s.n < 64 ? s.n : 64;
  double m = 0.0;
  if (lane < nblk) {
    const int64_t e0 = lane * s.n / nblk, e1 = (lane + 1) * s.n / nblk;
    m = (s.Pp[e1] - s.Pp[e0]) / static_cast<double>(e1 > e0 ? e1 - e0 : 1);
  }
  double sum = m, sq = m * m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
    sq += __shfl_down(sq, off, 64);
  }
  if (lane == 0) {
    s.st[0] = static_cast<uint8_t>(s.st[0] | kHead);
    s.st[s.n - 1] = static_cast<uint8_t>(s.st[s.n - 1] | kEnd);
    const double mean = sum / static_cast<double>(nblk);
    const double var = fmax(sq / static_cast<double>(nblk) - mean * mean, 0.0);
    s.hrec[0] = HeadRec{s.Pp[s.n] / static_cast<double>(s.n), static_cast<float>(0.67 * sqrt(var)), 0u};
    s.tile_head[0] = 0;
  }
}

template <class T> int Tv1dLevelSets3(const DVec& xv, const DVec& yv, double lam) {
  const int64_t n = yv.n;
  Runtime& rt = Runtime::Get();
  hipStream_t q = rt.stream();
  ProfScope prof("tv1d", n);
  const int64_t nb = (n + kTile - 1) / kTile;
  const int64_t npad = nb * kTile;
  const bool direct = nb <= kDirectTiles;
  auto bP = rt.Alloc((static_cast<size_t>(npad) + 4) * sizeof(double));
  auto bst = rt.Alloc(static_cast<size_t>(npad) * 2);
  auto bh = rt.Alloc(static_cast<size_t>(n) * sizeof(HeadRec));
  auto be = rt.Alloc(static_cast<size_t>(n) * sizeof(EndRec));
  // per-tile words, one allocation: tile_head, tile_l_in, tdone, tend (int32 each), dagg (u32), bagg (Int2)
  auto bt = rt.Alloc(static_cast<size_t>(nb) * (4 * sizeof(int32_t) + sizeof(unsigned) + sizeof(Int2)));
  auto bclip = rt.Alloc(static_cast<size_t>(nb) * sizeof(Clip3));
  auto bagg = rt.Alloc(static_cast<size_t>(nb) * sizeof(double));
  auto bcuts = rt.Alloc(static_cast<size_t>(kMaxLevels) * sizeof(unsigned long long));
  double* Pp = static_cast<double*>(bP->p) + 1;  // Pp + 1 is 16-byte aligned
  EPS_HIP(hipMemsetAsync(Pp, 0, sizeof(double), q));
  const bool aligned = reinterpret_cast<uintptr_t>(yv.data()) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(xv.data()) % 16 == 0;
  // fp64 prefix sums of y
  double* pagg = static_cast<double*>(bagg->p);
  hipLaunchKernelGGL(PrefixReduceKernel<T>, dim3(static_cast<unsigned>(nb)), dim3(kBlock), 0, q, yv.as<T>(), n,
                     aligned ? 1 : 0, pagg);
  if (!direct) RunAggScan<SumAlg>(nb, pagg);
  hipLaunchKernelGGL(PrefixApplyKernel<T>, dim3(static_cast<unsigned>(nb)), dim3(kBlock), 0, q, yv.as<T>(), n,
                     aligned ? 1 : 0, pagg, direct ? 1 : 0, Pp);

  TvS<T> s;
  s.y = yv.as<T>();
  s.x = xv.as<T>();
  s.Pp = Pp;
  uint8_t* stA = static_cast<uint8_t*>(bst->p);
  uint8_t* stB = stA + npad;
  s.hrec = static_cast<HeadRec*>(bh->p);
  s.erec = static_cast<EndRec*>(be->p);
  int32_t* tw = static_cast<int32_t*>(bt->p);
  s.tile_head = tw;
  int32_t* tile_l_in = tw + nb;
  s.tile_l_in = tile_l_in;
  s.tdone = tw + 2 * nb;
  s.tend = tw + 3 * nb;
  s.dagg = reinterpret_cast<unsigned*>(tw + 4 * nb);
  s.bagg = reinterpret_cast<Int2*>(tw + 5 * nb);
  s.agg_clip = static_cast<Clip3*>(bclip->p);
  s.cuts = static_cast<unsigned long long*>(bcuts->p);
  s.lam = lam;
  s.n = n;
  s.nb = nb;
  s.aligned = aligned;
  s.direct = direct;
  s.level = 0;
  EPS_HIP(hipMemsetAsync(stA, 0, static_cast<size_t>(npad) * 2, q));
  EPS_HIP(hipMemsetAsync(s.tile_head, 0xff, static_cast<size_t>(nb) * sizeof(int32_t), q));
  EPS_HIP(hipMemsetAsync(s.tdone, 0, static_cast<size_t>(nb) * sizeof(int32_t), q));
  EPS_HIP(hipMemsetAsync(s.cuts, 0, static_cast<size_t>(kMaxLevels) * sizeof(unsigned long long), q));
  s.st = stA;
  s.st2 = stB;
  hipLaunchKernelGGL(TvInitKernel<T>, dim3(1), dim3(64), 0, q, s);

  const dim3 grid(static_cast<unsigned>(nb)), block(kBlock);
  auto enqueue_level = [&](int level) {
    s.level = level;
    s.st = (level & 1) ? stB : stA;
    s.st2 = (level & 1) ? stA : stB;
    const unsigned long long* gate = level > 0 ? s.cuts + (level - 1) : nullptr;
    if (!direct) RunAggScan<MaxAlg>(nb, s.tile_head, gate, tile_l_in);
    hipLaunchKernelGGL((TvClipKernel<T, 0>), grid, block, 0, q, s);
    if (!direct) RunAggScan<Clip3Alg>(nb, s.agg_clip, gate);
    hipLaunchKernelGGL((TvClipKernel<T, 1>), grid, block, 0, q, s);
    if (!direct) RunAggScan<Dec3Alg>(nb, s.dagg, gate);
    hipLaunchKernelGGL(TvDecodeKernel<T>, grid, block, 0, q, s);
    if (!direct) RunAggScan<MinAlg>(nb, s.bagg, gate);
    hipLaunchKernelGGL(TvBoundKernel<T>, grid, block, 0, q, s);
  };
  // Levels are enqueued in batches; every kernel of a level returns at once when the level
  // before it made no cut, so a batch that overshoots costs a few empty launches, not a pass.
  int enq = 0, last = -1;
  std::vector<unsigned long long> h(kMaxLevels);
  while (last < 0) {
    const int batch = enq == 0 ? 6 : 2;
    EPS_CHECK_MSG(enq + batch < kMaxLevels, "tv1d: level-set recursion did not terminate");
    const int first = enq;
    for (int b = 0; b < batch; ++b) enqueue_level(enq++);
    EPS_HIP(hipMemcpyAsync(h.data() + first, s.cuts + first, static_cast<size_t>(batch) * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, q));
    EPS_HIP(hipStreamSynchronize(q));
    for (int lv = first; lv < enq; ++lv)
      if (h[lv] == 0) {
        last = lv;
        break;
      }
  }
  // every remaining region carries "did not split" now: write their x.  The state is the one the
  // boundary pass of level `last` wrote.
  s.level = last + 1;
  s.st = ((last + 1) & 1) ? stB : stA;
  s.st2 = ((last + 1) & 1) ? stA : stB;
  if (!direct) RunAggScan<MaxAlg>(nb, s.tile_head, nullptr, tile_l_in);
  hipLaunchKernelGGL(TvFlushKernel<T>, grid, block, 0, q, s);
  EPS_HIP(hipGetLastError());
  return last + 1;
}

int g_last_levels = 0;

}  // namespace

int Tv1dLastLevels() { return g_last_levels; }

void Tv1d(const DVec& x, const DVec& v, double lam) {
  EPS_CHECK(x.n == v.n && x.dt == v.dt);
  const int64_t n = x.n;
  if (n == 0) return;
  EPS_CHECK_MSG(n < (int64_t(1) << 31) - 1, "tv1d: n must be below 2^31");
  if (n == 1 || lam == 0) {  // tf_dp's trivial cases
    Copy(x, v);
    return;
  }
  static const char* form = std::getenv("EPSILON_HIP_TV");
  if (form != nullptr && form[0] == 'b') {
    g_last_levels = Tv1dBinary(x, v, lam);
    return;
  }
  if (x.dt == F32) g_last_levels = Tv1dLevelSets3<float>(x, v, lam);
  else g_last_levels = Tv1dLevelSets3<double>(x, v, lam);
}

}  // namespace k
}  // namespace eps

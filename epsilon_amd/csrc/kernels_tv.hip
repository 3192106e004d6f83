// K9: exact 1-D total-variation (fused-lasso) prox, parallel.
//
//   x = argmin 1/2 ||x - y||^2 + lam * sum_i |x[i+1] - x[i]|
//
// The reference calls glmgen's `tf_dp` (reference src/epsilon/prox/total_variation_1d.cc:8,21):
// Johnson's dynamic program, inherently sequential (a forward knot sweep and a backward fill).
// This kernel set computes the same unique minimiser with a divide-and-conquer over LEVEL SETS
// (Hochbaum 2001 / Chambolle-Darbon 2009, specialised to a chain):
//
//   For a contiguous region R whose neighbours are known to lie strictly above / below it, the
//   neighbour terms are linear and fold into the end samples (y'_l = y_l - lam*c_l, ...).
//   With tau = mean(y'_R):  {i : x*_i > tau} is the minimal minimiser of the BINARY chain problem
//       min_u  sum_i (tau - y'_i) u_i + lam * sum_i |u[i+1] - u[i]| ,   u in {0,1}^R ,
//   and since mean(x*_R) = tau, the region is constant (= tau) iff that set is empty.
//   Otherwise the runs of u are new regions, strictly ordered across every cut, and recurse.
//
// The binary chain problem is a 2-state Viterbi whose forward recursion on the cost difference
// d_i = cost(u_i=1) - cost(u_i=0) is  d_i = a_i + clip(d_{i-1}, -lam, lam)  - a composition of
// clamp-shift maps (p, lo, hi), which is associative => a parallel SCAN; the backward decode
// u_i = [d_i < -lam] or [d_i < lam and u_{i+1}] is a "first definite value to the right" scan.
// All decisions are taken in fp64 (data may be f32), so the partition matches the fp64 DP except
// at exact ties.
//
// Data layout (round 2: 17 bytes per sample and level instead of ~70).  The whole state of the
// recursion is ONE byte per sample,
//     bit 0 head of a region | bit 1 end | bits 2-3 side of the left neighbour (at heads)
//     bits 4-5 side of the right neighbour (at ends) | bits 6-7 class (0 / 1 / 2 = copy, 3 = done)
// plus a sparse table indexed by region head (tau, finished flag) that only region heads touch,
// plus the fp64 prefix sums of y that only region boundaries touch.  A level is three scans over
// the byte (and, for the first, over y):
//     forward   clamp-shift scan: reads y + state, writes the class into the state byte; the
//               region head of a sample (to find its tau) is the running maximum of head
//               positions - inside a tile a workgroup scan, across tiles a pre-scanned table;
//               samples of regions that finished one level earlier get x = tau here, once;
//     backward  decode scan: state -> state (u replaces the class);
//     backward  boundary scan (nearest new end / nearest old end): finds the cuts, writes the
//               next level's state byte into the second buffer and, at every new head, the
//               region record (tau from the prefix sums, "did not split" = finished).
// Each scan is reduce -> scan of the tile aggregates -> apply; tiles are read with 16-byte loads.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;
constexpr int kAggItems = 8;
constexpr int kAggTile = kBlock * kAggItems;
constexpr int kFwdItems = 8;               // samples per thread in the scans that read y
constexpr int kFwdTile = kBlock * kFwdItems;
constexpr int kByteItems = 16;             // samples per thread in the byte scans
constexpr int32_t kInf = 0x7fffffff;

// ---- scan algebra ---------------------------------------------------------------------------------

struct ClipMap {
  double p, lo, hi;
};
struct ClipAlg {
  using S = ClipMap;
  __device__ static S combine(S f, S g) {  // g after f
    S o;
    o.p = f.p + g.p;
    o.lo = fmin(fmax(f.lo + g.p, g.lo), g.hi);
    o.hi = fmin(fmax(f.hi + g.p, g.lo), g.hi);
    return o;
  }
  __device__ static S identity() { return S{0.0, -INFINITY, INFINITY}; }
};
struct DecodeAlg {  // first definite class in scan order wins
  using S = int;
  __device__ static S combine(S acc, S next) { return next != 2 ? next : acc; }
  __device__ static S identity() { return 2; }
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

// One step of a wave scan on the DPP path of the vector ALU (no LDS crossbar: ds_bpermute costs
// ~100 cycles per step and word, a DPP move a few): every 32-bit word of the value moves by the
// same lane pattern.  Lanes without a source (row_shr at the start of a row, rows outside
// ROW_MASK) get their own value back; the caller combines only where a source exists.
template <int CTRL, int ROW_MASK, class S> __device__ inline S DppMove(const S& v) {
  constexpr int W = (sizeof(S) + 3) / 4;
  int w[W] = {};
  memcpy(w, &v, sizeof(S));
#pragma unroll
  for (int k = 0; k < W; ++k) w[k] = __builtin_amdgcn_update_dpp(w[k], w[k], CTRL, ROW_MASK, 0xF, false);
  S o;
  memcpy(&o, w, sizeof(S));
  return o;
}

// Inclusive scan over the 64 lanes of a wave, in lane order (Alg::combine(earlier, later)):
// Hillis-Steele inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of rows 0 / 2 into
// rows 1 / 3 (row_bcast15) and lane 31 into rows 2 and 3 (row_bcast31).  (n = 1e8: 14.3 -> 13.9 ms.)
template <class Alg> __device__ inline typename Alg::S WaveInclusive(typename Alg::S v, int lane) {
  using S = typename Alg::S;
  const int in_row = lane & 15;
  {
    const S up = DppMove<0x111, 0xF>(v);
    if (in_row >= 1) v = Alg::combine(up, v);
  }
  {
    const S up = DppMove<0x112, 0xF>(v);
    if (in_row >= 2) v = Alg::combine(up, v);
  }
  {
    const S up = DppMove<0x114, 0xF>(v);
    if (in_row >= 4) v = Alg::combine(up, v);
  }
  {
    const S up = DppMove<0x118, 0xF>(v);
    if (in_row >= 8) v = Alg::combine(up, v);
  }
  {
    const S up = DppMove<0x142, 0xA>(v);
    if ((lane >> 4) & 1) v = Alg::combine(up, v);
  }
  {
    const S up = DppMove<0x143, 0xC>(v);
    if (lane >= 32) v = Alg::combine(up, v);
  }
  return v;
}

// Exclusive scan of one value per thread over the workgroup, in thread order: DPP scan inside a
// wave, the 4 wave totals through LDS.  *total = the workgroup aggregate.
template <class Alg>
__device__ inline typename Alg::S BlockExclusive(typename Alg::S mine, typename Alg::S* lds,
                                                 typename Alg::S* total) {
  using S = typename Alg::S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const S incl = WaveInclusive<Alg>(mine, lane);
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  S before = Alg::identity();
  for (int w = 0; w < wave; ++w) before = Alg::combine(before, lds[w]);
  S tot = lds[0];
  for (int w = 1; w < kBlock / 64; ++w) tot = Alg::combine(tot, lds[w]);
  *total = tot;
  S excl = ShflUp(incl, 1);
  if (lane == 0) excl = Alg::identity();
  excl = Alg::combine(before, excl);
  __syncthreads();
  return excl;
}

// ---- scan of the tile aggregates (exclusive, in place) -------------------------------------------

// one workgroup: every thread owns a contiguous chunk
template <class Alg>
__global__ __launch_bounds__(kBlock) void AggScanKernel(int64_t nb, typename Alg::S* agg) {
  using S = typename Alg::S;
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

// two-level form for many tiles (n = 1e8: tens of thousands of aggregates)
template <class Alg>
__global__ __launch_bounds__(kBlock) void AggReduceKernel(int64_t nb, const typename Alg::S* agg,
                                                          typename Alg::S* agg2) {
  using S = typename Alg::S;
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

template <class Alg>
__global__ __launch_bounds__(kBlock) void AggApplyKernel(int64_t nb, typename Alg::S* agg,
                                                         const typename Alg::S* agg2) {
  using S = typename Alg::S;
  __shared__ S lds[kBlock / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kAggTile + static_cast<int64_t>(threadIdx.x) * kAggItems;
  S item[kAggItems];
  S acc = Alg::identity();
#pragma unroll
  for (int k = 0; k < kAggItems; ++k) {
    item[k] = base + k < nb ? agg[base + k] : Alg::identity();
    acc = Alg::combine(acc, item[k]);
  }
  S total;
  S excl = BlockExclusive<Alg>(acc, lds, &total);
  S run = Alg::combine(agg2[blockIdx.x], excl);
#pragma unroll
  for (int k = 0; k < kAggItems; ++k) {
    if (base + k < nb) {
      agg[base + k] = run;
      run = Alg::combine(run, item[k]);
    }
  }
}

template <class Alg> void RunAggScan(int64_t nb, typename Alg::S* agg) {
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  if (nb > 2 * kAggTile) {
    const int64_t nb2 = (nb + kAggTile - 1) / kAggTile;
    auto buf = rt.Alloc(static_cast<size_t>(nb2) * sizeof(typename Alg::S));
    auto* agg2 = static_cast<typename Alg::S*>(buf->p);
    hipLaunchKernelGGL((AggReduceKernel<Alg>), dim3(static_cast<unsigned>(nb2)), dim3(kBlock), 0, s, nb, agg, agg2);
    hipLaunchKernelGGL((AggScanKernel<Alg>), dim3(1), dim3(kBlock), 0, s, nb2, agg2);
    hipLaunchKernelGGL((AggApplyKernel<Alg>), dim3(static_cast<unsigned>(nb2)), dim3(kBlock), 0, s, nb, agg, agg2);
  } else {
    hipLaunchKernelGGL((AggScanKernel<Alg>), dim3(1), dim3(kBlock), 0, s, nb, agg);
  }
}

// ---- tile scans ----------------------------------------------------------------------------------
// A tile op provides: Alg (S, combine, identity), kItems, kRev, a register context Ctx and
//   Load (tile, chunk, item[kItems], ctx)  block-cooperative (may synchronise): elements of the
//                                          natural positions c0 .. c0+kItems-1, c0 = (tile*kBlock + chunk)*kItems
//   Store(tile, chunk, incl[kItems], ctx)  inclusive results at the same natural positions.
// Scan order is natural (kRev = false) or reversed; with kRev the tile / chunk the workgroup /
// thread takes is mirrored, so that every thread still owns a contiguous, aligned chunk.

template <class Op>
__global__ __launch_bounds__(kBlock) void TileReduceKernel(Op op, int64_t nb, typename Op::Alg::S* agg) {
  using Alg = typename Op::Alg;
  using S = typename Alg::S;
  __shared__ S lds[kBlock / 64];
  const int64_t tile = Op::kRev ? nb - 1 - blockIdx.x : blockIdx.x;
  const int chunk = Op::kRev ? kBlock - 1 - static_cast<int>(threadIdx.x) : static_cast<int>(threadIdx.x);
  S item[Op::kItems];
  typename Op::Ctx ctx;
  op.Load(tile, chunk, item, ctx);
  S acc = Alg::identity();
#pragma unroll
  for (int k = 0; k < Op::kItems; ++k) acc = Alg::combine(acc, item[Op::kRev ? Op::kItems - 1 - k : k]);
  S total;
  BlockExclusive<Alg>(acc, lds, &total);
  if (threadIdx.x == 0) agg[blockIdx.x] = total;
}

template <class Op>
__global__ __launch_bounds__(kBlock) void TileApplyKernel(Op op, int64_t nb, const typename Op::Alg::S* agg) {
  using Alg = typename Op::Alg;
  using S = typename Alg::S;
  __shared__ S lds[kBlock / 64];
  const int64_t tile = Op::kRev ? nb - 1 - blockIdx.x : blockIdx.x;
  const int chunk = Op::kRev ? kBlock - 1 - static_cast<int>(threadIdx.x) : static_cast<int>(threadIdx.x);
  S item[Op::kItems];
  typename Op::Ctx ctx;
  op.Load(tile, chunk, item, ctx);
  S acc = Alg::identity();
#pragma unroll
  for (int k = 0; k < Op::kItems; ++k) acc = Alg::combine(acc, item[Op::kRev ? Op::kItems - 1 - k : k]);
  S total;
  const S excl = BlockExclusive<Alg>(acc, lds, &total);
  S run = Alg::combine(agg[blockIdx.x], excl);
#pragma unroll
  for (int k = 0; k < Op::kItems; ++k) {
    const int kk = Op::kRev ? Op::kItems - 1 - k : k;
    run = Alg::combine(run, item[kk]);
    item[kk] = run;
  }
  op.Store(tile, chunk, item, ctx);
}

template <class Op> void RunTileScan(const Op& op, int64_t n) {
  if (n <= 0) return;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  const int64_t tile = static_cast<int64_t>(kBlock) * Op::kItems;
  const int64_t nb = (n + tile - 1) / tile;
  using S = typename Op::Alg::S;
  auto buf = rt.Alloc(static_cast<size_t>(nb) * sizeof(S));
  S* agg = static_cast<S*>(buf->p);
  hipLaunchKernelGGL((TileReduceKernel<Op>), dim3(static_cast<unsigned>(nb)), dim3(kBlock), 0, s, op, nb, agg);
  RunAggScan<typename Op::Alg>(nb, agg);
  hipLaunchKernelGGL((TileApplyKernel<Op>), dim3(static_cast<unsigned>(nb)), dim3(kBlock), 0, s, op, nb, agg);
}

// ---- vector access helpers -----------------------------------------------------------------------

template <class T> struct Pack8;
template <> struct Pack8<float> {
  __device__ static void load(const float* p, float (&v)[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  __device__ static void store(float* p, const float (&v)[8]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct Pack8<double> {
  __device__ static void load(const double* p, double (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 a = reinterpret_cast<const double2*>(p)[q];
      v[2 * q] = a.x;
      v[2 * q + 1] = a.y;
    }
  }
  __device__ static void store(double* p, const double (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) reinterpret_cast<double2*>(p)[q] = make_double2(v[2 * q], v[2 * q + 1]);
  }
};

// ---- state byte ----------------------------------------------------------------------------------

__device__ inline int SideSign(int code) { return code == 1 ? 1 : (code == 2 ? -1 : 0); }
__device__ inline int SideCode(int sign) { return sign > 0 ? 1 : (sign < 0 ? 2 : 0); }
constexpr int kHead = 1, kEnd = 2;

// ---- prefix sums of y (fp64): Pp[i + 1] = sum_{k <= i} y_k, Pp[0] = 0 -----------------------------
// (Pp points one double behind a 16-byte aligned buffer, so the 8 results of a chunk are aligned)

template <class T> struct PrefixOp {
  using Alg = SumAlg;
  static constexpr int kItems = kFwdItems;
  static constexpr bool kRev = false;
  struct Ctx {};
  const T* y;
  double* Pp;
  int64_t n;
  bool aligned;
  __device__ void Load(int64_t tile, int chunk, double (&item)[kItems], Ctx&) const {
    const int64_t c0 = (tile * kBlock + chunk) * kItems;
    if (aligned && c0 + kItems <= n) {
      T v[8];
      Pack8<T>::load(y + c0, v);
#pragma unroll
      for (int k = 0; k < kItems; ++k) item[k] = static_cast<double>(v[k]);
    } else {
#pragma unroll
      for (int k = 0; k < kItems; ++k) item[k] = c0 + k < n ? static_cast<double>(y[c0 + k]) : 0.0;
    }
  }
  __device__ void Store(int64_t tile, int chunk, const double (&incl)[kItems], Ctx&) const {
    const int64_t c0 = (tile * kBlock + chunk) * kItems;
    if (c0 + kItems <= n) {
      Pack8<double>::store(Pp + c0 + 1, incl);
    } else {
#pragma unroll
      for (int k = 0; k < kItems; ++k)
        if (c0 + k < n) Pp[c0 + k + 1] = incl[k];
    }
  }
};

// ---- the recursion's shared state ------------------------------------------------------------------

template <class T> struct TvState {
  const T* y;
  T* x;
  const double* Pp;        // prefix sums
  uint8_t* st;             // state byte per sample (padded to a tile multiple)
  uint8_t* st2;            // next level's state (written by the boundary scan)
  double* tau_of;          // by region head
  uint8_t* fin_of;         // by region head: the region did not split -> constant
  int32_t* tile_head;      // per forward tile: last head position inside it (-1 none)
  const int32_t* tile_l_in;  // per forward tile: last head position before it
  unsigned long long* cuts;
  double lam;
  int64_t n;
  bool aligned;            // y and x are 16-byte aligned
};

// forward clamp-shift scan
template <class T, bool FLUSH> struct ClipOp {
  using Alg = ClipAlg;
  static constexpr int kItems = kFwdItems;
  static constexpr bool kRev = false;
  struct Ctx {
    unsigned long long bytes;   // the 8 state bytes
    unsigned fin_mask;          // samples whose region finished a level ago: x is written now
    double tau[kItems];         // (only read where fin_mask is set)
  };
  TvState<T> s;

  __device__ void Load(int64_t tile, int chunk, ClipMap (&item)[kItems], Ctx& ctx) const {
    __shared__ int32_t lds[kBlock / 64];
    const int64_t c0 = (tile * kBlock + chunk) * kItems;
    const unsigned long long bytes = *reinterpret_cast<const unsigned long long*>(s.st + c0);
    T yv[8];
    if (s.aligned && c0 + kItems <= s.n) {
      Pack8<T>::load(s.y + c0, yv);
    } else {
#pragma unroll
      for (int k = 0; k < kItems; ++k) yv[k] = c0 + k < s.n ? s.y[c0 + k] : T(0);
    }
    // region head of every sample = running maximum of head positions
    int32_t last = -1;
#pragma unroll
    for (int k = 0; k < kItems; ++k)
      if (((bytes >> (8 * k)) & kHead) && c0 + k < s.n) last = static_cast<int32_t>(c0 + k);
    int32_t total;
    int32_t l = BlockExclusive<MaxAlg>(last, lds, &total);
    const int32_t lin = s.tile_l_in[tile];
    l = l > lin ? l : lin;
    ctx.bytes = bytes;
    ctx.fin_mask = 0;
    double tau = 0.0;
    bool fin = false;
    int32_t lrec = -2;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int64_t i = c0 + k;
      const int b = static_cast<int>((bytes >> (8 * k)) & 0xff);
      ctx.tau[k] = 0.0;
      if (i >= s.n) {
        item[k] = ClipAlg::identity();
        continue;
      }
      if (b & kHead) l = static_cast<int32_t>(i);
      const int cls = b >> 6;
      if (cls == 3) {
        item[k] = ClipMap{0.0, 0.0, 0.0};  // finished regions are inert
        continue;
      }
      if (l != lrec) {
        tau = s.tau_of[l];
        fin = s.fin_of[l] != 0;
        lrec = l;
      }
      if (fin || FLUSH) {
        ctx.fin_mask |= 1u << k;
        ctx.tau[k] = tau;
        item[k] = ClipMap{0.0, 0.0, 0.0};
        continue;
      }
      double yp = static_cast<double>(yv[k]);
      if (b & kHead) yp -= s.lam * static_cast<double>(SideSign((b >> 2) & 3));
      if (b & kEnd) yp -= s.lam * static_cast<double>(SideSign((b >> 4) & 3));
      const double a = tau - yp;
      item[k] = (b & kHead) ? ClipMap{0.0, a, a} : ClipMap{a, a - s.lam, a + s.lam};
    }
  }

  __device__ void Store(int64_t tile, int chunk, const ClipMap (&incl)[kItems], Ctx& ctx) const {
    const int64_t c0 = (tile * kBlock + chunk) * kItems;
    unsigned long long out = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int b = static_cast<int>((ctx.bytes >> (8 * k)) & 0xff);
      int cls = b >> 6;
      if (c0 + k < s.n && cls != 3) {
        if (ctx.fin_mask & (1u << k)) {
          cls = 3;
        } else {
          const ClipMap m = incl[k];
          const double d = fmin(fmax(m.p, m.lo), m.hi);  // the composed map applied to 0
          if (b & kEnd) cls = d < 0.0 ? 1 : 0;           // region end: definite
          else if (d < -s.lam) cls = 1;
          else if (d >= s.lam) cls = 0;
          else cls = 2;
        }
      }
      out |= static_cast<unsigned long long>((b & 0x3f) | (cls << 6)) << (8 * k);
    }
    *reinterpret_cast<unsigned long long*>(s.st + c0) = out;
    if (ctx.fin_mask == 0xffu && s.aligned && c0 + kItems <= s.n) {
      T xv[8];
#pragma unroll
      for (int k = 0; k < kItems; ++k) xv[k] = static_cast<T>(ctx.tau[k]);
      Pack8<T>::store(s.x + c0, xv);
    } else if (ctx.fin_mask) {
#pragma unroll
      for (int k = 0; k < kItems; ++k)
        if (ctx.fin_mask & (1u << k)) s.x[c0 + k] = static_cast<T>(ctx.tau[k]);
    }
  }
};

// after the last level: write x for everything that is not written yet (no scan)
template <class T>
__global__ __launch_bounds__(kBlock) void TvFlushKernel(ClipOp<T, true> op) {
  ClipMap item[kFwdItems];
  typename ClipOp<T, true>::Ctx ctx;
  op.Load(blockIdx.x, threadIdx.x, item, ctx);
  op.Store(blockIdx.x, threadIdx.x, item, ctx);
}

// backward decode scan: u_i = first definite class at or to the right of i
template <class T> struct DecodeOp {
  using Alg = DecodeAlg;
  static constexpr int kItems = kByteItems;
  static constexpr bool kRev = true;
  struct Ctx {
    uint4 bytes;
  };
  TvState<T> s;
  __device__ void Load(int64_t tile, int chunk, int (&item)[kItems], Ctx& ctx) const {
    const int64_t c0 = (tile * kBlock + chunk) * kItems;
    ctx.bytes = *reinterpret_cast<const uint4*>(s.st + c0);
    const unsigned w[4] = {ctx.bytes.x, ctx.bytes.y, ctx.bytes.z, ctx.bytes.w};
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int cls = static_cast<int>((w[k >> 2] >> (8 * (k & 3) + 6)) & 3);
      item[k] = c0 + k < s.n ? (cls == 3 ? 0 : cls) : 2;
    }
  }
  __device__ void Store(int64_t tile, int chunk, const int (&incl)[kItems], Ctx& ctx) const {
    const int64_t c0 = (tile * kBlock + chunk) * kItems;
    unsigned w[4] = {ctx.bytes.x, ctx.bytes.y, ctx.bytes.z, ctx.bytes.w};
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int sh = 8 * (k & 3) + 6;
      const unsigned cls = (w[k >> 2] >> sh) & 3u;
      if (cls != 3u && c0 + k < s.n)
        w[k >> 2] = (w[k >> 2] & ~(3u << sh)) | (static_cast<unsigned>(incl[k] & 1) << sh);
    }
    *reinterpret_cast<uint4*>(s.st + c0) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// backward boundary scan: nearest NEW end / nearest OLD end at or to the right of every sample
template <class T> struct BoundOp {
  using Alg = MinAlg;
  static constexpr int kItems = kByteItems;
  static constexpr bool kRev = true;
  struct Ctx {
    uint4 bytes;
    unsigned new_head, new_end, cut_left, cut_right;  // bit k = sample c0 + k
  };
  TvState<T> s;
  __device__ void Load(int64_t tile, int chunk, Int2 (&item)[kItems], Ctx& ctx) const {
    const int64_t c0 = (tile * kBlock + chunk) * kItems;
    ctx.bytes = *reinterpret_cast<const uint4*>(s.st + c0);
    const unsigned w[4] = {ctx.bytes.x, ctx.bytes.y, ctx.bytes.z, ctx.bytes.w};
    // The bytes left and right of the chunk sit in the neighbouring lanes' registers (the scan runs
    // backwards: the chunk before this one belongs to lane + 1, the one after it to lane - 1);
    // only the lanes at a wave's edge read them from memory (two scalar byte loads per thread made
    // this kernel run at 1.4 TB/s).
    const int lane = threadIdx.x & 63;
    const unsigned up = __shfl_down(ctx.bytes.w, 1, 64), dn = __shfl_up(ctx.bytes.x, 1, 64);
    int bl = static_cast<int>(up >> 24), br = static_cast<int>(dn & 0xff);
    if (lane == 63) bl = c0 > 0 ? s.st[c0 - 1] : 0;
    if (lane == 0) br = c0 + kItems < s.n ? s.st[c0 + kItems] : 0;
    if (c0 == 0) bl = 0;
    if (c0 + kItems >= s.n) br = 0;
    ctx.new_head = ctx.new_end = ctx.cut_left = ctx.cut_right = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int64_t i = c0 + k;
      if (i >= s.n) {
        item[k] = MinAlg::identity();
        continue;
      }
      const int b = static_cast<int>((w[k >> 2] >> (8 * (k & 3))) & 0xff);
      const int bprev = k > 0 ? static_cast<int>((w[(k - 1) >> 2] >> (8 * ((k - 1) & 3))) & 0xff) : bl;
      const int bnext = k + 1 < kItems ? static_cast<int>((w[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xff) : br;
      const int cls = b >> 6;
      const bool active = cls != 3;
      const bool head = b & kHead, end = b & kEnd;
      // inside a region (not across its old boundary) both neighbours are active too
      const bool cr = active && !end && ((bnext >> 6) & 1) != (cls & 1);
      const bool cl = active && !head && ((bprev >> 6) & 1) != (cls & 1);
      if (head || cl) ctx.new_head |= 1u << k;
      if (end || cr) ctx.new_end |= 1u << k;
      if (cl) ctx.cut_left |= 1u << k;
      if (cr) ctx.cut_right |= 1u << k;
      item[k] = Int2{(end || cr) ? static_cast<int32_t>(i) : kInf, end ? static_cast<int32_t>(i) : kInf};
    }
  }
  __device__ void Store(int64_t tile, int chunk, const Int2 (&incl)[kItems], Ctx& ctx) const {
    const int64_t c0 = (tile * kBlock + chunk) * kItems;
    const unsigned w[4] = {ctx.bytes.x, ctx.bytes.y, ctx.bytes.z, ctx.bytes.w};
    unsigned o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int64_t i = c0 + k;
      if (i >= s.n) continue;
      const int b = static_cast<int>((w[k >> 2] >> (8 * (k & 3))) & 0xff);
      const int cls = b >> 6;
      const bool active = cls != 3;
      const int u = cls & 1;
      const bool nh = ctx.new_head & (1u << k), ne = ctx.new_end & (1u << k);
      // across a cut the u = 1 side lies strictly above the u = 0 side
      int clc = (b & kHead) ? ((b >> 2) & 3) : ((ctx.cut_left & (1u << k)) ? SideCode(u ? 1 : -1) : 0);
      int crc = (b & kEnd) ? ((b >> 4) & 3) : ((ctx.cut_right & (1u << k)) ? SideCode(u ? 1 : -1) : 0);
      const int nb = (nh ? kHead : 0) | (ne ? kEnd : 0) | (clc << 2) | (crc << 4) | (active ? 0 : (3 << 6));
      o[k >> 2] |= static_cast<unsigned>(nb) << (8 * (k & 3));
      if (nh && active) {
        // region record of the new region [i, r]
        const int32_t r = incl[k].a;
        const int brr = s.st[r];
        const int cr_sign = (brr & kEnd) ? SideSign((brr >> 4) & 3) : (((brr >> 6) & 1) ? 1 : -1);
        const double tot = s.Pp[static_cast<int64_t>(r) + 1] - s.Pp[i] -
                           s.lam * static_cast<double>(SideSign(clc) + cr_sign);
        s.tau_of[i] = tot / static_cast<double>(r - static_cast<int32_t>(i) + 1);
        s.fin_of[i] = ((b & kHead) && incl[k].a == incl[k].b) ? 1 : 0;
        atomicMax(&s.tile_head[i / kFwdTile], static_cast<int32_t>(i));
      }
    }
    *reinterpret_cast<uint4*>(s.st2 + c0) = make_uint4(o[0], o[1], o[2], o[3]);
    const int ncut = __popc(ctx.cut_right);
    if (ncut) atomicAdd(s.cuts, static_cast<unsigned long long>(ncut));  // few: one per cut point
  }
};

template <class T>
__global__ void TvInitKernel(TvState<T> s, int32_t* tile_head) {
  // one region [0, n-1] without neighbours
  s.st[0] = static_cast<uint8_t>(s.st[0] | kHead);
  s.st[s.n - 1] = static_cast<uint8_t>(s.st[s.n - 1] | kEnd);
  s.tau_of[0] = s.Pp[s.n] / static_cast<double>(s.n);
  s.fin_of[0] = 0;
  tile_head[0] = 0;
}

template <class T> int Tv1dLevelSets(const DVec& xv, const DVec& yv, double lam) {
  const int64_t n = yv.n;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("tv1d", n);
  const int64_t byte_tile = static_cast<int64_t>(kBlock) * kByteItems;
  const int64_t npad = (n + byte_tile - 1) / byte_tile * byte_tile;  // multiple of both tile sizes
  const int64_t nft = npad / kFwdTile;                               // forward tiles
  auto bP = rt.Alloc((static_cast<size_t>(npad) + 4) * sizeof(double));
  auto bst = rt.Alloc(npad), bst2 = rt.Alloc(npad);
  auto btau = rt.Alloc(static_cast<size_t>(n) * sizeof(double));
  auto bfin = rt.Alloc(n);
  auto bth = rt.Alloc(static_cast<size_t>(nft) * sizeof(int32_t));
  auto btl = rt.Alloc(static_cast<size_t>(nft) * sizeof(int32_t));
  auto bcount = rt.Alloc(sizeof(unsigned long long));
  double* Pp = static_cast<double*>(bP->p) + 1;  // Pp + 1 is 16-byte aligned
  EPS_HIP(hipMemsetAsync(Pp, 0, sizeof(double), s));
  const bool aligned = reinterpret_cast<uintptr_t>(yv.data()) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(xv.data()) % 16 == 0;
  PrefixOp<T> pop{yv.as<T>(), Pp, n, aligned};
  RunTileScan(pop, n);

  TvState<T> st;
  st.y = yv.as<T>();
  st.x = xv.as<T>();
  st.Pp = Pp;
  st.st = static_cast<uint8_t*>(bst->p);
  st.st2 = static_cast<uint8_t*>(bst2->p);
  st.tau_of = static_cast<double*>(btau->p);
  st.fin_of = static_cast<uint8_t*>(bfin->p);
  st.tile_head = static_cast<int32_t*>(bth->p);
  int32_t* tile_l_in = static_cast<int32_t*>(btl->p);
  st.tile_l_in = tile_l_in;
  st.cuts = static_cast<unsigned long long*>(bcount->p);
  st.lam = lam;
  st.n = n;
  st.aligned = aligned;
  EPS_HIP(hipMemsetAsync(st.st, 0, npad, s));
  EPS_HIP(hipMemsetAsync(st.st2, 0, npad, s));
  EPS_HIP(hipMemsetAsync(st.tile_head, 0xff, static_cast<size_t>(nft) * sizeof(int32_t), s));
  hipLaunchKernelGGL(TvInitKernel<T>, dim3(1), dim3(1), 0, s, st, st.tile_head);

  auto head_prescan = [&] {
    EPS_HIP(hipMemcpyAsync(tile_l_in, st.tile_head, static_cast<size_t>(nft) * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, s));
    RunAggScan<MaxAlg>(nft, tile_l_in);
  };
  int level = 0;
  for (;;) {
    ++level;
    EPS_CHECK_MSG(level < 60000, "tv1d: level-set recursion did not terminate");
    head_prescan();
    RunTileScan(ClipOp<T, false>{st}, n);
    RunTileScan(DecodeOp<T>{st}, n);
    EPS_HIP(hipMemsetAsync(st.cuts, 0, sizeof(unsigned long long), s));
    EPS_HIP(hipMemsetAsync(st.tile_head, 0xff, static_cast<size_t>(nft) * sizeof(int32_t), s));
    RunTileScan(BoundOp<T>{st}, n);
    std::swap(st.st, st.st2);
    unsigned long long h = 0;
    EPS_HIP(hipMemcpyAsync(&h, st.cuts, sizeof(h), hipMemcpyDeviceToHost, s));
    EPS_HIP(hipStreamSynchronize(s));
    if (h == 0) break;
  }
  // every remaining region carries "finished" now: write their x
  head_prescan();
  hipLaunchKernelGGL(TvFlushKernel<T>, dim3(static_cast<unsigned>(nft)), dim3(kBlock), 0, s,
                     ClipOp<T, true>{st});
  EPS_HIP(hipGetLastError());
  return level;
}

}  // namespace

// The round-2 binary recursion (one threshold per region and level), kept selectable
// (EPSILON_HIP_TV=binary) as the A/B partner of the three-threshold form in kernels_tv3.hip.
int Tv1dBinary(const DVec& x, const DVec& v, double lam) {
  if (x.dt == F32) return Tv1dLevelSets<float>(x, v, lam);
  return Tv1dLevelSets<double>(x, v, lam);
}

}  // namespace k
}  // namespace eps

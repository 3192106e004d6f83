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
// Region means come from one fp64 prefix sum of y, so a level costs a few streaming passes and
// the number of levels is ~log2(#constant pieces of the solution).
//
// All decisions are taken in fp64 (data may be f32), so the partition matches the fp64 DP except
// at exact ties.
#include <hip/hip_runtime.h>

#include <cmath>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;
constexpr int kItems = 8;
constexpr int kTile = kBlock * kItems;

// ---- generic 3-phase scan --------------------------------------------------------------------
// Op: struct with  using S;  __device__ S load(int64 i);  static S combine(S acc, S next);
//     static S identity();  __device__ void store(int64 i, S inclusive);
// REV scans from the last element to the first.

template <class S, class Op> __device__ inline S BlockExclusive(S mine, S* lds, S* total) {
  // inclusive Hillis-Steele over the 256 thread aggregates, in scan order = thread order
  const int t = threadIdx.x;
  lds[t] = mine;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    S v = lds[t];
    if (t >= off) v = Op::combine(lds[t - off], v);
    __syncthreads();
    lds[t] = v;
    __syncthreads();
  }
  S excl = t == 0 ? Op::identity() : lds[t - 1];
  *total = lds[kBlock - 1];
  __syncthreads();
  return excl;
}

template <class Op, bool REV>
__global__ __launch_bounds__(kBlock) void ScanReduceKernel(Op op, int64_t n,
                                                           typename Op::S* agg) {
  using S = typename Op::S;
  __shared__ S lds[kBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + static_cast<int64_t>(threadIdx.x) * kItems;
  S acc = Op::identity();
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t pos = base + k;  // position in scan order
    if (pos < n) acc = Op::combine(acc, op.load(REV ? n - 1 - pos : pos));
  }
  S total;
  BlockExclusive<S, Op>(acc, lds, &total);
  if (threadIdx.x == 0) agg[blockIdx.x] = total;
}

// exclusive scan of the block aggregates, in place, by one workgroup: every thread owns a
// contiguous chunk (sequential reduce, ONE workgroup scan of the 256 chunk totals, sequential
// write-back), so the number of barriers does not grow with the number of blocks
template <class Op>
__global__ __launch_bounds__(kBlock) void ScanAggKernel(int64_t nb, typename Op::S* agg) {
  using S = typename Op::S;
  __shared__ S lds[kBlock];
  const int64_t chunk = (nb + kBlock - 1) / kBlock;
  const int64_t b0 = static_cast<int64_t>(threadIdx.x) * chunk;
  int64_t b1 = b0 + chunk;
  if (b1 > nb) b1 = nb;
  S acc = Op::identity();
  for (int64_t i = b0; i < b1; ++i) acc = Op::combine(acc, agg[i]);
  S total;
  S run = BlockExclusive<S, Op>(acc, lds, &total);
  for (int64_t i = b0; i < b1; ++i) {
    const S v = agg[i];
    agg[i] = run;
    run = Op::combine(run, v);
  }
}

// Two-level form of the same step for many blocks (n = 1e8: 48 828 aggregates, which the single
// workgroup above walks in 256 serial chains of 191 dependent loads, ~200 us per scan and 6 ms of
// a 50 ms prox): a tile of aggregates per workgroup, the tile totals scanned by ScanAggKernel.
template <class Op>
__global__ __launch_bounds__(kBlock) void AggReduceKernel(int64_t nb, const typename Op::S* agg,
                                                          typename Op::S* agg2) {
  using S = typename Op::S;
  __shared__ S lds[kBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + static_cast<int64_t>(threadIdx.x) * kItems;
  S acc = Op::identity();
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t pos = base + k;
    if (pos < nb) acc = Op::combine(acc, agg[pos]);
  }
  S total;
  BlockExclusive<S, Op>(acc, lds, &total);
  if (threadIdx.x == 0) agg2[blockIdx.x] = total;
}

template <class Op>
__global__ __launch_bounds__(kBlock) void AggApplyKernel(int64_t nb, typename Op::S* agg,
                                                         const typename Op::S* agg2) {
  using S = typename Op::S;
  __shared__ S lds[kBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + static_cast<int64_t>(threadIdx.x) * kItems;
  S item[kItems];
  S acc = Op::identity();
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t pos = base + k;
    item[k] = pos < nb ? agg[pos] : Op::identity();
    acc = Op::combine(acc, item[k]);
  }
  S total;
  S excl = BlockExclusive<S, Op>(acc, lds, &total);
  S run = Op::combine(agg2[blockIdx.x], excl);
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t pos = base + k;
    if (pos < nb) {
      agg[pos] = run;  // exclusive prefix of block `pos`
      run = Op::combine(run, item[k]);
    }
  }
}

template <class Op, bool REV>
__global__ __launch_bounds__(kBlock) void ScanApplyKernel(Op op, int64_t n,
                                                          const typename Op::S* agg) {
  using S = typename Op::S;
  __shared__ S lds[kBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + static_cast<int64_t>(threadIdx.x) * kItems;
  S item[kItems];
  S acc = Op::identity();
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t pos = base + k;
    item[k] = pos < n ? op.load(REV ? n - 1 - pos : pos) : Op::identity();
    acc = Op::combine(acc, item[k]);
  }
  S total;
  S excl = BlockExclusive<S, Op>(acc, lds, &total);
  S run = Op::combine(agg[blockIdx.x], excl);
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t pos = base + k;
    if (pos < n) {
      run = Op::combine(run, item[k]);
      op.store(REV ? n - 1 - pos : pos, run);
    }
  }
}

template <class Op, bool REV> void RunScan(const Op& op, int64_t n) {
  if (n <= 0) return;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  const int64_t nb = (n + kTile - 1) / kTile;
  auto aggbuf = rt.Alloc(static_cast<size_t>(nb) * sizeof(typename Op::S));
  auto* agg = static_cast<typename Op::S*>(aggbuf->p);
  hipLaunchKernelGGL((ScanReduceKernel<Op, REV>), dim3(nb), dim3(kBlock), 0, s, op, n, agg);
  if (nb > 2 * kTile) {
    const int64_t nb2 = (nb + kTile - 1) / kTile;
    auto agg2buf = rt.Alloc(static_cast<size_t>(nb2) * sizeof(typename Op::S));
    auto* agg2 = static_cast<typename Op::S*>(agg2buf->p);
    hipLaunchKernelGGL((AggReduceKernel<Op>), dim3(nb2), dim3(kBlock), 0, s, nb, agg, agg2);
    hipLaunchKernelGGL((ScanAggKernel<Op>), dim3(1), dim3(kBlock), 0, s, nb2, agg2);
    hipLaunchKernelGGL((AggApplyKernel<Op>), dim3(nb2), dim3(kBlock), 0, s, nb, agg, agg2);
  } else {
    hipLaunchKernelGGL((ScanAggKernel<Op>), dim3(1), dim3(kBlock), 0, s, nb, agg);
  }
  hipLaunchKernelGGL((ScanApplyKernel<Op, REV>), dim3(nb), dim3(kBlock), 0, s, op, n, agg);
}

// ---- scan operators ---------------------------------------------------------------------------

template <class T> struct PrefixSumOp {  // P[i+1] = sum_{k<=i} y_k  (fp64)
  using S = double;
  const T* y;
  double* P;  // n + 1 entries, P[0] written by the launcher
  __device__ S load(int64_t i) const { return static_cast<double>(y[i]); }
  __device__ static S combine(S a, S b) { return a + b; }
  __device__ static S identity() { return 0.0; }
  __device__ void store(int64_t i, S v) const { P[i + 1] = v; }
};

struct ClipMap {
  double p, lo, hi;
};

// State of the divide-and-conquer, one entry per sample.
template <class T> struct TvState {
  const T* y;
  const double* P;
  int32_t* L;        // region start of sample i
  int32_t* R;        // region end of sample i
  int8_t* cl;        // at region starts: +1 neighbour below, -1 above, 0 none
  int8_t* cr;        // at region ends
  uint8_t* done;     // region finished (x written)
  uint8_t* s;        // forward classification: 0 force-0, 1 force-1, 2 copy from the right
  uint8_t* u;        // binary labelling of this level
  double lam;
  int64_t n;

  __device__ double Tau(int32_t l, int32_t r) const {
    const double tot = P[r + 1] - P[l] - lam * (static_cast<double>(cl[l]) + static_cast<double>(cr[r]));
    return tot / static_cast<double>(r - l + 1);
  }
  __device__ double Cost(int64_t i, int32_t l, int32_t r) const {  // a_i = tau - y'_i
    double yp = static_cast<double>(y[i]);
    if (i == l) yp -= lam * static_cast<double>(cl[l]);
    if (i == r) yp -= lam * static_cast<double>(cr[r]);
    return Tau(l, r) - yp;
  }
};

template <class T> struct ClipScanOp {  // forward: d_i, stores the classification s_i
  using S = ClipMap;
  TvState<T> st;
  __device__ S load(int64_t i) const {
    if (st.done[i]) return S{0.0, 0.0, 0.0};  // constant map: finished regions are inert
    const int32_t l = st.L[i], r = st.R[i];
    const double a = st.Cost(i, l, r);
    if (i == l) return S{0.0, a, a};  // region head: d_l = a_l, history cut
    return S{a, a - st.lam, a + st.lam};
  }
  __device__ static S combine(S f, S g) {  // g after f
    S o;
    o.p = f.p + g.p;
    o.lo = fmin(fmax(f.lo + g.p, g.lo), g.hi);
    o.hi = fmin(fmax(f.hi + g.p, g.lo), g.hi);
    return o;
  }
  __device__ static S identity() { return S{0.0, -INFINITY, INFINITY}; }
  __device__ void store(int64_t i, S m) const {
    if (st.done[i]) return;
    const double d = fmin(fmax(m.p, m.lo), m.hi);  // the composed map applied to 0
    uint8_t cls;
    if (i == st.R[i]) cls = d < 0.0 ? 1 : 0;  // region end: definite
    else if (d < -st.lam) cls = 1;
    else if (d >= st.lam) cls = 0;
    else cls = 2;
    st.s[i] = cls;
  }
};

template <class T> struct DecodeScanOp {  // backward: u_i = first definite class to the right
  using S = uint8_t;
  TvState<T> st;
  __device__ S load(int64_t i) const { return st.done[i] ? uint8_t(0) : st.s[i]; }
  __device__ static S combine(S acc, S next) { return next != 2 ? next : acc; }
  __device__ static S identity() { return 2; }
  __device__ void store(int64_t i, S v) const { st.u[i] = v; }
};

struct HeadScanOp {  // forward max of head positions -> new L
  using S = int32_t;
  const uint8_t* head;
  const uint8_t* done;
  int32_t* L;
  __device__ S load(int64_t i) const { return head[i] ? static_cast<int32_t>(i) : -1; }
  __device__ static S combine(S a, S b) { return a > b ? a : b; }
  __device__ static S identity() { return -1; }
  __device__ void store(int64_t i, S v) const {
    if (!done[i]) L[i] = v;
  }
};

struct EndScanOp {  // backward min of end positions -> new R
  using S = int32_t;
  const uint8_t* end;
  const uint8_t* done;
  int32_t* R;
  __device__ S load(int64_t i) const { return end[i] ? static_cast<int32_t>(i) : 0x7fffffff; }
  __device__ static S combine(S a, S b) { return a < b ? a : b; }
  __device__ static S identity() { return 0x7fffffff; }
  __device__ void store(int64_t i, S v) const {
    if (!done[i]) R[i] = v;
  }
};

// ---- elementwise passes ------------------------------------------------------------------------

template <class T>
__global__ __launch_bounds__(kBlock) void TvInitKernel(TvState<T> st) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
  if (i >= st.n) return;
  st.L[i] = 0;
  st.R[i] = static_cast<int32_t>(st.n - 1);
  st.cl[i] = 0;
  st.cr[i] = 0;
  st.done[i] = 0;
}

// New boundaries of this level: where u changes inside a region.  split[l] = level marks the
// region as not constant.
template <class T>
__global__ __launch_bounds__(kBlock) void TvSplitKernel(TvState<T> st, uint8_t* head,
                                                        uint8_t* end, int8_t* cl2, int8_t* cr2,
                                                        uint16_t* split, uint16_t level,
                                                        unsigned long long* cuts) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
  if (i >= st.n) return;
  if (st.done[i]) {
    head[i] = 0;
    end[i] = 0;
    return;
  }
  const int32_t l = st.L[i], r = st.R[i];
  const uint8_t ui = st.u[i];
  const bool old_head = i == l, old_end = i == r;
  const bool cut_left = !old_head && st.u[i - 1] != ui;
  const bool cut_right = !old_end && st.u[i + 1] != ui;
  head[i] = (old_head || cut_left) ? 1 : 0;
  end[i] = (old_end || cut_right) ? 1 : 0;
  // across a cut the u = 1 side lies strictly above the u = 0 side
  cl2[i] = old_head ? st.cl[i] : (cut_left ? (ui ? int8_t(1) : int8_t(-1)) : int8_t(0));
  cr2[i] = old_end ? st.cr[i] : (cut_right ? (ui ? int8_t(1) : int8_t(-1)) : int8_t(0));
  if (cut_right) {
    split[l] = level;
    atomicAdd(cuts, 1ull);  // few: one per cut point of this level
  }
}

// Regions that did not split are constant: write x = tau and retire them.
template <class T>
__global__ __launch_bounds__(kBlock) void TvFinishKernel(TvState<T> st, T* x,
                                                         const uint16_t* split, uint16_t level) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
  if (i >= st.n || st.done[i]) return;
  const int32_t l = st.L[i], r = st.R[i];
  if (split[l] != level) {
    x[i] = static_cast<T>(st.Tau(l, r));
    st.done[i] = 1;
  }
}

template <class T> int Tv1dLevelSets(const DVec& xv, const DVec& yv, double lam) {
  const int64_t n = yv.n;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("tv1d", n);
  auto alloc = [&](size_t bytes) { return rt.Alloc(bytes); };
  auto bP = alloc((n + 1) * sizeof(double));
  auto bL = alloc(n * sizeof(int32_t)), bR = alloc(n * sizeof(int32_t));
  auto bcl = alloc(n), bcr = alloc(n), bcl2 = alloc(n), bcr2 = alloc(n);
  auto bdone = alloc(n), bs = alloc(n), bu = alloc(n), bhead = alloc(n), bend = alloc(n);
  auto bsplit = alloc(n * sizeof(uint16_t));
  auto bcount = alloc(sizeof(unsigned long long));
  double* P = static_cast<double*>(bP->p);
  EPS_HIP(hipMemsetAsync(P, 0, sizeof(double), s));
  EPS_HIP(hipMemsetAsync(bsplit->p, 0, n * sizeof(uint16_t), s));
  PrefixSumOp<T> ps{yv.as<T>(), P};
  RunScan<PrefixSumOp<T>, false>(ps, n);

  TvState<T> st;
  st.y = yv.as<T>();
  st.P = P;
  st.L = static_cast<int32_t*>(bL->p);
  st.R = static_cast<int32_t*>(bR->p);
  st.cl = static_cast<int8_t*>(bcl->p);
  st.cr = static_cast<int8_t*>(bcr->p);
  st.done = static_cast<uint8_t*>(bdone->p);
  st.s = static_cast<uint8_t*>(bs->p);
  st.u = static_cast<uint8_t*>(bu->p);
  st.lam = lam;
  st.n = n;
  int8_t* cl2 = static_cast<int8_t*>(bcl2->p);
  int8_t* cr2 = static_cast<int8_t*>(bcr2->p);
  uint8_t* head = static_cast<uint8_t*>(bhead->p);
  uint8_t* end = static_cast<uint8_t*>(bend->p);
  uint16_t* split = static_cast<uint16_t*>(bsplit->p);
  auto* remaining = static_cast<unsigned long long*>(bcount->p);
  const unsigned grid = static_cast<unsigned>((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(TvInitKernel<T>, dim3(grid), dim3(kBlock), 0, s, st);

  int level = 0;
  for (;;) {
    ++level;
    EPS_CHECK_MSG(level < 60000, "tv1d: level-set recursion did not terminate");
    ClipScanOp<T> cop{st};
    RunScan<ClipScanOp<T>, false>(cop, n);
    DecodeScanOp<T> dop{st};
    RunScan<DecodeScanOp<T>, true>(dop, n);
    EPS_HIP(hipMemsetAsync(remaining, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(TvSplitKernel<T>, dim3(grid), dim3(kBlock), 0, s, st, head, end, cl2, cr2,
                       split, static_cast<uint16_t>(level), remaining);
    hipLaunchKernelGGL(TvFinishKernel<T>, dim3(grid), dim3(kBlock), 0, s, st, xv.as<T>(), split,
                       static_cast<uint16_t>(level));
    unsigned long long h = 0;
    EPS_HIP(hipMemcpyAsync(&h, remaining, sizeof(h), hipMemcpyDeviceToHost, s));
    EPS_HIP(hipStreamSynchronize(s));
    if (h == 0) break;
    HeadScanOp hop{head, st.done, st.L};
    RunScan<HeadScanOp, false>(hop, n);
    EndScanOp eop{end, st.done, st.R};
    RunScan<EndScanOp, true>(eop, n);
    std::swap(st.cl, cl2);
    std::swap(st.cr, cr2);
  }
  return level;
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
  if (x.dt == F32) g_last_levels = Tv1dLevelSets<float>(x, v, lam);
  else g_last_levels = Tv1dLevelSets<double>(x, v, lam);
}

}  // namespace k
}  // namespace eps

// One-shot peer-write exchange kernels for the column-sharded fused lasso sweep (SURVEY.md 8(e),
// mode E1; the loop being sharded is reference algorithms/prox_admm.cc:141-147).
//
// A sharded sweep has two places where every rank needs something from every other rank:
//   (1) the forward product A v' is a sum over the ranks' column slabs  (m floats, all-reduce);
//   (2) the cached inverse is applied by row slabs                       (m/G floats each, all-gather).
// Both messages are tens of KB, so they are latency-bound; a ring collective pays 2(G-1) hops and
// a kernel launch of its own.  Here each exchange rides INSIDE the kernel that produces the data:
//
//   PeerReduceExchangeKernel      sums this rank's per-workgroup partials of A v' (fixed order),
//                                 WRITES the m results into slot[rank] of every peer's window
//                                 (one hop over the direct xGMI links), then polls its own window
//                                 until all G slots carry this phase's tag and adds them in rank
//                                 order (identical bits on every rank), + the constant rhs.
//   PeerSlabApplyExchangeKernel   w_slab = scale * Dinv[:, slab]^T p (the cached inverse is
//                                 symmetric, so a row slab is read as contiguous columns), pushes
//                                 the slab to every peer and gathers the other slabs into w.
//
// An entry of a window is an 8-byte {tag, value} granule written by ONE system-scope store, so
// the data is its own flag: no fence between payload and flag, no ordering assumption on the
// fabric (MI355X_MICROARCH.md, Valid forms, R2).  tag = 2*epoch + phase, epoch = a device counter
// the fused pass increments once per sweep - kernel arguments are frozen when the sweep is
// replayed from a hipGraph, the counter is not.  Every (channel, source) slot exists TWICE, indexed by
// the parity of the epoch: a writer reaches epoch t + 2 - the next use of the same buffer - only
// after it has passed its poll of epoch t + 1 on that channel, i.e. after the reader has launched the
// kernel that pushes for epoch t + 1, which stream order puts behind the reader's kernel of epoch t
// that read the buffer.  (With ONE buffer per channel that argument needs a second exchange per
// sweep between two uses; a sweep with the replicated inverse apply - 2 ranks, or
// EPSILON_HIP_SHARDED_APPLY=r - has only the reduce exchange, and a delayed reader could then find
// its granule already overwritten with tag t + 1 and time out.)
//
// Every poll is bounded (kTimeoutTicks of the 100 MHz constant clock): a missing peer ends in an
// error word, never in a hung grid.  Waiting happens only in these two small kernels, never in the
// chip-filling streaming pass, so ranks that share one GPU (tests) cannot starve each other.
#include <hip/hip_runtime.h>

#include "comm.h"
#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int kBlock = 256;
constexpr unsigned long long kTimeoutTicks = 200000000ull;  // 2 s at 100 MHz

typedef unsigned long long u64;

__device__ inline u64 Granule(unsigned tag, unsigned bits) {
  return (static_cast<u64>(tag) << 32) | static_cast<u64>(bits);
}

__device__ inline void PushGranule(u64* dst, u64 g) {
  __hip_atomic_store(dst, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ inline u64 LoadGranule(const u64* src) {
  return __hip_atomic_load(const_cast<u64*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// After the first timed-out poll every later poll gives up at once (a device-side copy of the
// error word sits 64 bytes behind the epoch counter), so a missing peer costs one timeout, not
// one per kernel until the host looks.
__device__ inline unsigned* DevErr(const PeerView& pv) { return pv.epoch + 16; }

__device__ inline bool Failed(const PeerView& pv) {
  return __hip_atomic_load(DevErr(pv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

__device__ inline void ReportTimeout(const PeerView& pv, unsigned code) {
  __hip_atomic_store(DevErr(pv), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(pv.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// address of granule i of (channel, epoch parity, source) in the window of rank `dest`
__device__ inline u64* Slot(const PeerView& pv, int dest, int channel, unsigned epoch, int source,
                            long long i) {
  return pv.win[dest] +
         ((static_cast<long long>(channel) * 2 + (epoch & 1u)) * pv.G + source) * pv.slot + i;
}

__device__ inline bool PollBits(const PeerView& pv, int channel, unsigned epoch, int source,
                                long long i, unsigned tag, unsigned* bits, unsigned code) {
  const u64 t0 = wall_clock64();
  const u64* g = Slot(pv, pv.rank, channel, epoch, source, i);
  for (;;) {
    const u64 x = LoadGranule(g);
    if (static_cast<unsigned>(x >> 32) == tag) {
      *bits = static_cast<unsigned>(x);
      return true;
    }
    if (Failed(pv) || wall_clock64() - t0 > kTimeoutTicks) {
      ReportTimeout(pv, code);
      *bits = 0u;
      return false;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

// One value of the exchanged vectors: an f32 value is ONE granule (index i of its slot), an f64
// value TWO - {tag, high word} at 2 i and {tag, low word} at 2 i + 1, each its own flag - so a
// window slot of S granules carries S floats or S / 2 doubles.
template <class T> struct PeerVal;
template <> struct PeerVal<float> {
  static constexpr int kGran = 1;
  __device__ static void Push(const PeerView& pv, int dest, int channel, unsigned epoch, int source, long long i,
                              unsigned tag, float v) {
    PushGranule(Slot(pv, dest, channel, epoch, source, i), Granule(tag, __float_as_uint(v)));
  }
  __device__ static float Poll(const PeerView& pv, int channel, unsigned epoch, int source, long long i,
                               unsigned tag, unsigned code) {
    unsigned b;
    PollBits(pv, channel, epoch, source, i, tag, &b, code);
    return __uint_as_float(b);
  }
};
template <> struct PeerVal<double> {
  static constexpr int kGran = 2;
  __device__ static void Push(const PeerView& pv, int dest, int channel, unsigned epoch, int source, long long i,
                              unsigned tag, double v) {
    const u64 b = static_cast<u64>(__double_as_longlong(v));
    PushGranule(Slot(pv, dest, channel, epoch, source, 2 * i), Granule(tag, static_cast<unsigned>(b >> 32)));
    PushGranule(Slot(pv, dest, channel, epoch, source, 2 * i + 1), Granule(tag, static_cast<unsigned>(b)));
  }
  __device__ static double Poll(const PeerView& pv, int channel, unsigned epoch, int source, long long i,
                                unsigned tag, unsigned code) {
    unsigned hi, lo;
    PollBits(pv, channel, epoch, source, 2 * i, tag, &hi, code);
    PollBits(pv, channel, epoch, source, 2 * i + 1, tag, &lo, code);
    return __longlong_as_double(static_cast<long long>((static_cast<u64>(hi) << 32) | lo));
  }
};

// four consecutive entries of a vector: one 16-byte load in f32, two in f64
template <class T> struct Quad {
  T x, y, z, w;
};
__device__ inline Quad<float> LoadQuad(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return Quad<float>{v.x, v.y, v.z, v.w};
}
__device__ inline Quad<double> LoadQuad(const double* p) {
  const double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
  return Quad<double>{a.x, a.y, b.x, b.y};
}
template <class T> __device__ inline Quad<T> ZeroQuad() { return Quad<T>{T(0), T(0), T(0), T(0)}; }

__global__ void PeerBumpEpochKernel(PeerView pv) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *pv.epoch += 1u;
}

// y[r] = (sum_q  alpha * sum_k partial_q[k*rows + r])  + add[r],   q over ranks in order.
// A workgroup owns 32 rows: thread (rq, pl) = (t & 7, t >> 3) reads the float4 of rows 4 rq .. 4 rq + 3
// of the partial vectors k = pl, pl + 32, ... (16-byte loads, all of a thread's loads independent:
// 512 partials = 16 loads in flight per thread, 313 workgroups at m = 1e4 - the 20 MB of partials
// of an 8-way sharded sweep have to come in at chip rate, the exchange behind them is the
// latency that matters).  Summation order is fixed: k ascending per thread, then the 32 part
// lanes by three shuffle steps inside a wave and the 4 waves in order.
constexpr int kRQ = 8;                 // row quads per workgroup (32 rows)
constexpr int kPL = kBlock / kRQ;      // 32 part lanes

template <class T>
__global__ __launch_bounds__(kBlock) void PeerReduceExchangeKernel(
    PeerView pv, long long rows, int nparts, const T* __restrict__ partial, T alpha,
    const T* __restrict__ add, T* __restrict__ y) {
  __shared__ Quad<T> part[kBlock / 64][kRQ];
  const int t = threadIdx.x, rq = t & (kRQ - 1), pl = t >> 3, wave = t >> 6;
  const long long r0 = (static_cast<long long>(blockIdx.x) * kRQ + rq) * 4;  // rows % 4 == 0
  Quad<T> s = ZeroQuad<T>();
  if (r0 < rows) {
    const T* p = partial + r0;
    int k = pl;
    for (; k + 7 * kPL < nparts; k += 8 * kPL) {
      Quad<T> v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = LoadQuad(p + static_cast<long long>(k + u * kPL) * rows);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s.x += v[u].x;
        s.y += v[u].y;
        s.z += v[u].z;
        s.w += v[u].w;
      }
    }
    for (; k < nparts; k += kPL) {
      const Quad<T> v = LoadQuad(p + static_cast<long long>(k) * rows);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  }
  // the 8 part lanes of a wave sit 8 lanes apart
#pragma unroll
  for (int off = 32; off >= 8; off >>= 1) {
    s.x += __shfl_down(s.x, off, 64);
    s.y += __shfl_down(s.y, off, 64);
    s.z += __shfl_down(s.z, off, 64);
    s.w += __shfl_down(s.w, off, 64);
  }
  if ((t & 63) < kRQ) part[wave][rq] = s;
  __syncthreads();
  // The exchange is spread over the whole workgroup: thread (row, q) = (t & 31, t >> 5) pushes ONE
  // value to peer q and polls ONE value of source q - one or two stores and loads per lane and
  // round instead of 32 of each on 8 lanes (the kernel is latency-bound, not bandwidth-bound).
  static_assert(kRQ * 4 * PeerView::kMaxPeers == kBlock, "one (row, peer) pair per thread");
  __shared__ T mine_s[kRQ * 4];
  __shared__ T got[PeerView::kMaxPeers][kRQ * 4];
  const long long rbase = static_cast<long long>(blockIdx.x) * kRQ * 4;
  if (t < kRQ * 4) {
    const Quad<T> a = part[0][t >> 2], b = part[1][t >> 2], c = part[2][t >> 2], d = part[3][t >> 2];
    const int e = t & 3;
    const T av = e == 0 ? a.x : e == 1 ? a.y : e == 2 ? a.z : a.w;
    const T bv = e == 0 ? b.x : e == 1 ? b.y : e == 2 ? b.z : b.w;
    const T cv = e == 0 ? c.x : e == 1 ? c.y : e == 2 ? c.z : c.w;
    const T dv = e == 0 ? d.x : e == 1 ? d.y : e == 2 ? d.z : d.w;
    mine_s[t] = alpha * (((av + bv) + cv) + dv);
  }
  __syncthreads();
  const unsigned epoch = *pv.epoch;
  const unsigned tag = 2u * epoch + 1u;
  {
    const int row = t & (kRQ * 4 - 1), q = t >> 5;  // kRQ * 4 == 32
    const long long r = rbase + row;
    if (q < pv.G && r < rows) {
      PeerVal<T>::Push(pv, q, 0, epoch, pv.rehearse ? q : pv.rank, r, tag, mine_s[row]);
      got[q][row] = PeerVal<T>::Poll(pv, 0, epoch, q, r, tag, 1u);
    }
  }
  __syncthreads();
  if (t < kRQ * 4 && rbase + t < rows) {
    T acc = got[0][t];
    for (int q = 1; q < pv.G; ++q) acc += got[q][t];
    if (add) acc += add[rbase + t];
    y[rbase + t] = acc;
  }
}

// w[q*slab + j] for all q: this rank computes j < slab of its own slab
//   w_own[j] = scale * D[:, lo + j] . p      (0 for lo + j >= m, the padding of the last slab)
// CP columns per workgroup pass, the CP dot products reduced by wave shuffles and across the 4
// waves in a fixed order (GemvT2Kernel's scheme); the lanes that hold the results push them.  Afterwards every thread gathers a share of the G*slab granules of
// w from the local window into the plain vector the streaming pass reads.
template <int CP, class T>
__global__ __launch_bounds__(kBlock) void PeerSlabApplyExchangeKernel(
    PeerView pv, long long m, long long slab, long long lo, const T* __restrict__ D,
    long long ldd, T scale, const T* __restrict__ p, T* __restrict__ wpad) {
  // (p is read straight from global memory: a thread only ever needs the entries of its own row
  // chunks, nothing is shared between threads, and a workgroup takes one or two passes - staging
  // the m values in LDS first cost a third of the kernel's traffic)
  __shared__ T red[kBlock / 64][CP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned epoch = *pv.epoch;
  const unsigned tag = 2u * epoch + 2u;
  const long long npass = (slab + CP - 1) / CP;
  const long long nvec = m / 4;
  for (long long pass = blockIdx.x; pass < npass; pass += gridDim.x) {
    const long long j0 = pass * CP;
    T acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = T(0);
    // RU row chunks x CP columns = 8 independent loads of four entries per thread and step (f32;
    // half as many chunks in f64, where a chunk is two 16-byte loads)
    constexpr int RU = (sizeof(T) == 4 ? 8 : 4) / CP;
    static_assert(RU >= 1, "columns per pass");
    bool live[CP];
    const T* colp[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const long long col = lo + j0 + c;
      live[c] = j0 + c < slab && col < m;
      colp[c] = D + (live[c] ? col : 0) * ldd;
    }
    for (long long q0 = threadIdx.x; q0 < nvec; q0 += kBlock * RU) {
      Quad<T> a[RU][CP], xv[RU];
#pragma unroll
      for (int r = 0; r < RU; ++r) {
        const long long q = q0 + static_cast<long long>(r) * kBlock;
        const bool in = q < nvec;
#pragma unroll
        for (int c = 0; c < CP; ++c) a[r][c] = (in && live[c]) ? LoadQuad(colp[c] + q * 4) : ZeroQuad<T>();
        xv[r] = in ? LoadQuad(p + q * 4) : ZeroQuad<T>();
      }
#pragma unroll
      for (int r = 0; r < RU; ++r) {
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          acc[c] += a[r][c].x * xv[r].x;
          acc[c] += a[r][c].y * xv[r].y;
          acc[c] += a[r][c].z * xv[r].z;
          acc[c] += a[r][c].w * xv[r].w;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      T v = acc[c];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wave][c] = v;
    }
    __syncthreads();
    // thread (c, q) pushes column c's result to peer q: one value per lane
    if (threadIdx.x < CP * PeerView::kMaxPeers) {
      const int c = threadIdx.x % CP, q = threadIdx.x / CP;
      if (j0 + c < slab && q < pv.G) {
        const T t = scale * (((red[0][c] + red[1][c]) + red[2][c]) + red[3][c]);
        PeerVal<T>::Push(pv, q, 1, epoch, pv.rehearse ? q : pv.rank, j0 + c, tag, t);
        // the own slab goes straight into w: no workgroup of this grid ever waits for another
        // workgroup of the same grid, only for other GPUs
        if (q == 0 && !pv.rehearse) wpad[static_cast<long long>(pv.rank) * slab + j0 + c] = t;
      }
    }
    __syncthreads();
  }
  // gather the other ranks' slabs: value g = q*slab + j of channel 1
  const long long total = static_cast<long long>(pv.G) * slab;
  for (long long g = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; g < total;
       g += static_cast<long long>(gridDim.x) * kBlock) {
    const int q = static_cast<int>(g / slab);
    if (q == pv.rank && !pv.rehearse) continue;
    const long long j = g - q * slab;
    wpad[g] = PeerVal<T>::Poll(pv, 1, epoch, q, j, tag, 2u);
  }
}

}  // namespace

void PeerBumpEpoch(const PeerView& pv) {
  hipLaunchKernelGGL(PeerBumpEpochKernel, dim3(1), dim3(64), 0, Runtime::Get().stream(), pv);
  EPS_HIP(hipGetLastError());
}

// granules per value: a window slot of pv.slot granules carries pv.slot floats or half as many doubles
static int64_t GranulesPerValue(DType dt) { return dt == F64 ? 2 : 1; }

void PeerReduceExchange(const PeerView& pv, int64_t rows, int nparts, const DVec& partial,
                        double alpha, const DVec* add, const DVec& y) {
  const DType dt = partial.dt;
  EPS_CHECK(y.dt == dt && y.n == rows);
  EPS_CHECK(partial.n >= static_cast<int64_t>(nparts) * rows && nparts >= 1);
  EPS_CHECK_MSG(rows * GranulesPerValue(dt) <= pv.slot,
                "peer exchange: " << rows << " rows exceed the window slot of " << pv.slot << " granules");
  if (add) EPS_CHECK(add->n == rows && add->dt == dt);
  if (rows == 0) return;
  ProfScope prof("peer_reduce_exchange", rows, nparts);
  EPS_CHECK_MSG(rows % 4 == 0 && reinterpret_cast<uintptr_t>(partial.data()) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(y.data()) % 16 == 0,
                "peer exchange: rows must be a multiple of 4 and the buffers 16-byte aligned");
  const unsigned grid = static_cast<unsigned>((rows / 4 + kRQ - 1) / kRQ);
  if (dt == F32)
    hipLaunchKernelGGL(PeerReduceExchangeKernel<float>, dim3(grid), dim3(kBlock), 0, Runtime::Get().stream(), pv,
                       static_cast<long long>(rows), nparts, partial.as<float>(), static_cast<float>(alpha),
                       add ? add->as<float>() : nullptr, y.as<float>());
  else
    hipLaunchKernelGGL(PeerReduceExchangeKernel<double>, dim3(grid), dim3(kBlock), 0, Runtime::Get().stream(), pv,
                       static_cast<long long>(rows), nparts, partial.as<double>(), alpha,
                       add ? add->as<double>() : nullptr, y.as<double>());
  EPS_HIP(hipGetLastError());
}

bool PeerSlabApplySupported(const PeerView& pv, int64_t m, int64_t slab, const DVec& D, int64_t ldd) {
  return m % 4 == 0 && ldd % 4 == 0 && slab * GranulesPerValue(D.dt) <= pv.slot &&
         reinterpret_cast<uintptr_t>(D.data()) % 16 == 0;
}

void PeerSlabApplyExchange(const PeerView& pv, int64_t m, int64_t slab, int64_t lo, const DVec& D,
                           int64_t ldd, double scale, const DVec& p, const DVec& wpad) {
  EPS_CHECK(PeerSlabApplySupported(pv, m, slab, D, ldd));
  const DType dt = D.dt;
  EPS_CHECK(p.n == m && p.dt == dt && wpad.dt == dt && wpad.n >= slab * pv.G);
  EPS_CHECK(D.n >= (m - 1) * ldd + m && lo >= 0);
  EPS_CHECK(reinterpret_cast<uintptr_t>(p.data()) % 16 == 0);
  ProfScope prof("peer_slab_apply_exchange", m, slab);
  hipStream_t s = Runtime::Get().stream();
  // enough workgroups to fill the chip twice over: 4 columns per pass while that gives >= 512
  // passes, 2 below (at m = 1e4 and 8 ranks: 626 passes of 2 columns)
  const bool wide = slab / 4 >= 512;
  const int64_t cp = wide ? 4 : 2;
  int64_t grid = (slab + cp - 1) / cp;
  // at most four workgroups (16 waves of the 32) per CU: the polling tail must never fill the
  // chip (ranks that share one GPU in the tests need room to run beside each other)
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  const dim3 g(static_cast<unsigned>(grid)), b(kBlock);
  const long long mm = m, sl = slab, l0 = lo, ld = ldd;
  if (dt == F32) {
    if (wide)
      hipLaunchKernelGGL((PeerSlabApplyExchangeKernel<4, float>), g, b, 0, s, pv, mm, sl, l0, D.as<float>(), ld,
                         static_cast<float>(scale), p.as<float>(), wpad.as<float>());
    else
      hipLaunchKernelGGL((PeerSlabApplyExchangeKernel<2, float>), g, b, 0, s, pv, mm, sl, l0, D.as<float>(), ld,
                         static_cast<float>(scale), p.as<float>(), wpad.as<float>());
  } else {
    if (wide)
      hipLaunchKernelGGL((PeerSlabApplyExchangeKernel<4, double>), g, b, 0, s, pv, mm, sl, l0, D.as<double>(), ld,
                         scale, p.as<double>(), wpad.as<double>());
    else
      hipLaunchKernelGGL((PeerSlabApplyExchangeKernel<2, double>), g, b, 0, s, pv, mm, sl, l0, D.as<double>(), ld,
                         scale, p.as<double>(), wpad.as<double>());
  }
  EPS_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace eps

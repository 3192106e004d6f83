#include "device.h"

#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "comm.h"
#include "kernels.h"

namespace eps {

Buffer::~Buffer() {
  if (owned && p) Runtime::Get().Release(p, bytes);
}

Runtime& Runtime::Get() {
  static Runtime* rt = new Runtime();  // leaked on purpose: outlives static destructors
  return *rt;
}

Runtime::Runtime() {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    EPS_FATAL("no HIP device available: the epsilon_hip solver has no CPU fallback ("
              << hipGetErrorString(e) << ")");
  }
  const char* env = std::getenv("EPSILON_HIP_DEVICE");
  if (env == nullptr) env = std::getenv("LOCAL_RANK");
  device_ = env ? std::atoi(env) % count : 0;
  EPS_HIP(hipSetDevice(device_));
  EPS_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  EPS_HIP(hipMalloc(reinterpret_cast<void**>(&slots_dev_), 2 * kMaxSlots * sizeof(double)));
  EPS_HIP(hipHostMalloc(reinterpret_cast<void**>(&slots_host_), 3 * kMaxSlots * sizeof(double),
                        hipHostMallocDefault));
  EPS_HIP(hipMemsetAsync(slots_dev_, 0, 2 * kMaxSlots * sizeof(double), stream_));
  std::memset(slots_host_, 0, 3 * kMaxSlots * sizeof(double));
}

Runtime::~Runtime() {}

static size_t RoundUp(size_t bytes) {
  const size_t g = 256;
  return bytes == 0 ? g : (bytes + g - 1) / g * g;
}

std::shared_ptr<Buffer> Runtime::Alloc(size_t bytes) {
  bytes = RoundUp(bytes);
  auto b = std::make_shared<Buffer>();
  b->bytes = bytes;
  auto it = pool_.find(bytes);
  auto hit = holding_ ? hold_pool_.find(bytes) : hold_pool_.end();
  if (hit != hold_pool_.end()) {
    b->p = hit->second;
    hold_pool_.erase(hit);
    pooled_ -= bytes;
  } else if (it != pool_.end()) {
    b->p = it->second;
    pool_.erase(it);
    pooled_ -= bytes;
  } else {
    EPS_HIP(hipSetDevice(device_));
    hipError_t e = hipMalloc(&b->p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      TrimPool();
      EPS_HIP(hipMalloc(&b->p, bytes));
    }
  }
  in_use_ += bytes;
  return b;
}

void Runtime::Release(void* p, size_t bytes) {
  // Single-stream ordering makes immediate reuse safe: any kernel that touched `p` was
  // enqueued before whatever the next owner enqueues.
  (holding_ ? hold_pool_ : pool_).emplace(bytes, p);
  in_use_ -= bytes;
  pooled_ += bytes;
}

void Runtime::BeginHold() {
  EPS_CHECK_MSG(!holding_, "nested buffer holds");
  holding_ = true;
}

std::vector<std::pair<size_t, void*>> Runtime::EndHold() {
  holding_ = false;
  std::vector<std::pair<size_t, void*>> held(hold_pool_.begin(), hold_pool_.end());
  hold_pool_.clear();
  for (const auto& kv : held) pooled_ -= kv.first;  // neither pooled nor in use: the graph's
  return held;
}

void Runtime::ReturnHeld(std::vector<std::pair<size_t, void*>>* held) {
  for (const auto& kv : *held) {
    pool_.emplace(kv.first, kv.second);
    pooled_ += kv.first;
  }
  held->clear();
}

void Runtime::TrimPool() {
  EPS_HIP(hipStreamSynchronize(stream_));
  for (auto& kv : pool_) {
    (void)hipFree(kv.second);
    pooled_ -= kv.first;
  }
  pool_.clear();
}

void* Runtime::Scratch(size_t bytes) {
  if (bytes > scratch_bytes_) {
    if (scratch_) {
      EPS_HIP(hipStreamSynchronize(stream_));
      EPS_HIP(hipFree(scratch_));
    }
    scratch_bytes_ = RoundUp(bytes < (1u << 20) ? (1u << 20) : bytes);
    EPS_HIP(hipMalloc(&scratch_, scratch_bytes_));
  }
  return scratch_;
}

int Runtime::NewSlot() {
  EPS_CHECK_MSG(slots_used_ < kMaxSlots, "out of reduction slots");
  return slots_used_++;
}

void Runtime::ResetSlots() {
  if (slots_used_ > 0) {
    EPS_HIP(hipMemsetAsync(slots_dev_, 0, slots_used_ * sizeof(double), stream_));
    EPS_HIP(hipMemsetAsync(slots_dev_ + kMaxSlots, 0, slots_used_ * sizeof(double), stream_));
  }
  slots_used_ = 0;
}

void Runtime::FetchSlots() {
  FetchSlotsAsync();
  WaitSlots();
}

void Runtime::WaitSlots() {
  if (slots_event_) EPS_HIP(hipEventSynchronize(slots_event_));
  else EPS_HIP(hipStreamSynchronize(stream_));
}

void Runtime::FetchSlotsAsync() {
  if (slots_used_ > 0) {
    const bool sharded = ShardSpec::Get().active();
    // consensus form: this rank's own share of the sharded half is kept too (per-rank terms)
    if (sharded && ShardSpec::Get().consensus_terms())
      EPS_HIP(hipMemcpyAsync(slots_host_ + 2 * kMaxSlots, slots_dev_ + kMaxSlots,
                             slots_used_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    if (sharded) comm_->AllReduceSum(slots_dev_ + kMaxSlots, slots_used_, F64);
    EPS_HIP(hipMemcpyAsync(slots_host_, slots_dev_, slots_used_ * sizeof(double),
                           hipMemcpyDeviceToHost, stream_));
    if (sharded)
      EPS_HIP(hipMemcpyAsync(slots_host_ + kMaxSlots, slots_dev_ + kMaxSlots,
                             slots_used_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
  }
  if (!slots_event_) EPS_HIP(hipEventCreateWithFlags(&slots_event_, hipEventDisableTiming));
  EPS_HIP(hipEventRecord(slots_event_, stream_));
}

void Runtime::Sync() { EPS_HIP(hipStreamSynchronize(stream_)); }

size_t Runtime::ProfBegin(const std::string& tag) {
  ProfPending e;
  e.tag = tag;
  for (hipEvent_t* ev : {&e.a, &e.b}) {
    if (!prof_free_.empty()) {
      *ev = prof_free_.back();
      prof_free_.pop_back();
    } else {
      EPS_HIP(hipEventCreate(ev));
    }
  }
  EPS_HIP(hipEventRecord(e.a, stream_));
  prof_pending_.push_back(e);
  ++prof_open_;
  return prof_pending_.size() - 1;
}

void Runtime::ProfEnd(size_t index) {
  EPS_HIP(hipEventRecord(prof_pending_[index].b, stream_));
  --prof_open_;
  if (prof_open_ == 0 && prof_pending_.size() >= 8192) ProfCollect();
}

void Runtime::ProfCollect() {
  if (prof_pending_.empty()) return;
  EPS_HIP(hipStreamSynchronize(stream_));
  for (auto& e : prof_pending_) {
    float ms = 0;
    EPS_HIP(hipEventElapsedTime(&ms, e.a, e.b));
    ProfTotal& t = prof_totals_[e.tag];
    t.count += 1;
    t.ms += ms;
    prof_free_.push_back(e.a);
    prof_free_.push_back(e.b);
  }
  prof_pending_.clear();
}

void Runtime::ProfReset() {
  ProfCollect();
  prof_totals_.clear();
}

bool Runtime::ProfWanted(const char* name) const {
  if (prof_filter_.empty()) return true;
  size_t pos = 0;
  const std::string n(name);
  while (pos <= prof_filter_.size()) {
    size_t end = prof_filter_.find(',', pos);
    if (end == std::string::npos) end = prof_filter_.size();
    if (end > pos && n.compare(0, end - pos, prof_filter_, pos, end - pos) == 0) return true;
    pos = end + 1;
  }
  return false;
}

ProfScope::ProfScope(const char* name, int64_t a, int64_t b) {
  Runtime& rt = Runtime::Get();
  on = false;
  if (!rt.profiling() || rt.capturing()) return;
  std::string tag = name;
  if (a >= 0) tag += ":" + std::to_string(a);
  if (b >= 0) tag += "x" + std::to_string(b);
  on = rt.ProfWanted(tag.c_str());  // filter prefixes are matched against the full tag
  if (!on) return;
  index = rt.ProfBegin(tag);
}

ProfScope::~ProfScope() {
  if (!on) return;
  try {
    Runtime::Get().ProfEnd(index);
  } catch (...) {  // never throw out of a destructor (may run during unwinding)
  }
}

static thread_local DType g_current_dtype = F32;
DType CurrentDType() { return g_current_dtype; }
void SetCurrentDType(DType dt) { g_current_dtype = dt; }

// ---- DVec -------------------------------------------------------------------------------------

DVec DVec::Empty(int64_t n, DType dt) {
  EPS_CHECK(n >= 0);
  DVec v;
  v.n = n;
  v.dt = dt;
  v.buf = Runtime::Get().Alloc(static_cast<size_t>(n) * DTypeSize(dt));
  return v;
}

DVec DVec::Zeros(int64_t n, DType dt) {
  DVec v = Empty(n, dt);
  if (n > 0) EPS_HIP(hipMemsetAsync(v.data(), 0, v.bytes(), Runtime::Get().stream()));
  return v;
}

DVec DVec::Full(int64_t n, double val, DType dt) {
  DVec v = Empty(n, dt);
  k::Fill(v, val);
  return v;
}

namespace {
// Large host blobs (the data matrices the frontend ships as float64, constant.py:12-17): a plain
// hipMemcpy from pageable memory is staged by the runtime on ONE host thread (~6 GB/s measured:
// 0.33 s for the 1.9 GB of the 60000 x 4000 feature matrix), and in fp32 mode it moves twice the
// bytes the device keeps.  Here the host threads convert (or copy) the source in stripes into two
// pinned 32 MB buffers that the DMA engine drains alternately: the conversion of one chunk runs beside
// the transfer of the other and fp32 data crosses the link as fp32.
struct PinnedStage {
  void* buf[2] = {nullptr, nullptr};
  hipEvent_t drained[2] = {nullptr, nullptr};
  static constexpr size_t kBytes = size_t(32) << 20;
};

PinnedStage* GetPinnedStage() {
  static PinnedStage* st = [] {
    auto* p = new PinnedStage();  // lives as long as the process
    for (int i = 0; i < 2; ++i) {
      if (hipHostMalloc(&p->buf[i], PinnedStage::kBytes, hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&p->drained[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        delete p;
        return static_cast<PinnedStage*>(nullptr);
      }
    }
    return p;
  }();
  return st;
}

int HostThreads() {
  static const int n = [] {
    const char* e = std::getenv("EPSILON_HIP_HOST_THREADS");
    int v = e ? std::atoi(e) : 0;
    if (v <= 0) {
      v = static_cast<int>(std::thread::hardware_concurrency());
      if (v > 16) v = 16;  // (a container's CPU quota is usually far below the host's core count)
    }
    return v < 1 ? 1 : v;
  }();
  return n;
}

template <class D> void ParallelConvert(D* dst, const double* src, int64_t len) {
  const int T = static_cast<int>(std::min<int64_t>(HostThreads(), (len + (1 << 18) - 1) >> 18));
  auto work = [&](int t) {
    const int64_t lo = len * t / T, hi = len * (t + 1) / T;
    if constexpr (std::is_same<D, double>::value) {
      std::memcpy(dst + lo, src + lo, static_cast<size_t>(hi - lo) * sizeof(double));
    } else {
      for (int64_t i = lo; i < hi; ++i) dst[i] = static_cast<D>(src[i]);
    }
  };
  if (T <= 1) {
    work(0);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(T - 1);
  for (int t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}

// false: the pinned buffers are not available (the caller takes the plain route)
template <class D> bool UploadThroughPinned(D* dev, const double* src, int64_t n, hipStream_t s) {
  const auto tp = std::chrono::steady_clock::now();
  PinnedStage* st = GetPinnedStage();
  if (st == nullptr) return false;
  const double pin_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count();
  const int64_t chunk = static_cast<int64_t>(PinnedStage::kBytes / sizeof(D));
  int which = 0;
  bool used[2] = {false, false};
  static const bool trace = [] {
    const char* e = std::getenv("EPSILON_HIP_INIT_TRACE");
    return e && e[0] == '2';
  }();
  const auto t0 = std::chrono::steady_clock::now();
  double conv_ms = 0;
  for (int64_t off = 0; off < n; off += chunk, which ^= 1) {
    const int64_t len = std::min(chunk, n - off);
    if (used[which]) EPS_HIP(hipEventSynchronize(st->drained[which]));
    const auto c0 = std::chrono::steady_clock::now();
    ParallelConvert(static_cast<D*>(st->buf[which]), src + off, len);
    conv_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
    EPS_HIP(hipMemcpyAsync(dev + off, st->buf[which], static_cast<size_t>(len) * sizeof(D),
                           hipMemcpyHostToDevice, s));
    EPS_HIP(hipEventRecord(st->drained[which], s));
    used[which] = true;
  }
  EPS_HIP(hipStreamSynchronize(s));
  if (trace)
    std::fprintf(stderr, "[host] pinned upload %lld elements: pinned buffers ready after %.2f ms; %.2f ms (host convert / copy %.2f ms, %d threads)\n",
                 static_cast<long long>(n), pin_ms,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), conv_ms,
                 HostThreads());
  return true;
}
}  // namespace

int HostThreadCount() { return HostThreads(); }

std::shared_ptr<char> AllocHostBuffer(size_t bytes) {
  if (bytes >= (size_t(64) << 20)) {
    void* mem = nullptr;
    const size_t two_mb = size_t(2) << 20;
    EPS_CHECK_MSG(posix_memalign(&mem, two_mb, (bytes + two_mb - 1) & ~(two_mb - 1)) == 0, "host buffer: out of memory");
    (void)madvise(mem, bytes, MADV_HUGEPAGE);
    return std::shared_ptr<char>(static_cast<char*>(mem), [](char* q) { std::free(q); });
  }
  return std::shared_ptr<char>(new char[bytes > 0 ? bytes : 1], std::default_delete<char[]>());
}

void ParallelHostCopy(void* dst, const void* src, size_t bytes) {
  const int T = static_cast<int>(std::min<size_t>(static_cast<size_t>(HostThreads()), (bytes + (size_t(8) << 20) - 1) >> 23));
  if (T <= 1) {
    std::memcpy(dst, src, bytes);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(static_cast<size_t>(T));
  char* d = static_cast<char*>(dst);
  const char* sp = static_cast<const char*>(src);
  for (int t = 0; t < T; ++t)
    th.emplace_back([=] {  // stripes on 4 KB boundaries
      const size_t lo = (bytes / T * t) & ~size_t(4095), hi = t + 1 == T ? bytes : (bytes / T * (t + 1)) & ~size_t(4095);
      std::memcpy(d + lo, sp + lo, hi - lo);
    });
  for (auto& x : th) x.join();
}

namespace {
// dst (f64, pageable) <- n device entries of type S through the two pinned buffers: the DMA of
// chunk i runs while the host threads widen / copy chunk i - 1.  false: no pinned buffers.
template <class S> bool DownloadThroughPinned(double* dst, const S* dev, int64_t n, hipStream_t s) {
  PinnedStage* st = GetPinnedStage();
  if (st == nullptr) return false;
  const int64_t chunk = static_cast<int64_t>(PinnedStage::kBytes / sizeof(S));
  auto widen = [&](int64_t off, int64_t len, int which) {
    const S* src = static_cast<const S*>(st->buf[which]);
    double* out = dst + off;
    const int T = static_cast<int>(std::min<int64_t>(HostThreads(), (len + (1 << 18) - 1) >> 18));
    auto work = [=](int t) {
      const int64_t lo = len * t / T, hi = len * (t + 1) / T;
      if constexpr (std::is_same<S, double>::value) {
        std::memcpy(out + lo, src + lo, static_cast<size_t>(hi - lo) * sizeof(double));
      } else {
        for (int64_t i = lo; i < hi; ++i) out[i] = static_cast<double>(src[i]);
      }
    };
    if (T <= 1) {
      work(0);
      return;
    }
    std::vector<std::thread> th;
    th.reserve(static_cast<size_t>(T - 1));
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  };
  int which = 0;
  int64_t prev_off = 0, prev_len = 0;
  for (int64_t off = 0; off < n; off += chunk, which ^= 1) {
    const int64_t len = std::min(chunk, n - off);
    EPS_HIP(hipMemcpyAsync(st->buf[which], dev + off, static_cast<size_t>(len) * sizeof(S), hipMemcpyDeviceToHost, s));
    EPS_HIP(hipEventRecord(st->drained[which], s));
    if (prev_len > 0) {
      EPS_HIP(hipEventSynchronize(st->drained[which ^ 1]));
      widen(prev_off, prev_len, which ^ 1);
    }
    prev_off = off;
    prev_len = len;
  }
  if (prev_len > 0) {
    EPS_HIP(hipEventSynchronize(st->drained[which ^ 1]));
    widen(prev_off, prev_len, which ^ 1);
  }
  return true;
}
}  // namespace

DVec DVec::FromHost(const double* src, int64_t n, DType dt) {
  DVec v = Empty(n, dt);
  if (n == 0) return v;
  Runtime& rt = Runtime::Get();
  // 64 MB of doubles and more (pinning the two buffers costs 10-70 ms once per process: a 60 MB
  // matrix is quicker through the plain copy)
  static const bool pinned_on = [] {
    const char* e = std::getenv("EPSILON_HIP_PINNED_UPLOAD");
    return !(e && e[0] == '0');
  }();
  if (pinned_on && n >= (int64_t(1) << 23)) {
    const bool ok = dt == F64 ? UploadThroughPinned(v.as<double>(), src, n, rt.stream())
                              : UploadThroughPinned(v.as<float>(), src, n, rt.stream());
    if (ok) return v;
  }
  if (dt == F64) {
    EPS_HIP(hipMemcpyAsync(v.data(), src, n * sizeof(double), hipMemcpyHostToDevice,
                           rt.stream()));
    // pageable host memory: the copy is staged, the source may be reused after return only
    // once the stream has consumed it
    EPS_HIP(hipStreamSynchronize(rt.stream()));
  } else {
    // stage in chunks of <= 256 MiB of doubles, convert on the device
    const int64_t chunk = int64_t(1) << 25;
    auto stage = rt.Alloc(static_cast<size_t>(std::min(n, chunk)) * sizeof(double));
    for (int64_t off = 0; off < n; off += chunk) {
      int64_t len = std::min(chunk, n - off);
      EPS_HIP(hipMemcpyAsync(stage->p, src + off, len * sizeof(double),
                             hipMemcpyHostToDevice, rt.stream()));
      k::ConvertFromF64(v.Slice(off, len), static_cast<const double*>(stage->p));
      EPS_HIP(hipStreamSynchronize(rt.stream()));
    }
  }
  return v;
}

DVec DVec::Borrow(void* dev_ptr, int64_t n, DType dt) {
  DVec v;
  v.n = n;
  v.dt = dt;
  v.buf = std::make_shared<Buffer>();
  v.buf->p = dev_ptr;
  v.buf->bytes = static_cast<size_t>(n) * DTypeSize(dt);
  v.buf->owned = false;
  return v;
}

DVec DVec::Slice(int64_t start, int64_t len) const {
  EPS_CHECK(start >= 0 && len >= 0 && start + len <= n);
  DVec v = *this;
  v.offset = offset + static_cast<size_t>(start) * DTypeSize(dt);
  v.n = len;
  return v;
}

DVec DVec::Clone() const {
  DVec v = Empty(n, dt);
  if (n > 0) k::Copy(v, *this);
  return v;
}

void DVec::ToHost(double* dst) const {
  if (n == 0) return;
  Runtime& rt = Runtime::Get();
  // Large vectors (the 10^8-entry iterates of configs 3 and 5): in their own type over the link
  // into pinned buffers, widened to the boundary's float64 by the host threads meanwhile - a
  // pageable copy of converted doubles moved twice the bytes at a fifth of the rate.
  static const bool pinned_ok = [] {
    const char* e = std::getenv("EPSILON_HIP_PINNED_UPLOAD");
    return !(e && e[0] == '0');
  }();
  if (pinned_ok && n >= (int64_t(1) << 22)) {
    const bool done = dt == F64 ? DownloadThroughPinned(dst, as<double>(), n, rt.stream())
                                : DownloadThroughPinned(dst, as<float>(), n, rt.stream());
    if (done) return;
  }
  if (dt == F64) {
    EPS_HIP(hipMemcpyAsync(dst, data(), n * sizeof(double), hipMemcpyDeviceToHost,
                           rt.stream()));
    EPS_HIP(hipStreamSynchronize(rt.stream()));
    return;
  }
  const int64_t chunk = int64_t(1) << 25;
  auto stage = rt.Alloc(static_cast<size_t>(std::min(n, chunk)) * sizeof(double));
  for (int64_t off = 0; off < n; off += chunk) {
    int64_t len = std::min(chunk, n - off);
    k::ConvertToF64(static_cast<double*>(stage->p), Slice(off, len));
    EPS_HIP(hipMemcpyAsync(dst + off, stage->p, len * sizeof(double), hipMemcpyDeviceToHost,
                           rt.stream()));
    EPS_HIP(hipStreamSynchronize(rt.stream()));
  }
}

std::vector<double> DVec::ToHost() const {
  std::vector<double> out(n);
  ToHost(out.data());
  return out;
}

HostArray DVec::ToHostArray() const {
  HostArray a;
  a.n = static_cast<size_t>(n);
  a.mem = AllocHostBuffer(a.n * sizeof(double));
  ToHost(a.data());
  return a;
}

}  // namespace eps

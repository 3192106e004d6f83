// Device runtime: one HIP stream, a size-binned buffer pool, device scalar slots for
// deferred reductions, and DVec (a typed view of pooled HBM).
//
// The reference keeps every vector as an Eigen::VectorXd and allocates on each BlockVector
// operation (reference src/epsilon/vector/block_vector.cc:9-48).  Here all iterate state
// lives in HBM; temporaries come from a pool so that after the first sweep no hipMalloc
// happens, and norms are reduced into device-side "slots" that are read back with one
// copy per residual check instead of one sync per norm.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace eps {

class Runtime;

struct Buffer {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;  // false: borrowed device pointer (caller-resident blob)
  ~Buffer();
};

class Comm;          // collective backend (comm.h)
class PeerExchange;  // one-shot peer-write window (comm.h)

class Runtime {
 public:
  static Runtime& Get();

  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }

  std::shared_ptr<Buffer> Alloc(size_t bytes);
  void Release(void* p, size_t bytes);
  void TrimPool();
  // A "hold" fences the buffers a captured hipGraph touches off from everybody else: while it is
  // active every released buffer goes to a private pool (which allocations inside the hold may
  // re-use); EndHold() hands that pool to the caller, who keeps it for as long as the graph may
  // be replayed and gives it back with ReturnHeld().
  void BeginHold();
  std::vector<std::pair<size_t, void*>> EndHold();
  void ReturnHeld(std::vector<std::pair<size_t, void*>>* held);
  bool holding() const { return holding_; }
  size_t bytes_in_use() const { return in_use_; }
  size_t bytes_pooled() const { return pooled_; }

  // Reduction scratch (per-workgroup partials), at least `bytes` large.
  void* Scratch(size_t bytes);

  // Device scalar slots (doubles).  ResetSlots() rewinds the allocator; FetchSlots()
  // copies all used slots to the host with ONE async copy + stream sync.
  // Each slot has a replicated part and a sharded part (partial sum over this rank's slice of
  // sharded blocks); FetchSlots all-reduces the sharded parts in ONE collective.
  int NewSlot();
  double* SlotPtr(int i) { return slots_dev_ + i; }
  double* ShardSlotPtr(int i) { return slots_dev_ + kMaxSlots + i; }
  void ResetSlots();  // rewinds and zeroes (on the stream)
  void FetchSlots();
  // The same in two halves: FetchSlotsAsync enqueues the (all-reduce and) copy and records an
  // event, WaitSlots blocks the host until that copy has landed - work enqueued in between runs
  // on the device while the host waits (pipelined residual checks, admm.cc).
  void FetchSlotsAsync();
  void WaitSlots();
  double SlotValue(int i) const { return slots_host_[i] + slots_host_[kMaxSlots + i]; }
  // before the sum over ranks (valid when ShardSpec::consensus_terms() is set)
  double SlotLocalValue(int i) const { return slots_host_[i] + slots_host_[2 * kMaxSlots + i]; }

  void Sync();

  // ---- live kernel timing (HIP events on this stream; off unless eps_profile_enable) -------
  bool profiling() const { return profiling_; }
  void set_profiling(bool on) { profiling_ = on; }
  // only tags starting with one of these comma-separated prefixes are timed (empty: all)
  void set_prof_filter(const std::string& f) { prof_filter_ = f; }
  bool ProfWanted(const char* name) const;
  size_t ProfBegin(const std::string& tag);  // returns the entry index for ProfEnd
  void ProfEnd(size_t index);
  void ProfCollect();  // synchronises, folds finished event pairs into the totals
  void ProfReset();
  struct ProfTotal { int64_t count = 0; double ms = 0; };
  const std::map<std::string, ProfTotal>& prof_totals() const { return prof_totals_; }

  Comm* comm() { return comm_; }
  void set_comm(Comm* c) { comm_ = c; }
  PeerExchange* peer() { return peer_; }
  void set_peer(PeerExchange* p) { peer_ = p; }
  // true while the stream is being captured into a hipGraph (no events, no synchronisation)
  bool capturing() const { return capturing_; }
  void set_capturing(bool on) { capturing_ = on; }

  static constexpr int kMaxSlots = 4096;

 private:
  Runtime();
  ~Runtime();
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  std::multimap<size_t, void*> pool_;
  std::multimap<size_t, void*> hold_pool_;
  bool holding_ = false;
  size_t in_use_ = 0, pooled_ = 0;
  void* scratch_ = nullptr;
  size_t scratch_bytes_ = 0;
  double* slots_dev_ = nullptr;
  double* slots_host_ = nullptr;
  int slots_used_ = 0;
  hipEvent_t slots_event_ = nullptr;
  Comm* comm_ = nullptr;
  PeerExchange* peer_ = nullptr;
  bool capturing_ = false;
  bool profiling_ = false;
  std::string prof_filter_;
  struct ProfPending { std::string tag; hipEvent_t a, b; };
  std::vector<ProfPending> prof_pending_;
  int prof_open_ = 0;
  std::vector<hipEvent_t> prof_free_;
  std::map<std::string, ProfTotal> prof_totals_;
};

// RAII scope around one kernel launch (or a short launch sequence) for live timing.
struct ProfScope {
  bool on;
  size_t index = 0;
  explicit ProfScope(const char* name, int64_t a = -1, int64_t b = -1);
  ~ProfScope();
};

// Compute dtype of the solve being set up on this thread (f32 unless EPSILON_HIP_DTYPE=f64 or
// eps_set_option("dtype", ...)); maps without data of their own (scalars) take it from here.
DType CurrentDType();
void SetCurrentDType(DType dt);
// host threads for large host-side copies / conversions (EPSILON_HIP_HOST_THREADS; at most 16)
int HostThreadCount();
// `bytes` of host memory that is NOT value-initialised; from 64 MB on 2 MB-aligned with a
// transparent-huge-page hint (the page faults, not the bytes, bound the first write of a fresh
// buffer of several hundred MB).
std::shared_ptr<char> AllocHostBuffer(size_t bytes);
// dst[lo, hi) <- src, split over the host threads in stripes (large copies only)
void ParallelHostCopy(void* dst, const void* src, size_t bytes);

// Typed device vector view.  Copying a DVec shares the buffer (like shared_ptr).

// float64 values on the host in a buffer from AllocHostBuffer (results at the boundary)
struct HostArray {
  std::shared_ptr<char> mem;
  size_t n = 0;
  double* data() { return reinterpret_cast<double*>(mem.get()); }
  const double* data() const { return reinterpret_cast<const double*>(mem.get()); }
  size_t size() const { return n; }
};

struct DVec {
  std::shared_ptr<Buffer> buf;
  size_t offset = 0;  // bytes
  int64_t n = 0;
  DType dt = F32;

  bool defined() const { return buf != nullptr || n == 0; }
  void* data() const { return buf ? static_cast<char*>(buf->p) + offset : nullptr; }
  template <class T> T* as() const { return static_cast<T*>(data()); }
  size_t bytes() const { return static_cast<size_t>(n) * DTypeSize(dt); }

  static DVec Empty(int64_t n, DType dt);
  static DVec Zeros(int64_t n, DType dt);
  static DVec Full(int64_t n, double v, DType dt);
  // Host double -> device dt (converted on the device).
  static DVec FromHost(const double* src, int64_t n, DType dt);
  // Borrowed device memory (no copy, not freed).
  static DVec Borrow(void* dev_ptr, int64_t n, DType dt);
  DVec Slice(int64_t start, int64_t len) const;
  DVec Clone() const;
  std::vector<double> ToHost() const;  // synchronises
  void ToHost(double* dst) const;
  HostArray ToHostArray() const;  // the same into a buffer that is not value-initialised first
};

}  // namespace eps

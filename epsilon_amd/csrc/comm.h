// Collective backend of the sharded solve (mode E1).
//
// The reference has no distributed mode at all.  Here one process drives one
// GPU; a solve can be sharded over the ranks of a node: some block keys (variables, elementwise
// constraint rows) are *sharded* - each rank holds a slice - the others are *replicated*.
// Every map between two sharded keys must be elementwise (scalar / diagonal); a dense map from
// a sharded key into a replicated key is a contraction over the sharded dimension and yields a
// partial sum that is all-reduced.  For the headline lasso this is one all-reduce of m floats
// per sweep (the forward substitution A u) plus one of a handful of doubles per residual check.
//
// Backends: RCCL over xGMI (dlopen'ed, so single-GPU runs never touch it) and a host callback
// (staged through pinned memory) used by the multi-process tests that share one GPU or run
// the collective through gloo.
#pragma once

#include <map>
#include <set>
#include <string>
#include <vector>

#include "device.h"

namespace eps {

class Comm {
 public:
  virtual ~Comm() {}
  virtual int rank() const = 0;
  virtual int size() const = 0;
  // In-place sum over ranks, ordered on the runtime stream.
  virtual void AllReduceSum(void* dev_ptr, size_t count, DType dt) = 0;
  void AllReduceSum(const DVec& v) { AllReduceSum(v.data(), static_cast<size_t>(v.n), v.dt); }
  // recv[r*count .. (r+1)*count) = rank r's send[0..count); recv may not alias send.
  virtual void AllGather(const void* send_dev, void* recv_dev, size_t count, DType dt) = 0;
  // Maximum over ranks of one host value (synchronises).  Built on AllReduceSum of a one-hot
  // vector, so every backend - including the sum-only host callback - has it.
  double AllReduceMaxHost(double local);
};

typedef void (*HostAllReduceFn)(void* host_buf, size_t count, int dtype, void* ctx);

// RCCL: id = 128-byte ncclUniqueId from GetRcclUniqueId on rank 0.
void GetRcclUniqueId(void* out128);
Comm* NewRcclComm(int rank, int size, const void* id128);
Comm* NewHostCallbackComm(int rank, int size, HostAllReduceFn fn, void* ctx);

// ---- one-shot peer-write exchange over xGMI (last paragraph) ---------------------
//
// The per-sweep messages of the sharded lasso are m floats (40 KB at config 2): latency-bound.
// A ring all-reduce pays 2(G-1) dependent hops; the 8 GPUs of a node are fully connected, so
// every rank can instead WRITE its contribution straight into a slot of every peer's window
// (one hop), and every rank sums the G slots it received in rank order - the same bits on every
// rank.  The window is device memory shared through HIP IPC handles; an entry is an 8-byte
// {tag, value} granule written by ONE store, so the data is its own flag (no fence between a
// payload and a flag store, MI355X_MICROARCH.md "Valid forms" R2) and a reader polls until the
// tag equals the phase it waits for.  All polling is bounded: on a timeout the kernel sets an
// error word and finishes, the host raises at the next residual check.
//
// Kernels that use the window are in kernels_peer.hip; RCCL stays in charge of the large
// messages (the m x m Gram all-reduce at Init) and of the residual scalars.
struct PeerView {  // passed to kernels by value
  static constexpr int kMaxPeers = 8;
  unsigned long long* win[kMaxPeers];  // win[q]: rank q's window as mapped into THIS process
  int G = 1, rank = 0;
  // rehearsal (one process plays one rank of G, every window is its own): a push to "peer q"
  // lands in source slot q of the local window, so that every slot a poll waits for is filled
  int rehearse = 0;
  long long slot = 0;                   // granules per (channel, source rank) slot
  unsigned* epoch = nullptr;            // device counter, one increment per sweep
  unsigned* err = nullptr;              // host-mapped: != 0 after a timed-out poll
};

class PeerExchange {
 public:
  static constexpr int kChannels = 2;
  // Collective over `comm` (handles travel through its AllGather; every step's outcome is
  // agreed across the ranks, so either all of them get a window or none does).  Returns nullptr
  // and the reason in *why when the window cannot be set up or its self test fails.
  static PeerExchange* Create(Comm* comm, int64_t slot_floats, int rehearse_ranks, std::string* why);
  ~PeerExchange();
  const PeerView& view() const { return view_; }
  int64_t slot() const { return view_.slot; }
  // Collective: push a known pattern through both channels and check what arrives (throws).
  void SelfTest();
  // Throws if a kernel reported a timed-out poll since the last call (stream must be drained).
  void CheckError();
  // The device-side copy of the error word (read by the residual-norm kernel, so that a failure
  // on any rank reaches every rank through the check's all-reduce); ClearError resets both.
  const unsigned* device_error_word() const { return view_.epoch + 16; }
  void ClearError();

 private:
  PeerExchange() {}
  PeerView view_;
  void* local_ = nullptr;
  size_t bytes_ = 0;
  std::vector<void*> opened_;
  unsigned* err_host_ = nullptr;
};

// Which block keys are sharded in the solve being set up / run on this process.
class ShardSpec {
 public:
  static ShardSpec& Get();
  void Clear() {
    keys_.clear();
    consensus_terms_ = false;
  }
  void Add(const std::string& key) { keys_.insert(key); }
  bool active() const;                       // a communicator with size > 1 is installed
  bool IsSharded(const std::string& key) const {
    return keys_.count(key) != 0 || local_.count(key) != 0;
  }
  const std::set<std::string>& keys() const { return keys_; }
  // Consensus form (SURVEY.md 8(e) mode E2): an objective term all of whose variables are
  // sharded is a DIFFERENT term on every rank (f_g(x_g), its argument rows private to the rank)
  // instead of this rank's slice of one global term.
  void set_consensus_terms(bool on) { consensus_terms_ = on; }
  bool consensus_terms() const { return consensus_terms_; }
  // Keys whose meaning is local to one prox operator ("arg:<k>" rows of its H): set around that
  // operator's Init / Apply by LocalShardScope.
  void set_local(std::set<std::string> local) { local_ = std::move(local); }
  const std::set<std::string>& local() const { return local_; }
  // Global (summed over ranks) size of a sharded key, so that size-driven decisions (the fill
  // model of the block elimination) are identical on every rank.  0 = unknown.
  void set_global_dim(const std::string& key, int64_t d) { global_dim_[key] = d; }
  int64_t global_dim(const std::string& key) const {
    auto it = global_dim_.find(key);
    return it == global_dim_.end() ? 0 : it->second;
  }

 private:
  std::set<std::string> keys_;
  std::set<std::string> local_;
  bool consensus_terms_ = false;
  std::map<std::string, int64_t> global_dim_;
};

struct LocalShardScope {
  std::set<std::string> saved;
  explicit LocalShardScope(const std::set<std::string>& local) {
    saved = ShardSpec::Get().local();
    ShardSpec::Get().set_local(local);
  }
  ~LocalShardScope() { ShardSpec::Get().set_local(saved); }
};

}  // namespace eps

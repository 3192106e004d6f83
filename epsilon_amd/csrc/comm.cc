#include "comm.h"

#include <dlfcn.h>

#include <cstdlib>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <sstream>
#include <vector>

#include "kernels.h"

namespace eps {

ShardSpec& ShardSpec::Get() {
  static ShardSpec* s = new ShardSpec();
  return *s;
}

double Comm::AllReduceMaxHost(double local) {
  const int G = size();
  std::vector<double> h(static_cast<size_t>(G), 0.0);
  h[static_cast<size_t>(rank())] = local;
  DVec d = DVec::FromHost(h.data(), G, F64);
  AllReduceSum(d);
  h = d.ToHost();
  double mx = h[0];
  for (double v : h) mx = v > mx ? v : mx;
  return mx;
}

bool ShardSpec::active() const {
  Comm* c = Runtime::Get().comm();
  if (c == nullptr) return false;
  // EPSILON_HIP_FORCE_SHARDED=1 runs the sharded code path (and its collectives) on a
  // single-rank communicator: lets a 1-GPU box exercise the RCCL backend end to end.
  static const bool force = std::getenv("EPSILON_HIP_FORCE_SHARDED") != nullptr;
  return c->size() > 1 || force;
}

// ---- RCCL through dlopen -------------------------------------------------------------------------

namespace {

struct Id128 {
  char internal[128];
};

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ Id128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

RcclApi& Api() {
  static RcclApi api;
  if (api.handle) return api;
  const char* env = std::getenv("EPSILON_HIP_RCCL");
  void* h = nullptr;
  if (env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
  // prefer the copy a host framework (torch) already loaded, so one RCCL lives in the process
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  EPS_CHECK_MSG(h != nullptr, "cannot load librccl.so: " << dlerror());
  api.handle = h;
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  EPS_CHECK_MSG(api.GetUniqueId && api.CommInitRank && api.AllReduce && api.AllGather &&
                    api.CommDestroy,
                "librccl.so lacks the expected symbols");
  return api;
}

void CheckNccl(int rc, const char* what) {
  if (rc == 0) return;
  const char* msg = Api().GetErrorString ? Api().GetErrorString(rc) : "?";
  EPS_FATAL("RCCL " << what << " failed: " << msg << " (" << rc << ")");
}

constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclFloat64 = 8;

class RcclComm final : public Comm {
 public:
  RcclComm(int rank, int size, const void* id128) : rank_(rank), size_(size) {
    Id128 id;
    std::memcpy(&id, id128, sizeof(id));
    EPS_HIP(hipSetDevice(Runtime::Get().device()));
    CheckNccl(Api().CommInitRank(&comm_, size, id, rank), "ncclCommInitRank");
  }
  ~RcclComm() override {
    if (comm_) Api().CommDestroy(comm_);
  }
  int rank() const override { return rank_; }
  int size() const override { return size_; }
  void AllReduceSum(void* p, size_t count, DType dt) override {
    if (count == 0) return;
    CheckNccl(Api().AllReduce(p, p, count, dt == F32 ? kNcclFloat32 : kNcclFloat64, kNcclSum,
                              comm_, Runtime::Get().stream()),
              "ncclAllReduce");
  }
  void AllGather(const void* send, void* recv, size_t count, DType dt) override {
    if (count == 0) return;
    CheckNccl(Api().AllGather(send, recv, count, dt == F32 ? kNcclFloat32 : kNcclFloat64, comm_,
                              Runtime::Get().stream()),
              "ncclAllGather");
  }

 private:
  int rank_, size_;
  void* comm_ = nullptr;
};

class HostCallbackComm final : public Comm {
 public:
  HostCallbackComm(int rank, int size, HostAllReduceFn fn, void* ctx)
      : rank_(rank), size_(size), fn_(fn), ctx_(ctx) {}
  int rank() const override { return rank_; }
  int size() const override { return size_; }
  void AllReduceSum(void* p, size_t count, DType dt) override {
    if (count == 0) return;
    const size_t bytes = count * DTypeSize(dt);
    if (host_.size() < bytes) host_.resize(bytes);
    hipStream_t s = Runtime::Get().stream();
    EPS_HIP(hipMemcpyAsync(host_.data(), p, bytes, hipMemcpyDeviceToHost, s));
    EPS_HIP(hipStreamSynchronize(s));
    fn_(host_.data(), count, dt == F32 ? 0 : 1, ctx_);
    EPS_HIP(hipMemcpyAsync(p, host_.data(), bytes, hipMemcpyHostToDevice, s));
    EPS_HIP(hipStreamSynchronize(s));
  }
  void AllGather(const void* send, void* recv, size_t count, DType dt) override {
    // test backend: zero everything but the own slice, then sum
    if (count == 0) return;
    const size_t es = DTypeSize(dt);
    hipStream_t s = Runtime::Get().stream();
    EPS_HIP(hipMemsetAsync(recv, 0, count * es * size_, s));
    EPS_HIP(hipMemcpyAsync(static_cast<char*>(recv) + rank_ * count * es, send, count * es,
                           hipMemcpyDeviceToDevice, s));
    AllReduceSum(recv, count * size_, dt);
  }

 private:
  int rank_, size_;
  HostAllReduceFn fn_;
  void* ctx_;
  std::vector<char> host_;
};

}  // namespace

void GetRcclUniqueId(void* out128) {
  Id128 id;
  std::memset(&id, 0, sizeof(id));
  CheckNccl(Api().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, &id, sizeof(id));
}

Comm* NewRcclComm(int rank, int size, const void* id128) { return new RcclComm(rank, size, id128); }

Comm* NewHostCallbackComm(int rank, int size, HostAllReduceFn fn, void* ctx) {
  return new HostCallbackComm(rank, size, fn, ctx);
}

}  // namespace eps

// ---- PeerExchange ----------------------------------------------------------------------------------

namespace eps {

namespace {

// Window memory must be readable mid-kernel by a GPU other than the one that wrote it: take
// uncached / fine-grained device memory when the runtime offers it (what RCCL does for its own
// peer buffers).  Plain hipMalloc memory is the last resort and only good between ranks that share
// ONE device (the one-GPU rehearsals and tests): polled from another device, a plain coarse-grained
// line is exactly the stale-line case of MI355X_MICROARCH.md ("Inter-workgroup visibility") - the
// caller refuses it there, see PeerExchange::Create.
enum WindowMemory { kWindowUncached = 0, kWindowFineGrained = 1, kWindowPlain = 2 };

void* AllocWindow(size_t bytes, WindowMemory* kind) {
  void* p = nullptr;
  const char* force = std::getenv("EPSILON_HIP_PEER_WINDOW_MEMORY");  // tests: "plain"
  const bool plain_only = force && std::strcmp(force, "plain") == 0;
  if (!plain_only) {
#ifdef hipDeviceMallocUncached
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess && p) {
      *kind = kWindowUncached;
      return p;
    }
    (void)hipGetLastError();
#endif
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess && p) {
      *kind = kWindowFineGrained;
      return p;
    }
    (void)hipGetLastError();
  }
  EPS_HIP(hipMalloc(&p, bytes));
  *kind = kWindowPlain;
  return p;
}

// Identity of the physical device this process drives, as 16 bytes (the PCI bus id string,
// "0000:05:00.0"): two ranks with equal bytes share one GPU.
void DeviceIdentity(int device, unsigned char out[16]) {
  char bus[64] = {0};
  std::memset(out, 0, 16);
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) {
    (void)hipGetLastError();
    std::snprintf(bus, sizeof(bus), "ordinal:%d", device);
  }
  std::memcpy(out, bus, std::min<size_t>(16, std::strlen(bus)));
}

}  // namespace

// true iff `ok` holds on every rank (collective)
static bool AllRanksOk(Comm* comm, bool ok) {
  const double bad = ok ? 0.0 : 1.0;
  DVec d = DVec::FromHost(&bad, 1, F64);
  comm->AllReduceSum(d);
  return d.ToHost()[0] == 0.0;
}

PeerExchange* PeerExchange::Create(Comm* comm, int64_t slot_floats, int rehearse_ranks,
                                   std::string* why) {
  EPS_CHECK(comm != nullptr && slot_floats > 0);
  Runtime& rt = Runtime::Get();
  EPS_HIP(hipSetDevice(rt.device()));
  const bool rehearse = rehearse_ranks > 1;
  if (rehearse) EPS_CHECK_MSG(comm->size() == 1, "rank rehearsal needs a single-rank communicator");
  const int G = rehearse ? rehearse_ranks : comm->size();
  EPS_CHECK_MSG(G <= PeerView::kMaxPeers, "peer exchange supports at most " << PeerView::kMaxPeers
                                                                            << " ranks, got " << G);
  std::unique_ptr<PeerExchange> px(new PeerExchange());
  PeerView& v = px->view_;
  v.G = G;
  v.rank = comm->rank();
  v.rehearse = rehearse ? 1 : 0;
  v.slot = (slot_floats + 63) / 64 * 64;
  for (int q = 0; q < PeerView::kMaxPeers; ++q) v.win[q] = nullptr;
  // per rank: the IPC handle, 16 bytes of device identity, 1 byte of window memory kind
  constexpr int HB = static_cast<int>(sizeof(hipIpcMemHandle_t));
  constexpr int XB = HB + 16 + 1;
  const bool shared = !rehearse && comm->size() > 1;
  std::vector<double> enc(XB, 0.0);
  std::string err;
  WindowMemory mem_kind = kWindowPlain;
  // step 1 (local): window, epoch counter, error word, IPC handle
  try {
    // [channel][epoch parity][source rank][slot] granules, then one 256-byte line for the epoch
    // counter (kernels_peer.hip: why every slot exists twice)
    const size_t granules = static_cast<size_t>(kChannels) * 2 * G * v.slot;
    px->bytes_ = granules * sizeof(unsigned long long) + 256;
    px->local_ = AllocWindow(px->bytes_, &mem_kind);
    EPS_HIP(hipMemset(px->local_, 0, px->bytes_));
    v.epoch = reinterpret_cast<unsigned*>(static_cast<char*>(px->local_) +
                                          granules * sizeof(unsigned long long));
    EPS_HIP(hipHostMalloc(reinterpret_cast<void**>(&px->err_host_), 64, hipHostMallocMapped));
    *px->err_host_ = 0;
    void* err_dev = nullptr;
    EPS_HIP(hipHostGetDevicePointer(&err_dev, px->err_host_, 0));
    v.err = static_cast<unsigned*>(err_dev);
    if (shared) {
      hipIpcMemHandle_t mine;
      EPS_HIP(hipIpcGetMemHandle(&mine, px->local_));
      const unsigned char* mb = reinterpret_cast<const unsigned char*>(&mine);
      for (int i = 0; i < HB; ++i) enc[i] = mb[i];
      unsigned char ident[16];
      DeviceIdentity(rt.device(), ident);
      for (int i = 0; i < 16; ++i) enc[HB + i] = ident[i];
      enc[HB + 16] = static_cast<double>(mem_kind);
    }
  } catch (const std::exception& e) {
    err = e.what();
    (void)hipGetLastError();
  }
  if (!shared) {
    if (!err.empty()) {
      if (why) *why = err;
      return nullptr;
    }
    for (int q = 0; q < G; ++q) v.win[q] = static_cast<unsigned long long*>(px->local_);
  } else {
    // exchange the handles: one float per byte (exact under the sum-with-zeros all-gather of
    // the host-callback backend, which would not preserve arbitrary bit patterns)
    DVec send = DVec::FromHost(enc.data(), XB, F32);
    DVec recv = DVec::Zeros(static_cast<int64_t>(XB) * G, F32);
    comm->AllGather(send.data(), recv.data(), XB, F32);
    rt.Sync();
    if (!AllRanksOk(comm, err.empty())) {
      if (why) *why = err.empty() ? "a peer rank could not create its window" : err;
      return nullptr;
    }
    std::vector<double> all = recv.ToHost();
    // A window in plain (coarse-grained) device memory may only be polled from its own device:
    // every rank sees the same table, so every rank reaches the same verdict without a vote.
    for (int q = 0; q < G && err.empty(); ++q) {
      if (static_cast<int>(all[static_cast<size_t>(q) * XB + HB + 16]) != kWindowPlain) continue;
      for (int r = 0; r < G; ++r) {
        bool same = true;
        for (int i = 0; i < 16; ++i)
          same = same && all[static_cast<size_t>(q) * XB + HB + i] == all[static_cast<size_t>(r) * XB + HB + i];
        if (!same) {
          std::ostringstream os;
          os << "rank " << q << " could only get plain hipMalloc memory for its window and rank " << r
             << " sits on a different device: a plain window polled across devices may serve stale lines";
          err = os.str();
          break;
        }
      }
    }
    if (!err.empty()) {
      if (why) *why = err;
      return nullptr;
    }
    // step 2: map the peers' windows
    try {
      for (int q = 0; q < G; ++q) {
        if (q == v.rank) {
          v.win[q] = static_cast<unsigned long long*>(px->local_);
          continue;
        }
        hipIpcMemHandle_t h;
        unsigned char* hb = reinterpret_cast<unsigned char*>(&h);
        for (int i = 0; i < HB; ++i)
          hb[i] = static_cast<unsigned char>(all[static_cast<size_t>(q) * XB + i]);
        void* p = nullptr;
        EPS_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        px->opened_.push_back(p);
        v.win[q] = static_cast<unsigned long long*>(p);
      }
    } catch (const std::exception& e) {
      err = e.what();
      (void)hipGetLastError();
    }
    if (!AllRanksOk(comm, err.empty())) {
      if (why) *why = err.empty() ? "a peer rank could not map the windows" : err;
      return nullptr;
    }
  }
  // step 3: the exchange kernels on a known pattern, both channels
  try {
    px->SelfTest();
  } catch (const std::exception& e) {
    err = e.what();
    (void)hipGetLastError();
  }
  if (!AllRanksOk(comm, err.empty())) {
    if (why) *why = err.empty() ? "the self test failed on a peer rank" : err;
    return nullptr;
  }
  return px.release();
}

PeerExchange::~PeerExchange() {
  for (void* p : opened_) (void)hipIpcCloseMemHandle(p);
  if (local_) (void)hipFree(local_);
  if (err_host_) (void)hipHostFree(err_host_);
}

void PeerExchange::ClearError() {
  if (err_host_) *err_host_ = 0;
  (void)hipMemset(view_.epoch + 16, 0, sizeof(unsigned));
}

void PeerExchange::CheckError() {
  if (err_host_ && *err_host_ != 0) {
    const unsigned code = *err_host_;
    *err_host_ = 0;
    (void)hipMemset(view_.epoch + 16, 0, sizeof(unsigned));  // the device-side copy
    EPS_FATAL("peer exchange: a poll timed out (code " << code
                                                       << "): a peer rank did not deliver its part");
  }
}

void PeerExchange::SelfTest() {
  // all-reduce of (rank + 1) * (i % 7 + 1) over both code paths of the exchange kernels
  Runtime& rt = Runtime::Get();
  const int G = view_.G;
  const int64_t n = std::min<int64_t>(view_.slot, 1000);
  std::vector<double> h(n);
  const int me = view_.rehearse ? 0 : view_.rank;
  for (int64_t i = 0; i < n; ++i) h[i] = (me + 1.0) * static_cast<double>(i % 7 + 1);
  DVec part = DVec::FromHost(h.data(), n, F32);
  DVec out = DVec::Zeros(n, F32);
  k::PeerBumpEpoch(view_);
  k::PeerReduceExchange(view_, n, 1, part, 1.0, nullptr, out);
  rt.Sync();
  CheckError();
  std::vector<double> got = out.ToHost();
  double scale = 0;
  for (int q = 0; q < G; ++q) scale += view_.rehearse ? 1.0 : q + 1.0;
  for (int64_t i = 0; i < n; ++i)
    EPS_CHECK_MSG(got[i] == scale * static_cast<double>(i % 7 + 1),
                  "peer exchange self test: entry " << i << " is " << got[i] << ", expected "
                                                    << scale * static_cast<double>(i % 7 + 1));
  // channel 1: w = 2 * I p by slabs of 64 rows (the matrix is 2 I, so w = 2 p on every rank;
  // in a rehearsal every slab holds rank 0's rows)
  const int64_t slab = std::min<int64_t>(64, view_.slot), mt = slab * G;
  std::vector<double> Dh(static_cast<size_t>(mt) * mt, 0.0), ph(mt);
  for (int64_t i = 0; i < mt; ++i) {
    Dh[static_cast<size_t>(i) * mt + i] = 2.0;
    ph[i] = static_cast<double>(i % 11) - 5.0;
  }
  DVec D = DVec::FromHost(Dh.data(), mt * mt, F32);
  DVec p = DVec::FromHost(ph.data(), mt, F32);
  DVec w = DVec::Zeros(mt, F32);
  k::PeerBumpEpoch(view_);
  k::PeerSlabApplyExchange(view_, mt, slab, static_cast<int64_t>(me) * slab, D, mt, 1.0, p, w);
  rt.Sync();
  CheckError();
  got = w.ToHost();
  for (int64_t i = 0; i < mt; ++i) {
    const double want = 2.0 * ph[view_.rehearse ? i % slab : i];
    EPS_CHECK_MSG(got[i] == want, "peer exchange self test (slab apply): entry "
                                      << i << " is " << got[i] << ", expected " << want);
  }
}

}  // namespace eps

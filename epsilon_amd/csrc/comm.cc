#include "comm.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <vector>

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

// fmx_comm.cpp -- the path's one exchange BELOW the host-language seam: an RCCL all-gather of the ranks' result
// slices over xGMI, for callers that are not Python: a JVM that drives several GPUs from one process,
// or one process per GPU that has its own way of shipping 128 bytes between ranks.  (The Python mirror keeps using
// torch.distributed, which is RCCL too; findex_amd/distributed.py.)
//
// RCCL is looked up at first use with dlopen, not linked: a process that already holds an RCCL (torch ships one) gets
// that very library, a process that never gathers does not load it at all.
#include <fmx.h>

#include <dlfcn.h>

#include <cstring>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fmx_host.h"

namespace fmx {
namespace {

struct UniqueId { char internal[128]; };        // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef void *Comm;                              // ncclComm_t
constexpr int kNcclChar = 0;                     // ncclInt8 / ncclChar
static_assert(sizeof(UniqueId) == FMX_COMM_ID_BYTES, "unique id size");

struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
  int (*CommInitAll)(Comm *, int, const int *) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string why;
};

Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "?"); return; }
    auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p && r.why.empty()) r.why = std::string("RCCL lacks ") + n; return p; };
    r.GetUniqueId = reinterpret_cast<int (*)(UniqueId *)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<int (*)(Comm *, int, UniqueId, int)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<int (*)(Comm *, int, const int *)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<int (*)(Comm)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, Comm, hipStream_t)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<const char *(*)(int)>(sym("ncclGetErrorString"));
  });
  return &r;
}

int rccl_ready() {
  Rccl *r = rccl();
  if (!r->lib || !r->why.empty()) { set_error(r->why.empty() ? "RCCL unavailable" : r->why); return FMX_ERR_UNSUPPORTED; }
  return FMX_OK;
}

int nccl_fail(int rc, const char *what) {
  Rccl *r = rccl();
  set_error(std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
  return FMX_ERR_HIP;
}

struct CommSet {
  int n_ranks = 0;                     // ranks of the communicator
  std::vector<Comm> comm;              // this process's ranks (one per local device)
  std::vector<int> device;
  std::vector<hipStream_t> stream;     // one non-blocking stream per local rank for the collective
  ~CommSet() {
    Rccl *r = rccl();
    for (size_t i = 0; i < comm.size(); i++) {
      (void)hipSetDevice(device[i]);
      if (stream[i]) { (void)hipStreamSynchronize(stream[i]); (void)hipStreamDestroy(stream[i]); }
      if (comm[i] && r->CommDestroy) (void)r->CommDestroy(comm[i]);
    }
  }
};

#define HIP_TRY(call, what)                            \
  do {                                                 \
    hipError_t e__ = (call);                           \
    if (e__ != hipSuccess) return hip_fail(e__, what); \
  } while (0)

int add_streams(CommSet *c) {
  c->stream.assign(c->comm.size(), nullptr);
  for (size_t i = 0; i < c->comm.size(); i++) {
    HIP_TRY(hipSetDevice(c->device[i]), "hipSetDevice");
    HIP_TRY(hipStreamCreateWithFlags(&c->stream[i], hipStreamNonBlocking), "hipStreamCreate");
  }
  return FMX_OK;
}

}  // namespace
}  // namespace fmx

using namespace fmx;

extern "C" {

int fmx_comm_unique_id(void *id) {
  if (!id) { set_error("null argument"); return FMX_ERR_ARG; }
  int rc = rccl_ready();
  if (rc) return rc;
  const int e = rccl()->GetUniqueId(static_cast<UniqueId *>(id));
  return e ? nccl_fail(e, "ncclGetUniqueId") : FMX_OK;
}

int fmx_comm_create_rank(const fmx_index *idx, int n_ranks, int rank, const void *id, fmx_comm **out) {
  if (!idx || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) { set_error("bad argument"); return FMX_ERR_ARG; }
  *out = nullptr;
  int rc = rccl_ready();
  if (rc) return rc;
  const Index *h = reinterpret_cast<const Index *>(idx);
  HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
  std::unique_ptr<CommSet> c(new CommSet());
  c->n_ranks = n_ranks;
  c->comm.assign(1, nullptr);
  c->device.assign(1, h->device);
  UniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  const int e = rccl()->CommInitRank(&c->comm[0], n_ranks, uid, rank);
  if (e) return nccl_fail(e, "ncclCommInitRank");
  if ((rc = add_streams(c.get())) != FMX_OK) return rc;
  *out = reinterpret_cast<fmx_comm *>(c.release());
  return FMX_OK;
}

int fmx_comm_create_all(fmx_index *const *idxs, size_t n_idx, fmx_comm **out) {
  if (!idxs || !n_idx || !out) { set_error("null argument"); return FMX_ERR_ARG; }
  *out = nullptr;
  int rc = rccl_ready();
  if (rc) return rc;
  std::unique_ptr<CommSet> c(new CommSet());
  c->n_ranks = (int)n_idx;
  for (size_t r = 0; r < n_idx; r++) {
    if (!idxs[r]) { set_error("null index handle"); return FMX_ERR_ARG; }
    const int d = reinterpret_cast<const Index *>(idxs[r])->device;
    for (int seen : c->device)
      if (seen == d) { set_error("fmx_comm_create_all needs one handle per DEVICE (RCCL has one rank per GPU)"); return FMX_ERR_ARG; }
    c->device.push_back(d);
  }
  c->comm.assign(n_idx, nullptr);
  const int e = rccl()->CommInitAll(c->comm.data(), (int)n_idx, c->device.data());
  if (e) return nccl_fail(e, "ncclCommInitAll");
  if ((rc = add_streams(c.get())) != FMX_OK) return rc;
  *out = reinterpret_cast<fmx_comm *>(c.release());
  return FMX_OK;
}

int fmx_comm_free(fmx_comm *c) {
  delete reinterpret_cast<CommSet *>(c);
  return FMX_OK;
}

int fmx_comm_info(const fmx_comm *c, int *n_ranks, int *n_local) {
  if (!c) { set_error("null argument"); return FMX_ERR_ARG; }
  const CommSet *s = reinterpret_cast<const CommSet *>(c);
  if (n_ranks) *n_ranks = s->n_ranks;
  if (n_local) *n_local = (int)s->comm.size();
  return FMX_OK;
}

int fmx_allgather_dev(fmx_comm *c, const void *const *d_send, void *const *d_recv, size_t bytes) {
  if (!c || !d_send || !d_recv) { set_error("null argument"); return FMX_ERR_ARG; }
  CommSet *s = reinterpret_cast<CommSet *>(c);
  Rccl *r = rccl();
  const size_t nl = s->comm.size();
  for (size_t i = 0; i < nl; i++)
    if (bytes && (!d_send[i] || !d_recv[i])) { set_error("null slice pointer"); return FMX_ERR_ARG; }
  if (!bytes) return FMX_OK;
  int e = r->GroupStart();                         // one process, several ranks: the calls must be fused
  if (e) return nccl_fail(e, "ncclGroupStart");
  for (size_t i = 0; i < nl && !e; i++) {
    if (hipSetDevice(s->device[i]) != hipSuccess) { (void)r->GroupEnd(); set_error("hipSetDevice"); return FMX_ERR_HIP; }
    e = r->AllGather(d_send[i], d_recv[i], bytes, kNcclChar, s->comm[i], s->stream[i]);
  }
  const int e2 = r->GroupEnd();
  if (e) return nccl_fail(e, "ncclAllGather");
  if (e2) return nccl_fail(e2, "ncclGroupEnd");
  for (size_t i = 0; i < nl; i++) {
    HIP_TRY(hipSetDevice(s->device[i]), "hipSetDevice");
    HIP_TRY(hipStreamSynchronize(s->stream[i]), "hipStreamSynchronize(all-gather)");
  }
  return FMX_OK;
}

}  // extern "C"

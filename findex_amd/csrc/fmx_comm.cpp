// fmx_comm.cpp -- the path's one exchange BELOW the host-language seam: an RCCL all-gather of the ranks' result
// slices over xGMI, for callers that are not Python: a JVM that drives several GPUs from one process,
// or one process per GPU that has its own way of shipping 128 bytes between ranks.  (The Python mirror keeps using
// torch.distributed, which is RCCL too; findex_amd/distributed.py.)
//
// RCCL is looked up at first use with dlopen, not linked: a process that already holds an RCCL (torch ships one) gets
// that very library, a process that never gathers does not load it at all.
#include <fmx.h>

#include <dlfcn.h>

#include <cstdlib>
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
  int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
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
    r.Send = reinterpret_cast<int (*)(const void *, size_t, int, int, Comm, hipStream_t)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<int (*)(void *, size_t, int, int, Comm, hipStream_t)>(sym("ncclRecv"));
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
  std::vector<int> rank;               // the communicator rank of each local rank
  std::vector<hipStream_t> stream;     // one non-blocking stream per local rank for the collective
  std::vector<hipEvent_t> event;       // per local rank: "the producer stream has reached the call"
  // every per-rank vector has the communicator's local size from the start (null entries until made), so that a
  // communicator whose initialisation failed half-way is destroyed like any other
  explicit CommSet(size_t n_local) : comm(n_local, nullptr), device(n_local, 0), rank(n_local, 0), stream(n_local, nullptr), event(n_local, nullptr) {}
  ~CommSet() {
    Rccl *r = rccl();
    for (size_t i = 0; i < comm.size(); i++) {
      (void)hipSetDevice(device[i]);
      if (stream[i]) { (void)hipStreamSynchronize(stream[i]); (void)hipStreamDestroy(stream[i]); }
      if (event[i]) (void)hipEventDestroy(event[i]);
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
  for (size_t i = 0; i < c->comm.size(); i++) {
    HIP_TRY(hipSetDevice(c->device[i]), "hipSetDevice");
    HIP_TRY(hipStreamCreateWithFlags(&c->stream[i], hipStreamNonBlocking), "hipStreamCreate");
    HIP_TRY(hipEventCreateWithFlags(&c->event[i], hipEventDisableTiming), "hipEventCreate");
  }
  return FMX_OK;
}

// The error path of fmx_comm_create_* (a communicator destroyed before its streams exist) under test without a broken
// fabric: in a library compiled with -DFMX_FAULT_INJECTION (findex_amd/lib/libfmx_faults.so, which only the tests load)
// FMX_COMM_FAIL_INIT=1 reports the RCCL initialisation call as failed without making it.  The product library has no
// such switch.
bool fail_init_injected() {
#ifdef FMX_FAULT_INJECTION
  const char *e = getenv("FMX_COMM_FAIL_INIT");
  return e && e[0] == '1';
#else
  return false;
#endif
}

// The collective's stream of local rank i waits for the work the caller has enqueued on producer[i] so far (the
// search that writes d_send, whatever last read d_recv); producer == nullptr: the caller vouches that both are idle.
int order_after_producers(CommSet *s, void *const *producer) {
  if (!producer) return FMX_OK;
  for (size_t i = 0; i < s->comm.size(); i++) {
    HIP_TRY(hipSetDevice(s->device[i]), "hipSetDevice");
    HIP_TRY(hipEventRecord(s->event[i], static_cast<hipStream_t>(producer[i])), "hipEventRecord(producer stream)");
    HIP_TRY(hipStreamWaitEvent(s->stream[i], s->event[i], 0), "hipStreamWaitEvent");
  }
  return FMX_OK;
}

int finish_collective(CommSet *s, const char *what) {
  for (size_t i = 0; i < s->comm.size(); i++) {
    HIP_TRY(hipSetDevice(s->device[i]), "hipSetDevice");
    HIP_TRY(hipStreamSynchronize(s->stream[i]), what);
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
  std::unique_ptr<CommSet> c(new CommSet(1));
  c->n_ranks = n_ranks;
  c->device[0] = h->device;
  c->rank[0] = rank;
  UniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  const int e = fail_init_injected() ? 1 : rccl()->CommInitRank(&c->comm[0], n_ranks, uid, rank);
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
  std::unique_ptr<CommSet> c(new CommSet(n_idx));
  c->n_ranks = (int)n_idx;
  for (size_t r = 0; r < n_idx; r++) {
    if (!idxs[r]) { set_error("null index handle"); return FMX_ERR_ARG; }
    const int d = reinterpret_cast<const Index *>(idxs[r])->device;
    for (size_t q = 0; q < r; q++)
      if (c->device[q] == d) { set_error("fmx_comm_create_all needs one handle per DEVICE (RCCL has one rank per GPU)"); return FMX_ERR_ARG; }
    c->device[r] = d;
    c->rank[r] = (int)r;
  }
  const int e = fail_init_injected() ? 1 : rccl()->CommInitAll(c->comm.data(), (int)n_idx, c->device.data());
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

int fmx_allgather_dev(fmx_comm *c, const void *const *d_send, void *const *d_recv, size_t bytes, void *const *producer_streams) {
  if (!c || !d_send || !d_recv) { set_error("null argument"); return FMX_ERR_ARG; }
  CommSet *s = reinterpret_cast<CommSet *>(c);
  Rccl *r = rccl();
  const size_t nl = s->comm.size();
  for (size_t i = 0; i < nl; i++)
    if (bytes && (!d_send[i] || !d_recv[i])) { set_error("null slice pointer"); return FMX_ERR_ARG; }
  if (!bytes) return FMX_OK;
  int rc = order_after_producers(s, producer_streams);
  if (rc) return rc;
  int e = r->GroupStart();                         // one process, several ranks: the calls must be fused
  if (e) return nccl_fail(e, "ncclGroupStart");
  for (size_t i = 0; i < nl && !e; i++) {
    if (hipSetDevice(s->device[i]) != hipSuccess) { (void)r->GroupEnd(); set_error("hipSetDevice"); return FMX_ERR_HIP; }
    e = r->AllGather(d_send[i], d_recv[i], bytes, kNcclChar, s->comm[i], s->stream[i]);
  }
  const int e2 = r->GroupEnd();
  if (e) return nccl_fail(e, "ncclAllGather");
  if (e2) return nccl_fail(e2, "ncclGroupEnd");
  return finish_collective(s, "hipStreamSynchronize(all-gather)");
}

// Root-only delivery: every rank sends its slice to `root`, which receives them in rank order -- what north_star's
// "final RCCL gather of hit intervals" needs when one rank (the JVM's) consumes the answer: each of the root's links
// carries one slice in, nothing lands on the other ranks.  ncclSend / ncclRecv inside one group (RCCL has no gather
// primitive of its own; ncclGather in newer releases is this).
int fmx_gather_dev(fmx_comm *c, const void *const *d_send, void *const *d_recv, size_t bytes, int root, void *const *producer_streams) {
  if (!c || !d_send || !d_recv) { set_error("null argument"); return FMX_ERR_ARG; }
  CommSet *s = reinterpret_cast<CommSet *>(c);
  Rccl *r = rccl();
  if (root < 0 || root >= s->n_ranks) { set_error("root is not a rank of the communicator"); return FMX_ERR_ARG; }
  const size_t nl = s->comm.size();
  for (size_t i = 0; i < nl; i++)
    if (bytes && (!d_send[i] || (s->rank[i] == root && !d_recv[i]))) { set_error("null slice pointer"); return FMX_ERR_ARG; }
  if (!bytes) return FMX_OK;
  int rc = order_after_producers(s, producer_streams);
  if (rc) return rc;
  int e = r->GroupStart();
  if (e) return nccl_fail(e, "ncclGroupStart");
  for (size_t i = 0; i < nl && !e; i++) {
    if (hipSetDevice(s->device[i]) != hipSuccess) { (void)r->GroupEnd(); set_error("hipSetDevice"); return FMX_ERR_HIP; }
    e = r->Send(d_send[i], bytes, kNcclChar, root, s->comm[i], s->stream[i]);
    if (!e && s->rank[i] == root)
      for (int q = 0; q < s->n_ranks && !e; q++)
        e = r->Recv(static_cast<uint8_t *>(d_recv[i]) + (size_t)q * bytes, bytes, kNcclChar, q, s->comm[i], s->stream[i]);
  }
  const int e2 = r->GroupEnd();
  if (e) return nccl_fail(e, "ncclSend/ncclRecv");
  if (e2) return nccl_fail(e2, "ncclGroupEnd");
  return finish_collective(s, "hipStreamSynchronize(gather)");
}

}  // extern "C"

// fmx_api.cpp -- the C ABI of libfmx.so (include/fmx.h): file loaders, handle lifetime, and the
// host-pointer wrappers around the device entry points.  No compute happens on the host.
#include <fmx.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "fmx_host.h"
#include "fmx_hostpar.h"
#include "fmx_regex.h"

namespace fmx {

static thread_local std::string g_err;

void set_error(const std::string &msg) { g_err = msg; }

int hip_fail(hipError_t e, const char *what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return FMX_ERR_HIP;
}

static std::atomic<int> g_layout_pref{-1};       // fmx_config_set may race with fmx_open* on other threads
int layout_preference() { return g_layout_pref.load(std::memory_order_relaxed); }
static std::atomic<uint64_t> g_serial{0};
static std::atomic<int> g_force_superblocks{0};
static std::atomic<int> g_validate{0};
bool validate_device_operands() { return g_validate.load(std::memory_order_relaxed) != 0; }
bool force_superblocks() { return g_force_superblocks.load(std::memory_order_relaxed) != 0; }

static int arg_fail(const char *msg) {
  g_err = msg;
  return FMX_ERR_ARG;
}

struct StreamGuard {
  hipStream_t s = nullptr;
  ~StreamGuard() { if (s) (void)hipStreamDestroy(s); }
};

struct DevBuf {        // a device pointer borrowed from the call's context (Call::alloc)
  void *p = nullptr;
};

struct EventPair {     // the call context's two events (not owned)
  hipEvent_t a = nullptr, b = nullptr;
};

#define HIP_TRY(call, what)                            \
  do {                                                 \
    hipError_t e__ = (call);                           \
    if (e__ != hipSuccess) return hip_fail(e__, what); \
  } while (0)

static int use_device(const Index *h) {
  HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
  return FMX_OK;
}

static void free_ctx(CallCtx *c) {
  if (!c) return;
  for (void *b : c->buf) if (b) (void)hipFree(b);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->ev_a) (void)hipEventDestroy(c->ev_a);
  if (c->ev_b) (void)hipEventDestroy(c->ev_b);
  for (hipEvent_t e : c->chunk_ev) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

CallCtx *ctx_acquire(const Index *h) {
  CallCtx *c = nullptr;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->ctx_pool.empty()) { c = h->ctx_pool.back(); h->ctx_pool.pop_back(); }
  }
  if (!c) c = new (std::nothrow) CallCtx();
  if (!c) { g_err = "out of host memory"; return nullptr; }
  hipError_t e = hipSuccess;
  // non-blocking: no implicit ties to the legacy stream (another host thread may be capturing a graph, or torch may
  // be working on stream 0)
  if (!c->stream) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess && !c->ev_a) e = hipEventCreate(&c->ev_a);
  if (e == hipSuccess && !c->ev_b) e = hipEventCreate(&c->ev_b);
  if (e != hipSuccess) { (void)hip_fail(e, "hipStreamCreate/hipEventCreate"); free_ctx(c); return nullptr; }
  return c;
}

// Contexts holding more than kKeepBytes of scratch, and contexts beyond kKeep idle ones, are released.
void ctx_release(const Index *h, CallCtx *c) {
  constexpr size_t kKeep = 4;
  constexpr size_t kKeepBytes = 2048ull << 20;    // the reference-order match of a 100 k batch holds ~0.5 GiB (element slabs + raw results)
  bool keep = c->total() <= kKeepBytes;
  if (keep) {
    std::lock_guard<std::mutex> lk(h->mu);
    keep = h->ctx_pool.size() < kKeep;
    if (keep) h->ctx_pool.push_back(c);
  }
  if (!keep) free_ctx(c);
}

hipError_t ctx_scratch(CallCtx *c, int i, size_t bytes, void **out) {
  if (bytes < 16) bytes = 16;
  if (c->cap[i] < bytes) {
    if (c->buf[i]) { (void)hipFree(c->buf[i]); c->buf[i] = nullptr; c->cap[i] = 0; }
    const size_t want = bytes + bytes / 4;          // some slack: batches of similar size reuse the buffer
    hipError_t e = hipMalloc(&c->buf[i], want);
    if (e != hipSuccess) return e;
    c->cap[i] = want;
  }
  *out = c->buf[i];
  return hipSuccess;
}

// One host-pointer call: borrows a context from the handle (or makes one), hands out scratch buffers from it,
// runs the enqueue function on its stream between its two events, and returns the context on scope exit.
class Call {
 public:
  explicit Call(const Index *h) : h_(h) {}
  ~Call() { if (c_) ctx_release(h_, c_); }
  int init() {
    if (!c_) c_ = ctx_acquire(h_);
    return c_ ? FMX_OK : FMX_ERR_HIP;
  }
  // next scratch buffer of the call, at least `bytes` long
  hipError_t alloc(DevBuf &b, size_t bytes) {
    if (next_ >= CallCtx::kBufs) return hipErrorOutOfMemory;
    return ctx_scratch(c_, next_++, bytes, &b.p);
  }
  // pinned host staging of at least `bytes`
  hipError_t pinned(void **out, size_t bytes) {
    if (c_->pin_cap < bytes) {
      if (c_->pin) { (void)hipHostFree(c_->pin); c_->pin = nullptr; c_->pin_cap = 0; }
      hipError_t e = hipHostMalloc(&c_->pin, bytes, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      c_->pin_cap = bytes;
    }
    *out = c_->pin;
    return hipSuccess;
  }
  hipStream_t stream() const { return c_->stream; }
  hipError_t chunk_events(size_t n, hipEvent_t **out) {      // n events that live with the context
    while (c_->chunk_ev.size() < n) {
      hipEvent_t e = nullptr;
      const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
      if (rc != hipSuccess) return rc;
      c_->chunk_ev.push_back(e);
    }
    *out = c_->chunk_ev.data();
    return hipSuccess;
  }
  hipEvent_t ev_a() const { return c_->ev_a; }
  hipEvent_t ev_b() const { return c_->ev_b; }
  // Runs `enqueue` on the context's stream and records the device time between the two events.
  template <class F>
  int timed(F enqueue) {
    EventPair ev{c_->ev_a, c_->ev_b};
    int rc = enqueue(c_->stream, ev);
    if (rc != FMX_OK) { (void)hipStreamSynchronize(c_->stream); return rc; }
    HIP_TRY(hipStreamSynchronize(c_->stream), "hipStreamSynchronize");
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
      std::lock_guard<std::mutex> lk(h_->mu);
      h_->last_kernel_ms = ms;
      h_->launches++;
    }
    return FMX_OK;
  }

 private:
  const Index *h_;
  CallCtx *c_ = nullptr;
  int next_ = 0;
};

// ---- file formats
static uint64_t rd_u64(const uint8_t *p, bool be) {
  uint64_t v = 0;
  if (be) for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
  else for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
  return v;
}

// BWTLoader, bwtmerger.scala:144-174: int64 size, int64 eof, then `size` bytes; size+16 == file length.
// Checks the header and leaves the file positioned at the payload; the payload is then streamed to the
// device in chunks (stream_file_to_device) -- a 16 GiB .bwt never sits in host memory whole.
static int open_bwt_file(const char *path, bool be, FILE **out, uint64_t &n, uint64_t &eof) {
  FILE *f = std::fopen(path, "rb");
  if (!f) { g_err = std::string("File ") + path + " does not exists"; return FMX_ERR_IO; }
  std::unique_ptr<FILE, int (*)(FILE *)> guard(f, std::fclose);
  uint8_t hdr[16];
  if (std::fread(hdr, 1, 16, f) != 16) { g_err = std::string("File ") + path + " bad size"; return FMX_ERR_FORMAT; }
  n = rd_u64(hdr, be);
  eof = rd_u64(hdr + 8, be);
  if (fseeko(f, 0, SEEK_END) != 0) { g_err = "seek failed"; return FMX_ERR_IO; }
  const uint64_t flen = (uint64_t)ftello(f);
  if (n + 16 != flen) {
    g_err = std::string("File ") + path + " bad size " + std::to_string(n) + " != " + std::to_string(flen) + " + 16";
    return FMX_ERR_FORMAT;
  }
  if (fseeko(f, 16, SEEK_SET) != 0) { g_err = "seek failed"; return FMX_ERR_IO; }
  *out = guard.release();
  return FMX_OK;
}

// n bytes from the file's current position to device memory through two pinned staging buffers: the read of
// chunk i+1 overlaps the DMA of chunk i.
static int stream_file_to_device(FILE *f, void *dst, uint64_t n, hipStream_t st) {
  constexpr size_t kChunk = 32u << 20;
  struct Pin {
    void *p[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    ~Pin() {
      for (int i = 0; i < 2; i++) { if (p[i]) (void)hipHostFree(p[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
    }
  } pin;
  const size_t chunk = (size_t)std::min<uint64_t>(kChunk, n ? n : 1);
  for (int i = 0; i < 2; i++) {
    HIP_TRY(hipHostMalloc(&pin.p[i], chunk, hipHostMallocDefault), "hipHostMalloc(staging)");
    HIP_TRY(hipEventCreateWithFlags(&pin.ev[i], hipEventDisableTiming), "hipEventCreate");
  }
  bool used[2] = {false, false};
  int slot = 0;
  for (uint64_t o = 0; o < n; o += chunk, slot ^= 1) {
    const size_t len = (size_t)std::min<uint64_t>(chunk, n - o);
    if (used[slot]) HIP_TRY(hipEventSynchronize(pin.ev[slot]), "hipEventSynchronize");     // its last DMA is done
    if (std::fread(pin.p[slot], 1, len, f) != len) { g_err = "short read on the .bwt payload"; return FMX_ERR_IO; }
    HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(dst) + o, pin.p[slot], len, hipMemcpyHostToDevice, st), "H2D(bwt)");
    HIP_TRY(hipEventRecord(pin.ev[slot], st), "hipEventRecord");
    used[slot] = true;
  }
  HIP_TRY(hipStreamSynchronize(st), "hipStreamSynchronize");     // the staging buffers are freed on return
  return FMX_OK;
}

// AUXLoader, bwtmerger.scala:130-142: 256 int64 counts.
static int load_aux(const char *path, bool be, int64_t counts[256]) {
  FILE *f = std::fopen(path, "rb");
  if (!f) { g_err = std::string("File ") + path + " does not exists"; return FMX_ERR_IO; }
  uint8_t buf[2048];
  size_t got = std::fread(buf, 1, sizeof buf, f);
  int extra = std::fgetc(f);
  std::fclose(f);
  if (got != sizeof buf || extra != EOF) { g_err = std::string("File ") + path + " bad aux size"; return FMX_ERR_FORMAT; }
  for (int i = 0; i < 256; i++) counts[i] = (int64_t)rd_u64(buf + 8 * i, be);
  return FMX_OK;
}

Worker *worker_of(const Index *h) {
  std::lock_guard<std::mutex> lk(h->mu);
  if (!h->worker) h->worker.reset(new Worker());
  return h->worker.get();
}

static void destroy(Index *h) {
  if (!h) return;
  h->worker.reset();                  // finishes what was submitted, joins
  (void)hipSetDevice(h->device);
  if (h->d_bv) (void)hipFree(h->d_bv);
  if (h->d_chk) (void)hipFree(h->d_chk);
  if (h->d_sup) (void)hipFree(h->d_sup);
  if (h->d_bwt) (void)hipFree(h->d_bwt);
  if (h->d_cf) (void)hipFree(h->d_cf);
  if (h->d_slot) (void)hipFree(h->d_slot);
  if (h->d_counters) (void)hipFree(h->d_counters);
  if (h->d_ktab) (void)hipFree(h->d_ktab);
  if (h->d_kt_dense) (void)hipFree(h->d_kt_dense);
  if (h->d_kt_levels) (void)hipFree(h->d_kt_levels);
  if (h->d_jump) (void)hipFree(h->d_jump);
  if (h->d_row1) (void)hipFree(h->d_row1);
  if (h->d_row3) (void)hipFree(h->d_row3);
  if (h->d_sel_dir) (void)hipFree(h->d_sel_dir);
  if (h->d_sel_off) (void)hipFree(h->d_sel_off);
  if (h->d_sel_shift) (void)hipFree(h->d_sel_shift);
  for (CallCtx *c : h->ctx_pool) free_ctx(c);
  delete h;
}

// Common tail of the three open flavours: `src` is host or device memory holding n BWT bytes.
struct BlockSpec {       // fmx_open_block: NaiveBWTSearcher's constructor arguments beside the BWT
  const int64_t *bs;
  int first, skipped;    // BWT bytes at position 0 (-1 when that is the skipped row) and at the skipped row
};

static int open_common(const void *src, bool src_on_device, FILE *src_file, uint64_t n, uint64_t eof,
                       const int64_t *counts, int device, hipStream_t user_stream, fmx_index **out,
                       const BlockSpec *block = nullptr) {
  if (!out) return arg_fail("out is null");
  *out = nullptr;
  if (!src && !src_file && n) return arg_fail("bwt is null");
  if (n < 1 || eof >= n) return arg_fail("need n >= 1 and eof < n");
  if (n >= (1ull << 38)) { g_err = "n >= 2^38 is not supported"; return FMX_ERR_UNSUPPORTED; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) { g_err = "no HIP device available (libfmx has no CPU fallback)"; return FMX_ERR_HIP; }
  if (device < 0 || device >= ndev) return arg_fail("device index out of range");
  HIP_TRY(hipSetDevice(device), "hipSetDevice");
  Index *h = new (std::nothrow) Index();
  if (!h) { g_err = "out of host memory"; return FMX_ERR_NOMEM; }
  h->serial = ++g_serial;
  h->policy = default_policy();        // this handle's own copy from here on
  if (block) {
    h->block_mode = true;
    std::memcpy(h->block_bs, block->bs, sizeof h->block_bs);
    h->block_first = block->first;
    h->block_skipped = block->skipped;
  }
  h->device = device;
  h->n = n;
  h->eof = eof;
  h->nblocks = n / kBlockBits + 1;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->cu_count = prop.multiProcessorCount;
  StreamGuard own;
  hipStream_t st = user_stream;
  int rc = FMX_OK;
  do {
    if (!st) {
      if ((e = hipStreamCreate(&own.s)) != hipSuccess) { rc = hip_fail(e, "hipStreamCreate"); break; }
      st = own.s;
    }
    // device copy of the BWT: zero-padded to whole 128-position blocks, slot eof set to 0 (BWT' has the
    // EOF symbol there; the stored byte is a filler, bwtmerger.scala:799-806)
    const uint64_t padded = (n / kByteBlock + 2) * kByteBlock;
    if ((e = hipMalloc(&h->d_bwt, padded)) != hipSuccess) { rc = hip_fail(e, "hipMalloc(bwt)"); break; }
    if ((e = hipMemsetAsync((uint8_t *)h->d_bwt + n, 0, padded - n, st)) != hipSuccess) { rc = hip_fail(e, "memset"); break; }
    if (src_file) {
      if ((rc = stream_file_to_device(src_file, h->d_bwt, n, st)) != FMX_OK) break;
    } else {
      e = hipMemcpyAsync(h->d_bwt, src, n, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
      if (e != hipSuccess) { rc = hip_fail(e, "copy bwt"); break; }
    }
    if ((e = hipMemsetAsync((uint8_t *)h->d_bwt + eof, 0, 1, st)) != hipSuccess) { rc = hip_fail(e, "memset"); break; }
    const int pref = layout_preference();
    h->layout = pref == (int)kLayoutBytes ? kLayoutBytes : kLayoutOneHot;
    if (n >= kOneHotMaxN) {              // the one-hot block split is exact below 2^37 only
      if (pref == (int)kLayoutOneHot) { g_err = "layout onehot needs n < 2^37"; rc = FMX_ERR_UNSUPPORTED; break; }
      h->layout = kLayoutBytes;
    }
    rc = build_index(h, st, counts);
    if (rc == -1) {                      // one-hot vectors do not fit: fall back unless one-hot was forced
      if (pref == (int)kLayoutOneHot) { rc = FMX_ERR_NOMEM; break; }
      h->layout = kLayoutBytes;
      rc = build_index(h, st, counts);
    }
  } while (0);
  if (rc != FMX_OK) { destroy(h); return rc; }
  *out = reinterpret_cast<fmx_index *>(h);
  return FMX_OK;
}

static inline Index *H(fmx_index *p) { return reinterpret_cast<Index *>(p); }
static inline const Index *H(const fmx_index *p) { return reinterpret_cast<const Index *>(p); }

// Operands in, kernel, results out.  Small calls (the single-query forms a per-call adapter makes) pack all
// operands into one pinned staging buffer and travel as ONE copy each way; large calls copy each array
// directly between the caller's memory and its own device buffer.
struct HostIn { const void *src; size_t bytes; };
struct HostOut { void *dst; size_t bytes; };      // dst == nullptr: device scratch of that size, nothing comes back
constexpr size_t kSmallCall = 256u << 10;
constexpr size_t kTinyCall = 2048;

template <class Launch>
static int run_io(const Index *h, const HostIn *ins, int nin, const HostOut *outs, int nout, Launch launch) {
  Call call(h);
  int rc = call.init();
  if (rc != FMX_OK) return rc;
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t in_total = 0, out_total = 0;
  for (int j = 0; j < nin; j++) in_total += up16(ins[j].bytes);
  for (int j = 0; j < nout; j++) out_total += up16(outs[j].bytes);
  const void *din[CallCtx::kBufs] = {nullptr};
  void *dout[CallCtx::kBufs] = {nullptr};
  if (in_total + out_total <= kSmallCall) {
    void *pin = nullptr;
    HIP_TRY(call.pinned(&pin, kSmallCall), "hipHostMalloc");
    DevBuf a, b;
    HIP_TRY(call.alloc(a, kSmallCall), "hipMalloc");
    HIP_TRY(call.alloc(b, kSmallCall), "hipMalloc");
    uint8_t *hp = static_cast<uint8_t *>(pin);
    size_t o = 0;
    for (int j = 0; j < nin; j++) {
      if (ins[j].bytes) std::memcpy(hp + o, ins[j].src, ins[j].bytes);
      din[j] = static_cast<uint8_t *>(a.p) + o;
      o += up16(ins[j].bytes);
    }
    size_t oo = 0;
    for (int j = 0; j < nout; j++) { dout[j] = static_cast<uint8_t *>(b.p) + oo; oo += up16(outs[j].bytes); }
    uint8_t *hout = hp + in_total;                 // results come back behind the operands
    // The single-query forms (a per-call adapter's getPrevRange / occ / search): no copies at all -- the kernel reads its
    // few operand bytes from the page-locked staging buffer over the link and writes its results there (the buffer is
    // mapped into the device's address space like every hipHostMalloc allocation).  Two copy submissions less per call.
    // No events around the kernel either (19.7 against 24.4 against 33.1 us per fmx_occ_batch of one query):
    // fmx_stats_t.last_kernel_ms is not updated by such a call.  FMX_TINY_DIRECT=0: as every other small call; 1: with events.
    static const int tiny_direct = getenv("FMX_TINY_DIRECT") ? atoi(getenv("FMX_TINY_DIRECT")) : 2;
    if (tiny_direct && in_total + out_total <= kTinyCall) {
      o = 0;
      for (int j = 0; j < nin; j++) { din[j] = hp + o; o += up16(ins[j].bytes); }
      oo = 0;
      for (int j = 0; j < nout; j++) { dout[j] = hout + oo; oo += up16(outs[j].bytes); }
      if (tiny_direct == 2) {
        const hipError_t le = launch(call.stream(), din, dout);
        const hipError_t se = hipStreamSynchronize(call.stream());
        if (le != hipSuccess) return hip_fail(le, "kernel launch");
        if (se != hipSuccess) return hip_fail(se, "hipStreamSynchronize");
        { std::lock_guard<std::mutex> lk(h->mu); h->launches++; }
        rc = FMX_OK;
      } else {
        rc = call.timed([&](hipStream_t st, EventPair &ev) {
          HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
          HIP_TRY(launch(st, din, dout), "kernel launch");
          HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
          return (int)FMX_OK;
        });
      }
      if (rc != FMX_OK) return rc;
      oo = 0;
      for (int j = 0; j < nout; j++) {
        if (outs[j].bytes && outs[j].dst) std::memcpy(outs[j].dst, hout + oo, outs[j].bytes);
        oo += up16(outs[j].bytes);
      }
      return FMX_OK;
    }
    rc = call.timed([&](hipStream_t st, EventPair &ev) {
      if (in_total) HIP_TRY(hipMemcpyAsync(a.p, hp, in_total, hipMemcpyHostToDevice, st), "H2D");
      HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
      HIP_TRY(launch(st, din, dout), "kernel launch");
      HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
      if (out_total) HIP_TRY(hipMemcpyAsync(hout, b.p, out_total, hipMemcpyDeviceToHost, st), "D2H");
      return (int)FMX_OK;
    });
    if (rc != FMX_OK) return rc;
    oo = 0;
    for (int j = 0; j < nout; j++) {
      if (outs[j].bytes && outs[j].dst) std::memcpy(outs[j].dst, hout + oo, outs[j].bytes);
      oo += up16(outs[j].bytes);
    }
    return FMX_OK;
  }
  DevBuf bufs[CallCtx::kBufs];
  for (int j = 0; j < nin; j++) { HIP_TRY(call.alloc(bufs[j], ins[j].bytes), "hipMalloc"); din[j] = bufs[j].p; }
  for (int j = 0; j < nout; j++) { HIP_TRY(call.alloc(bufs[nin + j], outs[j].bytes), "hipMalloc"); dout[j] = bufs[nin + j].p; }
  return call.timed([&](hipStream_t st, EventPair &ev) {
    for (int j = 0; j < nin; j++)
      if (ins[j].bytes) HIP_TRY(hipMemcpyAsync(bufs[j].p, ins[j].src, ins[j].bytes, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
    HIP_TRY(launch(st, din, dout), "kernel launch");
    HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
    for (int j = 0; j < nout; j++)
      if (outs[j].bytes && outs[j].dst) HIP_TRY(hipMemcpyAsync(outs[j].dst, dout[j], outs[j].bytes, hipMemcpyDeviceToHost, st), "D2H");
    return (int)FMX_OK;
  });
}

}  // namespace fmx

using namespace fmx;

extern "C" {

const char *fmx_last_error(void) { return g_err.c_str(); }
int fmx_abi_version(void) { return FMX_ABI_VERSION; }

// fmx_config_set("pipeline", ..): large host batches in page-locked memory cut into chunks over three streams
static std::atomic<int> g_pipeline{0};

int fmx_config_set(const char *key, const char *value) {
  if (!key || !value) return arg_fail("null argument");
  if (std::strcmp(key, "layout") == 0) {
    if (std::strcmp(value, "auto") == 0) g_layout_pref.store(-1);
    else if (std::strcmp(value, "onehot") == 0) g_layout_pref.store((int)kLayoutOneHot);
    else if (std::strcmp(value, "bytes") == 0) g_layout_pref.store((int)kLayoutBytes);
    else return arg_fail("layout must be auto, onehot or bytes");
    return FMX_OK;
  }
  if (std::strcmp(key, "validate") == 0) {
    g_validate.store(std::strcmp(value, "0") != 0);
    return FMX_OK;
  }
  if (std::strcmp(key, "checkpoints") == 0) {
    if (std::strcmp(value, "auto") == 0) g_force_superblocks.store(0);
    else if (std::strcmp(value, "superblock") == 0) g_force_superblocks.store(1);
    else return arg_fail("checkpoints must be auto or superblock");
    return FMX_OK;
  }
  if (std::strcmp(key, "pipeline") == 0) {
    if (std::strcmp(value, "on") == 0) g_pipeline.store(1, std::memory_order_relaxed);
    else if (std::strcmp(value, "off") == 0) g_pipeline.store(0, std::memory_order_relaxed);
    else return arg_fail("pipeline must be on or off");
    return FMX_OK;
  }
  if (std::strcmp(key, "threads") == 0) {
    char *end = nullptr;
    const long v = std::strtol(value, &end, 10);
    if (end == value || *end || v < 0) return arg_fail("threads must be a non-negative integer");
    set_host_threads((unsigned)v);
    return FMX_OK;
  }
  // the table policy's keys: the defaults a handle copies when it is opened (fmx_index_config_set: one handle's own)
  const char *why = nullptr;
  const int pr = policy_set(default_policy(), key, value, &why);
  if (pr == 0) return FMX_OK;
  if (pr == 2) return arg_fail(why);
  return arg_fail("unknown configuration key");
}

int fmx_index_config_set(fmx_index *idx, const char *key, const char *value) {
  if (!idx || !key || !value) return arg_fail("null argument");
  const char *why = nullptr;
  const int pr = policy_set(H(idx)->policy, key, value, &why);
  if (pr == 0) return FMX_OK;
  if (pr == 2) return arg_fail(why);
  return arg_fail("not a per-handle key (ktab, jump, jump_pairs, jump_chars, tables_after, table_budget)");
}

int fmx_host_alloc(size_t bytes, void **out) {
  if (!out) return arg_fail("out is null");
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault), "hipHostMalloc");
  return FMX_OK;
}
int fmx_host_free(void *p) {
  if (p) HIP_TRY(hipHostFree(p), "hipHostFree");
  return FMX_OK;
}

int fmx_device_count(int *count) {
  if (!count) return arg_fail("count is null");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return FMX_OK;
}

// The X.fm sibling of X.bwt (BWTTempStorage.genFMFilename, bwtmerger.scala:33-36: the extension swapped).  The reference
// cannot open an index without it -- NaiveFMSearcher takes n = fm.size (bwtmerger.scala:339) and FMLoader throws on a bad
// header (:259-262): element size 4, size * 4 + 9 == file length.  This engine derives its rank dictionary from the .bwt
// and never reads the lists, so a MISSING .fm is fine; a .fm that is there is held to the reference's checks, and to
// describing the same n rows as the .bwt -- an index whose files do not belong together fails here as it does there.
static int check_fm_sibling(const char *bwt_path, bool be, uint64_t n) {
  std::string p(bwt_path);
  const size_t slash = p.find_last_of("/\\"), dot = p.find_last_of('.');
  if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) p.erase(dot);
  p += ".fm";
  FILE *f = std::fopen(p.c_str(), "rb");
  if (!f) return FMX_OK;
  std::unique_ptr<FILE, int (*)(FILE *)> guard(f, std::fclose);
  uint8_t hdr[9];
  if (std::fread(hdr, 1, 9, f) != 9) { g_err = "File " + p + " bad size"; return FMX_ERR_FORMAT; }
  const unsigned el = hdr[0];
  const uint64_t size = rd_u64(hdr + 1, be);
  if (el != 4) { g_err = "File " + p + " bad elSize " + std::to_string(el); return FMX_ERR_FORMAT; }
  if (fseeko(f, 0, SEEK_END) != 0) { g_err = "seek failed"; return FMX_ERR_IO; }
  const uint64_t flen = (uint64_t)ftello(f);
  if (size > (1ull << 40) || size * 4 + 9 != flen) {
    g_err = "File " + p + " bad size " + std::to_string(size) + " + 0x9 != " + std::to_string(flen) + "(filelen)";
    return FMX_ERR_FORMAT;
  }
  if (size != n) {
    g_err = "File " + p + " holds " + std::to_string(size) + " rows, " + bwt_path + " " + std::to_string(n) + ": not one index";
    return FMX_ERR_FORMAT;
  }
  return FMX_OK;
}

int fmx_open(const char *bwt_path, const char *aux_path, int big_endian, int device, fmx_index **out) {
  if (!bwt_path || !aux_path) return arg_fail("path is null");
  uint64_t n = 0, eof = 0;
  int64_t counts[256];
  FILE *f = nullptr;
  int rc = open_bwt_file(bwt_path, big_endian != 0, &f, n, eof);
  if (rc != FMX_OK) return rc;
  std::unique_ptr<FILE, int (*)(FILE *)> guard(f, std::fclose);
  rc = load_aux(aux_path, big_endian != 0, counts);
  if (rc != FMX_OK) return rc;
  rc = check_fm_sibling(bwt_path, big_endian != 0, n);
  if (rc != FMX_OK) return rc;
  return open_common(nullptr, false, f, n, eof, counts, device, nullptr, out);
}

int fmx_open_mem(const uint8_t *bwt, uint64_t n, uint64_t eof, const int64_t counts[256], int device,
                 fmx_index **out) {
  if (!counts) return arg_fail("counts is null");
  return open_common(bwt, false, nullptr, n, eof, counts, device, nullptr, out);
}

int fmx_open_dev(const void *d_bwt, uint64_t n, uint64_t eof, const int64_t *counts_or_null, int device, void *stream,
                 fmx_index **out) {
  return open_common(d_bwt, true, nullptr, n, eof, counts_or_null, device, (hipStream_t)stream, out);
}

// NaiveBWTSearcher(bwt, bucketStarts, rk0), findex.scala:459-506: the searcher BWTMerger2.calcGaps uses over one
// block's BWT.  The skipped row plays the part of the EOF slot; everything else that differs from NaiveFMSearcher
// is settled when the dictionary is built (build_index, block mode).
int fmx_open_block(const uint8_t *bwt, uint64_t n, const int64_t bucket_starts[256], uint64_t rk0, int device,
                   fmx_index **out) {
  if (!bwt || !bucket_starts) return arg_fail("null argument");
  if (n < 1 || rk0 >= n) return arg_fail("need n >= 1 and rk0 < n");
  for (int c = 0; c < 256; c++)
    if (bucket_starts[c] < 0 || (uint64_t)bucket_starts[c] > n) return arg_fail("bucket start out of range");
  BlockSpec spec{bucket_starts, rk0 == 0 ? -1 : (int)bwt[0], (int)bwt[rk0]};
  return open_common(bwt, false, nullptr, n, rk0, nullptr, device, nullptr, out, &spec);
}

int fmx_prepare(const fmx_index *idx, unsigned what) {
  if (!idx) return arg_fail("null argument");
  if (what & ~(unsigned)(FMX_PREPARE_KTAB | FMX_PREPARE_SELECT | FMX_PREPARE_JUMP | FMX_PREPARE_FRONTIER | FMX_PREPARE_SEARCH)) return arg_fail("unknown fmx_prepare flag");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc) return rc;
  CtxLease lease(h);
  if (!lease.c) return FMX_ERR_HIP;
  if (what & FMX_PREPARE_KTAB) {
    h->prepared_ktab.store(true, std::memory_order_relaxed);      // (its own flag: the row tables stay under their threshold)
    KTab kt;
    HIP_TRY(ktab_get(h, lease.c->stream, &kt), "k-mer table");
  }
  if (what & FMX_PREPARE_SELECT) HIP_TRY(select_prepare(h, lease.c->stream), "select directory");
  if (what & FMX_PREPARE_JUMP) {
    h->prepared_rows.store(true, std::memory_order_relaxed);      // from now on searches use (and may build) the row tables
    const uint4 *jt = nullptr;
    HIP_TRY(jump_get(h, lease.c->stream, &jt), "jump table");
    const unsigned long long *r3 = nullptr;
    HIP_TRY(row3_get(h, lease.c->stream, &r3), "three-step row table");      // beside the jump table, or instead of it
  }
  if (what & FMX_PREPARE_FRONTIER) {
    const unsigned long long *r1 = nullptr;
    HIP_TRY(row1_get(h, lease.c->stream, &r1), "row table");
  }
  // the literal search kernel this handle's tables select, calibrated here (its residency census: fmx_search.hip) so that no
  // _dev call ever has to read anything back
  if (what & (FMX_PREPARE_KTAB | FMX_PREPARE_JUMP | FMX_PREPARE_SEARCH)) HIP_TRY(search_calibrate(h, lease.c->stream), "search calibration");
  return FMX_OK;
}

int fmx_prepare_ex(fmx_index *idx, unsigned what, uint64_t budget_bytes) {
  if (!idx) return arg_fail("null argument");
  if (budget_bytes) {
    H(idx)->policy.budget_bytes.store(budget_bytes, std::memory_order_relaxed);
    H(idx)->policy.budget_ppb.store(0, std::memory_order_relaxed);
  }
  return fmx_prepare(idx, what);
}

int fmx_drop_tables(fmx_index *idx, unsigned what) {
  if (!idx) return arg_fail("null argument");
  if (!what || (what & ~(unsigned)(FMX_PREPARE_KTAB | FMX_PREPARE_JUMP | FMX_PREPARE_FRONTIER))) return arg_fail("fmx_drop_tables frees the k-mer table and the row tables (FMX_PREPARE_KTAB, FMX_PREPARE_JUMP, FMX_PREPARE_FRONTIER)");
  Index *h = H(idx);
  int rc = use_device(h);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  return drop_tables(h, what);
}

int fmx_close(fmx_index *idx) {
  if (!idx) return FMX_OK;            // closing nothing is not an error (a wrapper whose handle was already taken: hipfm.scala)
  destroy(H(idx));
  return FMX_OK;
}

int fmx_n(const fmx_index *idx, uint64_t *n) {
  if (!idx || !n) return arg_fail("null argument");
  *n = H(idx)->n;
  return FMX_OK;
}
int fmx_eof(const fmx_index *idx, uint64_t *eof) {
  if (!idx || !eof) return arg_fail("null argument");
  *eof = H(idx)->eof;
  return FMX_OK;
}
int fmx_cf(const fmx_index *idx, int c, uint64_t *out) {
  if (!idx || !out) return arg_fail("null argument");
  if (c < 0 || c > 255) return arg_fail("symbol out of range (reference: ArrayIndexOutOfBounds)");
  *out = H(idx)->cf[c];
  return FMX_OK;
}
int fmx_counts(const fmx_index *idx, int64_t out[256]) {
  if (!idx || !out) return arg_fail("null argument");
  std::memcpy(out, H(idx)->counts, sizeof H(idx)->counts);
  return FMX_OK;
}
int fmx_device(const fmx_index *idx, int *device) {
  if (!idx || !device) return arg_fail("null argument");
  *device = H(idx)->device;
  return FMX_OK;
}

// ---------------------------------------------------------------- device-pointer entry points
int fmx_occ_batch_dev(const fmx_index *idx, const void *d_c, const void *d_i, void *d_out, size_t k, void *stream) {
  if (!idx || (k && (!d_c || !d_i || !d_out))) return arg_fail("null argument");
  int rc = use_device(H(idx));
  if (rc) return rc;
  HIP_TRY(launch_occ(H(idx), d_c, d_i, d_out, k, (hipStream_t)stream), "k_occ");
  return FMX_OK;
}

int fmx_search_batch_ex_dev(const fmx_index *idx, const void *d_pat, const void *d_off, void *d_sp, void *d_ep, size_t k,
                            const fmx_search_opts *opts, void *stream) {
  const uint32_t fixed = opts ? opts->fixed_len : 0u;
  if (!idx || (k && ((!d_off && !fixed) || !d_sp || !d_ep))) return arg_fail("null argument");
  if (k && fixed && !d_pat) return arg_fail("pat is null");
  int rc = use_device(H(idx));
  if (rc) return rc;
  if (k && !fixed && validate_device_operands()) {
    bool ok = true;
    HIP_TRY(check_offsets(H(idx), d_off, k, (hipStream_t)stream, &ok), "k_check_offsets");
    if (!ok) return arg_fail("pattern offsets must be non-decreasing");
  }
  // packed: by the search kernel itself where it finishes every pattern, else in place behind it (d_sp's first k words)
  if (opts && (opts->packed & ~(FMX_SEARCH_PACKED | FMX_SEARCH_MISS_NONE))) return arg_fail("fmx_search_opts.packed: unknown bits");
  HIP_TRY(launch_search(H(idx), d_pat, fixed ? nullptr : d_off, d_sp, d_ep, k, (hipStream_t)stream, fixed,
                        (opts && (opts->packed & FMX_SEARCH_PACKED)) ? (uint64_t)opts->escape_cap : ~0ull,
                        (opts && (opts->packed & FMX_SEARCH_MISS_NONE)) ? kSearchMissNone : 0u), "k_search");
  return FMX_OK;
}

int fmx_search_batch_dev(const fmx_index *idx, const void *d_pat, const void *d_off, void *d_sp, void *d_ep, size_t k,
                         void *stream) {
  return fmx_search_batch_ex_dev(idx, d_pat, d_off, d_sp, d_ep, k, nullptr, stream);
}

size_t fmx_packed_words(size_t k, size_t escape_cap) { return k + 1 + 2 * escape_cap; }

int fmx_pack_intervals_dev(const fmx_index *idx, const void *d_sp, const void *d_ep, size_t k, size_t escape_cap, void *d_packed,
                           void *stream) {
  if (!idx || !d_packed || (k && (!d_sp || !d_ep))) return arg_fail("null argument");
  int rc = use_device(H(idx));
  if (rc) return rc;
  HIP_TRY(launch_pack_intervals(H(idx), d_sp, d_ep, k, escape_cap, d_packed, (hipStream_t)stream), "k_pack_intervals");
  return FMX_OK;
}

int fmx_unpack_intervals_dev(const fmx_index *idx, const void *d_packed, size_t k, size_t escape_cap, void *d_sp, void *d_ep,
                             void *stream) {
  if (!idx || !d_packed || (k && (!d_sp || !d_ep))) return arg_fail("null argument");
  int rc = use_device(H(idx));
  if (rc) return rc;
  HIP_TRY(launch_unpack_intervals(H(idx), d_packed, k, escape_cap, d_sp, d_ep, (hipStream_t)stream), "k_unpack_intervals");
  return FMX_OK;
}

// Host-side decode of the 8-byte form (format arithmetic on the caller's own result buffer: no search happens here).
int fmx_unpack_intervals(const uint64_t *packed, size_t k, size_t escape_cap, uint64_t *sp, uint64_t *ep) {
  if (!packed || (k && (!sp || !ep))) return arg_fail("null argument");
  for (size_t q = 0; q < k; q++) {
    const uint64_t a = packed[q] & ((1ull << 40) - 1);
    sp[q] = a;
    ep[q] = a + (packed[q] >> 40);
  }
  const uint64_t cnt = packed[k];
  for (uint64_t j = 0; j < cnt && j < escape_cap; j++) {
    const uint64_t q = packed[k + 1 + 2 * j];
    if (q >= k) { g_err = "packed intervals: escape entry names pattern " + std::to_string(q) + " of " + std::to_string(k); return FMX_ERR_FORMAT; }
    ep[q] = packed[k + 2 + 2 * j];
  }
  if (cnt > escape_cap) {
    g_err = std::to_string(cnt) + " intervals of 2^24 - 1 rows or more, the escape list holds " + std::to_string(escape_cap);
    return FMX_ERR_OVERFLOW;
  }
  return FMX_OK;
}

int fmx_prev_range_batch_dev(const fmx_index *idx, const void *d_sp, const void *d_ep, const void *d_c, void *d_sp1,
                             void *d_ep1, size_t k, void *stream) {
  if (!idx || (k && (!d_sp || !d_ep || !d_c || !d_sp1 || !d_ep1))) return arg_fail("null argument");
  int rc = use_device(H(idx));
  if (rc) return rc;
  HIP_TRY(launch_prev_range(H(idx), d_sp, d_ep, d_c, d_sp1, d_ep1, k, (hipStream_t)stream), "k_prev_range");
  return FMX_OK;
}

int fmx_lf_walk_batch_dev(const fmx_index *idx, const void *d_rows, size_t k, uint32_t len, void *d_out_bytes,
                          void *d_end_rows, void *stream) {
  if (!idx || (k && !d_rows)) return arg_fail("null argument");
  int rc = use_device(H(idx));
  if (rc) return rc;
  HIP_TRY(launch_lf_walk(H(idx), d_rows, k, len, d_out_bytes, d_end_rows, (hipStream_t)stream), "k_lf_walk");
  return FMX_OK;
}

int fmx_psi_batch_dev(const fmx_index *idx, const void *d_rows, void *d_out, size_t k, void *stream) {
  if (!idx || (k && (!d_rows || !d_out))) return arg_fail("null argument");
  int rc = use_device(H(idx));
  if (rc) return rc;
  HIP_TRY(launch_psi(H(idx), d_rows, d_out, k, (hipStream_t)stream), "k_psi");
  return FMX_OK;
}

int fmx_next_substr_batch_dev(const fmx_index *idx, const void *d_rows, size_t k, uint32_t len, void *d_out,
                              void *d_out_len, void *stream) {
  if (!idx || (k && (!d_rows || !d_out_len || (len && !d_out)))) return arg_fail("null argument");
  int rc = use_device(H(idx));
  if (rc) return rc;
  HIP_TRY(launch_next_substr(H(idx), d_rows, k, len, d_out, d_out_len, (hipStream_t)stream), "k_next_substr");
  return FMX_OK;
}

// ---------------------------------------------------------------- host-pointer entry points
int fmx_occ_batch(const fmx_index *idx, const uint8_t *c, const int64_t *i, uint64_t *out, size_t k) {
  if (!idx || (k && (!c || !i || !out))) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc || !k) return rc;
  const HostIn ins[] = {{c, k}, {i, k * 8}};
  const HostOut outs[] = {{out, k * 8}};
  return run_io(h, ins, 2, outs, 1, [&](hipStream_t st, const void *const *di, void *const *dout) {
    return launch_occ(h, di[0], di[1], dout[0], k, st);
  });
}

int fmx_search_batch(const fmx_index *idx, const uint8_t *pat, const uint64_t *off, uint64_t *sp, uint64_t *ep,
                     size_t k) {
  return fmx_search_batch_ex(idx, pat, off, sp, ep, k, nullptr);
}

int fmx_search_batch_ex(const fmx_index *idx, const uint8_t *pat, const uint64_t *off, uint64_t *sp, uint64_t *ep,
                        size_t k, const fmx_search_opts *opts) {
  const uint32_t fixed = opts ? opts->fixed_len : 0u;
  if (opts && (opts->packed & ~(FMX_SEARCH_PACKED | FMX_SEARCH_MISS_NONE))) return arg_fail("fmx_search_opts.packed: unknown bits");
  const bool packed = opts && (opts->packed & FMX_SEARCH_PACKED);
  const uint32_t sflags = (opts && (opts->packed & FMX_SEARCH_MISS_NONE)) ? kSearchMissNone : 0u;
  const size_t esc = packed ? (size_t)opts->escape_cap : 0;
  if (!idx || (k && ((!off && !fixed) || !sp || (!ep && !packed)))) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc) return rc;
  if (!k) { if (packed && sp) sp[0] = 0; return FMX_OK; }
  if (fixed || packed || sflags) {
    // ---- the lean forms of a host batch (round 4): equal-length patterns travel without offsets (8 B per pattern less
    // up the link) and the intervals come back in the 8-byte form (8 B per pattern less down): 56 -> 40 B per
    // 32-character pattern.  Whole arrays up, one chain of kernels, one array down, like the default path below.
    uint64_t lo = 0, total = (uint64_t)fixed * k;
    if (!fixed) {
      unsigned bad = off[k] < off[0];
      for (size_t q = 0; q < k; q++) bad |= off[q + 1] < off[q];
      if (bad) return arg_fail("pattern offsets must be non-decreasing");
      lo = off[0];
      total = off[k] - off[0];
    }
    if (total && !pat) return arg_fail("pat is null");
    const size_t out_words = packed ? fmx_packed_words(k, esc) : k;
    if (k < (128u << 10) && !lo) {
      HostIn ins[2] = {{total ? pat : nullptr, (size_t)total}, {off, fixed ? 0 : (k + 1) * 8}};
      HostOut outs[2] = {{sp, out_words * 8}, {packed ? nullptr : ep, k * 8}};
      return run_io(h, ins, 2, outs, 2, [&](hipStream_t st, const void *const *di, void *const *dout) {
        return launch_search(h, di[0], fixed ? nullptr : di[1], dout[0], dout[1], k, st, fixed, packed ? (uint64_t)esc : ~0ull, sflags);
      });
    }
    Call c0(h);
    if ((rc = c0.init()) != FMX_OK) return rc;
    DevBuf d_pat, d_off, d_sp, d_ep;
    HIP_TRY(c0.alloc(d_pat, (size_t)total + 16), "hipMalloc");
    HIP_TRY(c0.alloc(d_off, fixed ? 16 : (k + 1) * 8), "hipMalloc");
    HIP_TRY(c0.alloc(d_sp, out_words * 8), "hipMalloc");
    HIP_TRY(c0.alloc(d_ep, k * 8), "hipMalloc");
    if (total) HIP_TRY(hipMemcpy(d_pat.p, pat + lo, (size_t)total, hipMemcpyHostToDevice), "H2D(patterns)");
    if (!fixed) {
      if (lo) {
        std::vector<uint64_t> roff(k + 1);
        for (size_t q = 0; q <= k; q++) roff[q] = off[q] - lo;
        HIP_TRY(hipMemcpy(d_off.p, roff.data(), (k + 1) * 8, hipMemcpyHostToDevice), "H2D(offsets)");
      } else {
        HIP_TRY(hipMemcpy(d_off.p, off, (k + 1) * 8, hipMemcpyHostToDevice), "H2D(offsets)");
      }
    }
    rc = c0.timed([&](hipStream_t st, EventPair &ev) {
      HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
      HIP_TRY(launch_search(h, d_pat.p, fixed ? nullptr : d_off.p, d_sp.p, d_ep.p, k, st, fixed, packed ? (uint64_t)esc : ~0ull, sflags), "k_search");
      HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
      return (int)FMX_OK;
    });
    if (rc != FMX_OK) return rc;
    HIP_TRY(hipMemcpy(sp, d_sp.p, out_words * 8, hipMemcpyDeviceToHost), packed ? "D2H(packed)" : "D2H(sp)");
    if (!packed) HIP_TRY(hipMemcpy(ep, d_ep.p, k * 8, hipMemcpyDeviceToHost), "D2H(ep)");
    return FMX_OK;
  }
  // Offsets must be non-decreasing (a kernel would read a "negative" pattern as 2^64 bytes).  Walking a million of
  // them on the host takes 0.37 ms -- a quarter of a large call -- so large batches are checked where it is free:
  // chunk by chunk on the host while the uploads run (page-locked buffers), or by a kernel on the uploaded copy.
  constexpr size_t kPipelineMin = 128u << 10;     // patterns
  auto monotonic = [&](size_t a, size_t b) {      // off[a] <= off[a+1] <= .. <= off[b]
    unsigned bad = 0;
    for (size_t q = a; q < b; q++) bad |= off[q + 1] < off[q];
    return bad == 0;
  };
  if (k < kPipelineMin && !monotonic(0, k)) return arg_fail("pattern offsets must be non-decreasing");
  if (off[k] < off[0]) return arg_fail("pattern offsets must be non-decreasing");
  const uint64_t lo = off[0], total = off[k] - off[0];
  if (total && !pat) return arg_fail("pat is null");
  // offsets are rebased so that only the bytes in use travel
  std::vector<uint64_t> roff;
  const uint64_t *offp = off;
  if (lo) {
    roff.resize(k + 1);
    for (size_t q = 0; q <= k; q++) roff[q] = off[q] - lo;
    offp = roff.data();
  }
  // Small batches: one copy each way through run_io.  Large batches: whole arrays up with one synchronous copy each,
  // one chain of kernels, whole arrays down (1.30 ms for 1M x 32 bytes here, from pageable and page-locked memory alike:
  // the copies run at the link's 53 GB/s).  With fmx_config_set("pipeline", "on") large batches in page-locked caller
  // memory (fmx_host_alloc, or the caller's own hipHostMalloc / registered pages) are pipelined instead: the batch is
  // cut into chunks of patterns, chunk j's bytes and offsets go up on one stream, are searched on a second and come
  // back on a third, so the copies of one chunk run beside the kernels of another and both directions of the link are
  // busy.  Offsets stay absolute: every chunk is copied to its own place of one device image of the batch.  Off by
  // default: on this platform asynchronous copies are slower than synchronous ones (one chunk, no overlap at all:
  // 1.67 ms) and the overlap does not reliably win that back -- 1.08 to 1.7 ms with 4 chunks from box to box.
  if (k < kPipelineMin) {
    const HostIn ins[] = {{total ? pat + lo : nullptr, (size_t)total}, {offp, (k + 1) * 8}};
    const HostOut outs[] = {{sp, k * 8}, {ep, k * 8}};
    return run_io(h, ins, 2, outs, 2, [&](hipStream_t st, const void *const *di, void *const *dout) {
      return launch_search(h, di[0], di[1], dout[0], dout[1], k, st);
    });
  }
  Call c0(h), c1(h), c2(h);
  if ((rc = c0.init()) != FMX_OK || (rc = c1.init()) != FMX_OK || (rc = c2.init()) != FMX_OK) return rc;
  DevBuf d_pat, d_off, d_sp, d_ep;
  HIP_TRY(c0.alloc(d_pat, (size_t)total + 16), "hipMalloc");
  HIP_TRY(c0.alloc(d_off, (k + 1) * 8), "hipMalloc");
  HIP_TRY(c0.alloc(d_sp, k * 8), "hipMalloc");
  HIP_TRY(c0.alloc(d_ep, k * 8), "hipMalloc");
  auto pinned1 = [](const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
  };
  // both ends of a buffer: a partly registered one must not take the asynchronous path
  auto pinned = [&](const void *p, size_t bytes) {
    return pinned1(p) && (bytes == 0 || pinned1(static_cast<const uint8_t *>(p) + bytes - 1));
  };
  if (!g_pipeline.load(std::memory_order_relaxed) || !(pinned(sp, k * 8) && pinned(ep, k * 8) && pinned(offp, (k + 1) * 8) && (!total || pinned(pat + lo, (size_t)total)))) {
    // Pageable caller memory: "asynchronous" copies of it are staged piecewise by the runtime and block the calling
    // thread (measured: 8 chunks, 0.37 ms each, nothing overlapped), while one synchronous copy per array runs at
    // link speed.  So: whole arrays up, one kernel, whole arrays down.
    if (total) HIP_TRY(hipMemcpy(d_pat.p, pat + lo, (size_t)total, hipMemcpyHostToDevice), "H2D(patterns)");
    HIP_TRY(hipMemcpy(d_off.p, offp, (k + 1) * 8, hipMemcpyHostToDevice), "H2D(offsets)");
    bool ok = true;
    HIP_TRY(check_offsets(h, d_off.p, k, c0.stream(), &ok), "k_check_offsets");
    if (!ok) return arg_fail("pattern offsets must be non-decreasing");
    rc = c0.timed([&](hipStream_t st, EventPair &ev) {
      HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
      HIP_TRY(launch_search(h, d_pat.p, d_off.p, d_sp.p, d_ep.p, k, st), "k_search");
      HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
      return (int)FMX_OK;
    });
    if (rc != FMX_OK) return rc;
    HIP_TRY(hipMemcpy(sp, d_sp.p, k * 8, hipMemcpyDeviceToHost), "D2H(sp)");
    HIP_TRY(hipMemcpy(ep, d_ep.p, k * 8, hipMemcpyDeviceToHost), "D2H(ep)");
    return FMX_OK;
  }
  static const size_t nchunk = [] { const char *e = getenv("FMX_PIPE_CHUNKS"); const int v = e ? atoi(e) : 4; return (size_t)(v < 1 ? 1 : (v > 64 ? 64 : v)); }();
  static const bool trace = getenv("FMX_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto mark = [&](const char *what, size_t j) {
    if (trace) fprintf(stderr, "[fmx] search_batch %s %zu +%.3f ms\n", what, j,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  // Three streams, one per engine: `up` carries the chunks' uploads in order, one event after each; `run` waits for
  // chunk j's event and searches it; `down` waits for that search and sends the chunk's intervals back.  So the link
  // is busy in both directions beside the kernels, and no stream alternates between kernels and copies (every such
  // change of engine inside one stream costs a hand-over of ~0.1 ms: with the copies behind each chunk's kernels on
  // `run`, eight chunks took 2.4 ms where the kernels are 0.23 ms and the copies 0.75 ms).
  hipStream_t up = c1.stream(), run = c0.stream(), down = c2.stream();
  // Whatever ends this call early -- a failed enqueue, a bad offset -- the streams are drained first: copies still in
  // flight read and write the CALLER's buffers, and the device buffers go back to the handle's pool on return.
  struct Drain {
    hipStream_t a, b, c;
    bool armed = true;
    ~Drain() { if (armed) { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(c); } }
  } drain{up, run, down};
  for (size_t j = 0; j <= nchunk; j++)            // the chunks' byte ranges come from these: checked before any copy
    if (offp[k * j / nchunk] > total || (j && offp[k * j / nchunk] < offp[k * (j - 1) / nchunk]))
      return arg_fail("pattern offsets must be non-decreasing");
  hipEvent_t *cev = nullptr, *kev = nullptr;
  HIP_TRY(c0.chunk_events(nchunk, &cev), "hipEventCreate");
  HIP_TRY(c2.chunk_events(nchunk, &kev), "hipEventCreate");
  HIP_TRY(hipEventRecord(c0.ev_a(), run), "hipEventRecord");
  HIP_TRY(hipStreamWaitEvent(up, c0.ev_a(), 0), "hipStreamWaitEvent");      // the buffers' previous users are done
  for (size_t j = 0; j < nchunk; j++) {
    const size_t a = k * j / nchunk, b = k * (j + 1) / nchunk;
    const uint64_t b0 = offp[a], b1 = offp[b];
    if (b1 > b0)
      HIP_TRY(hipMemcpyAsync((uint8_t *)d_pat.p + b0, pat + lo + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, up), "H2D(patterns)");
    // chunk j needs offsets a .. b: entry b is also the next chunk's first, copied by both (same value)
    HIP_TRY(hipMemcpyAsync((uint64_t *)d_off.p + a, offp + a, (b - a + 1) * 8, hipMemcpyHostToDevice, up), "H2D(offsets)");
    HIP_TRY(hipEventRecord(cev[j], up), "hipEventRecord");
  }
  mark("uploads enqueued", nchunk);
  for (size_t j = 0; j < nchunk; j++) {
    const size_t a = k * j / nchunk, b = k * (j + 1) / nchunk;
    HIP_TRY(hipStreamWaitEvent(run, cev[j], 0), "hipStreamWaitEvent");
    if (a == b) continue;
    if (!monotonic(a, b)) return arg_fail("pattern offsets must be non-decreasing");      // while the uploads are under way
    HIP_TRY(launch_search(h, d_pat.p, (const uint64_t *)d_off.p + a, (uint64_t *)d_sp.p + a, (uint64_t *)d_ep.p + a, b - a, run),
            "k_search");
    HIP_TRY(hipEventRecord(kev[j], run), "hipEventRecord");
    HIP_TRY(hipStreamWaitEvent(down, kev[j], 0), "hipStreamWaitEvent");
    HIP_TRY(hipMemcpyAsync(sp + a, (uint64_t *)d_sp.p + a, (b - a) * 8, hipMemcpyDeviceToHost, down), "D2H(sp)");
    HIP_TRY(hipMemcpyAsync(ep + a, (uint64_t *)d_ep.p + a, (b - a) * 8, hipMemcpyDeviceToHost, down), "D2H(ep)");
    mark("chunk enqueued", j);
  }
  HIP_TRY(hipEventRecord(c2.ev_a(), down), "hipEventRecord");
  HIP_TRY(hipStreamWaitEvent(run, c2.ev_a(), 0), "hipStreamWaitEvent");     // `run` ends when the last download has
  HIP_TRY(hipEventRecord(c0.ev_b(), run), "hipEventRecord");
  HIP_TRY(hipStreamSynchronize(run), "hipStreamSynchronize");      // `run` waited for every upload and download: all three are idle
  drain.armed = false;
  mark("synchronized", 0);
  float ms = 0;
  if (hipEventElapsedTime(&ms, c0.ev_a(), c0.ev_b()) == hipSuccess) {
    std::lock_guard<std::mutex> lk(h->mu);
    h->last_kernel_ms = ms;            // for a pipelined call: the whole device side, copies included
    h->launches += nchunk;
  }
  return FMX_OK;
}

int fmx_search_batch_multi(fmx_index *const *idxs, size_t n_idx, const uint8_t *pat, const uint64_t *off,
                           uint64_t *sp, uint64_t *ep, size_t k) {
  if (!idxs || !n_idx || (k && (!off || !sp || !ep))) return arg_fail("null argument");
  for (size_t r = 0; r < n_idx; r++) {
    if (!idxs[r]) return arg_fail("null index handle");
    if (H(idxs[r])->n != H(idxs[0])->n || H(idxs[r])->eof != H(idxs[0])->eof) return arg_fail("the handles are not replicas of one index");
  }
  if (!k) return FMX_OK;
  for (size_t q = 0; q < k; q++)
    if (off[q + 1] < off[q]) return arg_fail("pattern offsets must be non-decreasing");
  // contiguous slices balanced by pattern bytes (by count when every pattern is empty)
  std::vector<size_t> cut(n_idx + 1, k);
  cut[0] = 0;
  const uint64_t total = off[k] - off[0];
  for (size_t r = 1; r < n_idx; r++) {
    if (!total) { cut[r] = k * r / n_idx; continue; }
    const uint64_t want = off[0] + total / n_idx * r;
    cut[r] = (size_t)(std::lower_bound(off, off + k + 1, want) - off);
    if (cut[r] < cut[r - 1]) cut[r] = cut[r - 1];
    if (cut[r] > k) cut[r] = k;
  }
  // slice r runs on handle r's own host thread (kept with the handle: a call wakes it, it does not create it)
  std::vector<int> rc(n_idx, FMX_OK);
  std::vector<std::string> msg(n_idx);
  std::vector<Worker *> busy;
  for (size_t r = 0; r < n_idx; r++) {
    const size_t a = cut[r], b = cut[r + 1];
    if (a == b) continue;
    Worker *w = worker_of(H(idxs[r]));
    w->submit([&, r, a, b]() {
      rc[r] = fmx_search_batch(idxs[r], pat, off + a, sp + a, ep + a, b - a);
      if (rc[r] != FMX_OK) msg[r] = fmx_last_error();        // the message is thread-local: carry it over
    });
    busy.push_back(w);
  }
  for (Worker *w : busy) w->wait();
  for (size_t r = 0; r < n_idx; r++)
    if (rc[r] != FMX_OK) { g_err = msg[r]; return rc[r]; }
  return FMX_OK;
}

int fmx_gather(fmx_index *const *idxs, size_t n_idx, const void *const *d_src, const size_t *cnt, size_t elem,
               void *dst) {
  if (!idxs || !n_idx || !d_src || !cnt || !elem || !dst) return arg_fail("null argument");
  std::vector<CallCtx *> ctx(n_idx, nullptr);
  size_t at = 0;
  int rc = FMX_OK;
  for (size_t r = 0; r < n_idx && rc == FMX_OK; r++) {
    if (!idxs[r]) { rc = arg_fail("null index handle"); break; }
    const Index *h = H(idxs[r]);
    if (cnt[r]) {
      if (!d_src[r]) { rc = arg_fail("null slice pointer"); break; }
      if ((rc = use_device(h)) != FMX_OK) break;
      ctx[r] = ctx_acquire(h);
      if (!ctx[r]) { rc = FMX_ERR_HIP; break; }
      const hipError_t e = hipMemcpyAsync(static_cast<uint8_t *>(dst) + at * elem, d_src[r], cnt[r] * elem,
                                          hipMemcpyDeviceToHost, ctx[r]->stream);
      if (e != hipSuccess) rc = hip_fail(e, "D2H(gather)");
    }
    at += cnt[r];
  }
  for (size_t r = 0; r < n_idx; r++) {
    if (!ctx[r]) continue;
    const Index *h = H(idxs[r]);
    (void)hipSetDevice(h->device);
    const hipError_t e = hipStreamSynchronize(ctx[r]->stream);
    if (e != hipSuccess && rc == FMX_OK) rc = hip_fail(e, "hipStreamSynchronize(gather)");
    ctx_release(h, ctx[r]);
  }
  return rc;
}

int fmx_prev_range_batch(const fmx_index *idx, const uint64_t *sp, const uint64_t *ep, const uint8_t *c,
                         uint64_t *sp1, uint64_t *ep1, size_t k) {
  if (!idx || (k && (!sp || !ep || !c || !sp1 || !ep1))) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc || !k) return rc;
  for (size_t q = 0; q < k; q++)
    if (sp[q] > ep[q] || ep[q] > h->n) return arg_fail("need sp <= ep <= n");
  const HostIn ins[] = {{sp, k * 8}, {ep, k * 8}, {c, k}};
  const HostOut outs[] = {{sp1, k * 8}, {ep1, k * 8}};
  return run_io(h, ins, 3, outs, 2, [&](hipStream_t st, const void *const *di, void *const *dout) {
    return launch_prev_range(h, di[0], di[1], di[2], dout[0], dout[1], k, st);
  });
}

// getIntervalPrevRange, findex.scala:37-51: one prev_range batch over the class, then the
// reference's filter (occ1 < occ2) and its prepend order (descending c).
int fmx_interval_prev_range(const fmx_index *idx, uint64_t sp, uint64_t ep, int cstart, int cend, uint64_t *out_sp,
                            uint64_t *out_ep, uint8_t *out_c, size_t *n_out) {
  if (!idx || !n_out) return arg_fail("null argument");
  *n_out = 0;
  if (cstart < 0 || cstart > 255 || (cstart <= cend && cend > 255))
    return arg_fail("symbol out of range (reference: ArrayIndexOutOfBounds)");
  if (cend < cstart) return FMX_OK;
  const size_t k = (size_t)(cend - cstart + 1);
  if (!out_sp || !out_ep) return arg_fail("null argument");
  std::vector<uint64_t> vsp(k, sp), vep(k, ep), sp1(k), ep1(k);
  std::vector<uint8_t> vc(k);
  for (size_t j = 0; j < k; j++) vc[j] = (uint8_t)(cstart + (int)j);
  int rc = fmx_prev_range_batch(idx, vsp.data(), vep.data(), vc.data(), sp1.data(), ep1.data(), k);
  if (rc) return rc;
  size_t w = 0;
  for (size_t j = k; j-- > 0;) {
    if (sp1[j] < ep1[j]) {
      out_sp[w] = sp1[j];
      out_ep[w] = ep1[j];
      if (out_c) out_c[w] = vc[j];
      w++;
    }
  }
  *n_out = w;
  return FMX_OK;
}

int fmx_lf_walk_batch(const fmx_index *idx, const uint64_t *rows, size_t k, uint32_t len, uint8_t *out_bytes,
                      uint64_t *end_rows) {
  if (!idx || (k && !rows)) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc || !k) return rc;
  for (size_t q = 0; q < k; q++)
    if (rows[q] >= h->n) return arg_fail("row out of range (reference: seek past .bwt / ArrayIndexOutOfBounds)");
  const HostIn ins[] = {{rows, k * 8}};
  const HostOut outs[] = {{out_bytes, out_bytes ? k * (size_t)len : 0}, {end_rows, end_rows ? k * 8 : 0}};
  return run_io(h, ins, 1, outs, 2, [&](hipStream_t st, const void *const *di, void *const *dout) {
    return launch_lf_walk(h, di[0], k, len, out_bytes ? dout[0] : nullptr, end_rows ? dout[1] : nullptr, st);
  });
}

int fmx_prev_substr(const fmx_index *idx, uint64_t sp, uint32_t len, uint8_t *out) {
  if (len && !out) return arg_fail("null argument");
  return fmx_lf_walk_batch(idx, &sp, 1, len, out, nullptr);
}

int fmx_psi_batch(const fmx_index *idx, const uint64_t *rows, uint64_t *out, size_t k) {
  if (!idx || (k && (!rows || !out))) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc || !k) return rc;
  for (size_t q = 0; q < k; q++)
    if (rows[q] >= h->n) return arg_fail("row out of range");
  const HostIn ins[] = {{rows, k * 8}};
  const HostOut outs[] = {{out, k * 8}};
  return run_io(h, ins, 1, outs, 1, [&](hipStream_t st, const void *const *di, void *const *dout) {
    return launch_psi(h, di[0], dout[0], k, st);
  });
}

// nextSubstr for k rows in one call: out is k x len bytes (row q's string at q*len, out_len[q] bytes of it used).
int fmx_next_substr_batch(const fmx_index *idx, const uint64_t *rows, size_t k, uint32_t len, uint8_t *out,
                          uint32_t *out_len) {
  if (!idx || (k && (!rows || !out_len || (len && !out)))) return arg_fail("null argument");
  const Index *h = H(idx);
  for (size_t q = 0; q < k; q++) {
    out_len[q] = 0;
    if (rows[q] >= h->n) return arg_fail("row out of range");
  }
  int rc = use_device(h);
  if (rc || !k) return rc;
  const HostIn ins[] = {{rows, k * 8}};
  const HostOut outs[] = {{out, k * (size_t)len}, {out_len, k * 4}};
  rc = run_io(h, ins, 1, outs, 2, [&](hipStream_t st, const void *const *di, void *const *dout) {
    return launch_next_substr(h, di[0], k, len, dout[0], dout[1], st);
  });
  if (rc) return rc;
  for (size_t q = 0; q < k; q++)                  // the kernel writes in walk order: ret.reverse, bwtmerger.scala:404
    std::reverse(out + q * (size_t)len, out + q * (size_t)len + out_len[q]);
  return FMX_OK;
}

int fmx_next_substr(const fmx_index *idx, uint64_t sp, uint32_t len, uint8_t *out, uint32_t *out_len) {
  if (!out_len) return arg_fail("null argument");
  return fmx_next_substr_batch(idx, &sp, 1, len, out, out_len);
}

int fmx_extract(const fmx_index *idx, uint64_t row, uint32_t len, int direction, uint8_t *out, uint32_t *out_len) {
  if (!out_len) return arg_fail("null argument");
  if (direction == 0) return arg_fail("direction must be positive (nextSubstr) or negative (prevSubstr)");
  if (direction > 0) return fmx_next_substr(idx, row, len, out, out_len);
  *out_len = 0;
  int rc = fmx_prev_substr(idx, row, len, out);
  if (rc == FMX_OK) *out_len = len;
  return rc;
}

// ---------------------------------------------------------------- .fm writer (FMCreator.create)
// Wire format, bwtmerger.scala:483-485,476-481: byte elSize (=4), int64 big-endian size, then
// `size` big-endian int32 entries.  elSize 8 is `???` in the reference (:465-469), so n must stay
// below 0xffffffff here too.
int fmx_write_fm(const fmx_index *idx, const char *path) {
  if (!idx || !path) return arg_fail("null argument");
  const Index *h = H(idx);
  if (h->n >= 0xffffffffull) { g_err = "the .fm format has no 8-byte entries (bwtmerger.scala:465-469)"; return FMX_ERR_UNSUPPORTED; }
  int rc = use_device(h);
  if (rc) return rc;
  Call call(h);
  if ((rc = call.init()) != FMX_OK) return rc;
  DevBuf dfm;
  HIP_TRY(call.alloc(dfm, h->n * 4), "hipMalloc(fm)");
  rc = call.timed([&](hipStream_t st, EventPair &ev) {
    HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
    HIP_TRY(launch_fm_fill(h, dfm.p, st), "k_fm_fill");
    HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
    return (int)FMX_OK;
  });
  if (rc) return rc;
  FILE *f = std::fopen(path, "wb");
  if (!f) { g_err = std::string("cannot create ") + path; return FMX_ERR_IO; }
  std::unique_ptr<FILE, int (*)(FILE *)> guard(f, std::fclose);
  uint8_t hdr[9];
  hdr[0] = 4;
  for (int i = 0; i < 8; i++) hdr[1 + i] = (uint8_t)(h->n >> (56 - 8 * i));
  if (std::fwrite(hdr, 1, 9, f) != 9) { g_err = "short write"; return FMX_ERR_IO; }
  const size_t chunk = 64u << 20;
  std::vector<uint8_t> buf(std::min<uint64_t>(chunk, h->n * 4));
  for (uint64_t o = 0; o < h->n * 4; o += chunk) {
    const size_t len = (size_t)std::min<uint64_t>(chunk, h->n * 4 - o);
    HIP_TRY(hipMemcpy(buf.data(), (const uint8_t *)dfm.p + o, len, hipMemcpyDeviceToHost), "D2H(fm)");
    if (std::fwrite(buf.data(), 1, len, f) != len) { g_err = "short write"; return FMX_ERR_IO; }
  }
  return FMX_OK;
}

// ---------------------------------------------------------------- statistics
int fmx_stats(const fmx_index *idx, fmx_stats_t *out) {
  if (!idx || !out) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc) return rc;
  unsigned long long cnt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  std::memset(out, 0, sizeof *out);
  HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  {   // the counters live in per-workgroup slots (fmx_device.h): sum them
    std::vector<unsigned long long> slots((size_t)kCounterSlots * kCounterStride);
    HIP_TRY(hipMemcpy(slots.data(), h->d_counters, kCounterBytes, hipMemcpyDeviceToHost), "D2H(counters)");
    for (uint32_t sl = 0; sl < kCounterSlots; sl++)
      for (int j = 0; j < 12; j++) cnt[j] += slots[(size_t)sl * kCounterStride + j];
  }
  std::lock_guard<std::mutex> lk(h->mu);
  out->rank_queries = cnt[0];
  out->backward_steps = cnt[1];
  out->launches = h->launches;
  out->last_kernel_ms = h->last_kernel_ms;
  out->index_bytes = h->index_bytes + h->sel_bytes + h->kt_bytes + h->jump_bytes + h->row1_bytes + h->row3_bytes;
  out->jump_lookups = cnt[10];
  out->jump_bytes = h->jump_bytes;
  out->row_lookups = cnt[11];
  out->row_bytes = h->row1_bytes + h->row3_bytes;
  out->n_blocks = h->nblocks;
  out->n_symbols = h->nslots;
  out->block_bytes = h->layout == kLayoutBytes ? kByteBlock + 4 : kBlockBytes;
  out->layout = h->layout;
  out->search_requests = cnt[2];
  out->frontier_requests = cnt[3];
  out->frontier_queue_writes = cnt[4];
  out->frontier_results = cnt[5];
  out->frontier_elements = cnt[6];
  out->frontier_queue_reads = cnt[7];
  out->frontier_records = cnt[8];
  out->ktab_lookups = cnt[9];
  out->ktab_k = h->kt.k;
  out->search_residency = h->search_residency.load();
  out->jump_chars = h->jump_bytes ? h->jump_chars : 0;
  out->build_ms = h->build_ms;
  out->tables_build_ms = h->tables_ms;
  out->tables_alloc_ms = (double)h->tables_alloc_us.load(std::memory_order_relaxed) * 1e-3;
  out->peak_table_build_bytes = h->peak_table_build_bytes.load(std::memory_order_relaxed);
  out->patterns_seen = h->patterns_seen.load(std::memory_order_relaxed);
  out->tables_held_bytes = h->tables_held.load(std::memory_order_relaxed);
  out->hbm_free_after_tables = h->hbm_free_after_tables.load(std::memory_order_relaxed);
  {
    const uint32_t ppm = h->policy.budget_ppb.load(std::memory_order_relaxed);
    uint64_t budget = h->policy.budget_bytes.load(std::memory_order_relaxed);
    size_t free_b = 0, total_b = 0;
    if (ppm && hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = (uint64_t)((double)(free_b + out->tables_held_bytes) * ((double)ppm * 1e-9));
    out->table_budget_bytes = budget;
  }
  return FMX_OK;
}

int fmx_last_kernel_ms(const fmx_index *idx, double *ms) {
  if (!idx || !ms) return arg_fail("null argument");
  const Index *h = H(idx);
  std::lock_guard<std::mutex> lk(h->mu);
  *ms = h->last_kernel_ms;
  return FMX_OK;
}

int fmx_stats_reset(fmx_index *idx) {
  if (!idx) return arg_fail("null argument");
  Index *h = H(idx);
  int rc = use_device(h);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  HIP_TRY(hipMemset(h->d_counters, 0, kCounterBytes), "memset(counters)");
  std::lock_guard<std::mutex> lk(h->mu);
  h->launches = 0;
  h->last_kernel_ms = 0;
  return FMX_OK;
}

}  // extern "C"

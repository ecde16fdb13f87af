// fmx_api.cpp -- the C ABI of libfmx.so (include/fmx.h): file loaders, handle lifetime, and the
// host-pointer wrappers around the device entry points.  No compute happens on the host.
#include <fmx.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "fmx_host.h"
#include "fmx_regex.h"

namespace fmx {

static thread_local std::string g_err;

void set_error(const std::string &msg) { g_err = msg; }

int hip_fail(hipError_t e, const char *what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return FMX_ERR_HIP;
}

static int g_layout_pref = -1;
int layout_preference() { return g_layout_pref; }

static int arg_fail(const char *msg) {
  g_err = msg;
  return FMX_ERR_ARG;
}

// ---- device buffers with scope lifetime
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

struct StreamGuard {
  hipStream_t s = nullptr;
  ~StreamGuard() { if (s) (void)hipStreamDestroy(s); }
};

struct EventPair {
  hipEvent_t a = nullptr, b = nullptr;
  ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
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

// ---- file formats
static uint64_t rd_u64(const uint8_t *p, bool be) {
  uint64_t v = 0;
  if (be) for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
  else for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
  return v;
}

// BWTLoader, bwtmerger.scala:144-174: int64 size, int64 eof, then `size` bytes; size+16 == file length.
static int load_bwt(const char *path, bool be, std::vector<uint8_t> &bwt, uint64_t &n, uint64_t &eof) {
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
  try { bwt.resize(n); } catch (...) { g_err = "out of host memory"; return FMX_ERR_NOMEM; }
  if (n && std::fread(bwt.data(), 1, n, f) != n) { g_err = std::string("short read on ") + path; return FMX_ERR_IO; }
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

static void destroy(Index *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->d_bv) (void)hipFree(h->d_bv);
  if (h->d_chk) (void)hipFree(h->d_chk);
  if (h->d_sup) (void)hipFree(h->d_sup);
  if (h->d_bwt) (void)hipFree(h->d_bwt);
  if (h->d_cf) (void)hipFree(h->d_cf);
  if (h->d_slot) (void)hipFree(h->d_slot);
  if (h->d_counters) (void)hipFree(h->d_counters);
  delete h;
}

// Common tail of the three open flavours: `src` is host or device memory holding n BWT bytes.
static int open_common(const void *src, bool src_on_device, uint64_t n, uint64_t eof, const int64_t *counts,
                       int device, hipStream_t user_stream, fmx_index **out) {
  if (!out) return arg_fail("out is null");
  *out = nullptr;
  if (!src && n) return arg_fail("bwt is null");
  if (n < 1 || eof >= n) return arg_fail("need n >= 1 and eof < n");
  if (n >= (1ull << 38)) { g_err = "n >= 2^38 is not supported"; return FMX_ERR_UNSUPPORTED; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) { g_err = "no HIP device available (libfmx has no CPU fallback)"; return FMX_ERR_HIP; }
  if (device < 0 || device >= ndev) return arg_fail("device index out of range");
  HIP_TRY(hipSetDevice(device), "hipSetDevice");
  Index *h = new (std::nothrow) Index();
  if (!h) { g_err = "out of host memory"; return FMX_ERR_NOMEM; }
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
    e = hipMemcpyAsync(h->d_bwt, src, n, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { rc = hip_fail(e, "copy bwt"); break; }
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

// Runs `enqueue` on a private stream bracketed by HIP events and records the device time.
template <class F>
static int timed(const Index *h, F enqueue) {
  StreamGuard sg;
  EventPair ev;
  HIP_TRY(hipStreamCreate(&sg.s), "hipStreamCreate");
  HIP_TRY(hipEventCreate(&ev.a), "hipEventCreate");
  HIP_TRY(hipEventCreate(&ev.b), "hipEventCreate");
  int rc = enqueue(sg.s, ev);
  if (rc != FMX_OK) return rc;
  HIP_TRY(hipStreamSynchronize(sg.s), "hipStreamSynchronize");
  float ms = 0;
  if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
    std::lock_guard<std::mutex> lk(h->mu);
    h->last_kernel_ms = ms;
    h->launches++;
  }
  return FMX_OK;
}

}  // namespace fmx

using namespace fmx;

extern "C" {

const char *fmx_last_error(void) { return g_err.c_str(); }
int fmx_abi_version(void) { return FMX_ABI_VERSION; }

int fmx_config_set(const char *key, const char *value) {
  if (!key || !value) return arg_fail("null argument");
  if (std::strcmp(key, "layout") == 0) {
    if (std::strcmp(value, "auto") == 0) g_layout_pref = -1;
    else if (std::strcmp(value, "onehot") == 0) g_layout_pref = (int)kLayoutOneHot;
    else if (std::strcmp(value, "bytes") == 0) g_layout_pref = (int)kLayoutBytes;
    else return arg_fail("layout must be auto, onehot or bytes");
    return FMX_OK;
  }
  return arg_fail("unknown configuration key");
}

int fmx_device_count(int *count) {
  if (!count) return arg_fail("count is null");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return FMX_OK;
}

int fmx_open(const char *bwt_path, const char *aux_path, int big_endian, int device, fmx_index **out) {
  if (!bwt_path || !aux_path) return arg_fail("path is null");
  std::vector<uint8_t> bwt;
  uint64_t n = 0, eof = 0;
  int64_t counts[256];
  int rc = load_bwt(bwt_path, big_endian != 0, bwt, n, eof);
  if (rc != FMX_OK) return rc;
  rc = load_aux(aux_path, big_endian != 0, counts);
  if (rc != FMX_OK) return rc;
  return open_common(bwt.data(), false, n, eof, counts, device, nullptr, out);
}

int fmx_open_mem(const uint8_t *bwt, uint64_t n, uint64_t eof, const int64_t counts[256], int device,
                 fmx_index **out) {
  if (!counts) return arg_fail("counts is null");
  return open_common(bwt, false, n, eof, counts, device, nullptr, out);
}

int fmx_open_dev(const void *d_bwt, uint64_t n, uint64_t eof, const int64_t *counts_or_null, int device, void *stream,
                 fmx_index **out) {
  return open_common(d_bwt, true, n, eof, counts_or_null, device, (hipStream_t)stream, out);
}

int fmx_close(fmx_index *idx) {
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

int fmx_search_batch_dev(const fmx_index *idx, const void *d_pat, const void *d_off, void *d_sp, void *d_ep, size_t k,
                         void *stream) {
  if (!idx || (k && (!d_off || !d_sp || !d_ep))) return arg_fail("null argument");
  int rc = use_device(H(idx));
  if (rc) return rc;
  HIP_TRY(launch_search(H(idx), d_pat, d_off, d_sp, d_ep, k, (hipStream_t)stream), "k_search");
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

// ---------------------------------------------------------------- host-pointer entry points
int fmx_occ_batch(const fmx_index *idx, const uint8_t *c, const int64_t *i, uint64_t *out, size_t k) {
  if (!idx || (k && (!c || !i || !out))) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc || !k) return rc;
  DevBuf dc, di, dout;
  HIP_TRY(dc.alloc(k), "hipMalloc");
  HIP_TRY(di.alloc(k * 8), "hipMalloc");
  HIP_TRY(dout.alloc(k * 8), "hipMalloc");
  return timed(h, [&](hipStream_t st, EventPair &ev) {
    HIP_TRY(hipMemcpyAsync(dc.p, c, k, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipMemcpyAsync(di.p, i, k * 8, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
    HIP_TRY(launch_occ(h, dc.p, di.p, dout.p, k, st), "k_occ");
    HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
    HIP_TRY(hipMemcpyAsync(out, dout.p, k * 8, hipMemcpyDeviceToHost, st), "D2H");
    return (int)FMX_OK;
  });
}

int fmx_search_batch(const fmx_index *idx, const uint8_t *pat, const uint64_t *off, uint64_t *sp, uint64_t *ep,
                     size_t k) {
  if (!idx || (k && (!off || !sp || !ep))) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc || !k) return rc;
  for (size_t q = 0; q < k; q++)
    if (off[q + 1] < off[q]) return arg_fail("pattern offsets must be non-decreasing");
  const uint64_t lo = off[0], total = off[k] - off[0];
  if (total && !pat) return arg_fail("pat is null");
  // offsets are rebased so that only the bytes in use travel
  std::vector<uint64_t> roff(k + 1);
  for (size_t q = 0; q <= k; q++) roff[q] = off[q] - lo;
  DevBuf dpat, doff, dsp, dep;
  HIP_TRY(dpat.alloc(total), "hipMalloc");
  HIP_TRY(doff.alloc((k + 1) * 8), "hipMalloc");
  HIP_TRY(dsp.alloc(k * 8), "hipMalloc");
  HIP_TRY(dep.alloc(k * 8), "hipMalloc");
  return timed(h, [&](hipStream_t st, EventPair &ev) {
    if (total) HIP_TRY(hipMemcpyAsync(dpat.p, pat + lo, total, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipMemcpyAsync(doff.p, roff.data(), (k + 1) * 8, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
    HIP_TRY(launch_search(h, dpat.p, doff.p, dsp.p, dep.p, k, st), "k_search");
    HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
    HIP_TRY(hipMemcpyAsync(sp, dsp.p, k * 8, hipMemcpyDeviceToHost, st), "D2H");
    HIP_TRY(hipMemcpyAsync(ep, dep.p, k * 8, hipMemcpyDeviceToHost, st), "D2H");
    return (int)FMX_OK;
  });
}

int fmx_prev_range_batch(const fmx_index *idx, const uint64_t *sp, const uint64_t *ep, const uint8_t *c,
                         uint64_t *sp1, uint64_t *ep1, size_t k) {
  if (!idx || (k && (!sp || !ep || !c || !sp1 || !ep1))) return arg_fail("null argument");
  const Index *h = H(idx);
  int rc = use_device(h);
  if (rc || !k) return rc;
  for (size_t q = 0; q < k; q++)
    if (sp[q] > ep[q] || ep[q] > h->n) return arg_fail("need sp <= ep <= n");
  DevBuf dsp, dep, dc, dsp1, dep1;
  HIP_TRY(dsp.alloc(k * 8), "hipMalloc");
  HIP_TRY(dep.alloc(k * 8), "hipMalloc");
  HIP_TRY(dc.alloc(k), "hipMalloc");
  HIP_TRY(dsp1.alloc(k * 8), "hipMalloc");
  HIP_TRY(dep1.alloc(k * 8), "hipMalloc");
  return timed(h, [&](hipStream_t st, EventPair &ev) {
    HIP_TRY(hipMemcpyAsync(dsp.p, sp, k * 8, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipMemcpyAsync(dep.p, ep, k * 8, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipMemcpyAsync(dc.p, c, k, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
    HIP_TRY(launch_prev_range(h, dsp.p, dep.p, dc.p, dsp1.p, dep1.p, k, st), "k_prev_range");
    HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
    HIP_TRY(hipMemcpyAsync(sp1, dsp1.p, k * 8, hipMemcpyDeviceToHost, st), "D2H");
    HIP_TRY(hipMemcpyAsync(ep1, dep1.p, k * 8, hipMemcpyDeviceToHost, st), "D2H");
    return (int)FMX_OK;
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
  DevBuf drows, dout, dend;
  HIP_TRY(drows.alloc(k * 8), "hipMalloc");
  if (out_bytes) HIP_TRY(dout.alloc(k * (size_t)len), "hipMalloc");
  if (end_rows) HIP_TRY(dend.alloc(k * 8), "hipMalloc");
  return timed(h, [&](hipStream_t st, EventPair &ev) {
    HIP_TRY(hipMemcpyAsync(drows.p, rows, k * 8, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
    HIP_TRY(launch_lf_walk(h, drows.p, k, len, dout.p, dend.p, st), "k_lf_walk");
    HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
    if (out_bytes && len) HIP_TRY(hipMemcpyAsync(out_bytes, dout.p, k * (size_t)len, hipMemcpyDeviceToHost, st), "D2H");
    if (end_rows) HIP_TRY(hipMemcpyAsync(end_rows, dend.p, k * 8, hipMemcpyDeviceToHost, st), "D2H");
    return (int)FMX_OK;
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
  DevBuf drows, dout;
  HIP_TRY(drows.alloc(k * 8), "hipMalloc");
  HIP_TRY(dout.alloc(k * 8), "hipMalloc");
  return timed(h, [&](hipStream_t st, EventPair &ev) {
    HIP_TRY(hipMemcpyAsync(drows.p, rows, k * 8, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
    HIP_TRY(launch_psi(h, drows.p, dout.p, k, st), "k_psi");
    HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
    HIP_TRY(hipMemcpyAsync(out, dout.p, k * 8, hipMemcpyDeviceToHost, st), "D2H");
    return (int)FMX_OK;
  });
}

int fmx_next_substr(const fmx_index *idx, uint64_t sp, uint32_t len, uint8_t *out, uint32_t *out_len) {
  if (!idx || !out_len || (len && !out)) return arg_fail("null argument");
  const Index *h = H(idx);
  *out_len = 0;
  if (sp >= h->n) return arg_fail("row out of range");
  int rc = use_device(h);
  if (rc) return rc;
  DevBuf dsp, dout, dlen;
  HIP_TRY(dsp.alloc(8), "hipMalloc");
  HIP_TRY(dout.alloc(len), "hipMalloc");
  HIP_TRY(dlen.alloc(4), "hipMalloc");
  std::vector<uint8_t> tmp(len ? len : 1);
  uint32_t w = 0;
  rc = timed(h, [&](hipStream_t st, EventPair &ev) {
    HIP_TRY(hipMemcpyAsync(dsp.p, &sp, 8, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(hipEventRecord(ev.a, st), "hipEventRecord");
    HIP_TRY(launch_next_substr(h, dsp.p, 1, len, dout.p, dlen.p, st), "k_next_substr");
    HIP_TRY(hipEventRecord(ev.b, st), "hipEventRecord");
    if (len) HIP_TRY(hipMemcpyAsync(tmp.data(), dout.p, len, hipMemcpyDeviceToHost, st), "D2H");
    HIP_TRY(hipMemcpyAsync(&w, dlen.p, 4, hipMemcpyDeviceToHost, st), "D2H");
    return (int)FMX_OK;
  });
  if (rc) return rc;
  for (uint32_t j = 0; j < w; j++) out[j] = tmp[w - 1 - j];   // ret.reverse, bwtmerger.scala:404
  *out_len = w;
  return FMX_OK;
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
  DevBuf dfm;
  HIP_TRY(dfm.alloc(h->n * 4), "hipMalloc(fm)");
  rc = timed(h, [&](hipStream_t st, EventPair &ev) {
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
  unsigned long long cnt[4] = {0, 0, 0, 0};
  HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  {   // the counters live in per-workgroup slots (fmx_device.h): sum them
    std::vector<unsigned long long> slots((size_t)kCounterSlots * kCounterStride);
    HIP_TRY(hipMemcpy(slots.data(), h->d_counters, kCounterBytes, hipMemcpyDeviceToHost), "D2H(counters)");
    for (uint32_t sl = 0; sl < kCounterSlots; sl++)
      for (int j = 0; j < 3; j++) cnt[j] += slots[(size_t)sl * kCounterStride + j];
  }
  std::lock_guard<std::mutex> lk(h->mu);
  out->rank_queries = cnt[0];
  out->backward_steps = cnt[1];
  out->launches = h->launches;
  out->last_kernel_ms = h->last_kernel_ms;
  out->index_bytes = h->index_bytes;
  out->n_blocks = h->nblocks;
  out->n_symbols = h->nslots;
  out->block_bytes = h->layout == kLayoutBytes ? kByteBlock + 4 : kBlockBytes;
  out->layout = h->layout;
  out->search_requests = cnt[2];
  out->build_ms = h->build_ms;
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

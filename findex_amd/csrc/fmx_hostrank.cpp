// fmx_hostrank.cpp -- SURVEY.md 8f-4: BWTMerger2.calcGaps' rank loop (bwtmerger.scala:981-1023).
//
// calcGaps walks the bytes of the older text once and keeps ONE running rank: curRank = bucketStarts(c) +
// searcher.occ(c, curRank - 1) -- every rank query needs the answer of the one before.  Such a chain cannot be
// batched, and measured on MI355X one lane group walks it at 0.99 us per step against 0.68 us on one host core
// with the reference's own structure (profiles/r02_calcgaps_chain.txt): the GPU is the wrong place for it.  What the
// path's rank dictionary can still give that loop is the structure itself: a sampled-count dictionary answers
// occ(c, i) with one count and a scan of at most 255 bytes lying right behind it -- where the reference's
// NaiveBWTSearcher binary-searches an inverted position list (findex.scala:479-505: ~log2(n / sigma) dependent
// probes).  So these two entry points answer on the HOST, from a host copy of the handle's BWT' cut into records of 256
// positions with every symbol's running count in front (built at first use).  They are not a fallback for anything: every batch entry point stays on the
// device and fails without one; this is the one dependent chain of the reference's construction code that the
// searcher seam (searcher: SuffixAlgo, bwtmerger.scala:981) exposes.
#include <fmx.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>

#include "fmx_host.h"

namespace fmx {
namespace {

constexpr uint32_t kHrShift = 8, kHrBlock = 1u << kHrShift;

#define HIP_TRY(call, what)                            \
  do {                                                 \
    hipError_t e__ = (call);                           \
    if (e__ != hipSuccess) return hip_fail(e__, what); \
  } while (0)

int host_rank_get(const Index *h, const HostRank **out) {
  std::lock_guard<std::mutex> lk(h->hr_mu);
  if (!h->hr) {
    if (h->n > (1ull << 31)) {
      set_error("the host-side rank dictionary serves block-sized indexes (n <= 2^31): it is meant for one merge block");
      return FMX_ERR_UNSUPPORTED;
    }
    std::unique_ptr<HostRank> hr(new (std::nothrow) HostRank());
    if (!hr) { set_error("out of host memory"); return FMX_ERR_NOMEM; }
    const uint32_t ns = h->nslots;
    const size_t nrec = (size_t)(h->n >> kHrShift) + 1;
    std::vector<uint8_t> raw;
    try {
      hr->stride = (size_t)ns * 4 + kHrBlock;
      hr->blob.assign(nrec * hr->stride, 0);
      raw.resize((size_t)h->n);
    } catch (const std::bad_alloc &) {
      set_error("out of host memory");
      return FMX_ERR_NOMEM;
    }
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
    HIP_TRY(hipMemcpy(raw.data(), h->d_bwt, (size_t)h->n, hipMemcpyDeviceToHost), "D2H(bwt)");
    uint32_t run[256] = {0};                // by slot
    for (size_t r = 0; r < nrec; r++) {
      uint8_t *rec = hr->blob.data() + r * hr->stride;
      std::memcpy(rec, run, (size_t)ns * 4);
      const uint64_t lo = (uint64_t)r << kHrShift, hi = std::min<uint64_t>(h->n, lo + kHrBlock);
      if (hi > lo) std::memcpy(rec + (size_t)ns * 4, raw.data() + lo, (size_t)(hi - lo));
      for (uint64_t p = lo; p < hi; p++) {
        const uint16_t sl = h->slot[raw[(size_t)p]];
        if (sl < kSlotEof) run[sl]++;
      }
    }
    h->hr = std::move(hr);
  }
  *out = h->hr.get();
  return FMX_OK;
}

// occ(c, i) as every searcher of the path answers it: #{p <= i : BWT'[p] == c}, i = -1 -> 0, i >= n clamped
inline uint64_t occ_host(const Index *h, const HostRank *hr, uint32_t c, int64_t i) {
  if (i < 0) return 0;
  const uint64_t x = (uint64_t)i >= h->n ? h->n : (uint64_t)i + 1;      // exclusive bound
  const uint16_t s = h->slot[c & 0xFFu];
  if (s == kSlotNone) return 0;
  if (s == kSlotEof) return x > h->eof ? 1 : 0;
  const uint8_t *rec = hr->blob.data() + (size_t)(x >> kHrShift) * hr->stride;
  uint32_t cnt;
  std::memcpy(&cnt, rec + (size_t)s * 4, 4);
  const uint8_t *p = rec + (size_t)h->nslots * 4, *e = p + (x & (kHrBlock - 1));
  const uint8_t want = (uint8_t)c;
  uint32_t m = 0;
  for (; p < e; p++) m += *p == want;                 // vectorised by the compiler (16 bytes per step)
  return cnt + m;
}

}  // namespace
}  // namespace fmx

using namespace fmx;

extern "C" {

int fmx_occ_host(const fmx_index *idx, int c, int64_t i, uint64_t *out) {
  if (!idx || !out) { set_error("null argument"); return FMX_ERR_ARG; }
  if (c < 0 || c > 255) { set_error("symbol out of range (reference: ArrayIndexOutOfBounds)"); return FMX_ERR_ARG; }
  const Index *h = reinterpret_cast<const Index *>(idx);
  const HostRank *hr = nullptr;
  int rc = host_rank_get(h, &hr);
  if (rc) return rc;
  *out = occ_host(h, hr, (uint32_t)c, i);
  return FMX_OK;
}

int fmx_calc_gaps_chain(const fmx_index *idx, const uint8_t *c, size_t k, uint64_t rank0, int last_char, uint64_t rklst,
                        uint64_t *ranks, size_t *done) {
  if (!idx || !done || (k && (!c || !ranks))) { set_error("null argument"); return FMX_ERR_ARG; }
  const Index *h = reinterpret_cast<const Index *>(idx);
  const HostRank *hr = nullptr;
  int rc = host_rank_get(h, &hr);
  if (rc) return rc;
  uint64_t cur = rank0;
  size_t j = 0;
  for (; j < k; j++) {
    const uint32_t ch = c[j];
    // curRank = if (curRank == 0) cFirst else cFirst + searcher.occ(c, curRank - 1), bwtmerger.scala:999-1001
    uint64_t r = h->cf[ch] + (cur == 0 ? 0 : occ_host(h, hr, ch, (int64_t)cur - 1));
    if (last_char >= 0 && (int)ch == last_char) {                  // :1003-1013
      if (r == rklst) { ranks[j] = r; break; }                      // the caller's KMP buffer / longSuffixCmp decides
      if (r > rklst) r += 1;
    }
    ranks[j] = r;
    cur = r;
  }
  *done = j;
  return FMX_OK;
}

}  // extern "C"

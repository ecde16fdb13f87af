// fmx_jump.hip -- the derived ROW tables: several backward steps of a search as ONE lookup, once the interval is a single
// row.  Also the per-handle table policy and budget (TablePolicy, table_room) every derived table is built under.
//
// SuffixAlgo.search (findex.scala:15-31) narrows (0, n) by a factor sigma per character; after ceil(log_sigma n) steps
// the interval of a pattern that still matches is one row [r, r + 1), and from there every further step is
//     c == BWT'[r] ?  [LF(r), LF(r) + 1)  :  empty          (getPrevRange on one row, findex.scala:32-36)
// -- the pattern is being compared, one character per dependent rank query, with the text that precedes suffix r.
// At C3 (n = 2^32, sigma = 128, 32-character patterns) that is 26 of a pattern's 28 memory requests.  The text before
// a row does not depend on the pattern, so it can be laid down once:
//     J[r]  = ( BWT'[r], BWT'[LF r], .., BWT'[LF^(jc-1) r] ;  LF^jc r )     jc <= 11 characters + a 40-bit row = 16 bytes
//             (bytes 0 .. 10 the characters, unused ones 0; bytes 11 .. 15 the row; jc = 9 by default, "jump_chars")
//     pairs : row r holds J[r] | J[LF^jc r], its own entry and that of the row it lands on, 32 bytes -- one 64-byte sector,
//             one request, up to 2 jc steps (the default for one-hot indexes of 2^30 rows and more when 32 n bytes fit the
//             handle's budget: "jump_pairs")
//     R3[r] = LF^3 r | three characters << 40                                  8 bytes: tails, intervals of 2 .. G rows, and
//             the whole one-row part where J does not fit (n = 2^34)
//     R1[r] = LF r | BWT'[r] << 40                                             8 bytes: the regex frontier's one-row elements
// A one-row search whose next jc characters equal J[r]'s lands on row LF^jc r with ONE request instead of jc (k_search4,
// fmx_search.hip).  A pattern that differs somewhere inside an entry misses there; the reference loop's values at the
// failing step are then a few rank steps away (walk_parked) -- so misses return the reference loop's values and count its
// steps, as with the k-mer table at the other end of the pattern (fmx_ktab.hip).
//
// Built on the device when it has a chance to pay (tables_due: fmx_prepare, or the search that brings the handle's patterns
// to the threshold), each table in ONE allocation: R3 first by walking (three rank queries per row), then J from it --
// floor(jc / 3) lookups of R3 and jc mod 3 rank queries per entry, twice that per pair (k_jump_build).  (Round 3 built J by
// doubling, J1 -> J2 -> J4 -> J8 through a second 16 n-byte buffer: 0.45 s of kernels and 3.5-4 s for two 64 GiB hipMallocs.)
// A table that does not fit the free memory or the handle's budget is simply not there.  Not for fmx_open_block handles
// (their skipped row and first-byte rule are not properties of an LF walk).
#include "fmx_device.h"
#include "fmx_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace fmx {

constexpr int kJThreads = 256;

// J[r]: one lane group per row walks the jc steps.  r3 != nullptr: floor(jc / 3) lookups of the three-step table R3 (one
// 8-byte word each, every lane of the group the same address) and jc mod 3 rank queries; else jc rank queries.
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kJThreads) void k_jump_build(DevIndex ix, const unsigned long long *__restrict__ r3, uint4 *__restrict__ out, uint32_t jc, uint32_t pairs) {
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint64_t ngroups = (uint64_t)gridDim.x * (kJThreads / G);
  // pairs: out[2 r] = J[r] and out[2 r + 1] = J[LF^jc r], the entry of the row the first one lands on -- 32 bytes, one
  // sector, one request for up to 2 jc steps of a search (k_search4<.., JT = 2>)
  for (uint64_t r0 = ((uint64_t)blockIdx.x * kJThreads + threadIdx.x) / G; r0 < ix.n; r0 += ngroups) {
    uint64_t r = r0;
    for (uint32_t part = 0; part <= pairs; part++) {
    unsigned long long lo = 0;      // characters 0 .. 7
    uint32_t hi = 0;                // characters 8 .. 10
    uint32_t s = 0;                 // characters walked so far
    const uint32_t threes = r3 ? jc / 3u : 0u;
    for (uint32_t q = 0; q < threes; q++) {
      const unsigned long long a = r3[r];
      const unsigned long long c3 = a >> 40;                 // three characters
      if (s < 8u) lo |= c3 << (8u * s);                      // (a shift of 8 s <= 56 bits: what runs over the top is in hi)
      if (s == 6u) hi |= (uint32_t)(c3 >> 16);
      if (s >= 8u) hi |= (uint32_t)c3 << (8u * (s - 8u));
      s += 3u;
      r = a & ((1ull << 40) - 1);
    }
    for (; s < jc; s++) {
      const uint32_t c = r == ix.eof ? 0u : ix.bwt[r];
      if (s < 8u) lo |= (unsigned long long)c << (8u * s);
      else hi |= c << (8u * (s - 8u));
      r = s_cf[c] + rank_excl<WIDE, LAYOUT>(ix, c, s_slot[c], r, lc);
    }
    if (lc.t == 0) out[pairs ? 2 * r0 + part : r0] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), hi | ((uint32_t)(r & 0xFFu) << 24), (uint32_t)(r >> 8));
    }
  }
}

// The regex frontier's row table: R[r] = LF r | BWT'[r] << 40, one 8-byte word per row.  An element of the frontier whose
// interval is one row [r, r + 1) steps with its state's byte c to [LF r, LF r + 1) if c == BWT'[r] and to nothing
// otherwise (getPrevRange on one row) -- with R that is one 8-byte load by the element's own lane instead of a rank query
// by a lane group (a 64-byte block through the LDS exchange, ~45 vector instructions): fmx_frontier.hip.  C4: 0.403 ->
// 0.387 ms per call.  (Tried on top and dropped: a third byte BWT'[LF r], with which an element that stepped this way
// drops those of its follows that cannot survive their own step, counting the reference's steps for them -- no gain on
// C4: the follows that die there are pushed by elements that were still wider than one row, 0.388 vs 0.382 ms.)
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kJThreads) void k_row1_init(DevIndex ix, unsigned long long *__restrict__ out) {
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint64_t ngroups = (uint64_t)gridDim.x * (kJThreads / G);
  for (uint64_t r = ((uint64_t)blockIdx.x * kJThreads + threadIdx.x) / G; r < ix.n; r += ngroups) {
    const uint32_t c = r == ix.eof ? 0u : ix.bwt[r];
    const uint64_t nxt = s_cf[c] + rank_excl<WIDE, LAYOUT>(ix, c, s_slot[c], r, lc);
    if (lc.t == 0) out[r] = nxt | ((unsigned long long)c << 40);
  }
}

// The three-step row table: R3[r] = LF^3 r | BWT'[r] << 40 | BWT'[LF r] << 48 | BWT'[LF^2 r] << 56, one 8-byte word per row --
// what the row jump table is, at a third of its size, for an index whose 16 n bytes of J do not fit beside the dictionary
// (n = 2^34: 256 GiB).  k_search_rows takes three steps of a one-row search with one load of it.  Built by walking:
// three rank queries per row, the first of them on consecutive rows (neighbouring blocks), the others at random.
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kJThreads) void k_row3_init(DevIndex ix, unsigned long long *__restrict__ out) {
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint64_t ngroups = (uint64_t)gridDim.x * (kJThreads / G);
  for (uint64_t r0 = ((uint64_t)blockIdx.x * kJThreads + threadIdx.x) / G; r0 < ix.n; r0 += ngroups) {
    uint64_t r = r0;
    unsigned long long chars = 0;
#pragma unroll
    for (uint32_t s = 0; s < 3; s++) {
      const uint32_t c = r == ix.eof ? 0u : ix.bwt[r];
      chars |= (unsigned long long)c << (8u * s);
      r = s_cf[c] + rank_excl<WIDE, LAYOUT>(ix, c, s_slot[c], r, lc);
    }
    if (lc.t == 0) out[r0] = r | (chars << 40);
  }
}

// ---- the table policy (fmx_host.h, TablePolicy): process defaults, copied by a handle at open
TablePolicy &default_policy() {
  static TablePolicy p;
  return p;
}

static bool parse_u64(const char *v, uint64_t *out) {
  char *end = nullptr;
  if (!*v || *v == '-') return false;
  const unsigned long long x = std::strtoull(v, &end, 10);
  if (end == v || *end) return false;
  *out = x;
  return true;
}

int policy_set(TablePolicy &p, const char *key, const char *value, const char **why) {
  auto is = [&](const char *k) { return std::strcmp(key, k) == 0; };
  auto val = [&](const char *v) { return std::strcmp(value, v) == 0; };
  if (is("ktab")) {
    if (val("auto")) p.ktab.store(1);
    else if (val("off")) p.ktab.store(0);
    else { *why = "ktab must be auto or off"; return 2; }
    return 0;
  }
  if (is("jump")) {
    if (val("auto")) p.jump_mode.store(7);
    else if (val("off")) p.jump_mode.store(0);
    else if (val("rows")) p.jump_mode.store(1);
    else if (val("jumps")) p.jump_mode.store(2);
    else if (val("rows3")) p.jump_mode.store(4);
    else { *why = "jump must be auto, rows, rows3, jumps or off"; return 2; }
    return 0;
  }
  if (is("jump_pairs")) {
    if (val("auto")) p.jump_pairs.store(-1);
    else if (val("on")) p.jump_pairs.store(1);
    else if (val("off")) p.jump_pairs.store(0);
    else { *why = "jump_pairs must be auto, on or off"; return 2; }
    return 0;
  }
  if (is("jump_chars")) {
    uint64_t v = 0;
    if (!parse_u64(value, &v) || v < 8 || v > 11) { *why = "jump_chars must be 8, 9, 10 or 11"; return 2; }
    p.jump_chars.store((int)v);
    return 0;
  }
  if (is("tables_after")) {
    if (val("auto")) { p.tables_after.store(-1); return 0; }
    uint64_t v = 0;
    if (!parse_u64(value, &v) || v > (1ull << 62)) { *why = "tables_after must be auto or a non-negative number of patterns"; return 2; }
    p.tables_after.store((long long)v);
    return 0;
  }
  if (is("table_budget")) {
    // "auto": none beyond the margins; "0.25" (a number with a point, 0 < f <= 1): that share of the free HBM; "N": N bytes
    if (val("auto")) { p.budget_bytes.store(~0ull); p.budget_ppb.store(0); return 0; }
    if (std::strchr(value, '.')) {
      char *end = nullptr;
      const double f = std::strtod(value, &end);
      if (end == value || *end || !(f > 0.0) || f > 1.0) { *why = "table_budget as a fraction must be in (0, 1]"; return 2; }
      p.budget_ppb.store((uint32_t)std::max(1.0, f * 1e9 + 0.5));
      p.budget_bytes.store(~0ull);
      return 0;
    }
    uint64_t v = 0;
    if (!parse_u64(value, &v)) { *why = "table_budget must be auto, a number of bytes, or a fraction of the free HBM like 0.5"; return 2; }
    p.budget_bytes.store(v);
    p.budget_ppb.store(0);
    return 0;
  }
  return 1;
}

// ---- when the derived tables are built.  Round 3 built all of them at a handle's FIRST search, whatever it was: a
// single getPrevRange-sized query on a C3-size handle waited 6 s and left 96 GiB behind.  Now a table is built by
// fmx_prepare, or by the search that brings the patterns the handle has been asked for to a threshold -- "auto": n / 64
// patterns (at least 65536) for the row tables, whose build is O(n) rank queries, i.e. when the searches themselves have
// done work of that order; 1024 patterns for the k-mer table (milliseconds to build).  "tables_after" = N: N patterns for
// both (0: at the first search, round 3's behaviour -- the tests use it).  Each table has its own "prepared" flag.
bool tables_due(const Index *h, uint64_t k, bool small_table) {
  // a search asks twice (k-mer table, then row tables): its patterns are counted by the second question
  const uint64_t seen = (small_table ? h->patterns_seen.load(std::memory_order_relaxed) : h->patterns_seen.fetch_add(k, std::memory_order_relaxed)) + k;
  if ((small_table ? h->prepared_ktab : h->prepared_rows).load(std::memory_order_relaxed)) return true;
  const long long after = h->policy.tables_after.load(std::memory_order_relaxed);
  const uint64_t need = after >= 0 ? (uint64_t)after : (small_table ? 1024ull : std::max<uint64_t>(65536, h->n / 64));
  return seen >= need;
}
void note_table_build(const Index *h, uint64_t bytes_held) {
  uint64_t cur = h->peak_table_build_bytes.load(std::memory_order_relaxed);
  while (bytes_held > cur && !h->peak_table_build_bytes.compare_exchange_weak(cur, bytes_held, std::memory_order_relaxed)) {}
}

// ---- the budget.  Until round 4 the only guard was "the table + 4-8 GiB are free": a 4 GiB BWT ended up holding 241 of the
// device's 288 GB, and a caller that needed 30 GiB afterwards (torch's allocator, a second index) got an OOM with no knob
// but "off".  A handle now counts what its derived tables hold (tables_held) against its budget -- bytes, or a share of the
// HBM that is free when the question is asked, its own tables counted as free -- and every build asks table_room first.
void tables_account(const Index *h, int64_t delta) {
  if (delta >= 0) h->tables_held.fetch_add((uint64_t)delta, std::memory_order_relaxed);
  else h->tables_held.fetch_sub((uint64_t)(-delta), std::memory_order_relaxed);
  size_t free_b = 0, total_b = 0;
  if (delta > 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess) h->hbm_free_after_tables.store(free_b, std::memory_order_relaxed);
  else if (delta > 0) (void)hipGetLastError();
}
hipError_t table_malloc(const Index *h, void **p, size_t bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipMalloc(p, bytes);
  h->tables_alloc_us.fetch_add((uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
  return e;
}
uint64_t table_room(const Index *h, uint64_t margin) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
  uint64_t room = free_b > margin ? free_b - margin : 0;
  const uint64_t held = h->tables_held.load(std::memory_order_relaxed);
  uint64_t budget = h->policy.budget_bytes.load(std::memory_order_relaxed);
  const uint32_t ppm = h->policy.budget_ppb.load(std::memory_order_relaxed);
  if (ppm) budget = (uint64_t)((double)(free_b + held) * ((double)ppm * 1e-9));
  if (budget != ~0ull) room = std::min(room, budget > held ? budget - held : 0);
  return room;
}

// 8 n bytes of row words by `launch`; leaves *slot null when the table is not wanted or does not fit (no error).
template <class Launch>
static void build_row_words(const Index *h, hipStream_t st, uint64_t margin, void **slot, uint64_t *slot_bytes, Launch launch) {
  const uint64_t bytes = h->n * 8;
  if (bytes > table_room(h, margin)) return;
  void *p = nullptr;
  hipError_t e = table_malloc(h, &p, bytes);
  if (e == hipSuccess) {
    const uint64_t per_wg = kJThreads / (h->layout == kLayoutBytes ? 8 : 4);
    const int grid = (int)std::min<uint64_t>((h->n + per_wg - 1) / per_wg, (uint64_t)h->cu_count * 8);
    launch(grid, static_cast<unsigned long long *>(p));
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) { *slot = p; *slot_bytes = bytes; note_table_build(h, bytes); tables_account(h, (int64_t)bytes); }
    else (void)hipFree(p);
  }
  if (e != hipSuccess) (void)hipGetLastError();
}

static bool rows_eligible(const Index *h) { return !h->block_mode && h->n >= 2 && h->nslots >= 1; }

// The three-step row table of a handle (nullptr: none yet / none at all): 8 n bytes.
hipError_t row3_get(const Index *h, hipStream_t st, const unsigned long long **out, bool build) {
  std::lock_guard<std::mutex> lk(h->r3_mu);
  if (!h->r3_ready && build) {
    const auto t0 = std::chrono::steady_clock::now();
    if ((h->policy.jump_mode.load(std::memory_order_relaxed) & 4) && rows_eligible(h))
      build_row_words(h, st, 8ull << 30, &h->d_row3, &h->row3_bytes, [&](int grid, unsigned long long *p) {
#define CALL(W, L) k_row3_init<W, L><<<grid, kJThreads, 0, st>>>(h->dev, p)
        FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
      });
    h->tables_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->r3_ready = true;
  }
  *out = static_cast<const unsigned long long *>(h->d_row3);
  return hipSuccess;
}

// The frontier's row table of a handle (nullptr: none), built at the first regex match: 8 n bytes.
hipError_t row1_get(const Index *h, hipStream_t st, const unsigned long long **out, bool build) {
  std::lock_guard<std::mutex> lk(h->r1_mu);
  if (!h->r1_ready && build) {
    const auto t0 = std::chrono::steady_clock::now();
    static const int forced = getenv("FMX_ROW1") ? atoi(getenv("FMX_ROW1")) : -1;      // 0 = off
    if (forced != 0 && (h->policy.jump_mode.load(std::memory_order_relaxed) & 1) && rows_eligible(h))
      build_row_words(h, st, 4ull << 30, &h->d_row1, &h->row1_bytes, [&](int grid, unsigned long long *p) {
#define CALL(W, L) k_row1_init<W, L><<<grid, kJThreads, 0, st>>>(h->dev, p)
        FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
      });
    h->tables_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->r1_ready = true;
  }
  *out = static_cast<const unsigned long long *>(h->d_row1);
  return hipSuccess;
}

// Called under h->jt_mu by jump_get.  Leaves h->d_jump null when the table is not wanted or does not fit.
static hipError_t build_jump(const Index *h, hipStream_t st) {
  static const int forced = getenv("FMX_JUMP") ? atoi(getenv("FMX_JUMP")) : -1;      // 0 = off, 1 = whenever it fits
  if (forced == 0 || (forced < 0 && !(h->policy.jump_mode.load(std::memory_order_relaxed) & 2))) return hipSuccess;
  if (!rows_eligible(h)) return hipSuccess;
  uint64_t bytes = h->n * 16;
  // the three-step table first: the search kernel uses it beside J, and J is built from it (two lookups instead of six
  // of the eight rank queries per row)
  const unsigned long long *r3 = nullptr;
  (void)row3_get(h, st, &r3, true);
  // The table, a margin for the callers' batches (the dictionary, the k-mer table and R3 are resident already), and the
  // handle's budget (table_room).
  // Pairs of entries (32 bytes per row: up to 2 jc steps per request, k_search4<.., JT = 2>) where the quad layout's index
  // is large enough for requests to be what binds (n >= 2^30) and twice the table fits beside everything else and inside the
  // budget; "jump_pairs" = "auto" | "on" | "off" (per handle: fmx_index_config_set); FMX_JUMP_PAIRS=0|1 overrides it (tests)
  const uint64_t room = table_room(h, 8ull << 30);
  const char *pe = getenv("FMX_JUMP_PAIRS");
  const int pcfg = pe ? (atoi(pe) != 0 ? 1 : 0) : h->policy.jump_pairs.load(std::memory_order_relaxed);
  const bool want_pairs = r3 && h->layout != kLayoutBytes && (pcfg < 0 ? h->n >= (1ull << 30) : pcfg != 0);
  const bool pairs = want_pairs && 2 * bytes <= room;
  if (pairs) bytes *= 2;
  if (bytes > room && !(forced == 1 && bytes <= table_room(h, 1ull << 28))) return hipSuccess;
  hipError_t e = hipSuccess;
  static const bool trace = getenv("FMX_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (trace) { (void)hipStreamSynchronize(st); fprintf(stderr, "[fmx] jump table %-12s +%.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
  };
  const uint32_t jc = (uint32_t)std::min(11, std::max(8, h->policy.jump_chars.load(std::memory_order_relaxed)));
  void *a = nullptr;
  e = table_malloc(h, &a, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }      // no table, no error
  mark("allocated");
  {
    const uint64_t per_wg = kJThreads / (h->layout == kLayoutBytes ? 8 : 4);
    const int grid = (int)std::min<uint64_t>((h->n + per_wg - 1) / per_wg, (uint64_t)h->cu_count * 8);
#define CALL(W, L) k_jump_build<W, L><<<grid, kJThreads, 0, st>>>(h->dev, r3, (uint4 *)a, jc, pairs ? 1u : 0u)
    FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  mark(r3 ? "built (R3)" : "built (walk)");
  if (e != hipSuccess) { (void)hipFree(a); return e; }
  h->d_jump = a;
  h->jump_bytes = bytes;
  h->jump_pairs = pairs;
  h->jump_chars = jc;
  note_table_build(h, bytes);
  tables_account(h, (int64_t)bytes);
  return hipSuccess;
}

// The jump table of a handle (nullptr: none yet / none at all).
hipError_t jump_get(const Index *h, hipStream_t st, const uint4 **out, bool build) {
  std::lock_guard<std::mutex> lk(h->jt_mu);
  if (!h->jt_ready && build) {
    const auto t0 = std::chrono::steady_clock::now();
    const double ms_before = h->tables_ms;
    const hipError_t e = build_jump(h, st);
    if (e != hipSuccess) { (void)hipGetLastError(); h->d_jump = nullptr; h->jump_bytes = 0; }      // searches walk every step
    // (row3_get, called inside build_jump, has added its share already: the whole span counts once)
    h->tables_ms = ms_before + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->jt_ready = true;
  }
  *out = static_cast<const uint4 *>(h->d_jump);
  return hipSuccess;
}

// fmx_drop_tables: frees derived tables (the caller guarantees that no call is using the handle).  They are built
// again by fmx_prepare or when the threshold is met anew.
int drop_tables(Index *h, unsigned what) {
  if (what & 1u) {      // the k-mer table
    std::lock_guard<std::mutex> lk(h->kt_mu);
    if (h->d_ktab) (void)hipFree(h->d_ktab);
    if (h->d_kt_dense) (void)hipFree(h->d_kt_dense);
    if (h->d_kt_levels) (void)hipFree(h->d_kt_levels);
    h->d_ktab = h->d_kt_dense = h->d_kt_levels = nullptr;
    tables_account(h, -(int64_t)h->kt_bytes);
    h->kt_bytes = 0;
    h->kt = KTab{};
    h->kt.sigma = h->nslots;
    h->kt_ready = false;
    h->prepared_ktab.store(false, std::memory_order_relaxed);
  }
  if (what & 4u) {
    { std::lock_guard<std::mutex> lk(h->jt_mu); if (h->d_jump) (void)hipFree(h->d_jump); tables_account(h, -(int64_t)h->jump_bytes); h->d_jump = nullptr; h->jump_bytes = 0; h->jump_pairs = false; h->jt_ready = false; }
    { std::lock_guard<std::mutex> lk(h->r3_mu); if (h->d_row3) (void)hipFree(h->d_row3); tables_account(h, -(int64_t)h->row3_bytes); h->d_row3 = nullptr; h->row3_bytes = 0; h->r3_ready = false; }
    h->prepared_rows.store(false, std::memory_order_relaxed);
  }
  if (what & 5u) h->patterns_seen.store(0, std::memory_order_relaxed);
  if (what & 8u) {
    std::lock_guard<std::mutex> lk(h->r1_mu);
    if (h->d_row1) (void)hipFree(h->d_row1);
    tables_account(h, -(int64_t)h->row1_bytes);
    h->d_row1 = nullptr; h->row1_bytes = 0; h->r1_ready = false;
  }
  return 0;
}

}  // namespace fmx

// fmx_jump.hip -- the row jump table: EIGHT backward steps of a search as ONE 16-byte lookup, once the interval is a
// single row.
//
// SuffixAlgo.search (findex.scala:15-31) narrows (0, n) by a factor sigma per character; after ceil(log_sigma n) steps
// the interval of a pattern that still matches is one row [r, r + 1), and from there every further step is
//     c == BWT'[r] ?  [LF(r), LF(r) + 1)  :  empty          (getPrevRange on one row, findex.scala:32-36)
// -- the pattern is being compared, one character per dependent rank query, with the text that precedes suffix r.
// At C3 (n = 2^32, sigma = 128, 32-character patterns) that is 26 of a pattern's 28 memory requests.  The text before
// a row does not depend on the pattern, so it can be laid down once:
//     J[r] = ( BWT'[r], BWT'[LF r], .., BWT'[LF^7 r] ;  LF^8 r )                   8 bytes + a row number = 16 bytes
// and a one-row search whose next eight characters equal J[r]'s lands on row LF^8 r with ONE request instead of eight
// (k_search4, fmx_search.hip).  A pattern that differs somewhere in those eight walks them the ordinary way -- so
// misses return the reference loop's values and count its steps, as with the k-mer table at the other end of the
// pattern (fmx_ktab.hip).
//
// Built on the device at a handle's first literal search (or by fmx_prepare), by doubling: J1[r] = (BWT'[r], LF r)
// from one rank query per row, then J2 = J1 o J1, J4 = J2 o J2, J8 = J4 o J4 -- three passes of random 16-byte
// gathers.  16 n bytes (64 GiB at C3, beside the 77 GiB dictionary), twice that while it is built; when the memory is
// not there (C5: n = 2^34) the handle simply has no jump table.  Not for fmx_open_block handles (their skipped row
// and first-byte rule are not properties of an LF walk).
#include "fmx_device.h"
#include "fmx_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace fmx {

constexpr int kJThreads = 256;

// J1[r] = (BWT'[r]; LF r) for r in [lo, hi): one lane group per row, LF r = C[c] + rank(c, r)
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kJThreads) void k_jump_init(DevIndex ix, uint4 *__restrict__ out, uint64_t lo, uint64_t hi) {
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint64_t ngroups = (uint64_t)gridDim.x * (kJThreads / G);
  for (uint64_t r = lo + ((uint64_t)blockIdx.x * kJThreads + threadIdx.x) / G; r < hi; r += ngroups) {
    const uint32_t c = r == ix.eof ? 0u : ix.bwt[r];
    const uint64_t nxt = s_cf[c] + rank_excl<WIDE, LAYOUT>(ix, c, s_slot[c], r, lc);
    if (lc.t == 0) out[r] = make_uint4(c, 0u, (uint32_t)nxt, (uint32_t)(nxt >> 32));
  }
}

// J(2m) = Jm o Jm: the m characters of row r, then the m characters of the row they lead to
__global__ __launch_bounds__(kJThreads) void k_jump_double(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n, uint32_t m) {
  const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += nth) {
    const uint4 a = in[r];
    const uint4 b = in[((uint64_t)a.w << 32) | a.z];
    uint4 o;
    o.x = m == 4 ? a.x : (a.x | (b.x << (8u * m)));
    o.y = m == 4 ? b.x : 0u;
    o.z = b.z;
    o.w = b.w;
    out[r] = o;
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

static std::atomic<int> g_jump_mode{7};      // bit 0: row table (R1), bit 1: row jump table (J8), bit 2: three-step row table (R3)
void jump_set_mode(int mode) { g_jump_mode.store(mode & 7, std::memory_order_relaxed); }

// Called under h->jt_mu by jump_get.  Leaves h->d_jump null when the table is not wanted or does not fit.
static hipError_t build_jump(const Index *h, hipStream_t st) {
  static const int forced = getenv("FMX_JUMP") ? atoi(getenv("FMX_JUMP")) : -1;      // 0 = off, 1 = whenever it fits
  if (forced == 0 || (forced < 0 && !(g_jump_mode.load(std::memory_order_relaxed) & 2))) return hipSuccess;
  if (h->block_mode || h->n < 2 || h->nslots < 1) return hipSuccess;
  const uint64_t bytes = h->n * 16;
  size_t free_b = 0, total_b = 0;
  hipError_t e = hipMemGetInfo(&free_b, &total_b);
  if (e != hipSuccess) return e;
  // twice the table while it is built, and a margin for the callers' batches; never more than half of what is free
  // once it stands (the dictionary and the k-mer table are resident already)
  if (2 * bytes + (8ull << 30) > free_b && !(forced == 1 && 2 * bytes + (1ull << 28) <= free_b)) return hipSuccess;
  static const bool trace = getenv("FMX_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (trace) { (void)hipStreamSynchronize(st); fprintf(stderr, "[fmx] jump table %-12s +%.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
  };
  void *a = nullptr, *b = nullptr;
  e = hipMalloc(&a, bytes);
  if (e == hipSuccess) e = hipMalloc(&b, bytes);
  if (e != hipSuccess) { if (a) (void)hipFree(a); (void)hipGetLastError(); return hipSuccess; }      // no table, no error
  mark("allocated");
  {
    const uint64_t per_wg = kJThreads / (h->layout == kLayoutBytes ? 8 : 4);
    const int grid = (int)std::min<uint64_t>((h->n + per_wg - 1) / per_wg, (uint64_t)h->cu_count * 8);
#define CALL(W, L) k_jump_init<W, L><<<grid, kJThreads, 0, st>>>(h->dev, (uint4 *)a, (uint64_t)0, h->n)
    FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
    e = hipGetLastError();
  }
  mark("J1");
  void *src = a, *dst = b;
  for (uint32_t m = 1; m < 8 && e == hipSuccess; m *= 2) {
    const int grid = (int)std::min<uint64_t>((h->n + kJThreads - 1) / kJThreads, (uint64_t)h->cu_count * 16);
    k_jump_double<<<grid, kJThreads, 0, st>>>((const uint4 *)src, (uint4 *)dst, h->n, m);
    e = hipGetLastError();
    std::swap(src, dst);
    mark("doubled");
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(dst);                  // the buffer the last pass read
  mark("freed");
  if (e != hipSuccess) { (void)hipFree(src); return e; }
  h->d_jump = src;
  h->jump_bytes = bytes;
  return hipSuccess;
}

// The jump table of a handle (nullptr: none), built on first use.
hipError_t jump_get(const Index *h, hipStream_t st, const uint4 **out) {
  std::lock_guard<std::mutex> lk(h->jt_mu);
  if (!h->jt_ready) {
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = build_jump(h, st);
    if (e != hipSuccess) { (void)hipGetLastError(); h->d_jump = nullptr; h->jump_bytes = 0; }      // searches walk every step
    h->tables_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->jt_ready = true;
  }
  *out = static_cast<const uint4 *>(h->d_jump);
  return hipSuccess;
}

// The frontier's row table of a handle (nullptr: none), built at the first regex match: 8 n bytes.
hipError_t row1_get(const Index *h, hipStream_t st, const unsigned long long **out) {
  std::lock_guard<std::mutex> lk(h->r1_mu);
  if (!h->r1_ready) {
    const auto t0 = std::chrono::steady_clock::now();
    static const int forced = getenv("FMX_ROW1") ? atoi(getenv("FMX_ROW1")) : -1;      // 0 = off
    size_t free_b = 0, total_b = 0;
    const uint64_t bytes = h->n * 8;
    if (forced != 0 && (g_jump_mode.load(std::memory_order_relaxed) & 1) && !h->block_mode && h->n >= 2 && h->nslots >= 1 &&
        hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes + (4ull << 30) <= free_b) {
      void *p = nullptr;
      hipError_t e = hipMalloc(&p, bytes);
      if (e == hipSuccess) {
        const uint64_t per_wg = kJThreads / (h->layout == kLayoutBytes ? 8 : 4);
        const int grid = (int)std::min<uint64_t>((h->n + per_wg - 1) / per_wg, (uint64_t)h->cu_count * 8);
#define CALL(W, L) k_row1_init<W, L><<<grid, kJThreads, 0, st>>>(h->dev, (unsigned long long *)p)
        FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) { h->d_row1 = p; h->row1_bytes = bytes; }
        else (void)hipFree(p);
      }
      if (e != hipSuccess) (void)hipGetLastError();      // no table: every element steps by rank query
    }
    h->tables_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->r1_ready = true;
  }
  *out = static_cast<const unsigned long long *>(h->d_row1);
  return hipSuccess;
}

// The three-step row table of a handle (nullptr: none), built at the first literal search that wants it: 8 n bytes.
hipError_t row3_get(const Index *h, hipStream_t st, const unsigned long long **out) {
  std::lock_guard<std::mutex> lk(h->r3_mu);
  if (!h->r3_ready) {
    const auto t0 = std::chrono::steady_clock::now();
    size_t free_b = 0, total_b = 0;
    const uint64_t bytes = h->n * 8;
    if ((g_jump_mode.load(std::memory_order_relaxed) & 4) && !h->block_mode && h->n >= 2 && h->nslots >= 1 &&
        hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes + (8ull << 30) <= free_b) {
      void *p = nullptr;
      hipError_t e = hipMalloc(&p, bytes);
      if (e == hipSuccess) {
        const uint64_t per_wg = kJThreads / (h->layout == kLayoutBytes ? 8 : 4);
        const int grid = (int)std::min<uint64_t>((h->n + per_wg - 1) / per_wg, (uint64_t)h->cu_count * 8);
#define CALL(W, L) k_row3_init<W, L><<<grid, kJThreads, 0, st>>>(h->dev, (unsigned long long *)p)
        FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) { h->d_row3 = p; h->row3_bytes = bytes; }
        else (void)hipFree(p);
      }
      if (e != hipSuccess) (void)hipGetLastError();      // no table: the lane groups walk every step
    }
    h->tables_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->r3_ready = true;
  }
  *out = static_cast<const unsigned long long *>(h->d_row3);
  return hipSuccess;
}

}  // namespace fmx

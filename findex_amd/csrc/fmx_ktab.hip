// fmx_ktab.hip -- the k-mer jump table: the first K backward steps of a search as ONE lookup.
//
// SuffixAlgo.search (findex.scala:15-31) starts every pattern from (0, n) and its first steps depend on nothing but
// the pattern's last characters: the interval after K steps is a function of the K-mer.  The reference's kernels
// already answer step 0 from C[] alone; this generalises that shortcut (the "ftab" of short-read aligners):
//     T[code] = (sp, ep, steps) after consuming the K symbols of `code`, most significant digit first,
// with symbols numbered densely over the ones that occur in the BWT (sigma' of them) and
// code = ((d0 * sigma' + d1) * sigma' + ...) + d(K-1), d0 = the pattern's LAST character.  A K-mer that does not occur
// keeps what the reference's loop holds when it stops -- the (sp, ep) of the first empty step and the number of
// steps taken -- so a miss returns the reference's values and counts the reference's steps.
//
// K is the largest with sigma'^K <= n / 8 (longer K-mers mostly do not occur) and a table of at most 16 GiB and a
// quarter of the free HBM: K = 4 at C3 (n = 2^32, sigma = 128: 4 GiB beside a 77 GiB dictionary), 12 at C2
// (n = 2^28, sigma = 4: 268 MB), 5 at C4 (sigma = 28).  Built on the device at the first search, level by level
// with the rank primitive (sigma'^K steps: ~10 ms at C3); every level is kept (the smaller ones cost 1/sigma' more)
// because the regex frontier steps through them one character at a time.
#include "fmx_device.h"
#include "fmx_host.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>

namespace fmx {

constexpr int kKtThreads = 256;

// level j+1 from level j: entry (code, c) = step(T_j[code], symbol c); one lane group per new entry
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kKtThreads) void k_ktab_level(DevIndex ix, const uint4 *__restrict__ prev, uint4 *__restrict__ next,
                                                            uint64_t n_prev, uint32_t sigma, uint32_t level,
                                                            const uint8_t *__restrict__ sym_of /* [sigma] */) {
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint64_t total = n_prev * sigma;
  const uint64_t ngroups = (uint64_t)gridDim.x * (kKtThreads / G);
  for (uint64_t e = ((uint64_t)blockIdx.x * kKtThreads + threadIdx.x) / G; e < total; e += ngroups) {
    const uint64_t code = e / sigma;
    const uint32_t c = sym_of[e % sigma];
    uint64_t sp, ep;
    uint32_t steps;
    if (level == 0) { sp = 0; ep = ix.n; steps = 0; }
    else {
      const uint4 p = prev[code];
      sp = (((uint64_t)p.y << 32) | p.x) & ((1ull << 56) - 1);
      steps = p.y >> 24;
      ep = ((uint64_t)p.w << 32) | p.z;
    }
    if (sp < ep) {                      // still alive: one more step of the reference's loop
      backward_step<WIDE, LAYOUT>(ix, c, s_slot[c], s_cf[c], lc, sp, ep);
      steps++;
    }
    if (lc.t == 0)
      next[e] = make_uint4((uint32_t)sp, (uint32_t)(sp >> 32) | (steps << 24), (uint32_t)ep, (uint32_t)(ep >> 32));
  }
}

// Chooses K, allocates and fills the levels.  Called under h->kt_mu by ktab_get.
static hipError_t build_ktab(const Index *h, hipStream_t st) {
  const uint32_t sigma = h->nslots;
  h->kt.k = 0;
  h->kt.sigma = sigma;
  static const int forced = getenv("FMX_KTAB") ? atoi(getenv("FMX_KTAB")) : -1;      // 0 = off, k > 0 = exactly k levels
  if (sigma < 2 || forced == 0 || !h->policy.ktab.load(std::memory_order_relaxed)) return hipSuccess;
  size_t free_b = 0, total_b = 0;
  hipError_t e = hipMemGetInfo(&free_b, &total_b);
  if (e != hipSuccess) return e;
  // at most 16 GiB, a quarter of the free HBM, and a quarter of what the handle's budget leaves (the row tables want the rest)
  const uint64_t room = table_room(h, 0);
  const uint64_t share = std::max<uint64_t>(room / 4, std::min<uint64_t>(room, 64ull << 20));
  const uint64_t max_bytes = std::min<uint64_t>(std::min<uint64_t>(16ull << 30, free_b / 4), share);
  uint32_t k = 0;
  uint64_t entries = 1, all = 0;
  for (;;) {
    const uint64_t nxt = entries * sigma;
    if (k >= 16 || nxt > h->n / 8 || nxt > (1ull << 32) || (all + nxt) * 16 > max_bytes) break;
    if (forced > 0 && k >= (uint32_t)forced) break;
    entries = nxt;
    all += nxt;
    k++;
  }
  if (forced > 0)       // a test may ask for more levels than the size rule gives (tiny indexes)
    while (k < (uint32_t)forced && k < 16 && (all + entries * sigma) * 16 <= max_bytes) { entries *= sigma; all += entries; k++; }
  if (k == 0) return hipSuccess;
  uint8_t dense[256], sym_of[256];
  for (int c = 0; c < 256; c++) {
    dense[c] = 0xFF;
    if (h->slot[c] < kSlotEof) { dense[c] = (uint8_t)h->slot[c]; sym_of[h->slot[c]] = (uint8_t)c; }    // slots number the present symbols densely
  }
  void *d_all = nullptr, *d_dense = nullptr, *d_sym = nullptr;
  e = table_malloc(h, &d_all, all * 16);
  if (e == hipSuccess) e = hipMalloc(&d_dense, 256);
  if (e == hipSuccess) e = hipMalloc(&d_sym, 256);
  if (e == hipSuccess) e = hipMemcpyAsync(d_dense, dense, 256, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_sym, sym_of, 256, hipMemcpyHostToDevice, st);
  uint64_t off = 0, n_prev = 1;
  const uint4 *prev = nullptr;
  for (uint32_t lv = 0; lv < k && e == hipSuccess; lv++) {
    uint4 *next = static_cast<uint4 *>(d_all) + off;
    const uint64_t n_next = n_prev * sigma;
    const uint64_t per_wg = kKtThreads / (h->layout == kLayoutBytes ? 8 : 4);
    const int grid = (int)std::min<uint64_t>((n_next + per_wg - 1) / per_wg, (uint64_t)h->cu_count * 8);
#define CALL(W, L) k_ktab_level<W, L><<<grid, kKtThreads, 0, st>>>(h->dev, prev, next, n_prev, sigma, lv, (const uint8_t *)d_sym)
    FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
    e = hipGetLastError();
    h->kt.level[lv] = next;
    prev = next;
    off += n_next;
    n_prev = n_next;
  }
  void *d_levels = nullptr;
  if (e == hipSuccess) e = hipMalloc(&d_levels, sizeof h->kt.level);
  if (e == hipSuccess) e = hipMemcpyAsync(d_levels, h->kt.level, sizeof h->kt.level, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);      // `dense` / `sym_of` go out of scope
  if (d_sym) (void)hipFree(d_sym);
  if (e != hipSuccess) {
    if (d_all) (void)hipFree(d_all);
    if (d_dense) (void)hipFree(d_dense);
    if (d_levels) (void)hipFree(d_levels);
    h->kt.k = 0;
    return e;
  }
  h->d_ktab = d_all;
  h->d_kt_dense = d_dense;
  h->d_kt_levels = d_levels;
  h->kt.level_dev = static_cast<const uint4 *const *>(d_levels);
  h->kt.k = k;
  h->kt.tab = h->kt.level[k - 1];
  h->kt.dense = static_cast<const uint8_t *>(d_dense);
  h->kt_bytes = all * 16 + 256;
  note_table_build(h, h->kt_bytes);
  tables_account(h, (int64_t)h->kt_bytes);
  return hipSuccess;
}

// The table of a handle (k == 0: none), built on first use.
hipError_t ktab_get(const Index *h, hipStream_t st, KTab *out, bool build) {
  std::lock_guard<std::mutex> lk(h->kt_mu);
  if (!h->kt_ready && !build) {      // not yet: this search walks its first steps on the rank dictionary
    *out = KTab{};
    out->sigma = h->nslots;
    return hipSuccess;
  }
  if (!h->kt_ready) {
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = build_ktab(h, st);
    if (e != hipSuccess) {
      // No table (out of memory for it, usually): searches walk every step on the rank dictionary, which is always
      // correct.  The failure must not stick to the handle -- or to the HIP runtime's last-error slot.
      (void)hipGetLastError();
      h->kt = KTab{};
      h->kt.sigma = h->nslots;
    }
    h->tables_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->kt_ready = true;
  }
  *out = h->kt;
  return hipSuccess;
}

}  // namespace fmx

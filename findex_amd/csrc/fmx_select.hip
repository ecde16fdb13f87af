// fmx_select.hip -- Psi (NaiveFMSearcher.getNextI / nextSubstr, bwtmerger.scala:390-405) as a select on the
// rank dictionary.
//
// The reference answers getNextI(row) with one read of its inverted list, fm[row] (the list IS the select
// structure).  Here fm[row] = position of the j-th occurrence of symbol c in BWT', with c the symbol whose bucket
// holds `row` and j = row - C[c]: a select on c's bit-vector.  Round 1 did it with one lane per query and a binary
// search over all block headers (~23 dependent probes at C3's size) plus a serial scan of the block.  Now:
//   * a sampled SELECT DIRECTORY, built on first use: per symbol, the block that holds every S-th occurrence
//     (S a power of two chosen per symbol so that S occurrences span about two blocks: S = 4 at sigma = 128,
//     S = 128 at sigma = 4; 4 bytes per sample, at most ~n bytes in all).  Query: dir[j/S], dir[j/S + 1] -- one
//     request -- bound the block;
//   * the lane group (quad / octet) probes the headers of that range four at a time -- one more round of requests,
//     more only where occurrences are locally sparse;
//   * the block is fetched 16 bytes per lane like a rank query, per-lane popcounts are prefix-summed inside the
//     group and the owning lane picks the bit with five halving steps.
// Three dependent requests per Psi step instead of ~23.
#include "fmx_device.h"
#include "fmx_host.h"

#include <chrono>

namespace fmx {

constexpr int kSelThreads = 256;

struct SelDir {
  const uint32_t *dir;       // all symbols' samples, slot after slot
  const uint64_t *off;       // [nslots + 1] first sample of each slot
  const uint8_t *shift;      // [nslots] log2 S
};

// occurrences of slot s before block b (one-hot: the block header; bytes layout: checkpoint [+ superblock count])
__device__ __forceinline__ uint64_t occ_before_block(const DevIndex &ix, uint32_t s, uint64_t b) {
  if (ix.layout == kLayoutBytes)
    return (ix.sup ? ix.sup[(b >> kSuperShift) * ix.nslots + s] : 0ull) + ix.chk[b * ix.nslots + s];
  return *reinterpret_cast<const uint64_t *>(ix.bv + ((uint64_t)s * ix.nblocks + b) * (kBlockBytes / 16));
}

// ---------------------------------------------------------------- directory build
// One thread per (slot, block): the samples i with  before(b) <= i * S < before(b + 1)  live in block b.
__global__ __launch_bounds__(256) void k_sel_build(DevIndex ix, const uint64_t *__restrict__ totals /* [nslots] */,
                                                    const uint64_t *__restrict__ off, const uint8_t *__restrict__ shift,
                                                    uint32_t *__restrict__ dir) {
  const uint64_t nb = ix.nblocks;
  const uint64_t total = (uint64_t)ix.nslots * nb;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t s = (uint32_t)(i / nb);
    const uint64_t b = i % nb;
    const uint64_t h0 = occ_before_block(ix, s, b);
    const uint64_t h1 = b + 1 < nb ? occ_before_block(ix, s, b + 1) : totals[s];
    const uint32_t k = shift[s];
    uint64_t smp = (h0 + ((1ull << k) - 1)) >> k;                 // first sample at or after h0
    for (; (smp << k) < h1; smp++) dir[off[s] + smp] = (uint32_t)b;
    if (b + 1 == nb) {                                            // the closing entry: no occurrence lies past the last block
      const uint64_t nsmp = (totals[s] + ((1ull << k) - 1)) >> k;
      dir[off[s] + nsmp] = (uint32_t)b;
    }
  }
}

// ---------------------------------------------------------------- the query
// Symbol whose bucket holds `row` (pos2char, bwtmerger.scala:376-385): the LAST c with cf[c] <= row -- symbols
// without occurrences share their start with the next present one.
__device__ __forceinline__ uint32_t owner_of(const uint64_t *s_cf, uint64_t row) {
  uint32_t lo = 0, hi = 255;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (s_cf[mid] <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// position of the `need`-th (0-based) set bit of v; v has more than `need` set bits
__device__ __forceinline__ uint32_t select_in_dword(uint32_t v, uint32_t need) {
  uint32_t pos = 0;
#pragma unroll
  for (uint32_t wd = 16; wd >= 1; wd >>= 1) {
    const uint32_t pc = (uint32_t)__builtin_popcount(v & ((1u << wd) - 1u));
    if (need >= pc) { need -= pc; v >>= wd; pos += wd; }
  }
  return pos;
}

// Psi for the whole lane group (every lane of the group passes the same row < n, gets the same answer).
template <uint32_t LAYOUT>
__device__ __forceinline__ uint64_t psi_group(const DevIndex &ix, const SelDir &sd, const uint64_t *s_cf, const uint16_t *s_slot,
                                              uint64_t row, const LaneConst &lc, uint32_t &c_out) {
  constexpr int G = Lay<LAYOUT>::G;
  const uint32_t c = owner_of(s_cf, row);
  c_out = c;
  const uint16_t s = s_slot[c];
  if (s >= kSlotEof) return ix.eof;            // bucket 0 = the EOF row (absent symbols own no row)
  const uint64_t j = row - s_cf[c];            // 0-based occurrence wanted
  const uint32_t k = sd.shift[s];
  const uint64_t e = sd.off[s] + (j >> k);
  uint64_t lo = sd.dir[e], hi = sd.dir[e + 1];
  const uint32_t t4 = lc.t & 3u;               // an octet probes with its first four lanes (the others repeat them)
  // narrow [lo, hi] to the last block whose count-before is <= j, four probes per round
  for (;;) {
    const uint64_t width = hi - lo + 1;
    if (width <= 4) {
      const uint64_t b = lo + t4;
      const bool le = t4 != 0 && b <= hi && occ_before_block(ix, s, b) <= j;
      uint32_t cnt = group_sum<G>((lc.t < 4u && le) ? 1u : 0u);
      lo += cnt;
      break;
    }
    const uint64_t q = lo + (width * (t4 + 1)) / 5;            // lo < q <= hi, increasing in t
    const bool le = occ_before_block(ix, s, q) <= j;
    const uint32_t cnt = group_sum<G>((lc.t < 4u && le) ? 1u : 0u);   // probes 0 .. cnt-1 are <= j
    const uint64_t q_lo = lo + (width * cnt) / 5, q_hi = lo + (width * (cnt + 1)) / 5;
    if (cnt < 4) hi = q_hi - 1;
    if (cnt > 0) lo = q_lo;
  }
  uint32_t need = (uint32_t)(j - occ_before_block(ix, s, lo));
  if (LAYOUT == kLayoutBytes) {
    // 16 block bytes per lane: matches per lane, prefix inside the octet, the owning lane picks its byte
    const uint4 w = load_line16((uint64_t)(uintptr_t)ix.bwt + lo * kByteBlock + lc.t * 16u);
    const uint32_t mine = match_count16(w, c, 16u, 0u);
    uint32_t before = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t v = (uint32_t)__shfl((int)mine, (int)((__lane_id() & ~7u) | u), 64);
      before += (uint32_t)u < lc.t ? v : 0u;
    }
    uint32_t found = 0;
    if (need >= before && need < before + mine) {
      uint32_t nd = need - before;
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
      const uint32_t pat = c * 0x01010101u;
      uint32_t res = 0;
      bool done = false;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const uint32_t z = ww[d] ^ pat;
        const uint32_t eq = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);   // 0x80 per equal byte
        const uint32_t pc = (uint32_t)__builtin_popcount(eq);
        if (!done) {
          if (nd < pc) { res = 4u * d + (select_in_dword(eq, nd) >> 3); done = true; }
          else nd -= pc;
        }
      }
      found = 1u + lc.t * 16u + res;          // +1: 0 means "not this lane"
    }
    const uint32_t at = group_or<G>(found) - 1u;
    return lo * kByteBlock + at;
  }
  // one-hot block: lane t holds dwords 4t .. 4t+3 of the block (dwords 0, 1 = header)
  const uint4 w = load_line16(block_addr(ix, s, (uint32_t)lo, lc));
  const uint32_t ww[4] = {lc.t == 0 ? 0u : w.x, lc.t == 0 ? 0u : w.y, w.z, w.w};
  const uint32_t mine = (uint32_t)(__builtin_popcount(ww[0]) + __builtin_popcount(ww[1]) + __builtin_popcount(ww[2]) +
                                   __builtin_popcount(ww[3]));
  const uint32_t c0 = group_bcast<G, 0>(mine), c1 = group_bcast<G, 1>(mine), c2 = group_bcast<G, 2>(mine);
  const uint32_t before = lc.t == 0 ? 0u : (lc.t == 1 ? c0 : (lc.t == 2 ? c0 + c1 : c0 + c1 + c2));
  uint32_t found = 0;
  if (need >= before && need < before + mine) {
    uint32_t nd = need - before, res = 0;
    bool done = false;
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t pc = (uint32_t)__builtin_popcount(ww[d]);
      if (!done) {
        if (nd < pc) { res = 32u * d + select_in_dword(ww[d], nd); done = true; }
        else nd -= pc;
      }
    }
    found = 1u + lc.t * 128u + res - 64u;      // payload position: dword (4t + d) starts at 32 * (4t + d - 2)
  }
  const uint32_t at = group_or<G>(found) - 1u;
  return lo * (uint64_t)kBlockBits + at;
}

template <uint32_t LAYOUT>
__global__ __launch_bounds__(kSelThreads) void k_psi(DevIndex ix, SelDir sd, const uint64_t *__restrict__ rows,
                                                      uint64_t *__restrict__ out, uint64_t k) {
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint64_t ngroups = (uint64_t)gridDim.x * (kSelThreads / G);
  for (uint64_t q = ((uint64_t)blockIdx.x * kSelThreads + threadIdx.x) / G; q < k; q += ngroups) {
    uint64_t row = rows[q];
    if (row >= ix.n) row = ix.n - 1;            // unvalidated device operands stay inside the index
    uint32_t c;
    const uint64_t p = psi_group<LAYOUT>(ix, sd, s_cf, s_slot, row, lc, c);
    if (lc.t == 0) out[q] = p;
  }
}

// NaiveFMSearcher.nextSubstr (bwtmerger.scala:394-405) for k independent starts: cp = getNextI(sp); then up to
// len times: b = bwt.read(cp), stop after b == 0, cp = getNextI(cp).  bwt.read(Psi(r)) is the symbol whose bucket
// holds r (that is what Psi selects), so the walk needs no BWT reads: emit owner(r), step r = Psi(r).  Bytes are
// written in walk order (the host reverses, :404).
template <uint32_t LAYOUT>
__global__ __launch_bounds__(kSelThreads) void k_next_substr(DevIndex ix, SelDir sd, const uint64_t *__restrict__ sps, uint64_t k,
                                                              uint32_t len, uint8_t *__restrict__ out,
                                                              uint32_t *__restrict__ out_len) {
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint64_t ngroups = (uint64_t)gridDim.x * (kSelThreads / G);
  for (uint64_t q = ((uint64_t)blockIdx.x * kSelThreads + threadIdx.x) / G; q < k; q += ngroups) {
    uint64_t row = sps[q];
    if (row >= ix.n) row = ix.n - 1;
    uint32_t w = 0;
    for (uint32_t st = 0; st < len; st++) {
      uint32_t c;
      const uint64_t nxt = psi_group<LAYOUT>(ix, sd, s_cf, s_slot, row, lc, c);
      if (lc.t == 0) out[q * len + w] = (uint8_t)c;
      w++;
      if (c == 0) break;
      row = nxt;
    }
    if (lc.t == 0) out_len[q] = w;
  }
}

// ---------------------------------------------------------------- host side
// Builds the directory on first use (most handles never extract text; C3's costs 4 GiB and a pass over the block
// headers).  Guarded by the handle's mutex; the pointers never change afterwards.
static hipError_t ensure_select(const Index *h, hipStream_t st, SelDir *out) {
  std::lock_guard<std::mutex> lk(h->sel_mu);
  if (!h->sel_ready) {
    const auto t_build = std::chrono::steady_clock::now();
    std::vector<uint64_t> off(h->nslots + 1, 0), totals(h->nslots ? h->nslots : 1, 0);
    std::vector<uint8_t> shift(h->nslots ? h->nslots : 1, 0);
    const double rows_per_block = h->layout == kLayoutBytes ? (double)kByteBlock : (double)kBlockBits;
    for (int c = 1; c < 256; c++) {
      const uint16_t s = h->slot[c];
      if (s >= kSlotEof) continue;
      const uint64_t cnt = (uint64_t)h->counts[c];
      totals[s] = cnt;
      // S occurrences span about two blocks (four in the bytes layout, whose blocks are small): S ~ 2 * density * block
      const double want = (h->layout == kLayoutBytes ? 4.0 : 2.0) * rows_per_block * (double)cnt / (double)h->n;
      uint32_t k = 0;
      while (k < 8 && (double)(2u << k) <= want) k++;
      shift[s] = (uint8_t)k;
    }
    for (uint32_t s = 0; s < h->nslots; s++) off[s + 1] = off[s] + ((totals[s] + ((1ull << shift[s]) - 1)) >> shift[s]) + 1;
    const uint64_t entries = off[h->nslots];
    hipError_t e = table_malloc(h, &h->d_sel_dir, (entries ? entries : 1) * 4 + 16);
    if (e == hipSuccess) e = hipMalloc(&h->d_sel_off, off.size() * 8);
    if (e == hipSuccess) e = hipMalloc(&h->d_sel_shift, shift.size());
    uint64_t *d_tot = nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&d_tot, totals.size() * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_sel_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_sel_shift, shift.data(), shift.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tot, totals.data(), totals.size() * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && h->nslots) {
      const uint64_t work = (uint64_t)h->nslots * h->nblocks;
      const int grid = (int)std::min<uint64_t>((work + 255) / 256, (uint64_t)h->cu_count * 32);
      k_sel_build<<<grid, 256, 0, st>>>(h->dev, d_tot, (const uint64_t *)h->d_sel_off, (const uint8_t *)h->d_sel_shift,
                                         (uint32_t *)h->d_sel_dir);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);           // the host vectors go out of scope
    if (d_tot) (void)hipFree(d_tot);
    if (e != hipSuccess) return e;
    h->sel_bytes = entries * 4;
    tables_account(h, (int64_t)h->sel_bytes);      // (Psi cannot do without it: counted against the budget, never refused by it)
    h->sel_ready = true;
    h->tables_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
  }
  out->dir = (const uint32_t *)h->d_sel_dir;
  out->off = (const uint64_t *)h->d_sel_off;
  out->shift = (const uint8_t *)h->d_sel_shift;
  return hipSuccess;
}

hipError_t select_prepare(const Index *h, hipStream_t st) {
  SelDir d;
  return ensure_select(h, st, &d);
}

static inline int sel_grid(const Index *h, uint64_t k, int per_block) {
  uint64_t want = (k + per_block - 1) / per_block;
  const uint64_t cap = (uint64_t)h->cu_count * 8;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

hipError_t launch_psi(const Index *h, const void *d_rows, void *d_out, uint64_t k, hipStream_t st) {
  if (!k) return hipSuccess;
  SelDir sd;
  hipError_t e = ensure_select(h, st, &sd);
  if (e != hipSuccess) return e;
  if (h->layout == kLayoutBytes)
    k_psi<kLayoutBytes><<<sel_grid(h, k, kSelThreads / 8), kSelThreads, 0, st>>>(h->dev, sd, (const uint64_t *)d_rows, (uint64_t *)d_out, k);
  else
    k_psi<kLayoutOneHot><<<sel_grid(h, k, kSelThreads / 4), kSelThreads, 0, st>>>(h->dev, sd, (const uint64_t *)d_rows, (uint64_t *)d_out, k);
  return hipGetLastError();
}

hipError_t launch_next_substr(const Index *h, const void *d_sps, uint64_t k, uint32_t len, void *d_out,
                              void *d_out_len, hipStream_t st) {
  if (!k) return hipSuccess;
  SelDir sd;
  hipError_t e = ensure_select(h, st, &sd);
  if (e != hipSuccess) return e;
  if (h->layout == kLayoutBytes)
    k_next_substr<kLayoutBytes><<<sel_grid(h, k, kSelThreads / 8), kSelThreads, 0, st>>>(h->dev, sd, (const uint64_t *)d_sps, k, len,
                                                                                         (uint8_t *)d_out, (uint32_t *)d_out_len);
  else
    k_next_substr<kLayoutOneHot><<<sel_grid(h, k, kSelThreads / 4), kSelThreads, 0, st>>>(h->dev, sd, (const uint64_t *)d_sps, k, len,
                                                                                          (uint8_t *)d_out, (uint32_t *)d_out_len);
  return hipGetLastError();
}

}  // namespace fmx

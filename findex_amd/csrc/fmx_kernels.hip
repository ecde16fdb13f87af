// fmx_kernels.hip -- query kernels over the rank dictionary (gfx950, wave64).
//
// K2 occ_batch, K3 literal backward search (generic form), K4 prev_range_batch, LF walks.
// All are HBM-bound integer work: one 64-byte block per rank query, fetched whole by a quad of
// lanes (an octet and two lines in the bytes layout; fmx_device.h); C[] and the symbol->slot map
// are staged in LDS per workgroup.  "group" below = the lanes that serve one query.
// Reference semantics: SuffixAlgo (findex.scala:9-52), NaiveFMSearcher (bwtmerger.scala:335-421).
#include <type_traits>

#include "fmx_device.h"
#include "fmx_host.h"

namespace fmx {

constexpr int kThreads = 256;                       // 4 waves, 64 quads (32 octets) per workgroup

struct Tables {
  uint64_t cf[256];
  uint16_t slot[256];
};

__device__ __forceinline__ void stage_tables(const DevIndex &ix, Tables &tb) {
  for (int c = threadIdx.x; c < 256; c += blockDim.x) {
    tb.cf[c] = ix.cf[c];
    tb.slot[c] = ix.slot[c];
  }
  __syncthreads();
}

template <bool WIDE, uint32_t LAYOUT>
__device__ __forceinline__ void step(const DevIndex &ix, const Tables &tb, uint32_t c, const LaneConst &lc,
                                     uint64_t &sp, uint64_t &ep) {
  backward_step<WIDE, LAYOUT>(ix, c, tb.slot[c], tb.cf[c], lc, sp, ep);
}

// ---------------------------------------------------------------- K2: occ_batch
// SuffixAlgo.occ(c,i) = rank_excl(c, i+1); i < 0 -> 0; i >= n clamps to n-1.
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kThreads) void k_occ(DevIndex ix, const uint8_t *__restrict__ c,
                                                   const int64_t *__restrict__ i, uint64_t *__restrict__ out,
                                                   uint64_t k, unsigned long long *__restrict__ counters) {
  __shared__ Tables tb;
  stage_tables(ix, tb);
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint32_t t = lc.t;
  uint32_t done = 0;
  auto boundary = [&](int64_t key) { return key < 0 ? 0 : ((uint64_t)key >= ix.n ? ix.n : (uint64_t)key + 1); };
  // A group takes 4 consecutive queries per trip: lane u reads the operands of query u (so a wave reads
  // 64 consecutive operands with one coalesced instruction each, instead of every lane of a group asking
  // for the same one), the group broadcasts them, requests the 4 blocks together and lane u stores answer u.
  const uint64_t ngroups = (uint64_t)gridDim.x * (kThreads / G);
  const uint64_t gid = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  const uint32_t tu = t < 4u ? t : 3u;
  for (uint64_t base = gid * 4; base < k; base += ngroups * 4) {
    const uint64_t myq = base + tu;
    const bool own = myq < k;
    const uint32_t myc = own ? c[myq] : 0u;
    const uint64_t myx = own ? boundary(i[myq]) : 0;
    RankReq r[4];
    uint32_t cc[4];
    auto issue = [&](auto U) {
      constexpr int u = decltype(U)::value;
      cc[u] = group_bcast<G, u>(myc);
      const uint64_t x = group_bcast64<G, u>(myx);
      r[u] = rank_issue<LAYOUT>(ix, base + u < k ? tb.slot[cc[u]] : kSlotNone, x, lc);
    };
    issue(std::integral_constant<int, 0>{}); issue(std::integral_constant<int, 1>{});
    issue(std::integral_constant<int, 2>{}); issue(std::integral_constant<int, 3>{});
    uint64_t mine = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t v = rank_complete<WIDE, LAYOUT>(r[u], cc[u], lc);
      if (t == (uint32_t)u) mine = v;
    }
    if (t < 4u && own) { out[myq] = mine; done++; }
  }
  counters_add(counters, done, 0, 0);
}

// ---------------------------------------------------------------- K4: prev_range_batch
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kThreads) void k_prev_range(DevIndex ix, const uint64_t *__restrict__ sp_in,
                                                          const uint64_t *__restrict__ ep_in,
                                                          const uint8_t *__restrict__ c, uint64_t *__restrict__ sp1,
                                                          uint64_t *__restrict__ ep1, uint64_t k,
                                                          unsigned long long *__restrict__ counters) {
  __shared__ Tables tb;
  stage_tables(ix, tb);
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint32_t t = lc.t;
  uint32_t done = 0;
  // 4 consecutive steps per group and trip, operands read and results stored one per lane (as in k_occ)
  const uint64_t ngroups = (uint64_t)gridDim.x * (kThreads / G);
  const uint64_t gid = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  const uint32_t tu = t < 4u ? t : 3u;
  for (uint64_t base = gid * 4; base < k; base += ngroups * 4) {
    const uint64_t myq = base + tu;
    const bool own = myq < k;
    // the device-pointer entry point cannot validate its operands on the host: keep them inside the index
    uint64_t mysp = own ? sp_in[myq] : 0, myep = own ? ep_in[myq] : 0;
    if (mysp > ix.n) mysp = ix.n;
    if (myep > ix.n) myep = ix.n;
    const uint32_t myc = own ? c[myq] : 0u;
    RankReq r1[4], r2[4];
    uint32_t cc[4];
    auto issue = [&](auto U) {
      constexpr int u = decltype(U)::value;
      cc[u] = group_bcast<G, u>(myc);
      const uint16_t slot = base + u < k ? tb.slot[cc[u]] : kSlotNone;
      r1[u] = rank_issue<LAYOUT>(ix, slot, group_bcast64<G, u>(mysp), lc);
      r2[u] = rank_issue<LAYOUT>(ix, slot, group_bcast64<G, u>(myep), lc);
    };
    issue(std::integral_constant<int, 0>{}); issue(std::integral_constant<int, 1>{});
    issue(std::integral_constant<int, 2>{}); issue(std::integral_constant<int, 3>{});
    uint64_t osp = 0, oep = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t cfc = tb.cf[cc[u]];
      const uint64_t a1 = cfc + rank_complete<WIDE, LAYOUT>(r1[u], cc[u], lc);
      const uint64_t a2 = cfc + rank_complete<WIDE, LAYOUT>(r2[u], cc[u], lc);
      if (t == (uint32_t)u) { osp = a1; oep = a2; }
    }
    if (t < 4u && own) { sp1[myq] = osp; ep1[myq] = oep; done++; }
  }
  counters_add(counters, 2ull * done, done, 0);
}

// ---------------------------------------------------------------- K3: literal backward search
// SuffixAlgo.search (findex.scala:15-31): one pattern per lane group, groups walk the batch with a
// grid stride and pick up their next pattern as soon as the current one ends (last byte consumed
// or interval empty), so early exits do not idle lanes.  The next pattern's offsets and the next
// pattern byte are requested a step early; only the two rank lines are on the dependent chain.
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kThreads) void k_search(DevIndex ix, const uint8_t *__restrict__ pat,
                                                      const PatOff off, uint64_t *__restrict__ sp_out,
                                                      uint64_t *__restrict__ ep_out, uint64_t k,
                                                      unsigned long long *__restrict__ counters) {
  __shared__ Tables tb;
  stage_tables(ix, tb);
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint32_t t = lc.t;
  const uint64_t noct = (uint64_t)gridDim.x * (kThreads / G);      // lane groups in the grid
  uint64_t p = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  bool active = p < k;
  uint64_t base = 0, sp = 0, ep = ix.n;
  int64_t i = -1;          // index of the byte to consume next
  uint32_t c = 0;
  uint64_t nb = 0, ne = 0; // offsets of the pattern this group takes next
  uint32_t steps = 0;
  if (active) {
    uint64_t e0;
    off.get(p, base, e0);
    i = (int64_t)(e0 - base) - 1;
    if (i >= 0) c = pat[base + i];
    if (p + noct < k) off.get(p + noct, nb, ne);
  }
  while (__builtin_amdgcn_ballot_w64(active)) {
    if (active) {
      if (i >= 0 && sp < ep) {
        const uint32_t cn = i > 0 ? pat[base + i - 1] : 0;   // next byte, off the critical path
        step<WIDE, LAYOUT>(ix, tb, c, lc, sp, ep);
        c = cn;
        i--;
        steps++;
      } else {
        if (t == 0) { sp_out[p] = sp; ep_out[p] = ep; }
        p += noct;
        active = p < k;
        if (active) {
          base = nb;
          i = (int64_t)(ne - nb) - 1;
          sp = 0;
          ep = ix.n;
          if (i >= 0) c = pat[base + i];
          if (p + noct < k) off.get(p + noct, nb, ne);
        }
      }
    }
  }
  counters_add(counters, t == 0 ? 2ull * steps : 0ull, t == 0 ? steps : 0u, 0);
}

// ---------------------------------------------------------------- LF walk (prevSubstr / getPrevI)
// NaiveFMSearcher.prevSubstr (bwtmerger.scala:409-419): emit BWT'[row], row = cf(b)+occ(b,row-1).
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kThreads) void k_lf_walk(DevIndex ix, const uint64_t *__restrict__ rows, uint64_t k,
                                                       uint32_t len, uint8_t *__restrict__ out_bytes,
                                                       uint64_t *__restrict__ end_rows,
                                                       unsigned long long *__restrict__ counters) {
  __shared__ Tables tb;
  stage_tables(ix, tb);
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint32_t t = lc.t;
  const uint64_t noct = (uint64_t)gridDim.x * (kThreads / G);      // lane groups in the grid
  uint32_t done = 0;
  // two walks per group, stepped together: their (dependent) chains overlap
  for (uint64_t q = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) / G; q < k; q += 2 * noct) {
    const uint64_t q2 = q + noct;
    const bool two = q2 < k;
    uint64_t ra = rows[q], rb = two ? rows[q2] : 0;
    if (ra >= ix.n) ra = ix.n - 1;          // unvalidated device operands stay inside the index
    if (rb >= ix.n) rb = ix.n - 1;
    for (uint32_t s = 0; s < len; s++) {
      const uint32_t ba = ra == ix.eof ? 0u : ix.bwt[ra];
      const uint32_t bb = two ? (rb == ix.eof ? 0u : ix.bwt[rb]) : 0u;
      if (out_bytes && t == 0) {
        out_bytes[q * len + s] = (uint8_t)ba;
        if (two) out_bytes[q2 * len + s] = (uint8_t)bb;
      }
      const RankReq qa = rank_issue<LAYOUT>(ix, tb.slot[ba], ra, lc);
      RankReq qb = qa;
      if (two) qb = rank_issue<LAYOUT>(ix, tb.slot[bb], rb, lc);
      ra = tb.cf[ba] + rank_complete<WIDE, LAYOUT>(qa, ba, lc);
      if (two) rb = tb.cf[bb] + rank_complete<WIDE, LAYOUT>(qb, bb, lc);
    }
    if (end_rows && t == 0) {
      end_rows[q] = ra;
      if (two) end_rows[q2] = rb;
    }
    done += two ? 2 * len : len;
  }
  counters_add(counters, t == 0 ? done : 0u, 0, 0);
}

// ---------------------------------------------------------------- .fm payload (FMCreator)
// FMCreator.create (bwtmerger.scala:452-532) bucket-sorts BWT positions by symbol; entry r of the
// result is the position p with LF(p) = r.  One LF step per position, scattered as 4-byte
// big-endian ints (the wire format, :476-481).
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kThreads) void k_fm_fill(DevIndex ix, uint64_t p0, uint64_t p1, uint32_t *__restrict__ fm) {
  __shared__ Tables tb;
  stage_tables(ix, tb);
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint64_t noct = (uint64_t)gridDim.x * (kThreads / G);      // lane groups in the grid
  for (uint64_t p = p0 + (((uint64_t)blockIdx.x * kThreads + threadIdx.x) / G); p < p1; p += 2 * noct) {
    const uint64_t pb = p + noct;
    const bool two = pb < p1;
    const uint32_t b1 = p == ix.eof ? 0u : ix.bwt[p];
    const uint32_t b2 = two ? (pb == ix.eof ? 0u : ix.bwt[pb]) : 0u;
    const RankReq r1 = rank_issue<LAYOUT>(ix, tb.slot[b1], p, lc);
    RankReq r2 = r1;
    if (two) r2 = rank_issue<LAYOUT>(ix, tb.slot[b2], pb, lc);
    const uint64_t v1 = tb.cf[b1] + rank_complete<WIDE, LAYOUT>(r1, b1, lc);
    if (lc.t == 0) fm[v1] = __builtin_bswap32((uint32_t)p);
    if (two) {
      const uint64_t v2 = tb.cf[b2] + rank_complete<WIDE, LAYOUT>(r2, b2, lc);
      if (lc.t == 0) fm[v2] = __builtin_bswap32((uint32_t)pb);
    }
  }
}

// Psi (getNextI) and nextSubstr: fmx_select.hip.

// ---------------------------------------------------------------- operand check for the device-pointer search
__global__ __launch_bounds__(kThreads) void k_check_offsets(const uint64_t *__restrict__ off, uint64_t k,
                                                             unsigned int *__restrict__ bad) {
  for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < k; q += (uint64_t)gridDim.x * blockDim.x)
    if (off[q + 1] < off[q]) atomicOr(bad, 1u);
}

hipError_t check_offsets(const Index *h, const void *d_off, uint64_t k, hipStream_t st, bool *ok) {
  unsigned int *d_bad = nullptr, bad = 0;
  hipError_t e = hipMallocAsync((void **)&d_bad, sizeof bad, st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d_bad, 0, sizeof bad, st);
  if (e == hipSuccess) {
    uint64_t want = (k + kThreads - 1) / kThreads, cap = (uint64_t)h->cu_count * 8;
    k_check_offsets<<<(int)(want < cap ? want : cap), kThreads, 0, st>>>((const uint64_t *)d_off, k, d_bad);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFreeAsync(d_bad, st);
  *ok = bad == 0;
  return e;
}

// ---------------------------------------------------------------- launchers
// Counts fit 32 bits iff n <= 2^32; the one-hot kernels then reduce the block header with the popcounts.
static inline int grid_for(const Index *h, uint64_t k, int per_block) {
  uint64_t want = (k + per_block - 1) / per_block;
  uint64_t cap = (uint64_t)h->cu_count * 8;   // 8 x 256 threads fill a CU's 32 wave slots
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

hipError_t launch_occ(const Index *h, const void *d_c, const void *d_i, void *d_out, uint64_t k, hipStream_t st) {
  if (!k) return hipSuccess;
#define CALL(W, L)                                                                                              \
  k_occ<W, L><<<grid_for(h, k, kThreads / Lay<L>::G), kThreads, 0, st>>>(h->dev, (const uint8_t *)d_c,         \
                                                                         (const int64_t *)d_i, (uint64_t *)d_out, k, h->d_counters)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  return hipGetLastError();
}

hipError_t launch_prev_range(const Index *h, const void *d_sp, const void *d_ep, const void *d_c, void *d_sp1,
                             void *d_ep1, uint64_t k, hipStream_t st) {
  if (!k) return hipSuccess;
#define CALL(W, L)                                                                                              \
  k_prev_range<W, L><<<grid_for(h, k, kThreads / Lay<L>::G), kThreads, 0, st>>>(                               \
      h->dev, (const uint64_t *)d_sp, (const uint64_t *)d_ep, (const uint8_t *)d_c, (uint64_t *)d_sp1, (uint64_t *)d_ep1, k, h->d_counters)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  return hipGetLastError();
}

// generic search kernel: one pattern per group, byte-at-a-time pattern reads (FMX_SEARCH_VARIANT=1,
// and batches of 2^32 patterns or more)
hipError_t launch_search_v1(const Index *h, const void *d_pat, PatOff po, void *d_sp, void *d_ep, uint64_t k,
                            hipStream_t st) {
  if (!k) return hipSuccess;
#define CALL(W, L)                                                                                              \
  k_search<W, L><<<grid_for(h, k, kThreads / Lay<L>::G), kThreads, 0, st>>>(                                   \
      h->dev, (const uint8_t *)d_pat, po, (uint64_t *)d_sp, (uint64_t *)d_ep, k, h->d_counters)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  return hipGetLastError();
}

// ---------------------------------------------------------------- the 8-byte form of a batch's intervals (fmx.h)
// word q = sp | w << 40 with w = min(ep - sp, 0xFFFFFF); w == 0xFFFFFF: the interval is that wide or wider and its ep is
// in the escape list behind the k words: word k counts the wide intervals, pairs (q, ep) follow (the first escape_cap of
// them; their order is whatever the atomics made it, unpacking scatters by q).  In place when d_packed == d_sp: a thread
// reads its own sp before it writes its word.
__global__ __launch_bounds__(kThreads) void k_pack_intervals(const uint64_t *sp, const uint64_t *__restrict__ ep, uint64_t k,
                                                             uint64_t escape_cap, unsigned long long *packed) {
  const uint64_t nth = (uint64_t)gridDim.x * kThreads;
  for (uint64_t q = (uint64_t)blockIdx.x * kThreads + threadIdx.x; q < k; q += nth) {
    const uint64_t a = sp[q], b = ep[q];
    const uint64_t w = b - a;                       // sp <= ep always (occ is monotone); a miss has w == 0
    unsigned long long word = a | (kPackWide << 40);
    if (w < kPackWide) word = a | (w << 40);
    else {
      const unsigned long long slot = atomicAdd(packed + k, 1ull);
      if (slot < escape_cap) { packed[k + 1 + 2 * slot] = q; packed[k + 2 + 2 * slot] = b; }
    }
    packed[q] = word;
  }
}
// (sp, ep) from the words, then the escape list's entries on top (pass 0 / pass 1: two launches, so that a wide
// interval's ep is written after its word has been expanded)
__global__ __launch_bounds__(kThreads) void k_unpack_intervals(const unsigned long long *__restrict__ packed, uint64_t k,
                                                               uint64_t escape_cap, uint64_t *__restrict__ sp,
                                                               uint64_t *__restrict__ ep, int pass) {
  const uint64_t nth = (uint64_t)gridDim.x * kThreads;
  if (pass == 0) {
    for (uint64_t q = (uint64_t)blockIdx.x * kThreads + threadIdx.x; q < k; q += nth) {
      const unsigned long long w = packed[q];
      const uint64_t a = w & ((1ull << 40) - 1);
      sp[q] = a;
      ep[q] = a + (w >> 40);
    }
  } else {
    const uint64_t cnt = packed[k] < escape_cap ? packed[k] : escape_cap;
    for (uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x; j < cnt; j += nth) {
      const uint64_t q = packed[k + 1 + 2 * j];
      if (q < k) ep[q] = packed[k + 2 + 2 * j];
    }
  }
}

hipError_t launch_pack_intervals(const Index *h, const void *d_sp, const void *d_ep, uint64_t k, uint64_t escape_cap,
                                 void *d_packed, hipStream_t st) {
  hipError_t e = hipMemsetAsync(static_cast<unsigned long long *>(d_packed) + k, 0, 8, st);
  if (e != hipSuccess || !k) return e;
  k_pack_intervals<<<grid_for(h, k, kThreads), kThreads, 0, st>>>((const uint64_t *)d_sp, (const uint64_t *)d_ep, k, escape_cap,
                                                                   (unsigned long long *)d_packed);
  return hipGetLastError();
}

hipError_t launch_unpack_intervals(const Index *h, const void *d_packed, uint64_t k, uint64_t escape_cap, void *d_sp,
                                   void *d_ep, hipStream_t st) {
  if (!k) return hipSuccess;
  k_unpack_intervals<<<grid_for(h, k, kThreads), kThreads, 0, st>>>((const unsigned long long *)d_packed, k, escape_cap,
                                                                     (uint64_t *)d_sp, (uint64_t *)d_ep, 0);
  if (escape_cap)
    k_unpack_intervals<<<grid_for(h, escape_cap, kThreads), kThreads, 0, st>>>((const unsigned long long *)d_packed, k, escape_cap,
                                                                                (uint64_t *)d_sp, (uint64_t *)d_ep, 1);
  return hipGetLastError();
}

hipError_t launch_lf_walk(const Index *h, const void *d_rows, uint64_t k, uint32_t len, void *d_out, void *d_end,
                          hipStream_t st) {
  if (!k) return hipSuccess;
#define CALL(W, L)                                                                                              \
  k_lf_walk<W, L><<<grid_for(h, k, kThreads / Lay<L>::G), kThreads, 0, st>>>(                                  \
      h->dev, (const uint64_t *)d_rows, k, len, (uint8_t *)d_out, (uint64_t *)d_end, h->d_counters)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  return hipGetLastError();
}

hipError_t launch_fm_fill(const Index *h, void *d_fm, hipStream_t st) {
#define CALL(W, L)                                                                                              \
  k_fm_fill<W, L><<<grid_for(h, h->n, kThreads / Lay<L>::G), kThreads, 0, st>>>(h->dev, (uint64_t)0, h->n, (uint32_t *)d_fm)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  return hipGetLastError();
}

}  // namespace fmx

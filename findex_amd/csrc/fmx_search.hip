// fmx_search.hip -- K3: batched literal backward search (SuffixAlgo.search, findex.scala:15-31).
//
// The dominant kernel of the headline benchmark.  Per backward step an octet of lanes fetches the
// two rank-dictionary lines of (c, sp) and (c, ep) and turns them into the next interval; the
// only dependent chain is line -> popcount -> next line address.  What the kernel does to keep
// the memory system busy while each chain waits:
//   * U patterns per octet are stepped in the same loop trip: all 2U lines are requested before
//     any is consumed (U x the lines in flight per wave at the same wave count);
//   * nothing on the step path waits for a load issued in the same trip except those lines:
//     pattern bytes are read 4 at a time, two dwords ahead; the descriptor of the pattern an
//     octet takes next (offset, length, last 4 bytes; written by the k_prep pre-pass) is
//     requested one whole pattern earlier;
//   * C[] and each symbol's bit-vector base address sit in LDS as one 16-byte entry per symbol.
// Octets pick up their next pattern as soon as one ends, so early exits do not idle lanes.
#include "fmx_device.h"
#include "fmx_host.h"

namespace fmx {

constexpr int kSThreads = 256;
constexpr int kSOctets = kSThreads / kOctet;

struct PatDesc {      // 16 bytes, one per pattern
  uint64_t end;       // offset one past the pattern's last byte
  uint32_t len;
  uint32_t tail4;     // byte j = pat[end-1-j] (the first four bytes the search consumes)
};

// Bytes pat[pos-1], pat[pos-2], pat[pos-3], pat[pos-4] in byte lanes 0..3 (fewer when pos < 4).
__device__ __forceinline__ uint32_t fetch4(const uint8_t *__restrict__ pat, uint64_t pos) {
  if (pos >= 4) {
    uint32_t d;
    __builtin_memcpy(&d, pat + pos - 4, 4);          // unaligned dword load
    return __builtin_bswap32(d);
  }
  uint32_t r = 0;
  for (uint32_t j = 0; j < (uint32_t)pos; j++) r |= (uint32_t)pat[pos - 1 - j] << (8 * j);
  return r;
}

// Pre-pass: one descriptor per pattern, and a verdict on the batch's shape: `ragged` is set when some
// group of 8 consecutive patterns (one lockstep batch of k_search4) has lengths that differ by more
// than a quarter of its longest -- then the dynamic-refill kernel serves the call instead.
__global__ __launch_bounds__(256) void k_prep(const uint8_t *__restrict__ pat, const uint64_t *__restrict__ off,
                                               PatDesc *__restrict__ desc, uint32_t k, uint32_t *__restrict__ ragged) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t rounds = (k + stride - 1) / stride;          // same trip count for every lane: DPP below
  for (uint32_t rd = 0; rd < rounds; rd++) {
    const uint32_t q = rd * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t len32 = 0;
    if (q < k) {
      const uint64_t b = off[q], e = off[q + 1];
      const uint64_t len = e - b;
      PatDesc d;
      d.end = e;
      d.len = (uint32_t)len;
      d.tail4 = len ? fetch4(pat, e) : 0;
      desc[q] = d;
      len32 = d.len;
    }
    // max and min over the octet = over one lockstep batch (patterns past k count as the last length)
    uint32_t mx = len32, mn = q < k ? len32 : 0xFFFFFFFFu;
    mx = max(mx, dpp<kDppXor1>(mx)); mx = max(mx, dpp<kDppXor2>(mx)); mx = max(mx, dpp<kDppHalfMirror>(mx));
    mn = min(mn, dpp<kDppXor1>(mn)); mn = min(mn, dpp<kDppXor2>(mn)); mn = min(mn, dpp<kDppHalfMirror>(mn));
    if (q < k && (threadIdx.x & 7) == 0 && (uint64_t)(mx - mn) * 4 > mx) *ragged = 1u;
  }
}

// ---------------------------------------------------------------- lean single-pattern-per-octet kernel
// One pattern per octet, written for instruction count: the profile of the first version
// (FMX_SEARCH_VARIANT=1, fmx_kernels.hip) showed the kernel bound by vector-instruction issue
// (about 165 vector instructions per backward step and wave), not by HBM.  What this one does:
//   * the rank primitive of fmx_device.h (5 instructions per payload dword);
//   * C[] and each symbol's bit-vector base address in LDS as one 16-byte entry per symbol;
//   * pattern bytes read 4 at a time, two dwords ahead; the descriptor of the pattern an octet
//     takes next (offset, length, last 4 bytes; written by the k_prep pre-pass) is requested
//     one whole pattern earlier, so nothing on the step path waits for a load issued in the
//     same trip except the two rank lines;
//   * the retire / refill block is skipped with one wave-uniform test when no octet ends.
// Tried and dropped (slower, VALU-bound): 2 or 3 patterns per octet in one trip; skipping the
// second line load when sp and ep share a block (the extra select costs more than the request).
template <bool WIDE>
__global__ __launch_bounds__(kSThreads) void k_search3(DevIndex ix, const uint8_t *__restrict__ pat,
                                                        const PatDesc *__restrict__ desc,
                                                        uint64_t *__restrict__ sp_out, uint64_t *__restrict__ ep_out,
                                                        uint32_t k, unsigned long long *__restrict__ counters,
                                                        const uint32_t *__restrict__ ragged) {
  if (*ragged == 0u) return;          // uniform batch: k_search4 serves it
  __shared__ uint4 s_tab[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) {
    const uint64_t cf = ix.cf[c];
    const uint16_t s = ix.slot[c];
    uint64_t vb = 0;
    if (s < kSlotEof) vb = (uint64_t)(uintptr_t)ix.bv + (uint64_t)s * ix.nblocks * kBlockBytes;
    else if (s == kSlotEof) vb = 1;
    s_tab[c] = make_uint4((uint32_t)cf, (uint32_t)(cf >> 32), (uint32_t)vb, (uint32_t)(vb >> 32));
  }
  __syncthreads();
  const LaneConst lc = lane_const();
  const uint32_t t = lc.t;
  const uint32_t lane_off = t * 16;
  const uint32_t octet = (blockIdx.x * kSThreads + threadIdx.x) >> 3;
  const uint32_t stride = gridDim.x * kSOctets;

  uint32_t pid = octet;
  bool act = pid < k;
  uint32_t left = 0, ch = 0, nx = 0, nch = 0, steps = 0;
  uint64_t cur = 0, sp = 0, ep = ix.n;
  PatDesc nd;
  nd.end = 0; nd.len = 0; nd.tail4 = 0;
  if (act) {
    const PatDesc d = desc[pid];
    cur = d.end;
    left = d.len;
    ch = d.tail4;
    nch = 4;
    if (left > 4) nx = fetch4(pat, cur - 4);
    if ((uint64_t)pid + stride < k) nd = desc[pid + stride];
  }

  while (__builtin_amdgcn_ballot_w64(act)) {
    const bool stepping = act && left > 0 && sp < ep;
    if (stepping) {
      const uint4 e = s_tab[ch & 0xFFu];
      const uint64_t cfc = ((uint64_t)e.y << 32) | e.x;
      const uint64_t vb = ((uint64_t)e.w << 32) | e.z;
      // byte cursor bookkeeping (runs while the two lines are in flight on the common path)
      auto next_char = [&]() {
        ch >>= 8;
        nch -= 1;
        left -= 1;
        cur -= 1;
        if (nch == 0) {
          ch = nx;
          nch = 4;
          if (left > 4) nx = fetch4(pat, cur - 4);
        }
      };
      if (vb > 1) {
        uint32_t b1, b2, m1, m2;
        split960(sp, b1, m1);
        split960(ep, b2, m2);
        const uint64_t base = vb + lane_off;
        const uint4 w1 = load_line16(base + (uint64_t)b1 * kBlockBytes);
        const uint4 w2 = load_line16(base + (uint64_t)b2 * kBlockBytes);
        next_char();
        sp = cfc + rank_finish<WIDE>(w1, m1, lc);
        ep = cfc + rank_finish<WIDE>(w2, m2, lc);
      } else {                       // symbol absent from the BWT, or the EOF symbol 0
        next_char();
        const uint64_t r1 = (vb == 1 && sp > ix.eof) ? 1 : 0;
        const uint64_t r2 = (vb == 1 && ep > ix.eof) ? 1 : 0;
        sp = cfc + r1;
        ep = cfc + r2;
      }
      steps++;
    }
    // retire / refill: skipped by the whole wave when no octet ended this trip
    if (__builtin_amdgcn_ballot_w64(act && !stepping)) {
      if (act && !stepping) {
        if (t == 0) { sp_out[pid] = sp; ep_out[pid] = ep; }
        const uint64_t np = (uint64_t)pid + stride;
        act = np < k;
        if (act) {
          pid = (uint32_t)np;
          cur = nd.end;
          left = nd.len;
          ch = nd.tail4;
          nch = 4;
          sp = 0;
          ep = ix.n;
          if (left > 4) nx = fetch4(pat, cur - 4);
          if (np + stride < k) nd = desc[np + stride];
        }
      }
    }
  }
  if (t == 0 && steps) { atomicAdd(&counters[0], 2ull * steps); atomicAdd(&counters[1], (unsigned long long)steps); }
}


// ---------------------------------------------------------------- lockstep batches + single-row shortcut
// A wave takes 8 consecutive patterns (one per octet), steps them together and retires them
// together.  What that buys: the choice between the two step bodies below is wave-uniform.
//   * general step: two rank queries (sp and ep lines);
//   * single-row step, taken when every stepping octet holds an interval of exactly one row
//     (sigma = 128, n = 2^32: from the 6th of 32 steps on): one rank query plus one bit test --
//     [sp, sp+1) maps to [C[c] + rank(c, sp), + BWT'[sp] == c), and BWT'[sp] == c is bit sp of c's
//     own vector, i.e. a bit of the line already fetched.  Same result as getPrevRange
//     (findex.scala:32-36), half the popcount work and one line request instead of two.
// Octets whose pattern ends early idle until the batch ends (patterns of one batch should have
// similar lengths; the benchmark's do).
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kSThreads) void k_search4(DevIndex ix, const uint8_t *__restrict__ pat,
                                                        const PatDesc *__restrict__ desc,
                                                        uint64_t *__restrict__ sp_out, uint64_t *__restrict__ ep_out,
                                                        uint32_t k, unsigned long long *__restrict__ counters,
                                                        const uint32_t *__restrict__ ragged, uint32_t serve_ragged) {
  // k_prep's verdict: very uneven batches go to the dynamic-refill kernel (one-hot layout only)
  if (!serve_ragged && *ragged != 0u) return;
  // per symbol: {C[c], x} with x = byte address of the symbol's bit-vector (one-hot layout) or its
  // slot (bytes layout); x = 0 absent symbol, x = 1 the EOF symbol
  __shared__ uint4 s_tab[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) {
    const uint64_t cf = ix.cf[c];
    const uint16_t s = ix.slot[c];
    uint64_t vb = 0;
    if (s < kSlotEof) vb = LAYOUT == kLayoutBytes ? (uint64_t)s + 2 : (uint64_t)(uintptr_t)ix.bv + (uint64_t)s * ix.nblocks * kBlockBytes;
    else if (s == kSlotEof) vb = 1;
    s_tab[c] = make_uint4((uint32_t)cf, (uint32_t)(cf >> 32), (uint32_t)vb, (uint32_t)(vb >> 32));
  }
  __syncthreads();
  const LaneConst lc = lane_const();
  const uint32_t t = lc.t;
  const uint32_t lane_off = t * 16;
  const uint32_t wave = (blockIdx.x * kSThreads + threadIdx.x) >> 6;
  const uint32_t nwaves = gridDim.x * (kSThreads / 64);
  const uint32_t oct = (threadIdx.x & 63) >> 3;
  const uint32_t nbatch = (k + 7) / 8;
  uint32_t steps = 0;
  PatDesc nd;
  nd.end = 0; nd.len = 0; nd.tail4 = 0;
  if (wave < nbatch && wave * 8 + oct < k) nd = desc[wave * 8 + oct];
  for (uint32_t batch = wave; batch < nbatch; batch += nwaves) {
    const uint32_t pid = batch * 8 + oct;
    const bool act = pid < k;
    uint64_t cur = nd.end, sp = 0, ep = ix.n;
    uint32_t left = act ? nd.len : 0u, ch = nd.tail4, nch = 4, nx = 0;
    if (left > 4) nx = fetch4(pat, cur - 4);
    {
      const uint64_t np = (uint64_t)(batch + nwaves) * 8 + oct;     // descriptor of the next batch
      if (np < k) nd = desc[np];
    }
    auto next_char = [&]() {
      ch >>= 8;
      nch -= 1;
      left -= 1;
      cur -= 1;
      if (nch == 0) {
        ch = nx;
        nch = 4;
        if (left > 4) nx = fetch4(pat, cur - 4);
      }
    };
    // symbols without a vector: absent (x = 0) or the EOF symbol (x = 1)
    auto step_special = [&](uint64_t cfc, uint64_t vb) {
      next_char();
      const uint64_t r1 = (vb == 1 && sp > ix.eof) ? 1 : 0;
      const uint64_t r2 = (vb == 1 && ep > ix.eof) ? 1 : 0;
      sp = cfc + r1;
      ep = cfc + r2;
    };
    for (;;) {
      const bool stepping = left > 0 && sp < ep;
      if (!__builtin_amdgcn_ballot_w64(stepping)) break;
      const bool wide_iv = stepping && (ep - sp) != 1;
      if (!__builtin_amdgcn_ballot_w64(wide_iv)) {
        // ---- every stepping octet holds one row: one rank query + one bit (byte) test
        if (stepping) {
          const uint32_t c = ch & 0xFFu;
          const uint4 e = s_tab[c];
          const uint64_t cfc = ((uint64_t)e.y << 32) | e.x;
          const uint64_t vb = ((uint64_t)e.w << 32) | e.z;
          if (vb > 1) {
            if (LAYOUT == kLayoutBytes) {
              const ByteRankReq q1 = byte_rank_issue(ix, (uint16_t)(vb - 2), sp, lc);
              next_char();
              const uint32_t bidx = q1.rem & 15u;                    // byte of this lane that holds row sp
              const uint32_t comp = bidx >> 2;
              const uint32_t word = comp < 2u ? (comp == 0u ? q1.w.x : q1.w.y) : (comp == 2u ? q1.w.z : q1.w.w);
              const uint32_t byte = __builtin_amdgcn_ubfe(word, 8u * (bidx & 3u), 8u);
              const uint32_t bit = ((q1.rem >> 4) == t && byte == c) ? 1u : 0u;
              sp = cfc + byte_rank_finish(q1, c, lc);
              ep = sp + octet_or(bit);
            } else {
              uint32_t b1, m1;
              split960(sp, b1, m1);
              const uint4 w1 = load_line16(vb + lane_off + (uint64_t)b1 * kBlockBytes);
              next_char();
              const uint32_t d = (m1 >> 5) + 2;                    // dword of the line that holds bit sp
              const uint32_t comp = d & 3u;
              const uint32_t word = comp < 2u ? (comp == 0u ? w1.x : w1.y) : (comp == 2u ? w1.z : w1.w);
              uint32_t bit = __builtin_amdgcn_ubfe(word, m1, 1u);   // offset taken mod 32
              bit = (d >> 2) == t ? bit : 0u;
              sp = cfc + rank_finish<WIDE>(w1, m1, lc);
              ep = sp + octet_or(bit);
            }
          } else {
            step_special(cfc, vb);
          }
          steps++;
        }
      } else if (stepping) {
        // ---- general step: two rank queries
        const uint32_t c = ch & 0xFFu;
        const uint4 e = s_tab[c];
        const uint64_t cfc = ((uint64_t)e.y << 32) | e.x;
        const uint64_t vb = ((uint64_t)e.w << 32) | e.z;
        if (sp == 0 && ep == ix.n && vb > 1) {
          // first step of every pattern: rank(c, 0) = 0 and rank(c, n) = the symbol's count, i.e. the
          // interval is the symbol's whole bucket [C[c], C[c+1]) -- no line needed
          const uint4 e2 = s_tab[(c + 1) & 0xFFu];
          next_char();
          sp = cfc;
          ep = c == 255u ? ix.n : (((uint64_t)e2.y << 32) | e2.x);
        } else if (vb > 1) {
          if (LAYOUT == kLayoutBytes) {
            const ByteRankReq q1 = byte_rank_issue(ix, (uint16_t)(vb - 2), sp, lc);
            const ByteRankReq q2 = byte_rank_issue(ix, (uint16_t)(vb - 2), ep, lc);
            next_char();
            sp = cfc + byte_rank_finish(q1, c, lc);
            ep = cfc + byte_rank_finish(q2, c, lc);
          } else {
            uint32_t b1, b2, m1, m2;
            split960(sp, b1, m1);
            split960(ep, b2, m2);
            const uint64_t base = vb + lane_off;
            const uint4 w1 = load_line16(base + (uint64_t)b1 * kBlockBytes);
            const uint4 w2 = load_line16(base + (uint64_t)b2 * kBlockBytes);
            next_char();
            sp = cfc + rank_finish<WIDE>(w1, m1, lc);
            ep = cfc + rank_finish<WIDE>(w2, m2, lc);
          }
        } else {
          step_special(cfc, vb);
        }
        steps++;
      }
    }
    if (act && t == 0) { sp_out[pid] = sp; ep_out[pid] = ep; }
  }
  if (t == 0 && steps) { atomicAdd(&counters[0], 2ull * steps); atomicAdd(&counters[1], (unsigned long long)steps); }
}

// v1 kernel (fmx_kernels.hip), kept for A/B runs
hipError_t launch_search_v1(const Index *h, const void *d_pat, const void *d_off, void *d_sp, void *d_ep, uint64_t k,
                            hipStream_t st);

static int search_variant() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("FMX_SEARCH_VARIANT");
    v = e ? atoi(e) : 3;      // measured (tools/ragged_bench.py): lockstep wins or ties on every mix tried
  }
  return v;
}

// Resident workgroups per CU for a kernel: the grid is sized to what is resident so that every
// octet starts at once and the static pattern striding stays balanced.
template <class K>
static int blocks_per_cu(K kernel) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kSThreads, 0) != hipSuccess || nb < 1) nb = 1;
  return nb > 8 ? 8 : nb;
}

template <bool WIDE>
static hipError_t launch_v3w(const Index *h, const uint8_t *pat, const PatDesc *desc, uint64_t *sp, uint64_t *ep,
                             uint32_t k, const uint32_t *ragged, hipStream_t st) {
  static const int per_cu = blocks_per_cu(k_search3<WIDE>);
  uint64_t want = ((uint64_t)k + kSOctets - 1) / kSOctets;
  uint64_t cap = (uint64_t)h->cu_count * per_cu;
  int grid = (int)(want < cap ? (want ? want : 1) : cap);
  k_search3<WIDE><<<grid, kSThreads, 0, st>>>(h->dev, pat, desc, sp, ep, k, h->d_counters, ragged);
  return hipGetLastError();
}

template <bool WIDE, uint32_t LAYOUT>
static hipError_t launch_v4w(const Index *h, const uint8_t *pat, const PatDesc *desc, uint64_t *sp, uint64_t *ep,
                             uint32_t k, const uint32_t *ragged, uint32_t serve_ragged, hipStream_t st) {
  static const int per_cu = blocks_per_cu(k_search4<WIDE, LAYOUT>);
  uint64_t want = ((uint64_t)k + kSOctets - 1) / kSOctets;
  uint64_t cap = (uint64_t)h->cu_count * per_cu;
  int grid = (int)(want < cap ? (want ? want : 1) : cap);
  k_search4<WIDE, LAYOUT><<<grid, kSThreads, 0, st>>>(h->dev, pat, desc, sp, ep, k, h->d_counters, ragged, serve_ragged);
  return hipGetLastError();
}

// FMX_SEARCH_VARIANT: 1 generic step kernel; 2 always the dynamic-refill kernel (one-hot layout);
// 3 (default) always the lockstep kernel; 0: k_prep decides per call -- the lockstep kernel for batches
// whose groups of 8 have similar lengths, the dynamic-refill kernel for very uneven ones (both are
// launched; the one that is not wanted returns at once).
hipError_t launch_search(const Index *h, const void *d_pat, const void *d_off, void *d_sp, void *d_ep, uint64_t k,
                         hipStream_t st) {
  if (!k) return hipSuccess;
  const int variant = search_variant();
  if (variant == 1 || k > 0xFFFFFFF0ull || (variant == 2 && h->layout != kLayoutOneHot))
    return launch_search_v1(h, d_pat, d_off, d_sp, d_ep, k, st);
  PatDesc *desc = nullptr;
  hipError_t e = hipMallocAsync((void **)&desc, (k + 1) * sizeof(PatDesc), st);
  if (e != hipSuccess) return e;
  uint32_t *ragged = reinterpret_cast<uint32_t *>(desc + k);        // the flag lives behind the descriptors
  int pg = (int)((k + 255) / 256);
  if (pg > h->cu_count * 8) pg = h->cu_count * 8;
  e = hipMemsetAsync(ragged, 0, sizeof(PatDesc), st);
  if (e == hipSuccess) {
    k_prep<<<pg, 256, 0, st>>>((const uint8_t *)d_pat, (const uint64_t *)d_off, desc, (uint32_t)k, ragged);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    const bool wide = h->n > (1ull << 32);
    const bool onehot = h->layout == kLayoutOneHot;
    const uint8_t *p = (const uint8_t *)d_pat;
    uint64_t *osp = (uint64_t *)d_sp, *oep = (uint64_t *)d_ep;
    const uint32_t kk = (uint32_t)k;
    if (variant == 2) {                       // force: make the refill kernel see "ragged"
      e = hipMemsetAsync(ragged, 1, 1, st);
      if (e == hipSuccess) e = wide ? launch_v3w<true>(h, p, desc, osp, oep, kk, ragged, st) : launch_v3w<false>(h, p, desc, osp, oep, kk, ragged, st);
    } else {
      // lockstep kernel: serves everything when forced (variant 3) or when no refill kernel exists
      const uint32_t serve_all = (variant == 3 || !onehot) ? 1u : 0u;
      if (!onehot) e = launch_v4w<true, kLayoutBytes>(h, p, desc, osp, oep, kk, ragged, serve_all, st);   // counts are 64-bit sums there
      else e = wide ? launch_v4w<true, kLayoutOneHot>(h, p, desc, osp, oep, kk, ragged, serve_all, st)
                    : launch_v4w<false, kLayoutOneHot>(h, p, desc, osp, oep, kk, ragged, serve_all, st);
      if (e == hipSuccess && !serve_all)
        e = wide ? launch_v3w<true>(h, p, desc, osp, oep, kk, ragged, st) : launch_v3w<false>(h, p, desc, osp, oep, kk, ragged, st);
    }
  }
  hipError_t e2 = hipFreeAsync(desc, st);
  return e != hipSuccess ? e : e2;
}

}  // namespace fmx

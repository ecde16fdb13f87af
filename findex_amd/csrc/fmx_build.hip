// fmx_build.hip -- K1: build the rank dictionary in HBM from the raw BWT bytes.
//
// This is the device analogue of the reference's rank-structure build, FMCreator.create
// (bwtmerger.scala:452-532: one pass over the BWT with the byte at stream index eof replaced by
// 0, positions bucketed by symbol).  Instead of inverted position lists it emits, per symbol,
// one-hot bit-vector blocks with a running count in each block header (layout: fmx_device.h).
//
// Three launches over "superchunks" of kSuper blocks (= kSuper*448 BWT positions):
//   k_hist : per-superchunk symbol histogram            (reads n bytes)
//   k_scan : exclusive scan of the histograms per symbol (tiny)
//   k_fill : per superchunk, block by block: bits via LDS atomicOr, headers from the running
//            counts, whole 64-byte blocks written with coalesced dword stores
//            (reads n bytes, writes nslots * nblocks * 64 bytes)
#include "fmx_device.h"
#include "fmx_host.h"

#include <algorithm>
#include <string>

namespace fmx {

constexpr int kBuildThreads = 256;
constexpr uint32_t kSuper = 256;                    // blocks per superchunk
constexpr uint32_t kPW = kBlockPayloadDwords;      // 14 payload dwords per one-hot block
constexpr uint32_t kBD = kBlockBytes / 4;          // 16 dwords per block: [hdr lo, hdr hi, payload]

__global__ __launch_bounds__(kBuildThreads) void k_hist(const uint8_t *__restrict__ bwt, uint64_t n, uint64_t eof,
                                                         uint64_t *__restrict__ hist /* [nsuper][256] */) {
  __shared__ uint32_t h[4][256];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) (&h[0][0])[i] = 0;
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * kSuper * kBlockBits;
  uint64_t hi = lo + (uint64_t)kSuper * kBlockBits;
  if (hi > n) hi = n;
  for (uint64_t p = lo + threadIdx.x; p < hi; p += blockDim.x) {
    if (p != eof) atomicAdd(&h[wave][bwt[p]], 1u);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 256; c += blockDim.x)
    hist[(uint64_t)blockIdx.x * 256 + c] = (uint64_t)h[0][c] + h[1][c] + h[2][c] + h[3][c];
}

// One workgroup per symbol: exclusive scan of that symbol's per-superchunk counts, 256 superchunks
// per trip with a carried total; hist becomes the exclusive prefix, totals[c] the sum.
__global__ __launch_bounds__(256) void k_scan(uint64_t *__restrict__ hist, uint64_t nsuper,
                                               uint64_t *__restrict__ totals) {
  __shared__ uint64_t buf[256];
  __shared__ uint64_t carry;
  const int c = blockIdx.x;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint64_t s0 = 0; s0 < nsuper; s0 += 256) {
    const uint64_t s = s0 + threadIdx.x;
    const uint64_t v = s < nsuper ? hist[s * 256 + c] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {              // Hillis-Steele inclusive scan in LDS
      const uint64_t add = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
      __syncthreads();
      buf[threadIdx.x] += add;
      __syncthreads();
    }
    const uint64_t base = carry;
    if (s < nsuper) hist[s * 256 + c] = base + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = base + buf[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[c] = carry;
}

__global__ __launch_bounds__(kBuildThreads) void k_fill(const uint8_t *__restrict__ bwt, uint64_t n, uint64_t eof,
                                                         uint64_t nblocks, uint32_t nslots,
                                                         const uint16_t *__restrict__ slot_of /* [256] */,
                                                         const uint16_t *__restrict__ sym_of /* [nslots] */,
                                                         const uint64_t *__restrict__ base /* [nsuper][256] */,
                                                         uint32_t *__restrict__ bv) {
  extern __shared__ uint32_t lds[];
  constexpr uint32_t pw = kPW;
  uint32_t *bits = lds;                                      // [nslots][14]
  uint64_t *run = reinterpret_cast<uint64_t *>(lds + ((nslots * pw + 1) & ~1u));               // [nslots]
  uint16_t *s_slot = reinterpret_cast<uint16_t *>(run + nslots);                               // [256]
  for (int c = threadIdx.x; c < 256; c += blockDim.x) s_slot[c] = slot_of[c];
  for (uint32_t s = threadIdx.x; s < nslots; s += blockDim.x) run[s] = base[(uint64_t)blockIdx.x * 256 + sym_of[s]];
  const uint64_t b0 = (uint64_t)blockIdx.x * kSuper;
  uint64_t b1 = b0 + kSuper;
  if (b1 > nblocks) b1 = nblocks;
  const uint32_t nwords = nslots * pw;
  for (uint64_t b = b0; b < b1; b++) {
    for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) bits[i] = 0;
    __syncthreads();
    const uint64_t p0 = b * kBlockBits;
    for (uint32_t r = threadIdx.x; r < kBlockBits; r += blockDim.x) {
      const uint64_t p = p0 + r;
      if (p < n && p != eof) {
        const uint16_t s = s_slot[bwt[p]];
        if (s < kSlotEof) atomicOr(&bits[s * pw + (r >> 5)], 1u << (r & 31));
      }
    }
    __syncthreads();
    // write nslots blocks of kBD dwords: [hdr lo, hdr hi, payload dwords]
    for (uint32_t i = threadIdx.x; i < nslots * kBD; i += blockDim.x) {
      const uint32_t s = i / kBD, w = i % kBD;
      uint32_t v;
      if (w == 0) v = (uint32_t)run[s];
      else if (w == 1) v = (uint32_t)(run[s] >> 32);
      else v = bits[s * pw + w - 2];
      bv[((uint64_t)s * nblocks + b) * kBD + w] = v;
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < nslots; s += blockDim.x) {
      uint32_t pc = 0;
      for (uint32_t w = 0; w < pw; w++) pc += __builtin_popcount(bits[s * pw + w]);
      run[s] += pc;
    }
    __syncthreads();
  }
}

// ---- bytes layout (fmx_device.h): per-superblock histogram = k_hist with the superblock as the
// chunk; then one workgroup per superblock walks its 2^15 blocks, writing each block's checkpoint row
// (counts since the superblock start) before adding the block's own bytes.
__global__ __launch_bounds__(kBuildThreads) void k_hist_bytes(const uint8_t *__restrict__ bwt, uint64_t n,
                                                               uint64_t *__restrict__ hist /* [nsb][256] */) {
  __shared__ uint32_t h[4][256];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) (&h[0][0])[i] = 0;
  __syncthreads();
  const uint64_t lo = ((uint64_t)blockIdx.x << kSuperShift) * kByteBlock;
  uint64_t hi = lo + ((uint64_t)kByteBlock << kSuperShift);
  if (hi > n) hi = n;
  for (uint64_t p = lo + threadIdx.x; p < hi; p += blockDim.x) atomicAdd(&h[wave][bwt[p]], 1u);   // slot eof holds 0
  __syncthreads();
  for (int c = threadIdx.x; c < 256; c += blockDim.x)
    hist[(uint64_t)blockIdx.x * 256 + c] = (uint64_t)h[0][c] + h[1][c] + h[2][c] + h[3][c];
}

__global__ __launch_bounds__(kBuildThreads) void k_fill_bytes(const uint8_t *__restrict__ bwt, uint64_t nbb,
                                                               uint32_t nslots, const uint16_t *__restrict__ slot_of,
                                                               const uint16_t *__restrict__ sym_of,
                                                               const uint64_t *__restrict__ base /* [nsb][256] */,
                                                               uint32_t *__restrict__ chk, uint64_t *__restrict__ sup) {
  __shared__ uint32_t run[256];       // per slot: count since the superblock start (sup != nullptr) or since row 0
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_slot[c] = slot_of[c]; run[c] = 0; }
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < nslots; s += blockDim.x) {
    const uint64_t before = base[(uint64_t)blockIdx.x * 256 + sym_of[s]];
    if (sup) sup[(uint64_t)blockIdx.x * nslots + s] = before;
    else run[s] = (uint32_t)before;   // absolute checkpoints: every count fits 32 bits
  }
  __syncthreads();
  const uint64_t b0 = (uint64_t)blockIdx.x << kSuperShift;
  uint64_t b1 = b0 + (1ull << kSuperShift);
  if (b1 > nbb) b1 = nbb;
  // two blocks per trip: threads 0..127 take block b, 128..255 block b+1 (bwt is zero-padded)
  for (uint64_t b = b0; b < b1; b += 2) {
    for (uint32_t s = threadIdx.x; s < nslots; s += blockDim.x) chk[b * nslots + s] = run[s];
    __syncthreads();
    const uint32_t half = threadIdx.x >> 7;
    uint16_t sl = kSlotNone;
    if (b + half < b1) sl = s_slot[bwt[(b + half) * kByteBlock + (threadIdx.x & 127)]];
    if (half == 0 && sl < kSlotEof) atomicAdd(&run[sl], 1u);
    __syncthreads();
    if (b + 1 < b1)
      for (uint32_t s = threadIdx.x; s < nslots; s += blockDim.x) chk[(b + 1) * nslots + s] = run[s];
    __syncthreads();
    if (half == 1 && sl < kSlotEof) atomicAdd(&run[sl], 1u);
    __syncthreads();
  }
}

// Builds the whole device side of an index from h->d_bwt / h->n / h->eof:
//   1. symbol histogram on the device (authoritative; EOF slot excluded),
//   2. validation against the caller's .aux counts when given,
//   3. symbol->slot map, C[] (NaiveFMSearcher.cf, bwtmerger.scala:346-352: counts(0):=1 then
//      exclusive prefix sums), table upload, rank dictionary allocation,
//   4. the fill pass.
// Returns an FMX_* status.
int build_index(Index *h, hipStream_t st, const int64_t *given_counts) {
  // Layout: one-hot vectors when they fit comfortably (decided from n and a sigma upper bound before
  // the histogram, re-checked after), the bytes+checkpoints layout otherwise or when forced.
  bool bytes_layout = h->layout == kLayoutBytes;
  h->nblocks = bytes_layout ? h->n / kByteBlock + 1 : h->n / kBlockBits + 1;
  const uint64_t nsuper = bytes_layout ? ((h->nblocks >> kSuperShift) + 1) : (h->nblocks + kSuper - 1) / kSuper;
  uint64_t *d_hist = nullptr, *d_tot = nullptr;
  uint16_t *d_sym = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;   // (hist + scan), (fill): kernels only
  int rc = 0;
  hipError_t e = hipSuccess;
  uint64_t tot[256];
  uint16_t sym_of[256] = {0};
#define FMX_TRY(call, what)                       \
  if ((e = (call)) != hipSuccess) {               \
    rc = hip_fail(e, what);                       \
    break;                                        \
  }
  do {
    FMX_TRY(hipEventCreate(&ev0), "hipEventCreate");
    FMX_TRY(hipEventCreate(&ev1), "hipEventCreate");
    FMX_TRY(hipEventCreate(&ev2), "hipEventCreate");
    FMX_TRY(hipEventCreate(&ev3), "hipEventCreate");
    FMX_TRY(hipMalloc(&d_hist, nsuper * 256 * sizeof(uint64_t)), "hipMalloc(hist)");
    FMX_TRY(hipMalloc(&d_tot, 256 * sizeof(uint64_t)), "hipMalloc(totals)");
    FMX_TRY(hipMalloc(&d_sym, 256 * sizeof(uint16_t)), "hipMalloc(sym)");
    FMX_TRY(hipEventRecord(ev0, st), "hipEventRecord");
    if (bytes_layout) k_hist_bytes<<<(int)nsuper, kBuildThreads, 0, st>>>((const uint8_t *)h->d_bwt, h->n, d_hist);
    else k_hist<<<(int)nsuper, kBuildThreads, 0, st>>>((const uint8_t *)h->d_bwt, h->n, h->eof, d_hist);
    FMX_TRY(hipGetLastError(), "k_hist");
    k_scan<<<256, 256, 0, st>>>(d_hist, nsuper, d_tot);
    FMX_TRY(hipGetLastError(), "k_scan");
    FMX_TRY(hipEventRecord(ev1, st), "hipEventRecord");
    FMX_TRY(hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, st), "copy totals");
    FMX_TRY(hipStreamSynchronize(st), "sync(hist)");

    if (bytes_layout) tot[0] -= 1;          // the EOF slot itself holds the one 0 byte of BWT'
    if (tot[0] != 0) {
      set_error("BWT holds byte 0 outside the EOF slot: findex escapes 0 on input (bwtreader.scala:136-155) and its "
                "FMCreator gives symbol 0 exactly one slot (bwtmerger.scala:440-450)");
      rc = 6 /* FMX_ERR_UNSUPPORTED */;
      break;
    }
    if (given_counts && !h->block_mode) {
      bool ok = given_counts[0] == 0;
      for (int c = 1; c < 256 && ok; c++) ok = (uint64_t)given_counts[c] == tot[c];
      if (!ok) {
        set_error("symbol counts (.aux) do not describe this BWT");
        rc = 2 /* FMX_ERR_FORMAT */;
        break;
      }
    }
    h->nslots = 0;
    uint64_t run = 1;                       // counts(0) := 1, bwtmerger.scala:348
    h->cf[0] = 0;
    h->counts[0] = 0;
    h->slot[0] = kSlotEof;
    for (int c = 1; c < 256; c++) {
      h->counts[c] = (int64_t)tot[c];
      h->cf[c] = run;
      run += tot[c];
      if (tot[c]) { sym_of[h->nslots] = (uint16_t)c; h->slot[c] = (uint16_t)h->nslots++; }
      else h->slot[c] = kSlotNone;
    }
    if (h->block_mode) {
      // NaiveBWTSearcher (findex.scala:459-506): cf = the caller's bucket starts (:478); occ counts the symbol in
      // bwt[0..key] without row rk0 -- the rank dictionary with that row as its "EOF" slot -- for every symbol,
      // 0 included (no byte 0 here, so 0 everywhere).  Its binary search returns (iend - istart) when it ends on the
      // bucket's last slot and reads 0 there (:500-502, meant for the hole the skipped row leaves): a bucket
      // without a hole whose ONLY entry is position 0 reads the same, so such a symbol -- the block's first byte,
      // if it occurs nowhere else and is not the skipped row's byte -- answers 0 for every key.
      for (int c = 0; c < 256; c++) h->cf[c] = (uint64_t)h->block_bs[c];
      h->slot[0] = kSlotNone;
      const int c0 = h->block_first;
      if (c0 > 0 && tot[c0] == 1 && h->block_skipped != c0 && h->slot[c0] < kSlotEof) {
        // drop its vector: compact the slot numbering
        const uint16_t dead = h->slot[c0];
        for (int c = 1; c < 256; c++)
          if (h->slot[c] < kSlotEof && h->slot[c] > dead) h->slot[c]--;
        for (uint32_t sl = dead; sl + 1 < h->nslots; sl++) sym_of[sl] = sym_of[sl + 1];
        h->nslots--;
        h->slot[c0] = kSlotNone;
      }
    }
    uint64_t max_count = 0;
    for (int c = 1; c < 256; c++) max_count = std::max<uint64_t>(max_count, tot[c]);
    const bool abs_chk = bytes_layout && max_count < (1ull << 32) && !force_superblocks();      // absolute 32-bit checkpoints, no superblocks
    const uint64_t bv_bytes = bytes_layout ? (uint64_t)h->nslots * (h->nblocks * 4 + (abs_chk ? 0 : nsuper * 8))
                                           : (uint64_t)h->nslots * h->nblocks * kBD * 4;
    size_t free_b = 0, total_b = 0;
    FMX_TRY(hipMemGetInfo(&free_b, &total_b), "hipMemGetInfo");
    if (bv_bytes + (64ull << 20) > free_b) {
      set_error("rank dictionary needs " + std::to_string(bv_bytes >> 20) + " MiB of device memory, " +
                std::to_string(free_b >> 20) + " MiB free");
      rc = bytes_layout ? 4 /* FMX_ERR_NOMEM */ : -1 /* retry with the bytes layout */;
      break;
    }
    if (bytes_layout) {
      FMX_TRY(hipMalloc(&h->d_chk, (uint64_t)h->nslots * h->nblocks * 4 + 16), "hipMalloc(checkpoints)");
      if (!abs_chk) FMX_TRY(hipMalloc(&h->d_sup, (uint64_t)h->nslots * nsuper * 8 + 16), "hipMalloc(superblocks)");
    } else {
      FMX_TRY(hipMalloc(&h->d_bv, bv_bytes ? bv_bytes : 16), "hipMalloc(rank dictionary)");
    }
    FMX_TRY(hipMalloc(&h->d_cf, sizeof h->cf), "hipMalloc(cf)");
    FMX_TRY(hipMalloc(&h->d_slot, sizeof h->slot), "hipMalloc(slot)");
    FMX_TRY(hipMalloc((void **)&h->d_counters, kCounterBytes + kCensusBytes + kCalibScratchBytes + kTixBytes), "hipMalloc(counters)");
    FMX_TRY(hipMemsetAsync(h->d_counters, 0, kCounterBytes + kCensusBytes + kCalibScratchBytes + kTixBytes, st), "memset(counters)");
    FMX_TRY(hipMemcpyAsync(h->d_cf, h->cf, sizeof h->cf, hipMemcpyHostToDevice, st), "copy cf");
    FMX_TRY(hipMemcpyAsync(h->d_slot, h->slot, sizeof h->slot, hipMemcpyHostToDevice, st), "copy slot");
    FMX_TRY(hipMemcpyAsync(d_sym, sym_of, sizeof sym_of, hipMemcpyHostToDevice, st), "copy sym");
    FMX_TRY(hipEventRecord(ev2, st), "hipEventRecord");
    if (bytes_layout) {
      if (h->nslots) {
        k_fill_bytes<<<(int)nsuper, kBuildThreads, 0, st>>>((const uint8_t *)h->d_bwt, h->nblocks, h->nslots,
                                                           (const uint16_t *)h->d_slot, d_sym, d_hist,
                                                           (uint32_t *)h->d_chk, (uint64_t *)h->d_sup);
        FMX_TRY(hipGetLastError(), "k_fill_bytes");
      }
    } else if (h->nslots) {
      size_t lds = (size_t)((h->nslots * (kBlockBits / 32) + 1) & ~1u) * 4 + (size_t)h->nslots * 8 + 256 * 2;
      k_fill<<<(int)nsuper, kBuildThreads, lds, st>>>((const uint8_t *)h->d_bwt, h->n, h->eof, h->nblocks, h->nslots,
                                                       (const uint16_t *)h->d_slot, d_sym, d_hist,
                                                       (uint32_t *)h->d_bv);
      FMX_TRY(hipGetLastError(), "k_fill");
    }
    FMX_TRY(hipEventRecord(ev3, st), "hipEventRecord");
    FMX_TRY(hipStreamSynchronize(st), "sync(fill)");
    float ms_a = 0, ms_b = 0;                 // the allocations between the two phases are the driver's time, not the build's
    FMX_TRY(hipEventElapsedTime(&ms_a, ev0, ev1), "hipEventElapsedTime");
    FMX_TRY(hipEventElapsedTime(&ms_b, ev2, ev3), "hipEventElapsedTime");
    h->build_ms = ms_a + ms_b;
    h->index_bytes = bv_bytes + h->n + sizeof h->cf + sizeof h->slot;
    h->dev.bv = (const uint4 *)h->d_bv;
    h->dev.chk = (const uint32_t *)h->d_chk;
    h->dev.sup = (const uint64_t *)h->d_sup;
    h->dev.layout = h->layout;
    h->dev.nslots = h->nslots;
    h->dev.bwt = (const uint8_t *)h->d_bwt;
    h->dev.cf = (const uint64_t *)h->d_cf;
    h->dev.slot = (const uint16_t *)h->d_slot;
    h->dev.n = h->n;
    h->dev.eof = h->eof;
    h->dev.nblocks = h->nblocks;
  } while (0);
#undef FMX_TRY
  if (d_hist) (void)hipFree(d_hist);
  if (d_tot) (void)hipFree(d_tot);
  if (d_sym) (void)hipFree(d_sym);
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  if (ev2) (void)hipEventDestroy(ev2);
  if (ev3) (void)hipEventDestroy(ev3);
  return rc;
}

}  // namespace fmx

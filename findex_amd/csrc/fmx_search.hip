// fmx_search.hip -- K3: batched literal backward search (SuffixAlgo.search, findex.scala:15-31).
//
// The dominant kernel of the headline benchmark.  Per backward step a group of lanes (a quad in the
// one-hot layout, an octet in the bytes layout) fetches the rank-dictionary block(s) of (c, sp) and
// (c, ep) and turns them into the next interval; the only dependent chain is block -> popcount ->
// next block address.  What the kernel does so that nothing else sits on that chain:
//   * pattern bytes are read 4 at a time, two dwords ahead, with branch-free address arithmetic; the
//     pattern a group takes next is prepared while the current batch is searched: its offsets are
//     requested two batches ahead and its last 4 bytes (which depend on the offsets) one batch ahead
//     (round 1 did this with a pre-pass kernel that wrote 16-byte descriptors: one more launch, 32 bytes
//     of traffic per pattern and a scratch buffer whose ownership had to be tracked per stream);
//   * C[] and each symbol's bit-vector base address sit in LDS as one 16-byte entry per symbol;
//   * the rank primitive of fmx_device.h (5 vector instructions per payload dword).
// History (profiles/): the first version (FMX_SEARCH_VARIANT=1, fmx_kernels.hip) ran about
// 165 vector instructions per step and wave and was bound by instruction issue, not HBM.  Tried and
// dropped on the way here:
//   * 2 or 3 patterns per lane group in one trip (with quads: 116 registers, half the waves,
//     0.69 -> 0.72 ms -- one request per quad already saturates the memory system);
//   * a dynamic-refill variant that hands a group its next pattern as soon as one ends (at most 4 %
//     faster than lockstep batches on lengths uniform in 1..64, slower otherwise);
//   * forming the batches from patterns sorted by length (idle lanes issue no requests, and requests
//     are the limit: ragged batches already run at 91 % of the uniform rate; the sort added its 90 us);
//   * with the row jump table (round 3): every lane group on its own step number, so that a group whose pattern
//     differs from its row's text inside a jump does not make the fifteen that jumped wait for its eight steps
//     (66 registers, per-group pattern cursors: 0.278 ms against 0.259 ms in lockstep on C3 -- with a jump table the
//     kernel is bound by instruction issue, 88 % of the SIMDs' issue slots, not by waiting).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "fmx_device.h"
#include "fmx_host.h"

namespace fmx {

constexpr int kSThreads = 256;

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

// ---------------------------------------------------------------- lockstep batches + single-row step
// A wave takes 64/G consecutive patterns (one per lane group), steps them together and retires them
// together.  What that buys: everything that steers the loop is wave-uniform (scalar branches).
//   * step number `it` is the same for every group of the wave, so the pattern cursor (which dword
//     of the pattern is current, when to fetch the next) is scalar bookkeeping;
//   * first step (peeled): (0, n) maps to the symbol's whole bucket [C[c], C[c+1]) -- no block needed;
//   * general step: two rank queries (sp and ep blocks);
//   * single-row step, taken when every stepping group holds an interval of exactly one row
//     (sigma = 128, n = 2^32: from the 6th of 32 steps on): one rank query plus one bit test --
//     [sp, sp+1) maps to [C[c] + rank(c, sp), + BWT'[sp] == c), and BWT'[sp] == c is bit sp of c's
//     own vector, i.e. a bit of the block already fetched.  Same result as getPrevRange
//     (findex.scala:32-36), half the popcount work and one request instead of two.
// Groups whose pattern ends early idle until the batch ends.

// Chunk j of a pattern = its bytes pat[end-1-4j-i], i = 0..3, in byte lanes 0..3: the four bytes the
// search consumes at steps 4j..4j+3.  Branch-free; lanes whose pattern has no such chunk read their
// own offset entry instead (any valid address: the value is never used).
__device__ __forceinline__ uint32_t pat_chunk(const uint8_t *__restrict__ pat, const uint64_t *__restrict__ own,
                                              uint64_t end, uint32_t len, uint32_t j) {
  const uint32_t have = len > 4u * j ? len - 4u * j : 0u;     // pattern bytes left at chunk j
  const uint64_t pos = end - 4ull * j;                        // valid when have > 0 (then pos >= have >= 1)
  // have >= 4: the dword below pos; have in 1..3 (the pattern starts inside that dword): the dword at
  // the pattern's start, shifted -- it lies in [pos - have, pos - have + 4), inside the buffer only
  // if len >= 4, which holds for every chunk but chunk 0 (j >= 1 and have >= 1 give len >= 5)
  const uint32_t back = have >= 4u ? 4u : have;
  const uint8_t *src = have ? pat + (pos - back) : reinterpret_cast<const uint8_t *>(own);
  uint32_t d;
  __builtin_memcpy(&d, src, 4);                               // unaligned dword load
  // bytes wanted: src[back-1] .. src[0] -> lanes 0 .. back-1
  return __builtin_bswap32(d) >> (8u * (4u - (back ? back : 4u)));
}

// KT = characters the k-mer jump table (fmx_ktab.hip) answers with one lookup at the start of a search: 0 (no table),
// 4, 8 or 12 -- a compile-time constant, so that the step loop below starts at a constant step number and is
// compiled exactly as without the table, and the pattern pipeline prefetches exactly the KT tail bytes the table
// is indexed with (4 at sigma = 128: nothing more than before).
// JT: the handle has a row jump table (fmx_jump.hip): once every stepping group of the wave holds one row, eight steps
// at a time are ONE 16-byte lookup for every group whose next eight pattern characters are the ones its row's entry
// names; the others walk those eight steps as before while the ones that jumped wait.
// RW > 0: the handle has a row table and no row jump table (fmx_jump.hip, row1_get / row3_get): a group whose interval has
// become ONE ROW hands its pattern to the wave's rows list, and in the wave's next rows phase a LANE finishes it -- from
// there on a search needs no rank query (rows_phase below; until round 4 a second and a third launch).  RW = 1: handed
// over at once; RW = 3 (the table takes three steps per word): when the steps left are a multiple of three, after up to
// two more one-row steps by the group.
// R3T (with JT): the handle also has the three-step row table (fmx_jump.hip): a one-row group that is not at a chunk
// boundary, or has fewer than eight characters left, takes three steps with one 8-byte lookup instead of three rank
// queries -- at C3 the three steps between the wide part of a search and its first aligned jump.
#ifndef FMX_SEARCH_WAVES
#define FMX_SEARCH_WAVES 6      // waves per SIMD the search kernel is compiled for (register budget 512 / waves, in eights)
#endif
#ifdef FMX_SEARCHLOG
// Diagnostic build only (tools/search_wave_timeline.py): begin and end of every wave of the last k_search4 launch on
// the constant 100 MHz clock, and the batches it searched.
__device__ unsigned long long g_searchlog[1u << 15][4];
#endif
// G2 (round 5, one-hot layout only): a pattern is served by a PAIR of lanes instead of a quad -- 32 patterns per wave, the
// dictionary's 64-byte block fetched as two 32-byte halves (fmx_device.h, Blk2).  tools/c3_halfbatch.py: with the same 1M
// pattern slots holding 16 / 8 / 4 / 2 real patterns per batch a C3 launch takes 0.132 / 0.102 / 0.093 / 0.088 ms -- its
// time is the round trips of its waves' lockstep batches (a wave works through ~10 of them, ~8.5 us each whatever they
// hold), not its requests.  Twice the patterns per batch is half the batches per wave.
#ifndef FMX_SEARCH_WAVES_G2
#define FMX_SEARCH_WAVES_G2 5
#endif
template <bool WIDE, uint32_t LAYOUT, uint32_t KT, uint32_t JT, uint32_t RW, bool R3T, bool G2 = false>
__global__ __launch_bounds__(kSThreads) __attribute__((amdgpu_waves_per_eu(G2 ? FMX_SEARCH_WAVES_G2 : FMX_SEARCH_WAVES, 8))) void k_search4(DevIndex ix, const uint4 *__restrict__ ktab, const uint8_t *__restrict__ kdense,
                                                        uint32_t ksigma, const uint4 *__restrict__ jtab, const uint32_t jc,
                                                        const unsigned long long *__restrict__ r3tab, const uint8_t *__restrict__ pat,
                                                        const PatOff po,
                                                        uint64_t *__restrict__ sp_out, uint64_t *__restrict__ ep_out,
                                                        uint32_t k, unsigned long long *__restrict__ counters, const uint64_t pk_cap,
                                                        const uint32_t spin) {
  static_assert(!G2 || LAYOUT == kLayoutOneHot, "pairs of lanes serve the one-hot layout only");
  constexpr int G = G2 ? 2 : Lay<LAYOUT>::G;     // lanes per pattern
  constexpr uint32_t P = 64 / G;                 // patterns per wave
  constexpr uint32_t RG = G2 ? 4u : (uint32_t)G; // rows of an interval a group can look up in the row tables at once (a pair's lanes take two rows each)
  constexpr uint32_t R = LAYOUT == kLayoutBytes ? 2u : 1u;    // memory requests per rank query
#ifdef FMX_SEARCHLOG
  const unsigned long long sl_t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long sl_t1 = 0;
  uint32_t sl_batches = 0;
#endif
  // the residency census (fmx_device.h): when this workgroup began
  if (threadIdx.x == 0 && blockIdx.x < kCensusBlocks) {
    counters[(size_t)kCounterSlots * kCounterStride + 2u * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0) counters[(size_t)kCounterSlots * kCounterStride + 2u * kCensusBlocks] = gridDim.x;      // whose entries these are
  }
  if (spin & 0xFFFFu) {
    // a CALIBRATION launch (search_calibrate, from fmx_prepare): every workgroup stays resident for `spin` ticks of the
    // 100 MHz clock whatever its batch holds, so that "began before the first one ended" means "was resident beside it"
    const unsigned long long c0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - c0 < (spin & 0xFFFFu)) __builtin_amdgcn_s_sleep(32);
  }
  // per symbol: {C[c], x} with x = byte address of the symbol's bit-vector (one-hot layout) or its
  // slot + 2 (bytes layout); x = 0 absent symbol, x = 1 the EOF symbol
  __shared__ uint4 s_tab[256];
  __shared__ uint8_t s_dense[KT ? 256 : 4];      // byte -> dense symbol id of the k-mer table (0xFF: not in it)
  __shared__ uint16_t s_slot[KT ? 256 : 4];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) {
    if (KT) { s_dense[c] = kdense[c]; s_slot[c] = ix.slot[c]; }
    const uint64_t cf = ix.cf[c];
    const uint16_t s = ix.slot[c];
    uint64_t vb = 0;
    if (s < kSlotEof) vb = LAYOUT == kLayoutBytes ? (uint64_t)s + 2 : (uint64_t)(uintptr_t)ix.bv + (uint64_t)s * ix.nblocks * kBlockBytes;
    else if (s == kSlotEof) vb = 1;
    s_tab[c] = make_uint4((uint32_t)cf, (uint32_t)(cf >> 32), (uint32_t)vb, (uint32_t)(vb >> 32));
  }
  __syncthreads();
  const LaneConst lc = lane_const<G2 ? 4 : G>();      // (a pair's lane takes its positions from t itself: rank_finish_g2)
  const uint32_t t = G2 ? (threadIdx.x & 1u) : lc.t;
  const uint32_t lane_off = t * 16;
  uint32_t steps = 0, reqs = 0;     // reqs: memory requests for rank-dictionary lines (counters[2])
  // A rank query of the one-hot layout for this lane group, whatever its width: rank_excl(x) of the symbol whose vector begins at
  // vb, and (bit) BWT'[x] == that symbol.  Issue and finish are separate so that a step can have two blocks in flight.
  struct Blk { uint4 a, b; };
  // (a pair fetches the block's upper 32 bytes only when the query looks there: positions 192 and up -- payload dwords 6 .. 13;
  // `upto` = the largest in-block position asked of this block.  43 % of the queries are answered by the lower half alone.)
  auto blk_load = [&](uint64_t vb, uint32_t b1, uint32_t upto) -> Blk {
    const uint64_t at = vb + lane_off + (uint64_t)b1 * kBlockBytes;
    Blk w;
    w.a = load_line16(at);
    w.b = make_uint4(0, 0, 0, 0);
    if constexpr (G2) {
      if (upto >= 192u) w.b = load_line16(at + 32);      // (the same 64-byte line: served by the L2 miss the lower half started -- not counted as a request)
    }
    return w;
  };
  auto blk_rank = [&](const Blk &w, uint32_t m1) -> uint64_t {
    if constexpr (G2) return rank_finish_g2<WIDE>(Blk2{w.a, w.b}, m1, t);
    else return rank_finish<WIDE>(w.a, m1, lc);
  };
  auto blk_bit = [&](const Blk &w, uint32_t m1) -> uint32_t {
    if constexpr (G2) return payload_bit_g2(Blk2{w.a, w.b}, m1, t);
    else return payload_bit(w.a, m1, lc);
  };
  constexpr uint32_t RB = 1u;                     // memory requests (distinct lines) per one-hot block
  // lane 0 / lane 1 of the group to all its lanes
  auto gbc0 = [&](uint32_t v) -> uint32_t { if constexpr (G2) return pair_bcast<0>(v); else return group_bcast<G, 0>(v); };
  auto gbc1 = [&](uint32_t v) -> uint32_t { if constexpr (G2) return pair_bcast<1>(v); else return group_bcast<G, 1>(v); };
  const uint32_t wave = (blockIdx.x * kSThreads + threadIdx.x) >> 6;
  const uint32_t nwaves = gridDim.x * (kSThreads / 64);
  const uint32_t grp = (threadIdx.x & 63) / G;
  const uint32_t nbatch = (k + P - 1) / P;
  // The last rounds are DRAWN (round 5).  The batches are strided statically over the waves, every wave's pipeline knowing
  // two batches ahead which ones are its own -- and the launch ends with its slowest wave: the timelines
  // (profiles/r05_c3_wave_timeline.txt, r05_c5_c2_wave_timeline.txt) have the waves leave their batch loops 20-50 us apart
  // (C2 83-110 us, C5 244-294 us p10-p99).  So when the host hands the launch a ticket area (bits 16-23 of `spin`; fmx_device.h,
  // kTixAreas) and there are four rounds or more, all rounds but the last two full ones are strided as before and the rest --
  // two to three rounds' worth -- is a pool: a wave that has done its share draws batch after batch from it (one returning
  // atomic each, issued a batch ahead) until it is empty.  A drawn batch starts cold (its offsets, then its bytes: two
  // round trips the strided ones have behind them), which is why only the end is drawn; the strided part keeps its code
  // and its registers (tools/r05_tickets.patch drew every batch: +2.5-4 % for the drawing, -6-7 % for its state in the loop).
  // (not in the quads' kernels with a row jump table: the second copy of the batch body costs them 4 vector and 9 scalar
  // registers spilled at six waves per SIMD -- C3 by quads 0.137-0.140 -> 0.143 ms; their large batches go to the pairs)
  constexpr bool kPool = G2 || RW != 0u;
  const uint32_t tix_area = kPool ? (spin >> 16) & 0xFFu : 0u;
  // fmx.h, FMX_SEARCH_MISS_NONE (bit 31 of `spin`): a pattern that a row-table lookup finds to MISS -- its one row's text differs
  // from it -- is reported as (0, 0), None in the reference (findex.scala:30), instead of being parked and walked to the reference
  // loop's values at its failing step; the steps the reference's loop made on it are known from where the texts differ.
  // (Kernels with a row jump table only: in the rows kernels -- C5, C2 -- the walks are 2 % of a launch, and one more scalar
  // that lives through their loops cost C2 3 %.)
  const bool miss_none = JT != 0u && (spin >> 31) != 0u;
  uint32_t nstatic = nbatch;
  if (tix_area) {
    const uint32_t rounds = nbatch / nwaves, held = (spin >> 24) & 0xFu;      // held: full rounds that go to the pool with the partial one
    if (rounds >= held + 2u) nstatic = (rounds - held) * nwaves;
  }
  // Pattern pipeline.  A wave's 16 (8) patterns lie one behind the other in the pattern buffer, so their bytes are ONE
  // contiguous span: it is fetched with one coalesced wave-level load (16 bytes per lane, up to 1 KiB) while the batch
  // before it is searched, parked in the wave's own LDS area, and every chunk of pattern bytes the search consumes is an
  // LDS read.  Round 4 (profiles/r04_c3_bound.md): the memory system answers ~50 G cache-line requests per second that
  // miss the CUs' own L1 / address-translation caches, whatever they ask for -- and the chunk loads of the former
  // pipeline (a dword per group and four steps, two per row-jump lookup: ~35 line requests per batch beside its ~100
  // table lookups) were a quarter of the kernel's requests.  A span of 1 KiB covers batches whose patterns average 63
  // bytes; a longer one is read chunk by chunk from global memory as before (`staged` is wave-uniform).
  // The offsets of a batch are requested two batches ahead and only LOOKED AT one batch later (raw values are carried
  // over: any arithmetic on them here would put the wait for the load right behind it).
  // The bytes layout keeps the former pipeline (chunks from global memory, the next batch's tail requested a batch
  // ahead): its waves hold 8 patterns, not 16, so the staging costs the same instructions and registers for half the
  // lines saved -- C5's share of this kernel went from 0.181 to 0.217 ms with it (62 -> 70 registers, 8 -> 7 waves).
  constexpr bool kStage = LAYOUT != kLayoutBytes;
  constexpr uint32_t kStageBytes = G2 ? 2048 : 1024, kStagePad = 16;      // (a pair-of-lanes wave's 32 patterns: two 16-byte loads per lane)
  // two areas per wave: the batch being searched reads one while the next batch's span is parked in the other as soon as
  // it has arrived (it arrives with the batch's first table lookup: no registers hold it across the search)
  __shared__ __attribute__((aligned(16))) uint32_t s_pat[kSThreads / 64][2][kStage ? (kStagePad + kStageBytes + 16) / 4 : 4];
  uint32_t par = 0;                 // which area holds the current batch
  // Patterns that are found to MISS by a table lookup (their one row's text differs from the pattern within the
  // lookup's characters) still owe the reference loop's values at the failing step: a few one-row rank steps.  Taking
  // them where they arise would make the whole wave execute them while the groups that jumped wait; a kernel with a
  // row jump table (and no hand-over to k_search_rows) therefore parks them in a list in LDS and walks them densely, a
  // lane group each, when 48 have come together and before the wave ends (walk_parked below) -- round 3 parked them in
  // the output arrays for a second launch, k_search_defer: 26-30 us behind the 150 of this one at C3.
  // Round 5: a kernel WITHOUT a row jump table but with a row table (RW: the three-step table where J does not fit -- C5,
  // n = 2^34 -- or the frontier's one-step table) finishes every pattern itself too.  Until round 4 it parked a pattern whose
  // interval had become one row in the OUTPUT arrays and two more launches picked them up (k_search_rows: one lane per
  // pattern, 64 chains per wave; k_search_defer: the failing step).  Now the wave keeps those patterns in a second list in
  // LDS and, whenever 64 have come together (eight batches of the bytes layout), turns into what k_search_rows was for
  // one phase: every LANE walks one pattern through the row table, three (one) steps per 8-byte word; the patterns whose
  // row disagrees go on to the walk list above and are finished by lane groups.  One launch instead of three (C5: two
  // launch ramps and tails, and a round trip of the parked state through the output arrays, gone).
  constexpr bool kFold = (JT && RW == 0u) || RW != 0u;          // the wave has a walk list
  constexpr uint32_t kParkCap = RW ? 128u : (G2 ? 96u : 64u);   // (a rows phase may hand over 64 at once; a pair-of-lanes batch parks up to 32)
  __shared__ uint64_t s_park_row[kFold ? kSThreads / 64 : 1][kFold ? kParkCap : 1];
  __shared__ uint32_t s_park_pid[kFold ? kSThreads / 64 : 1][kFold ? kParkCap : 1];
  __shared__ uint32_t s_park_it[kFold ? kSThreads / 64 : 1][kFold ? kParkCap : 1];
  uint32_t npark = 0;               // entries in this wave's list (wave-uniform)
  constexpr uint32_t kRowsCap = 64u + P;                        // the one-row patterns waiting for their lane: a phase starts at 64
  __shared__ uint64_t s_rows_row[RW ? kSThreads / 64 : 1][RW ? kRowsCap : 1];
  __shared__ uint64_t s_rows_end[RW ? kSThreads / 64 : 1][RW ? kRowsCap : 1];      // end of the pattern in the pattern buffer
  __shared__ uint32_t s_rows_pid[RW ? kSThreads / 64 : 1][RW ? kRowsCap : 1];
  __shared__ uint32_t s_rows_it[RW ? kSThreads / 64 : 1][RW ? kRowsCap : 1];
  __shared__ uint32_t s_rows_len[RW ? kSThreads / 64 : 1][RW ? kRowsCap : 1];
  uint32_t nrows = 0;               // entries in it (wave-uniform)
  uint32_t rsteps = 0, rlooks = 0;  // steps taken and row-table words fetched by this LANE in rows phases
  // A pattern's final interval (the lane that holds it calls).  pk_cap != ~0: straight into the 8-byte form (fmx.h) -- word
  // q of sp_out, wide intervals appended to the escape list behind word k -- instead of a pass of k_pack_intervals over
  // both arrays behind the search (every variant of this kernel finishes all of its patterns itself since round 5).
  auto emit = [&](uint32_t q, uint64_t a, uint64_t b) {
    if (pk_cap != ~0ull) {
      unsigned long long *pk = reinterpret_cast<unsigned long long *>(sp_out);
      const uint64_t w = b - a;
      if (w >= kPackWide) {
        const unsigned long long slot = atomicAdd(pk + k, 1ull);
        if (slot < pk_cap) { pk[(uint64_t)k + 1 + 2 * slot] = q; pk[(uint64_t)k + 2 + 2 * slot] = b; }
      }
      pk[q] = a | ((w < kPackWide ? w : kPackWide) << 40);
    } else {
      sp_out[q] = a;
      ep_out[q] = b;
    }
  };
  const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t lane64 = threadIdx.x & 63u;
  const uint64_t pat_addr = (uint64_t)(uintptr_t)pat;
  auto load_off_raw = [&](uint64_t bt, uint64_t &v0, uint64_t &v1) {      // batches past the end read the last pattern's offsets
    const uint64_t pid = bt * P + grp;
    const uint64_t *p = po.at(pid < k ? pid : (uint64_t)k - 1);
    v0 = p[0];
    v1 = p[1];
  };
  // (end, len) of group `grp` in batch bt from the raw offsets; groups past the end of the batch list get an empty
  // pattern at the end of the last one, so that a wave's span is always [begin of its lane 0, end of its lane 63)
  auto fix_off = [&](uint64_t bt, uint64_t v0, uint64_t v1, uint64_t &e, uint32_t &len) {
    const uint64_t pid = bt * P + grp;
    const uint64_t q = pid < k ? pid : (uint64_t)k - 1;
    const uint64_t b = po.fixed ? q * po.fixed : v0;
    e = po.fixed ? (q + 1) * po.fixed : v1;
    len = pid < k ? (uint32_t)(e - b) : 0u;
  };
  struct Stage { uint4 w; uint4 w2; uint64_t base; bool ok; };      // base: offset in the pattern buffer of LDS byte kStagePad (16-byte aligned address); w2: the second KiB (G2)
  auto stage_issue = [&](uint64_t e, uint32_t len) {
    const uint64_t b = e - len;
    const uint64_t b0 = ((uint64_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const uint64_t e1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e, 63);
    Stage st;
    const uint64_t al = (pat_addr + b0) & ~15ull;
    const uint64_t span = pat_addr + e1 - al;                 // bytes from the aligned start to the end of the last pattern
    st.ok = kStage && span <= kStageBytes;
    st.base = al - pat_addr;                                  // (wraps below zero when the buffer itself is unaligned: only differences are used)
    st.w = make_uint4(0, 0, 0, 0);
    if (st.ok && 16ull * lane64 < span) st.w = load_line16(al + 16u * lane64);      // the 16-byte block that holds the span's last byte is the last one read
    st.w2 = make_uint4(0, 0, 0, 0);
    if constexpr (G2) {
      if (st.ok && 1024ull + 16ull * lane64 < span) st.w2 = load_line16(al + 1024u + 16u * lane64);
    }
    return st;
  };
  auto stage_park = [&](const Stage &st, uint32_t area) {
    if (st.ok) *reinterpret_cast<uint4 *>(s_pat[wave_in_wg][area] + (kStagePad + 16u * lane64) / 4) = st.w;
    if constexpr (G2) {
      if (st.ok) *reinterpret_cast<uint4 *>(s_pat[wave_in_wg][area] + (kStagePad + 1024u + 16u * lane64) / 4) = st.w2;
    }
  };
  // the tail of a pattern: its first NT chunks (chunk i = the four bytes the search consumes at steps 4i .. 4i+3,
  // first one in byte lane 0); NT = 1 without the table, KT / 4 with it
  constexpr uint32_t NT = KT ? KT / 4 : 1;
  struct Tail { uint32_t c[NT]; };
  uint64_t end0, end1, raw2a, raw2b;
  uint32_t len0, len1;
  // lookups in the k-mer table (counters[9]), the row jump table (counters[10]) and the three-step row table (counters[11]):
  // counted per wave in scalar registers (a ballot's population count), not per lane
  uint32_t ktl = 0, jtl = 0, r3l = 0;
  {
    uint64_t a0, a1;
    load_off_raw(wave, a0, a1);
    fix_off(wave, a0, a1, end0, len0);
    load_off_raw((uint64_t)wave + nwaves, a0, a1);
    fix_off((uint64_t)wave + nwaves, a0, a1, end1, len1);
  }
  Stage cur = stage_issue(end0, len0);
  stage_park(cur, par);
  // without staging (the bytes layout) a batch's tail -- the chunks its k-mer lookup is made of -- is requested while the
  // batch before it is searched; a staging kernel's rare unstaged batch (a span over 1 KiB) fetches it when it starts
  Tail tail_ahead;
#pragma unroll
  for (uint32_t i = 0; i < NT; i++) tail_ahead.c[i] = (!kStage && len0 > 4u * i) ? fetch4(pat, end0 - 4ull * i) : 0u;
  // The parked patterns, P at a time, a lane group each: one-row steps (one rank query + one bit test, as in the step
  // loop) from the row and step they were parked with until the interval is empty -- or, should one not fail after all,
  // to its end -- leaving the reference loop's final values in the output arrays and counting its steps.
  auto walk_parked = [&](auto last_tag) {
    constexpr bool kLast = decltype(last_tag)::value;          // the walk before the wave ends (not the one that makes room inside the batch loop)
    while (npark) {                                            // wave-uniform
      const uint32_t take = npark < P ? npark : P;
      const bool actw = grp < take;
      const uint32_t slot = npark - take + (actw ? grp : 0u);
      npark -= take;
      const uint32_t wpid = actw ? s_park_pid[wave_in_wg][slot] : 0u;
      uint32_t wit = actw ? s_park_it[wave_in_wg][slot] : 0u;
      uint64_t wsp = actw ? s_park_row[wave_in_wg][slot] : 0ull;
      uint32_t wj = (uint32_t)(wsp >> 56);                       // steps that are known to succeed (the lookup that parked it saw them)
      const bool wnone = wj == 0xFFu;                            // FMX_SEARCH_MISS_NONE: parked as None -- nothing to walk, (0, 0) to write
      wj = wnone ? 0u : wj;
      wsp &= (1ull << 56) - 1;
      if constexpr (R3T && kLast) {
        // ... three of them at a time by the three-step row table: one 8-byte load where the step loop below spends three
        // rank queries (a pattern that misses in the middle of a nine-character entry: 3 dependent requests, not 5).  Only
        // in the last walk -- where nearly all of them happen: a wave parks ~1.6 patterns per batch of a workload with
        // 10 % misses and the list holds 64 -- because a second copy of this loop inside the batch loop costs the step
        // loop a register it does not have (the dictionary's lane address was spilled and re-read at every rank step).
        // Round 5 also tried the last walk with TWO patterns per lane group, every request of both in flight before either is
        // waited for (17 parked patterns per wave = one round instead of two): no change at C3 (0.1389-0.1395 against
        // 0.1373-0.1391 ms on one box) -- the waves' last walks overlap other waves' batches until the very end.
        while (__builtin_amdgcn_ballot_w64(actw && wj >= 3u)) {
          const bool go = actw && wj >= 3u;
          if (go) {
            wsp = r3tab[wsp] & ((1ull << 40) - 1);
            wj -= 3u;
            wit += 3u;
            steps += 3u;
          }
          r3l += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(go && t == 0));
        }
      }
      uint64_t wep = wsp + ((actw && !wnone) ? 1u : 0u);
      uint64_t wbegin = 0, wend = 0;
      if (actw) po.get(wpid, wbegin, wend);
      const uint32_t wlen = (uint32_t)(wend - wbegin);
      for (;;) {                                               // eight steps at a time
        const uint32_t wrem = wlen - wit;
        if (!__builtin_amdgcn_ballot_w64(actw && wsp < wep && wrem != 0u)) break;
        const uint32_t nst = wrem < 8u ? wrem : 8u;
        uint64_t chars = 0;                                    // the next nst characters, the one of step `wit` in the low byte
        if (actw && wsp < wep) {
          if (nst == 8u) {
            uint32_t lo, hi;
            __builtin_memcpy(&lo, pat + (wend - wit - 8), 4);
            __builtin_memcpy(&hi, pat + (wend - wit - 4), 4);
            chars = ((uint64_t)__builtin_bswap32(lo) << 32) | __builtin_bswap32(hi);
          } else {
            for (uint32_t s8 = 0; s8 < nst; s8++) chars |= (uint64_t)pat[wend - wit - 1 - s8] << (8u * s8);
          }
        }
        for (uint32_t s8 = 0; s8 < 8; s8++) {
          const bool stepping = actw && wsp < wep && s8 < nst;
          if (!__builtin_amdgcn_ballot_w64(stepping)) break;
          if (stepping) {
            const uint32_t c = (uint32_t)(chars >> (8u * s8)) & 0xFFu;
            const uint4 en = s_tab[c];
            const uint64_t cfc = ((uint64_t)en.y << 32) | en.x;
            const uint64_t vb = ((uint64_t)en.w << 32) | en.z;
            if (vb > 1) {
              if (LAYOUT == kLayoutBytes) {
                const ByteRankReq q1 = byte_rank_issue(ix, (uint16_t)(vb - 2), wsp, lc);
                wsp = cfc + byte_rank_finish(q1, c, lc);
                wep = wsp + byte_match_bit(q1, c, lc);
              } else {
                uint32_t b1, m1;
                split448(wsp, b1, m1);
                const Blk w1 = blk_load(vb, b1, m1);
                wsp = cfc + blk_rank(w1, m1);
                wep = wsp + blk_bit(w1, m1);
              }
              reqs += R * RB;
            } else {
              const uint64_t r1 = cfc + ((vb == 1 && wsp > ix.eof) ? 1u : 0u);
              wep = cfc + ((vb == 1 && wep > ix.eof) ? 1u : 0u);
              wsp = r1;
            }
            steps++;
          }
        }
        wit += nst;                                            // (meaningless once the interval is empty: the loop ends then)
      }
      if (actw && t == 0) emit(wpid, wsp, wep);
    }
  };
  // A rows phase (RW kernels): 64 of the waiting one-row patterns (all of them when the wave is about to end), one per LANE.
  // A one-row search needs no rank query -- it compares the pattern with the text in front of its row's suffix, which the
  // row table holds: R3[r] = (BWT'[r], BWT'[LF r], BWT'[LF^2 r]; LF^3 r), three steps per 8-byte word (RW = 3; the hand-over
  // left a multiple of three steps), or R1[r] = (BWT'[r]; LF r), one step per word (RW = 1) -- so it needs no lane group
  // either: 64 dependent chains per wave where the lane groups keep 8 (16).  A pattern that gets through is finished here;
  // one whose row disagrees FAILS within the word's steps, and the reference loop's values at the failing step are a rank
  // query: it goes to the walk list, at the step the word began with, for the lane groups.
  auto rows_phase = [&](const bool all) {
    if constexpr (RW != 0u) {
      while (nrows >= (all ? 1u : 64u)) {                        // wave-uniform
        const uint32_t take = nrows < 64u ? nrows : 64u;
        const bool on = lane64 < take;
        const uint32_t slot = nrows - take + (on ? lane64 : 0u);
        nrows -= take;
        uint64_t row = on ? s_rows_row[wave_in_wg][slot] : 0ull;
        const uint64_t rend = on ? s_rows_end[wave_in_wg][slot] : 0ull;
        const uint32_t rpid = on ? s_rows_pid[wave_in_wg][slot] : 0u;
        const uint32_t rlen = on ? s_rows_len[wave_in_wg][slot] : 0u;
        uint32_t rit = on ? s_rows_it[wave_in_wg][slot] : 0u;
        bool live = on;
        while (__builtin_amdgcn_ballot_w64(live)) {
          const uint32_t rem = rlen - rit;
          bool fail = false;                                       // this lane's pattern goes to the walk list now
          if constexpr (RW == 3u) {
            const bool tm = live && rem >= 3u;
            unsigned long long re = 0;
            uint32_t d = 0;
            if (tm) {                                              // pat[rend - rit - 3 .. rend - rit + 1): inside the pattern (rit >= 1)
              re = r3tab[row];
              __builtin_memcpy(&d, pat + (rend - rit - 3), 4);
            }
            if (tm && (uint32_t)(re >> 40) == __builtin_bswap32(d << 8)) {
              rlooks++;
              row = re & ((1ull << 40) - 1);
              rit += 3;
              rsteps += 3;
            } else if (live) {
              rlooks += tm ? 1u : 0u;
              if (rem == 0u) emit(rpid, row, row + 1);             // through
              else fail = true;                                    // the failing step is within these three (or a tail of one or two)
              live = false;
            }
          } else {
            const bool rm = live && rem != 0u;
            unsigned long long re = 0;
            uint32_t c = 0;
            if (rm) {
              re = r3tab[row];                                     // (the one-step table is passed in the same argument)
              c = pat[rend - rit - 1];
            }
            const uint32_t c2 = (uint32_t)(re >> 40) & 0xFFu;
            if (rm && c == c2 && c2 != 0u) {
              rlooks++;
              row = re & ((1ull << 40) - 1);
              rit++;
              rsteps++;
            } else if (live) {
              rlooks += rm ? 1u : 0u;
              if (rem == 0u) emit(rpid, row, row + 1);
              else fail = true;                                    // the failing step (or the end-of-text row): a rank query
              live = false;
            }
          }
          const unsigned long long fm = __builtin_amdgcn_ballot_w64(fail);
          if (fm) {
            const uint32_t ps = npark + (uint32_t)__builtin_popcountll(fm & ((1ull << lane64) - 1ull));
            if (fail) { s_park_row[wave_in_wg][ps] = row; s_park_pid[wave_in_wg][ps] = rpid; s_park_it[wave_in_wg][ps] = rit; }
            npark += (uint32_t)__builtin_popcountll(fm);
          }
        }
      }
    }
  };
#ifdef FMX_SEARCHLOG
  sl_t1 = __builtin_amdgcn_s_memrealtime();
#endif
  uint32_t batch = wave;
  Stage nxt_stage;
  {
    // One batch, written once and compiled twice: STAGED = its bytes are in LDS; else (a span longer than the LDS area)
    // they are read chunk by chunk from global memory.  Two copies of the code, so that no value of the staged path is
    // ever a merge with the result of a global load -- the compiler waits for ALL outstanding loads (`vmcnt(0)`) where it
    // meets such a value, and the next batch's loads issued below would be among them.
    // AHEAD: a strided batch, which requests the next one's bytes and the offsets of the one after it; a drawn one does not
    auto search_one_batch = [&](auto staged_tag, auto ahead_tag) {
    constexpr bool STAGED = decltype(staged_tag)::value;
    constexpr bool AHEAD = decltype(ahead_tag)::value;
    const uint32_t pid = batch * P + grp;
    const bool act = pid < k;
    const uint64_t *own = po.at(act ? pid : 0u);      // any valid address (pat_chunk)
    const uint64_t end = end0;
    const uint32_t len = act ? len0 : 0u;
    const uint64_t cur_base = cur.base;
    const uint32_t *const spat = s_pat[wave_in_wg][par];
    // chunk j of this group's pattern (see pat_chunk)
    auto chunk = [&](uint32_t j) -> uint32_t {
      if constexpr (!STAGED) {
        return pat_chunk(pat, own, end, len, j);
      } else {
        const uint32_t have = len > 4u * j ? len - 4u * j : 0u;
        const uint32_t o = have ? (uint32_t)(end - 4ull * j - cur_base) + (kStagePad - 4u) : 0u;      // LDS byte of the dword that ends where the chunk ends
        const uint32_t lo = spat[o >> 2], hi = spat[(o >> 2) + 1];
        const uint32_t r = __builtin_bswap32(__builtin_amdgcn_alignbyte(hi, lo, o & 3u));
        return have >= 4u ? r : (r & ((1u << (8u * have)) - 1u));
      }
    };
    // the four characters that steps at .. at + 3 consume, the first in byte lane 0, zeros behind the pattern's start -- for
    // any `at` (chunk(j) = chars4(4 j)): what a row-jump lookup compares, and where the cursor is set again behind one
    auto chars4 = [&](uint32_t at) -> uint32_t {
      const uint32_t have = len > at ? len - at : 0u;
      if constexpr (STAGED) {
        const uint32_t o = have ? (uint32_t)(end - at - cur_base) + (kStagePad - 4u) : 0u;
        const uint32_t lo = spat[o >> 2], hi = spat[(o >> 2) + 1];
        const uint32_t r = __builtin_bswap32(__builtin_amdgcn_alignbyte(hi, lo, o & 3u));
        return have >= 4u ? r : (r & ((1u << (8u * have)) - 1u));
      } else {
        uint32_t r = 0;
        if (have >= 4u) {
          uint32_t d;
          __builtin_memcpy(&d, pat + (end - at - 4), 4);
          r = __builtin_bswap32(d);
        } else {
          for (uint32_t j = 0; j < have; j++) r |= (uint32_t)pat[end - at - 1 - j] << (8u * j);
        }
        return r;
      }
    };
    Tail tailq;
#pragma unroll
    for (uint32_t i = 0; i < NT; i++) {
      if constexpr (STAGED) tailq.c[i] = chunk(i);
      else if constexpr (kStage) tailq.c[i] = len > 4u * i ? fetch4(pat, end - 4ull * i) : 0u;
      else tailq.c[i] = act ? tail_ahead.c[i] : 0u;
    }
    uint32_t ch = tailq.c[0];                                 // chunk 0
    uint32_t nx = chunk(KT ? KT / 4 + 1 : 1);                 // the chunk after the current one
    // the next batch's bytes and the offsets of the one after it: requested behind this batch's first lookup (below)
    auto issue_ahead = [&]() {
      if constexpr (AHEAD) {
        nxt_stage = stage_issue(end1, len1);
        if constexpr (!kStage) {
#pragma unroll
          for (uint32_t i = 0; i < NT; i++) tail_ahead.c[i] = len1 > 4u * i ? fetch4(pat, end1 - 4ull * i) : 0u;
        }
        load_off_raw((uint64_t)batch + 2ull * nwaves, raw2a, raw2b);
      }
    };
    if (KT == 0) { issue_ahead(); if constexpr (AHEAD) stage_park(nxt_stage, par ^ 1u); }
    uint64_t sp = 0, ep = ix.n;
    // symbols without a vector: absent (x = 0) or the EOF symbol (x = 1)
    auto special = [&](uint64_t cfc, uint64_t vb, uint64_t x) { return cfc + ((vb == 1 && x > ix.eof) ? 1u : 0u); };
    if (KT == 0) {
      // ---- step 0: rank(c, 0) = 0 and rank(c, n) = the symbol's count -- the interval is the symbol's
      // whole bucket [C[c], C[c+1]), no block needed
      if (len > 0) {
        const uint32_t c = ch & 0xFFu;
        const uint4 e = s_tab[c];
        const uint4 e2 = s_tab[(c + 1) & 0xFFu];
        const uint64_t cfc = ((uint64_t)e.y << 32) | e.x;
        const uint64_t vb = ((uint64_t)e.w << 32) | e.z;
        const uint64_t nxt = c == 255u ? ix.n : (((uint64_t)e2.y << 32) | e2.x);
        sp = vb > 1 ? cfc : special(cfc, vb, 0);
        ep = vb > 1 ? nxt : special(cfc, vb, ix.n);
        steps++;
      }
      ch >>= 8;
    } else {
      // ---- the first KT steps from the k-mer table: T[code] = (sp, ep, steps) after the KT characters of `code`
      // (for a k-mer that does not occur: the reference loop's values at its first empty step and the steps it
      // took).  A pattern shorter than KT, or with a byte that has no bit-vector among its last KT, walks those
      // steps the plain way below.
      uint32_t code = 0;
      bool elig = act && len >= KT;
#pragma unroll
      for (uint32_t j = 0; j < KT; j++) {
        const uint32_t d = s_dense[(tailq.c[j >> 2] >> (8u * (j & 3u))) & 0xFFu];
        elig = elig && d != 0xFFu;
        code = code * ksigma + d;
      }
      // every lane loads (a group that is not eligible reads entry 0 and ignores it): a load under `if (elig)` is merged with
      // a default value right behind it, and the compiler puts the wait for the load there -- before the loads below
      const uint4 ent = ktab[elig ? code : 0u];
      issue_ahead();
      if (elig) {
        sp = (((uint64_t)ent.y << 32) | ent.x) & ((1ull << 56) - 1);
        ep = ((uint64_t)ent.w << 32) | ent.z;
        steps += ent.y >> 24;           // the reference's loop ran this many steps on these characters
      }
      ktl += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(elig && t == 0));
      if constexpr (AHEAD) stage_park(nxt_stage, par ^ 1u);      // it was requested beside the entry and has arrived with it
      if (__builtin_amdgcn_ballot_w64(act && !elig)) {
        for (uint32_t j = 0; j < KT; j++) {
          const bool stepping = act && !elig && j < len && sp < ep;
          if (!__builtin_amdgcn_ballot_w64(stepping)) break;
          if (stepping) {
            const uint32_t c = (tailq.c[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
            const uint4 e = s_tab[c];
            if constexpr (G2) {      // (backward_step of fmx_device.h is written for the layout's own lane group)
              const uint64_t cfc = ((uint64_t)e.y << 32) | e.x, vb = ((uint64_t)e.w << 32) | e.z;
              if (vb > 1) {
                uint32_t b1, b2, m1, m2;
                split448(sp, b1, m1);
                split448(ep, b2, m2);
                const Blk w1 = blk_load(vb, b1, b2 != b1 ? m1 : (m1 > m2 ? m1 : m2));
                Blk w2 = w1;
                if (b2 != b1) { w2 = blk_load(vb, b2, m2); reqs += RB; }
                sp = cfc + blk_rank(w1, m1);
                ep = cfc + blk_rank(w2, m2);
                reqs += RB;
              } else {
                const uint64_t r1 = cfc + ((vb == 1 && sp > ix.eof) ? 1u : 0u);
                ep = cfc + ((vb == 1 && ep > ix.eof) ? 1u : 0u);
                sp = r1;
              }
            } else {
              reqs += backward_step<WIDE, LAYOUT>(ix, c, s_slot[c], ((uint64_t)e.y << 32) | e.x, lc, sp, ep);
            }
            steps++;
          }
        }
      }
      ch = chunk(KT / 4);              // the chunk step KT starts
    }
    uint32_t skip = 0;                                         // steps this group has jumped over and still sits out
    uint32_t cursor_it = KT ? KT : 1u;                         // the step (ch, nx) stand for
    bool deferred = false;                                     // this group's pattern was handed on (walk list / rows list)
    for (uint32_t it = KT ? KT : 1u;; it++) {                  // `it` is wave-uniform
      bool alive = it < len && sp < ep;
      if constexpr (RW != 0u) {      // one row: the rest is a lane's, in the wave's next rows phase
        const bool hand = alive && (ep - sp) == 1 && (RW == 1u || (len - it) % RW == 0u);
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(hand && t == 0);
        if (hm) {
          const uint32_t slot = nrows + (uint32_t)__builtin_popcountll(hm & ((1ull << lane64) - 1ull));
          if (hand && t == 0) {
            s_rows_row[wave_in_wg][slot] = sp; s_rows_end[wave_in_wg][slot] = end; s_rows_pid[wave_in_wg][slot] = pid;
            s_rows_it[wave_in_wg][slot] = it; s_rows_len[wave_in_wg][slot] = len;
          }
          nrows += (uint32_t)__builtin_popcountll(hm);
        }
        if (hand) {
          deferred = true;
          ep = sp;
          alive = false;
        }
      }
      if (!__builtin_amdgcn_ballot_w64(alive)) break;
      // the cursor (`ch`: what is left of the current chunk, `nx`: the next chunk) set for step ni; it is maintained step by
      // step while groups step and set afresh (cursor_it says for which step it stands) after the clock has jumped
      auto cursor_to = [&](uint32_t ni) {
        const uint32_t c0 = chars4(ni), a = ni & 3u;
        ch = a ? (c0 & ((1u << (8u * (4u - a))) - 1u)) : c0;
        nx = chunk((ni >> 2) + 1);
        cursor_it = ni;
      };
      // the twelve characters steps at .. at + 11 consume (p0 first): what a lookup in the row jump table compares
      auto chars12 = [&](uint32_t at, bool want, uint32_t &q0, uint32_t &q1, uint32_t &q2) {
        if constexpr (STAGED) {        // the 12 bytes that end where step `at` reads, in one burst of four LDS dwords
          const uint32_t o = want ? (uint32_t)(end - at - cur_base) + (kStagePad - 12u) : 0u;
          const uint32_t w0 = o >> 2, sh = o & 3u;
          const uint32_t d0 = spat[w0], d1 = spat[w0 + 1], d2 = spat[w0 + 2], d3 = spat[w0 + 3];
          q0 = __builtin_bswap32(__builtin_amdgcn_alignbyte(d3, d2, sh));
          q1 = __builtin_bswap32(__builtin_amdgcn_alignbyte(d2, d1, sh));
          q2 = __builtin_bswap32(__builtin_amdgcn_alignbyte(d1, d0, sh));
        } else {
          q0 = chars4(at); q1 = chars4(at + 4u); q2 = chars4(at + 8u);
        }
      };
      const uint32_t rem = len - it;
      bool lookedup = false;                                   // this group has taken steps by table lookup in this iteration
      bool park_now = false;                                   // ... or was found to miss by one: it is parked below
      uint32_t missj = 0;                                      // ... having agreed with its row's text for this many steps first
      uint32_t park_ahead = 0;                                 // ... from the step this many behind the wave's clock (a pair's first entry agreed)
      if (JT && !__builtin_amdgcn_ballot_w64(alive && (skip != 0u || (ep - sp) > (uint64_t)RG))) {
        // ---- every live group holds at most G rows (one, as a rule: sigma = 128, n = 2^32 -- from the 6th step on) and none
        // is sitting out: a group with jc or more characters left looks its rows up in the row jump table -- J[r] = the
        // jc characters an LF walk from r reads and the row it ends on (fmx_jump.hip).  The pattern's characters come from
        // the staged span at any offset (round 3's lookups had to start on a chunk boundary of the pattern and held
        // eight characters; nine fit C3's 32 - 5 = 27 one-row steps exactly: three lookups where there were three and a
        // three-step word); between an entry's arrival and the next entry's request stand a comparison and a select.
        const bool can = alive && rem >= jc;
        if (__builtin_amdgcn_ballot_w64(can)) {
          // JT == 2: the table holds PAIRS of entries, J[r] and J[LF^jc r] side by side in 32 bytes -- one request (a sector)
          // for up to 2 jc steps: the group's even lanes take the first entry and the pattern's next jc characters, its odd
          // lanes the second entry and the jc characters behind those (staged batches of the quad layout only)
          constexpr bool kPair = JT == 2u && STAGED && (G == 4 || G == 2);
          const uint64_t width = ep - sp;
          const bool single = !__builtin_amdgcn_ballot_w64(can && width != 1u);      // every group that looks up holds one row
          const uint32_t half = (kPair && single) ? (t & 1u) : 0u;
          const bool can2 = kPair && single && can && rem >= 2u * jc;
          uint32_t p0, p1, p2;
          chars12(it + (half && can2 ? jc : 0u), can, p0, p1, p2);      // (LDS: they arrive long before the entries requested below)
          const uint32_t m2 = jc > 8u ? ((1u << (8u * (jc - 8u))) - 1u) : 0u;
          bool jumped = false, pair_both = false, pair_then_miss = false;
          uint64_t rowj = 0;
          uint32_t nrows = 1;
          // ONE 16-byte load per row, everything taken out of it unconditionally: with the row used only under `if (hit)`,
          // the compiler sank that half of the load behind the comparison -- two dependent loads per lookup (round 4,
          // profiles/r04_c3_bound.md)
          if (single) {
            // one row per group: every lane of the group loads the same entry, the row comes straight out of it
            uint4 je = make_uint4(0, 0, 0, 0);
            if (can) je = jtab[(JT == 2u ? 2ull * sp : sp) + (half && can2 ? 1u : 0u)];
            const uint32_t d0 = je.x ^ p0, d1 = je.y ^ p1, d2 = (je.z ^ p2) & m2;
            jumped = can && (d0 | d1 | d2) == 0u;
            // (an entry's character of step s is its byte s: the first byte that differs is the step that fails)
            missj = d0 ? (uint32_t)__builtin_ctz(d0) >> 3 : (d1 ? 4u + ((uint32_t)__builtin_ctz(d1) >> 3) : 8u + ((uint32_t)__builtin_ctz(d2 | 0x80000000u) >> 3));
            rowj = (uint64_t)(je.z >> 24) | ((uint64_t)je.w << 8);
            if constexpr (kPair) {
              // what the two halves found, to every lane of the group: the first entry's verdict decides whether the group
              // moves at all, the second one's whether it moves jc steps or 2 jc
              const uint32_t ok = jumped ? 1u : 0u;
              const bool ok1 = gbc0(ok) != 0u, ok2 = can2 && gbc1(ok) != 0u;
              const uint32_t rlo1 = gbc0((uint32_t)rowj), rhi1 = gbc0((uint32_t)(rowj >> 32));
              const uint32_t rlo2 = gbc1((uint32_t)rowj), rhi2 = gbc1((uint32_t)(rowj >> 32));
              const uint32_t mj1 = gbc0(missj), mj2 = gbc1(missj);
              jumped = can && ok1;
              pair_both = ok1 && ok2;
              pair_then_miss = can2 && ok1 && !ok2;            // jc steps, and parked behind them with what the second entry saw
              rowj = pair_both ? (((uint64_t)rhi2 << 32) | rlo2) : (((uint64_t)rhi1 << 32) | rlo1);
              missj = pair_then_miss ? mj2 : mj1;
            }
            jtl += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(can && t == 0));
          } else {
            // two to G rows somewhere: lane t looks up row sp + t; the rows whose characters are the pattern's go on to
            // LF^jc of themselves -- LF keeps the order of rows that carry the same character, so they land side by side:
            // the new interval begins at the first survivor's image and has as many rows as there are survivors
            const uint32_t lane64g = threadIdx.x & 63u, gbase = lane64g - t;
            if constexpr (G2) {
              // (a pair: lane t takes rows sp + t and sp + 2 + t -- up to four rows, like a quad)
              const bool mn0 = can && (uint64_t)t < width, mn1 = can && (uint64_t)(2u + t) < width;
              uint4 j0 = make_uint4(0, 0, 0, 0), j1 = make_uint4(0, 0, 0, 0);
              if (mn0) j0 = jtab[JT == 2u ? 2ull * (sp + t) : sp + t];
              if (mn1) j1 = jtab[JT == 2u ? 2ull * (sp + 2u + t) : sp + 2u + t];
              const uint32_t d0 = j0.x ^ p0, d1 = j0.y ^ p1, d2 = (j0.z ^ p2) & m2;
              const bool hit0 = mn0 && (d0 | d1 | d2) == 0u;
              const bool hit1 = mn1 && ((j1.x ^ p0) | (j1.y ^ p1) | ((j1.z ^ p2) & m2)) == 0u;
              missj = d0 ? (uint32_t)__builtin_ctz(d0) >> 3 : (d1 ? 4u + ((uint32_t)__builtin_ctz(d1) >> 3) : 8u + ((uint32_t)__builtin_ctz(d2 | 0x80000000u) >> 3));      // (lane 0's first entry: row sp's)
              const uint32_t hm = ((uint32_t)(__builtin_amdgcn_ballot_w64(hit0) >> gbase) & 3u) | (((uint32_t)(__builtin_amdgcn_ballot_w64(hit1) >> gbase) & 3u) << 2);
              const uint32_t fr = hm ? (uint32_t)__builtin_ctz(hm) : 0u;           // the first surviving row: sp + fr, held by lane fr & 1 as its entry fr >> 1
              const uint4 js = (fr >> 1) ? j1 : j0;
              const int first = (int)(gbase + (fr & 1u));
              const uint32_t rlo = (uint32_t)__shfl((int)((js.z >> 24) | (js.w << 8)), first, 64), rhi = (uint32_t)__shfl((int)(js.w >> 24), first, 64);
              jumped = can && hm != 0u;
              rowj = ((uint64_t)rhi << 32) | rlo;
              nrows = (uint32_t)__builtin_popcount(hm);
              jtl += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mn0)) + (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mn1));
            } else {
            const bool mine = can && (uint64_t)t < width;
            uint4 je = make_uint4(0, 0, 0, 0);
            if (mine) je = jtab[JT == 2u ? 2ull * (sp + t) : sp + t];      // (the first entry of a pair)
            const uint32_t d0 = je.x ^ p0, d1 = je.y ^ p1, d2 = (je.z ^ p2) & m2;
            const bool hit = mine && (d0 | d1 | d2) == 0u;
            missj = d0 ? (uint32_t)__builtin_ctz(d0) >> 3 : (d1 ? 4u + ((uint32_t)__builtin_ctz(d1) >> 3) : 8u + ((uint32_t)__builtin_ctz(d2 | 0x80000000u) >> 3));      // (lane 0's: the entry of row sp)
            const uint32_t hm = (uint32_t)(__builtin_amdgcn_ballot_w64(hit) >> gbase) & ((1u << G) - 1u);
            const int first = (int)(gbase + (hm ? (uint32_t)__builtin_ctz(hm) : 0u));
            const uint32_t rlo = (uint32_t)__shfl((int)((je.z >> 24) | (je.w << 8)), first, 64), rhi = (uint32_t)__shfl((int)(je.w >> 24), first, 64);
            jumped = can && hm != 0u;
            rowj = ((uint64_t)rhi << 32) | rlo;
            nrows = (uint32_t)__builtin_popcount(hm);
            jtl += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));      // an entry per row looked up
            }
          }
          if (can) {
            if (!jumped && width == 1u) {
              // The pattern differs from its one row's text within these characters: it misses, and what is left to find is
              // where -- the reference loop's values at the failing step.  Walking there here would hold up the whole wave
              // (every lane executes the steps, the groups that jumped wait): the group parks its state in the wave's
              // walk list and retires; walk_parked walks the parked patterns P at a time, densely.
              park_now = true;
              deferred = true;
            }                                                        // (wider and no row agrees: it steps on and ends within jc steps)
            const uint32_t took = jumped ? (pair_both ? 2u * jc : jc) : 0u;
            if (pair_then_miss) { park_now = true; deferred = true; park_ahead = jc; }
            ep = (jumped && !pair_then_miss) ? rowj + nrows : ((deferred && !pair_then_miss) ? sp : (pair_then_miss ? rowj : ep));       // parked: not alive any more
            sp = jumped ? rowj : sp;
            steps += took;
            skip = pair_then_miss ? 0u : took;
            lookedup = jumped;
          }
        }
      }
      if (JT && R3T) {
        // ---- a group that holds at most G rows and has not just jumped takes THREE steps with the three-step row table,
        // lane t row sp + t as above: the tail of its pattern (fewer than jc characters left) at any step; with more left
        // -- it waits for the other groups of the wave to become narrow, or free, too -- only at every third step, so that
        // the groups that wait this way come free together (taken at any step, their three-step rests would interleave
        // and the row jump table, which wants every group free at once, would never be reached).
        const uint64_t width = ep - sp;
        const bool want3 = alive && skip == 0u && !deferred && !lookedup && width >= 1u && width <= (uint64_t)RG && rem >= 3u &&
                           (rem < jc || it % 3u == 0u);
        if (__builtin_amdgcn_ballot_w64(want3)) {
          const uint32_t three = chars4(it) & 0xFFFFFFu;
          const uint32_t lane64 = threadIdx.x & 63u, base = lane64 - t;
          unsigned long long re = 0;                      // lane 0's: row sp's word (what a miss is located with)
          uint32_t hm, lo3, hi3;
          bool mine;
          if constexpr (G2) {                             // a pair: rows sp + t and sp + 2 + t
            mine = want3 && (uint64_t)t < width;
            const bool mine1 = want3 && (uint64_t)(2u + t) < width;
            unsigned long long re1 = 0;
            if (mine) re = r3tab[sp + t];
            if (mine1) re1 = r3tab[sp + 2u + t];
            const bool h0 = mine && (uint32_t)(re >> 40) == three, h1 = mine1 && (uint32_t)(re1 >> 40) == three;
            hm = ((uint32_t)(__builtin_amdgcn_ballot_w64(h0) >> base) & 3u) | (((uint32_t)(__builtin_amdgcn_ballot_w64(h1) >> base) & 3u) << 2);
            const uint32_t fr = hm ? (uint32_t)__builtin_ctz(hm) : 0u;
            const unsigned long long rs = (fr >> 1) ? re1 : re;
            const int first = (int)(base + (fr & 1u));
            lo3 = (uint32_t)__shfl((int)(uint32_t)rs, first, 64);
            hi3 = (uint32_t)__shfl((int)(uint32_t)(rs >> 32), first, 64);
            r3l += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine1));      // (the first rows' words are counted below)
          } else {
          mine = want3 && (uint64_t)t < width;
          if (mine) re = r3tab[sp + t];
          const bool hit = mine && (uint32_t)(re >> 40) == three;
          hm = (uint32_t)(__builtin_amdgcn_ballot_w64(hit) >> base) & ((1u << G) - 1u);
          const int first = (int)(base + (hm ? (uint32_t)__builtin_ctz(hm) : 0u));
          lo3 = (uint32_t)__shfl((int)(uint32_t)re, first, 64);
          hi3 = (uint32_t)__shfl((int)(uint32_t)(re >> 32), first, 64);
          }
          if (want3) {
            const bool took = hm != 0u;
            if (took) {
              sp = (((uint64_t)hi3 << 32) | lo3) & ((1ull << 40) - 1);
              ep = sp + (uint32_t)__builtin_popcount(hm);
              steps += 3;
              skip = 3u;
              lookedup = true;
            } else if (width == 1u) {                                          // it fails within these three: the walk finds where
              park_now = true;
              deferred = true;
              ep = sp;
              missj = (uint32_t)__builtin_ctz((((uint32_t)(re >> 40) ^ three) & 0xFFFFFFu) | 0x80000000u) >> 3;       // (0 .. 2; lane 0's word is row sp's)
            }                                                                  // (wider and no row agrees: it steps on and ends within three steps)
          }
          r3l += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));      // a lane per row looked up
        }
      }
      if (JT && __builtin_amdgcn_ballot_w64(park_now)) {
        // (sp is still the row the lookup was made with: a parked group's sp is not touched again)
        if constexpr (kFold) {      // (park_now is only ever set by a table lookup: JT)
          // (miss_none: the entry is parked as "None" -- no row, 0xFF known-good steps -- and the steps the reference made on the
          // pattern are counted here: the missj that agree with the row's text and the one that does not; the walk then only
          // writes (0, 0) for it.  Writing it here costs the step loop 5-17 spilled vector registers.)
          if (miss_none && park_now) steps += missj + 1u;
          const unsigned long long pm = __builtin_amdgcn_ballot_w64(park_now && t == 0);
          const uint32_t lane64p = threadIdx.x & 63u;
          const uint32_t slot = npark + (uint32_t)__builtin_popcountll(pm & ((1ull << lane64p) - 1ull));
          if (park_now && t == 0) { s_park_row[wave_in_wg][slot] = miss_none ? (0xFFull << 56) : (sp | ((uint64_t)missj << 56)); s_park_pid[wave_in_wg][slot] = pid; s_park_it[wave_in_wg][slot] = it + park_ahead; }
          npark += (uint32_t)__builtin_popcountll(pm);
        }
      }
      if ((JT || R3T) && !__builtin_amdgcn_ballot_w64(alive && skip == 0u && !deferred)) {
        // ---- nobody steps in this iteration: every live group has jumped or is sitting out.  The clock goes to the first
        // group that is free again (1 .. 11 steps on).
        bool cand = alive && !deferred && skip != 0u;      // the smallest `skip` among them, a bit at a time
        uint32_t adv = 0;
#pragma unroll
        for (int bit = (JT == 2u ? 4 : 3); bit >= 0; bit--) {
          const bool z = cand && ((skip >> bit) & 1u) == 0u;
          if (__builtin_amdgcn_ballot_w64(z)) cand = z;
          else adv |= 1u << bit;
        }
        if (!__builtin_amdgcn_ballot_w64(cand)) adv = 1u;                       // (everybody parked: the loop ends at the top)
        skip -= skip >= adv ? adv : skip;
        it += adv - 1u;
        continue;
      }
      if constexpr (JT || R3T) {
        if (cursor_it != it) cursor_to(it);                                     // (wave-uniform; the clock only jumps with a row table)
      }
      const bool stepping = alive && skip == 0u && !deferred;
      skip -= skip ? 1u : 0u;
      const bool wide_iv = stepping && (ep - sp) != 1;
      if (!__builtin_amdgcn_ballot_w64(wide_iv)) {
        // ---- every stepping group holds one row: one rank query + one bit (byte) test
        if (stepping) {
          const uint32_t c = ch & 0xFFu;
          const uint4 e = s_tab[c];
          const uint64_t cfc = ((uint64_t)e.y << 32) | e.x;
          const uint64_t vb = ((uint64_t)e.w << 32) | e.z;
          if (vb > 1) {
            if (LAYOUT == kLayoutBytes) {
              const ByteRankReq q1 = byte_rank_issue(ix, (uint16_t)(vb - 2), sp, lc);
              sp = cfc + byte_rank_finish(q1, c, lc);
              ep = sp + byte_match_bit(q1, c, lc);
            } else {
              uint32_t b1, m1;
              split448(sp, b1, m1);
              const Blk w1 = blk_load(vb, b1, m1);
              sp = cfc + blk_rank(w1, m1);
              ep = sp + blk_bit(w1, m1);
            }
            reqs += R * RB;
          } else {
            const uint64_t r1 = special(cfc, vb, sp);
            ep = special(cfc, vb, ep);
            sp = r1;
          }
          steps++;
        }
      } else if (stepping) {
        // ---- general step: two rank queries
        const uint32_t c = ch & 0xFFu;
        const uint4 e = s_tab[c];
        const uint64_t cfc = ((uint64_t)e.y << 32) | e.x;
        const uint64_t vb = ((uint64_t)e.w << 32) | e.z;
        if (vb > 1) {
          if (LAYOUT == kLayoutBytes) {
            // narrow intervals: sp and ep share a 128-position block -- one block line and one checkpoint, both ranks
            const ByteRankReq q1 = byte_rank_issue(ix, (uint16_t)(vb - 2), sp, lc);
            ByteRankReq q2 = q1;
            const bool two = (ep >> 7) != (sp >> 7);
            if (two) q2 = byte_rank_issue(ix, (uint16_t)(vb - 2), ep, lc);
            else q2.rem = (uint32_t)ep & 127u;
            sp = cfc + byte_rank_finish(q1, c, lc);
            ep = cfc + byte_rank_finish(q2, c, lc);
            reqs += two ? 2 * R : R;
          } else {
            uint32_t b1, b2, m1, m2;
            split448(sp, b1, m1);
            split448(ep, b2, m2);
            const Blk w1 = blk_load(vb, b1, b2 != b1 ? m1 : (m1 > m2 ? m1 : m2));
            Blk w2 = w1;                                       // narrow intervals: sp and ep share a block
            if (b2 != b1) { w2 = blk_load(vb, b2, m2); reqs += RB; }
            sp = cfc + blk_rank(w1, m1);
            ep = cfc + blk_rank(w2, m2);
            reqs += RB;
          }
        } else {
          const uint64_t r1 = special(cfc, vb, sp);
          ep = special(cfc, vb, ep);
          sp = r1;
        }
        steps++;
      }
      // pattern cursor: scalar bookkeeping, one dword fetch every 4th step, two chunks ahead
      ch >>= 8;
      if ((it & 3u) == 3u) {
        ch = nx;
        nx = chunk((it >> 2) + 2);
      }
      cursor_it = it + 1u;
    }
    if (act && t == 0 && !deferred) emit(pid, sp, ep);
    };      // search_one_batch
    auto one_batch = [&](auto ahead_tag) {
      if constexpr (kStage) {
        if (cur.ok) search_one_batch(std::true_type{}, ahead_tag);
        else search_one_batch(std::false_type{}, ahead_tag);
      } else {
        search_one_batch(std::false_type{}, ahead_tag);
      }
      if constexpr (RW != 0u) {
        if (nrows >= 64u) {
          if (npark > kParkCap - 64u) walk_parked(std::false_type{});         // room for all a rows phase may hand over
          rows_phase(false);
        }
      } else {
        if (kFold && npark > kParkCap - P) walk_parked(std::false_type{});    // room for a whole batch's groups
      }
    };
    // ---- the strided rounds
    for (; batch < nstatic; batch += nwaves) {
#ifdef FMX_SEARCHLOG
      sl_batches++;
#endif
      one_batch(std::true_type{});
      cur = nxt_stage;
      par ^= 1u;
      end0 = end1; len0 = len1;
      fix_off((uint64_t)batch + 2ull * nwaves, raw2a, raw2b, end1, len1);
    }
    // ---- the pool: batches nstatic .. nbatch - 1 by ticket, from kTixShards counters: the waves of eight consecutive workgroups
    // (one per XCD) share a counter, wave w draws from counter (w / 32) mod 64, and ticket j of counter c is pool batch c + 64 j
    // (fewer counters when the grid has fewer than 64 such groups).
    // Every wave draws until its ticket is past its counter's share -- one failing draw each -- so a counter ends at its share
    // + its waves, and the wave that drew the last of those sets it back to zero for the stream's next launch (every other
    // draw from it has returned by then: the counter serialises them).
    if constexpr (kPool) {
    if (nstatic != nbatch) {
      const uint32_t groups = (nwaves + 31u) >> 5;                                                      // of 32 waves (the last may be short)
      const uint32_t nshards = groups < kTixShards ? groups : kTixShards;
      const uint32_t shard = (wave >> 5) % nshards;
      unsigned long long *tix = counters + (kCounterBytes + kCensusBytes + kCalibScratchBytes) / 8 +
                                ((size_t)(tix_area - 1u) * kTixShards + shard) * kTixStride;
      const uint32_t npool = nbatch - nstatic;
      const uint32_t pool = npool > shard ? (npool - shard + nshards - 1u) / nshards : 0u;               // this counter's batches
      const uint32_t mine = groups > shard ? (groups - shard + nshards - 1u) / nshards : 0u;             // ... its groups of waves
      const uint32_t drawers = mine * 32u - ((groups - 1u) % nshards == shard ? groups * 32u - nwaves : 0u);
      uint32_t drawn = 0;
      if (lane64 == 0) drawn = (uint32_t)atomicAdd(tix, 1ull);
      for (;;) {
        const uint32_t tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
        if (tk >= pool) {
          if (tk == pool + drawers - 1u && lane64 == 0) atomicExch(tix, 0ull);
          break;
        }
        if (lane64 == 0) drawn = (uint32_t)atomicAdd(tix, 1ull);      // the next one, looked at when this batch is done
        batch = nstatic + shard + nshards * tk;
#ifdef FMX_SEARCHLOG
        sl_batches++;
#endif
        {
          uint64_t a0, a1;
          load_off_raw(batch, a0, a1);
          fix_off(batch, a0, a1, end0, len0);
        }
        cur = stage_issue(end0, len0);
        stage_park(cur, par);
#pragma unroll
        for (uint32_t i = 0; i < NT; i++) tail_ahead.c[i] = (!kStage && len0 > 4u * i) ? fetch4(pat, end0 - 4ull * i) : 0u;
        one_batch(std::false_type{});
      }
    }
    }
  }
  if constexpr (RW != 0u) {
    if (npark > kParkCap - 64u) walk_parked(std::false_type{});
    rows_phase(true);
  }
  // Round 5 also tried the LAST walk of a bytes-layout wave by pairs of lanes (64 of the block's 128 bytes per lane, 32 patterns
  // per round): a C5 wave's ~20 parked patterns are three rounds of dependent steps by its octets -- 17.7 us at the median, at the
  // very end of the launch (profiles/r05_c5_c2_wave_timeline.txt) -- and would be one.  It compiles to 73 vector registers where
  // the octets' walk has 71, i.e. six waves per SIMD instead of seven; held to seven by the occupancy attribute the compiler
  // also cuts the scalar budget to 94 (21 spills): 0.328-0.332 ms against 0.321 on one box.  Taken out.
#ifdef FMX_SEARCHLOG
  const unsigned long long sl_tw = __builtin_amdgcn_s_memrealtime();      // the last walk begins
#endif
  if (kFold) walk_parked(std::true_type{});
#ifdef FMX_SEARCHLOG
  const unsigned long long sl_t2 = __builtin_amdgcn_s_memrealtime();
#endif
  counters_add(counters, (t == 0 ? 2ull * steps : 0ull) + 2ull * rsteps, (t == 0 ? steps : 0u) + rsteps, t == 0 ? reqs : 0u);
  if (RW) {      // the rows phases' row-table words (counters[11])
    const unsigned long long rl = wave_sum((unsigned long long)rlooks);
    if ((threadIdx.x & 63u) == 0 && rl) atomicAdd(counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride + 11, rl);
  }
#ifdef FMX_SEARCHLOG
  if ((threadIdx.x & 63u) == 0 && wave < (1u << 15)) {
    unsigned long long *e = g_searchlog[wave];
    e[0] = sl_t0; e[1] = ((sl_t1 - sl_t0) & 0xFFFFFFFFull) | ((sl_tw - sl_t0) << 32); e[2] = sl_t2; e[3] = __builtin_amdgcn_s_memrealtime() | ((unsigned long long)sl_batches << 48);
  }
#endif
  if (threadIdx.x == 0 && blockIdx.x < kCensusBlocks)       // ... and when its first wave ended
    counters[(size_t)kCounterSlots * kCounterStride + 2u * blockIdx.x + 1u] = __builtin_amdgcn_s_memrealtime();
  if (KT) {
    const unsigned long long lookups = ktl;
    if ((threadIdx.x & 63u) == 0 && lookups)
      atomicAdd(counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride + 9, lookups);
  }
  if (JT) {
    const unsigned long long lookups = jtl;
    if ((threadIdx.x & 63u) == 0 && lookups)
      atomicAdd(counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride + 10, lookups);
  }
  if (JT && R3T) {
    const unsigned long long lookups = r3l;
    if ((threadIdx.x & 63u) == 0 && lookups)
      atomicAdd(counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride + 11, lookups);
  }
}

hipError_t launch_search_v1(const Index *h, const void *d_pat, PatOff po, void *d_sp, void *d_ep, uint64_t k,
                            hipStream_t st);

static int search_variant() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("FMX_SEARCH_VARIANT");
    v = e ? atoi(e) : 4;      // 1: generic step kernel; anything else: k_search4
  }
  return v;
}

// Resident workgroups per CU for a kernel: the grid is sized to what is resident so that every
// group starts at once and the static batch striding stays balanced.
template <class K>
static int blocks_per_cu(K kernel) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kSThreads, 0) != hipSuccess || nb < 1) nb = 1;
  return nb > 8 ? 8 : nb;
}

// What the occupancy query says is an upper bound: on this chip a kernel with 81..96 scalar registers gets one
// workgroup per CU fewer than it answers (MI355X_MICROARCH.md, "Residency"; there is no API for the scalar register
// count), and the surplus workgroups run as a second generation behind the first -- C5's bytes-layout kernel: 8 asked
// for, 7 resident, a launch of 0.205 ms instead of 0.170.  So every instantiation of k_search4 is calibrated by a
// CENSUS on the device it runs on: a calibration launch of the full grid (one empty pattern; every workgroup spins
// 60 us) leaves each workgroup's begin and end times behind the counters (fmx_device.h); the host counts the workgroups
// that began before the first one ended and divides by the CUs.  A census taken while other work held part of the
// device counts late workgroups that waited for THAT: only an answer of the query's number or up to two below it is
// believed, and only when two launches in a row give it (at most four are made).
// Round 4 took the readings inside fmx_search_batch_dev -- a 64 KB copy and a stream synchronisation on the CALLER's
// stream in the 2nd to 6th full-size call, against the header's promise that _dev entry points only enqueue work
// (ADVICE r4).  Now search_calibrate does it: called by fmx_prepare, and by the search that builds
// a handle's tables at its threshold (that call allocates and synchronises anyway, and says so); no other call reads
// anything back.  An instantiation that was never calibrated keeps the query's answer.
struct Residency {
  std::atomic<int> admitted{0};            // workgroups per CU that were resident at once; 0 = not measured yet
  std::mutex mu;                           // one calibration at a time per instantiation and device
};
constexpr uint32_t kCalibSpin = 6000;      // 60 us
static int census_read(const Index *h, int grid, int api, hipStream_t st) {
  std::vector<unsigned long long> t(2 * (size_t)kCensusBlocks + 2);
  if (hipMemcpyAsync(t.data(), h->d_counters + (size_t)kCounterSlots * kCounterStride, kCensusBytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (t[2 * (size_t)kCensusBlocks] != (unsigned long long)grid) return 0;      // the last launch to write here was not the calibration (another stream's search)
  const int nb = std::min<int>(grid, (int)kCensusBlocks);
  unsigned long long first_begin = ~0ull, first_end = ~0ull;
  for (int i = 0; i < nb; i++) {
    if (!t[2 * i] || t[2 * i + 1] < t[2 * i]) return 0;      // overwritten half-way by another launch
    first_begin = std::min(first_begin, t[2 * i]);
    first_end = std::min(first_end, t[2 * i + 1]);
  }
  if (first_end < first_begin + kCalibSpin / 2) return 0;
  int resident = 0;
  for (int i = 0; i < nb; i++) resident += t[2 * i] < first_end ? 1 : 0;
  const int got = resident / std::max(1, h->cu_count);
  return (got >= api - 2 && got >= 1) ? std::min(api, got) : 0;      // fewer: the device was not this launch's alone
}

// The ticket area (1 .. kTixAreas; 0: none) of the stream a search is enqueued on: k_search4, "The last rounds are DRAWN".
// FMX_SEARCH_TICKETS=0: none for anybody (every batch strided, as until round 5).
static uint32_t ticket_area(const Index *h, hipStream_t st) {
  static const bool off = getenv("FMX_SEARCH_TICKETS") && atoi(getenv("FMX_SEARCH_TICKETS")) == 0;
  if (off) return 0u;
  if (st == hipStreamPerThread) return 0u;      // ONE handle value, a stream per host thread: launches "on it" may run side by side
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return 0u; }
  if (cs != hipStreamCaptureStatusNone) return 0u;      // a graph may be replayed on any stream, beside anything
  static_assert(sizeof(h->tix_owner) / sizeof(h->tix_owner[0]) == kTixAreas, "one owner per ticket area");
  std::lock_guard<std::mutex> lk(h->tix_mu);
  uint32_t free_at = 0;
  for (uint32_t i = 0; i < kTixAreas; i++) {
    if (h->tix_used[i] && h->tix_owner[i] == (void *)st) return i + 1u;
    if (!h->tix_used[i] && !free_at) free_at = i + 1u;
  }
  if (free_at) { h->tix_used[free_at - 1u] = true; h->tix_owner[free_at - 1u] = (void *)st; }
  return free_at;
}

// full rounds of batches that are drawn with the partial last one (FMX_SEARCH_POOL_ROUNDS: 1 .. 15; 2 by measurement -- C5 0.318-0.320 /
// 0.314-0.316 / 0.315-0.316 ms, C3text 0.425 / 0.423 / 0.435 ms with 1 / 2 / 4: every drawn batch starts cold)
static uint32_t pool_rounds() {
  static const uint32_t v = [] { const char *e = getenv("FMX_SEARCH_POOL_ROUNDS"); const int x = e ? atoi(e) : 2; return (uint32_t)std::max(1, std::min(x, 15)); }();
  return v;
}

// cal: this is search_calibrate's call -- nothing is searched, the instantiation is calibrated
template <bool WIDE, uint32_t LAYOUT, uint32_t KT, uint32_t JT, uint32_t RW, bool R3T = false, bool G2 = false>
static hipError_t launch_v4kj(const Index *h, const KTab &kt, const uint4 *jt, const unsigned long long *r1, const uint8_t *pat,
                              const PatOff off, uint64_t *sp, uint64_t *ep, uint32_t k, hipStream_t st, uint64_t pk_cap, uint32_t flags, bool cal) {
  static const int api = blocks_per_cu(k_search4<WIDE, LAYOUT, KT, JT, RW, R3T, G2>);
  // FMX_SEARCH_WGS: fewer resident workgroups per CU (an experiment on how throughput follows the chains in flight)
  static const int forced = getenv("FMX_SEARCH_WGS") ? std::max(1, std::min(atoi(getenv("FMX_SEARCH_WGS")), api)) : 0;
  static Residency res[16];
  Residency &rs = res[(unsigned)h->device & 15u];
  constexpr uint64_t per_wg = kSThreads / (G2 ? 2 : Lay<LAYOUT>::G);
  if (cal) {
    std::lock_guard<std::mutex> lk(rs.mu);
    if (!forced && !rs.admitted.load() && h->cu_count * api <= (int)kCensusBlocks) {
      static const bool trace = getenv("FMX_TRACE") != nullptr;
      // scratch behind the census entries: two zero words (the offsets of one empty pattern) and the two output words
      unsigned long long *scr = h->d_counters + (kCounterBytes + kCensusBytes) / 8;
      const PatOff po{(const uint64_t *)scr, 0ull};
      const int grid = h->cu_count * api;
      { const hipError_t e0 = hipMemsetAsync(scr, 0, kCalibScratchBytes, st); if (e0 != hipSuccess) return e0; }      // (the offsets MUST be zeros: the kernel reads pat[off])
      int last = 0, got = 0;
      for (int attempt = 0; attempt < 4 && !rs.admitted.load(); attempt++) {
        k_search4<WIDE, LAYOUT, KT, JT, RW, R3T, G2><<<grid, kSThreads, 0, st>>>(h->dev, KT ? kt.level[KT - 1] : nullptr, kt.dense, kt.sigma, jt, h->jump_chars,
                                                                             (R3T || RW) ? r1 : nullptr, (const uint8_t *)h->d_bwt, po, (uint64_t *)(scr + 2), (uint64_t *)(scr + 3),
                                                                             1u, h->d_counters, ~0ull, kCalibSpin);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        got = census_read(h, grid, api, st);
        if (trace) fprintf(stderr, "[fmx] k_search4<%d,%u,%u,%d,%u,%d,%d> census: %d of the %d workgroups per CU the occupancy query allows were resident\n",
                           (int)WIDE, LAYOUT, KT, (int)JT, RW, (int)R3T, (int)G2, got, api);
        if (got && got == last) rs.admitted.store(got);
        last = got;
      }
      if (!rs.admitted.load()) rs.admitted.store(-1);      // no two agreeing readings (a shared device): the query's answer stands, and is not asked again
    }
    const int m = rs.admitted.load();
    h->search_residency.store(forced ? ((uint32_t)forced | 0x100u) : ((uint32_t)(m > 0 ? m : api) | (m > 0 ? 0x100u : 0u)));
    return hipSuccess;
  }
  const int measured = std::max(0, rs.admitted.load());
  const int per_cu = forced ? forced : (measured ? measured : api);
  uint64_t want = ((uint64_t)k + per_wg - 1) / per_wg;
  uint64_t cap = (uint64_t)h->cu_count * per_cu;
  int grid = (int)(want < cap ? (want ? want : 1) : cap);
  h->search_residency.store((uint32_t)per_cu | ((forced || measured) ? 0x100u : 0u));
  // ONE launch in every configuration (round 5: the kernels without a row jump table used to be followed by k_search_rows and
  // k_search_defer); the 8-byte form, where asked for, is written by the kernel itself
  const uint32_t area = ticket_area(h, st);
  {
    static const bool trace = getenv("FMX_TRACE") != nullptr;
    if (trace) {      // (what the kernel will make of it: k_search4, "The last rounds are DRAWN")
      constexpr uint32_t Pw = 64u / (G2 ? 2u : (uint32_t)Lay<LAYOUT>::G);
      const uint64_t nbatch = ((uint64_t)k + Pw - 1) / Pw, nw = (uint64_t)grid * (kSThreads / 64), rounds = nbatch / nw;
      const bool pooled = (G2 || RW != 0u) && area && rounds >= pool_rounds() + 2u;
      fprintf(stderr, "[fmx] k_search4<%d,%u,%u,%d,%u,%d,%d>: %llu batches over %llu waves, the last %llu drawn from ticket area %u\n", (int)WIDE, LAYOUT, KT,
              (int)JT, RW, (int)R3T, (int)G2, (unsigned long long)nbatch, (unsigned long long)nw,
              (unsigned long long)(pooled ? nbatch - (rounds - pool_rounds()) * nw : 0), pooled ? area : 0u);
    }
  }
  k_search4<WIDE, LAYOUT, KT, JT, RW, R3T, G2><<<grid, kSThreads, 0, st>>>(h->dev, KT ? kt.level[KT - 1] : nullptr, kt.dense, kt.sigma, jt, h->jump_chars,
                                                                       (R3T || RW) ? r1 : nullptr, pat, off, sp, ep, k, h->d_counters, pk_cap, (area << 16) | (pool_rounds() << 24) | ((flags & kSearchMissNone) ? 0x80000000u : 0u));
  return hipGetLastError();
}
template <bool WIDE, uint32_t LAYOUT, uint32_t KT>
static hipError_t launch_v4k(const Index *h, const KTab &kt, const uint8_t *pat, const PatOff off, uint64_t *sp, uint64_t *ep,
                             uint32_t k, hipStream_t st, uint64_t pk_cap, uint32_t flags, bool cal) {
  // the row tables are built by fmx_prepare or by the search that brings the handle's patterns to the threshold
  // (fmx_jump.hip, tables_due); until then -- a per-call adapter's single queries -- every step is walked on the dictionary
  // (the search that finds them due builds them -- it allocates and synchronises, fmx.h says so -- and calibrates the
  // kernel they select right away: later calls only enqueue)
  bool built_now = false;
  const bool due = !cal && tables_due(h, k, false);
  if (due) { std::lock_guard<std::mutex> lk(h->jt_mu); built_now = !h->jt_ready; }
  const uint4 *jt = nullptr;
  hipError_t e = jump_get(h, st, &jt, due);
  if (e != hipSuccess) return e;
  if (built_now) {
    const unsigned long long *r3b = nullptr;
    (void)row3_get(h, st, &r3b, true);
    if ((e = search_calibrate(h, st)) != hipSuccess) return e;
  }
  // With a row jump table the lane groups finish the one-row part themselves, eight steps per lookup, in lockstep
  // (C3: 0.240 ms; handing it to k_search_rows: 0.248 ms -- both run at ~37 G requests/s, and the hand-over costs two
  // more launches: C2 0.129 -> 0.155 ms).  Without one (it does not fit: C5, n = 2^34) a table of a third of the size
  // serves it with one lane per pattern: the three-step row table, or the regex frontier's one-step row table where
  // the handle has that one already (C5: 49 -> 75 G rank queries/s with R1).
  static const int rows = getenv("FMX_ROWS") ? atoi(getenv("FMX_ROWS")) : -1;      // 0: never k_search_rows, 1: R1 (and J) whenever there is a row table
  if (rows == 1) {
    const unsigned long long *r1 = nullptr;
    if ((e = row1_get(h, st, &r1, due)) != hipSuccess) return e;
    if (r1) return launch_v4kj<WIDE, LAYOUT, KT, 0u, 1>(h, kt, h->jump_pairs ? nullptr : jt, r1, pat, off, sp, ep, k, st, pk_cap, flags, cal);
  }
  if (jt) {      // and the three-step table beside it, for the steps no aligned jump covers
    const unsigned long long *r3 = nullptr;
    if (rows != 0 && (e = row3_get(h, st, &r3, due)) != hipSuccess) return e;
    if (h->jump_pairs) {    // (pairs are built from the three-step table: it is there)
      // large batches on the one-hot layout: a PAIR of lanes per pattern, 32 patterns per wave (k_search4<.., G2>) -- half the
      // lockstep batches per wave; a batch that does not give every CU four workgroups of pairs keeps the quads, which spread
      // it over twice the workgroups (C3 by batch size, pairs / quads: 125 k 0.0296 / 0.0335 ms, 250 k 0.0484 / 0.0493,
      // 500 k 0.0751 / 0.0744, 1 M 0.127 / 0.133, 4 M 0.482 / 0.499)
      // (FMX_SEARCH_G2=0 / 1: never / whenever the instantiation exists)
      if constexpr (LAYOUT == kLayoutOneHot) {
        const char *g2e = getenv("FMX_SEARCH_G2");      // (looked at per launch: the tests switch it inside one process)
        const int g2 = g2e ? atoi(g2e) : -1;
        if (r3 && g2 != 0) {
          if (cal) {                  // fmx_prepare calibrates both instantiations: which one a search takes depends on its size
            const hipError_t ec = launch_v4kj<WIDE, LAYOUT, KT, 2u, 0, true, true>(h, kt, jt, r3, pat, off, sp, ep, k, st, pk_cap, flags, true);
            if (ec != hipSuccess) return ec;
          } else if (g2 == 1 || (uint64_t)k >= (uint64_t)h->cu_count * 512) {
            return launch_v4kj<WIDE, LAYOUT, KT, 2u, 0, true, true>(h, kt, jt, r3, pat, off, sp, ep, k, st, pk_cap, flags, false);
          }
        }
      }
      return r3 ? launch_v4kj<WIDE, LAYOUT, KT, 2u, 0, true>(h, kt, jt, r3, pat, off, sp, ep, k, st, pk_cap, flags, cal)
                : launch_v4kj<WIDE, LAYOUT, KT, 2u, 0>(h, kt, jt, nullptr, pat, off, sp, ep, k, st, pk_cap, flags, cal);
    }
    // single entries (an index under 2^30 rows, or one whose budget has no room for pairs): the pairs of LANES serve it just the same
    if constexpr (LAYOUT == kLayoutOneHot) {
      const char *g2e = getenv("FMX_SEARCH_G2");
      const int g2 = g2e ? atoi(g2e) : -1;
      if (r3 && g2 != 0) {
        if (cal) {
          const hipError_t ec = launch_v4kj<WIDE, LAYOUT, KT, 1u, 0, true, true>(h, kt, jt, r3, pat, off, sp, ep, k, st, pk_cap, flags, true);
          if (ec != hipSuccess) return ec;
        } else if (g2 == 1 || (uint64_t)k >= (uint64_t)h->cu_count * 512) {
          return launch_v4kj<WIDE, LAYOUT, KT, 1u, 0, true, true>(h, kt, jt, r3, pat, off, sp, ep, k, st, pk_cap, flags, false);
        }
      }
    }
    return r3 ? launch_v4kj<WIDE, LAYOUT, KT, 1u, 0, true>(h, kt, jt, r3, pat, off, sp, ep, k, st, pk_cap, flags, cal)
              : launch_v4kj<WIDE, LAYOUT, KT, 1u, 0>(h, kt, jt, nullptr, pat, off, sp, ep, k, st, pk_cap, flags, cal);
  }
  if (rows != 0) {
    const unsigned long long *r3 = nullptr, *r1 = nullptr;
    bool have1;
    { std::lock_guard<std::mutex> lk(h->r1_mu); have1 = h->d_row1 != nullptr; }
    if (!have1 && (e = row3_get(h, st, &r3, due)) != hipSuccess) return e;
    if (r3) return launch_v4kj<WIDE, LAYOUT, KT, 0u, 3>(h, kt, nullptr, r3, pat, off, sp, ep, k, st, pk_cap, flags, cal);
    if ((e = row1_get(h, st, &r1, due)) != hipSuccess) return e;
    if (r1) return launch_v4kj<WIDE, LAYOUT, KT, 0u, 1>(h, kt, nullptr, r1, pat, off, sp, ep, k, st, pk_cap, flags, cal);
  }
  return launch_v4kj<WIDE, LAYOUT, KT, 0u, 0>(h, kt, nullptr, nullptr, pat, off, sp, ep, k, st, pk_cap, flags, cal);
}

template <bool WIDE, uint32_t LAYOUT>
static hipError_t launch_v4(const Index *h, const uint8_t *pat, const PatOff off, uint64_t *sp, uint64_t *ep,
                            uint32_t k, hipStream_t st, uint64_t pk_cap, uint32_t flags = 0u, bool cal = false) {
  KTab kt;
  const hipError_t e = ktab_get(h, st, &kt, !cal && tables_due(h, k, true));
  if (e != hipSuccess) return e;
  // the search uses the table's levels in steps of four characters (all levels are kept: fmx_ktab.hip)
  if (kt.k >= 12) return launch_v4k<WIDE, LAYOUT, 12>(h, kt, pat, off, sp, ep, k, st, pk_cap, flags, cal);
  if (kt.k >= 8) return launch_v4k<WIDE, LAYOUT, 8>(h, kt, pat, off, sp, ep, k, st, pk_cap, flags, cal);
  if (kt.k >= 4) return launch_v4k<WIDE, LAYOUT, 4>(h, kt, pat, off, sp, ep, k, st, pk_cap, flags, cal);
  return launch_v4k<WIDE, LAYOUT, 0>(h, kt, pat, off, sp, ep, k, st, pk_cap, flags, cal);
}

// One launch per call: no scratch, nothing to own per stream, so concurrent calls on one handle need no lock.
// d_off == nullptr: a batch of k patterns of fixed_len bytes each, one behind the other (fmx_search_opts.fixed_len).
// pack_cap != ~0: the intervals in the 8-byte form (fmx.h) with room for pack_cap escape entries -- d_sp receives the
// packed words (its first k words double as the kernels' sp array where a set of kernels cannot pack by itself), d_ep is
// scratch.
hipError_t launch_search(const Index *h, const void *d_pat, const void *d_off, void *d_sp, void *d_ep, uint64_t k,
                         hipStream_t st, uint32_t fixed_len, uint64_t pack_cap, uint32_t flags) {
  if (pack_cap != ~0ull) {       // the count of wide intervals (word k): zero before anything appends to the list
    const hipError_t e0 = hipMemsetAsync(static_cast<unsigned long long *>(d_sp) + k, 0, 8, st);
    if (e0 != hipSuccess) return e0;
  }
  if (!k) return hipSuccess;
  const PatOff po{d_off ? (const uint64_t *)d_off : (const uint64_t *)h->d_cf, d_off ? 0ull : (uint64_t)fixed_len};
  if (search_variant() == 1 || k > 0xFFFFFFF0ull) {
    hipError_t e1 = launch_search_v1(h, d_pat, po, d_sp, d_ep, k, st);
    if (e1 == hipSuccess && pack_cap != ~0ull) e1 = launch_pack_intervals(h, d_sp, d_ep, k, pack_cap, d_sp, st);
    return e1;
  }
  hipError_t e = hipSuccess;
#define CALL(W, L) e = launch_v4<W, L>(h, (const uint8_t *)d_pat, po, (uint64_t *)d_sp, (uint64_t *)d_ep, (uint32_t)k, st, pack_cap, flags)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  return e;
}

// The residency census of the k_search4 instantiation this handle's tables select now (see Residency above).
hipError_t search_calibrate(const Index *h, hipStream_t st) {
  if (search_variant() == 1) return hipSuccess;
  hipError_t e = hipSuccess;
  const PatOff po{(const uint64_t *)h->d_cf, 0ull};
#define CALL(W, L) e = launch_v4<W, L>(h, (const uint8_t *)h->d_bwt, po, nullptr, nullptr, 1u, st, ~0ull, 0u, true)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  return e;
}

}  // namespace fmx

#ifdef FMX_SEARCHLOG
extern "C" int fmx_debug_searchlog(void *out, size_t bytes) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fmx::g_searchlog), std::min(bytes, sizeof fmx::g_searchlog)) == hipSuccess ? 0 : -1;
}
#endif

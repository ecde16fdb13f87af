// fmx_frontier.hip -- K5: Glushkov SA-interval frontier expansion for a batch of regexes.
//
// Reference: ReTree._matchSA, re2/retree.scala:618-653.  There one priority queue of
// StatePoint(len, sp, ep, state) is popped serially; each pop is one getPrevRange; a non-empty
// range either emits SAResult (isLast) or pushes one StatePoint per entry of state.follows.
// Every frontier element is independent of the others, so the device keeps the whole batch's
// frontier in two HBM work queues (SoA) and expands it level by level (level = len):
//   - one frontier element per lane group (a quad in the one-hot layout, an octet in the bytes
//     layout), stepped with the same rank primitive as the literal search;
//   - survivors are compacted into the next queue: per-group push counts are prefix-summed
//     across the wave, one atomicAdd per wave reserves the slots, and the group's lanes write
//     the follows in parallel; results are compacted the same way with __ballot.
// The set of getPrevRange calls, and so the result multiset, equals the reference's whenever
// its maxBranching / maxIterations limits do not bind.
#include <fmx.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "fmx_device.h"
#include "fmx_host.h"
#include "fmx_nfa.h"
#include "fmx_regex.h"

namespace fmx {

constexpr int kFThreads = 256;

struct Queue {           // SoA frontier queue in HBM; every element of a level has len == level
  uint32_t *state;       // global CharNode id
  uint64_t *sp;
  uint64_t *ep;
};

struct FStat {           // per-lane partial sums for the frontier counters (fmx_device.h, slots 3..7)
  uint32_t reqs = 0, pushes = 0, emits = 0, reads = 0;
};

constexpr uint32_t kStageCap = 192;     // survivors a wave stages in LDS before reserving queue slots
constexpr uint32_t kStageSmall = 12;    // follows lists up to this length go through the stage (16 groups x 12 <= cap)

struct Stage {           // 20 bytes per entry: 4 waves x 192 entries + the symbol tables keep 8 workgroups per CU
  uint32_t state[kStageCap];
  uint64_t sp[kStageCap];
  uint64_t ep[kStageCap];
};

// The queues and the result buffer are cut into kSub slices with one tail counter each, every counter
// on its own 128-byte line: a single tail cannot take the appends of a whole level (same-address device
// atomics complete at ~100 per microsecond -- with one tail a 3.7 M-element level spent 250 us on 25 k
// appends, and every level paid ~60 us for the 6144 waves' final flush).  A wave appends to slice
// (wave + number of its earlier appends) % kSub, so slices stay balanced even when one wave produces
// everything; at the next level slice j is read by the waves with id % kSub == j.
constexpr uint32_t kSub = 64;
struct alignas(128) PaddedCount {
  unsigned long long v;
  unsigned long long pad[15];
};

struct FrontierCtl {     // device-resident counters
  // Level L reads count[L % 3], appends to count[(L+1) % 3] and clears count[(L+2) % 3] (its
  // predecessor's input), so a chain of level launches needs no host round trip in between.
  PaddedCount count[3][kSub];
  PaddedCount res_count[kSub];
  unsigned long long overflow;     // bit 0: queue, bit 1: results
  // The level the next grid launch works on is level_base + its launch number, so that a chain of launches
  // can be replayed as one hipGraph with fixed kernel arguments; levels >= max_level do nothing.
  uint32_t level_base;
  uint32_t max_level;
};

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
  // inclusive Hillis-Steele over the 64 lanes, then shift
  const uint32_t lane = __lane_id();
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= (uint32_t)d) x += y;
  }
  total = __shfl(x, 63, 64);
  return x - v;
}

// The work of one wave on one queue slice at one level: it takes the elements part*P + g + r*nparts*P
// (g = lane group, r = round) of slice `sub`, one element per group and round.  Rounds are
// software-pipelined: the queue entry of round r+2 and the state record of round r+1 are requested before
// round r's rank blocks, so a round waits for one memory latency (the rank blocks), not four in a row
// (entry -> record -> blocks -> follows).  Wave-level operations only: the caller may be the grid kernel
// (one slice per wave) or the single-workgroup tail kernel (several slices per wave, many levels).
// ALL = true (tail kernel): the elements are those of all slices, numbered through `s_prefix` (exclusive
// prefix sums of the slice counts, kSub + 1 entries in LDS), cur_count their total.
template <bool WIDE, uint32_t LAYOUT, bool ALL = false>
__device__ __forceinline__ void frontier_slice(const DevIndex &ix, const NfaTables &nfa, const Queue &cur, const Queue &nxt,
                                               uint32_t level, uint64_t sub_cap, fmx_result *__restrict__ res,
                                               uint64_t seg_cap, FrontierCtl *__restrict__ ctl, const uint64_t *s_cf,
                                               const uint16_t *s_slot, Stage &stg, uint32_t w, uint32_t sub,
                                               uint64_t part, uint64_t nparts, uint64_t cur_count, uint32_t &appends,
                                               uint32_t &stepped, FStat &fs, const uint64_t *s_prefix = nullptr) {
  constexpr int G = Lay<LAYOUT>::G;
  constexpr uint32_t P = 64 / G;             // elements per wave and round
  const LaneConst lc = lane_const<G>();
  const uint32_t t = lc.t;
  if (!ALL && cur_count > sub_cap) cur_count = sub_cap;
  if (cur_count == 0) return;                // wave-uniform
  const uint64_t in_off = (uint64_t)sub * sub_cap;
  const uint64_t ngrp = nparts * P;
  const uint64_t first = part * P + (threadIdx.x & 63u) / G;
  PaddedCount *next_count = ctl->count[(level + 1) % 3];
  uint32_t staged = 0;                       // wave-uniform
  auto flush = [&]() {
    if (!staged) return;
    __builtin_amdgcn_wave_barrier();
    const uint32_t so = (w + appends++) % kSub;
    unsigned long long base = 0;
    if (__lane_id() == 0) base = atomicAdd(&next_count[so].v, (unsigned long long)staged);
    base = __shfl(base, 0, 64);
    const uint64_t out_off = (uint64_t)so * sub_cap;
    for (uint32_t i = __lane_id(); i < staged; i += 64) {
      const unsigned long long at = base + i;
      if (at < sub_cap) {
        nxt.state[out_off + at] = stg.state[i];
        nxt.sp[out_off + at] = stg.sp[i];
        nxt.ep[out_off + at] = stg.ep[i];
      } else {
        atomicOr(&ctl->overflow, 1ull);
      }
    }
    __builtin_amdgcn_wave_barrier();
    staged = 0;
  };
  struct Entry { uint32_t state; uint64_t sp, ep; };
  auto load_entry = [&](uint64_t q) {        // out-of-range rounds read element 0 (valid, unused)
    uint64_t i = q < cur_count ? q : 0;
    if (ALL) {                               // the slice that holds logical element i: last s with prefix[s] <= i
      uint32_t sl = 0;
#pragma unroll
      for (uint32_t step = kSub / 2; step; step >>= 1)
        if (s_prefix[sl + step] <= i) sl += step;
      i = (uint64_t)sl * sub_cap + (i - s_prefix[sl]);
    } else {
      i += in_off;
    }
    Entry e;
    e.state = cur.state[i]; e.sp = cur.sp[i]; e.ep = cur.ep[i];
    return e;
  };
  auto load_rec = [&](uint32_t state) {
    const uint4 *p = reinterpret_cast<const uint4 *>(nfa.st + state);
    const uint4 a = p[0], b = p[1];
    StateRec r;
    r.fol_off = a.x; r.fol_cnt = a.y; r.regex = a.z; r.c_emit = a.w;
    r.f[0] = b.x; r.f[1] = b.y; r.f[2] = b.z; r.f[3] = b.w;
    return r;
  };
  // all groups of a wave run the same number of rounds so that the wave-wide scans stay convergent
  const uint64_t rounds = (cur_count + ngrp - 1) / ngrp;
  Entry e0 = load_entry(first), e1 = load_entry(first + ngrp);
  StateRec rec0 = load_rec(e0.state);
  for (uint64_t rd = 0; rd < rounds; rd++) {
    const uint64_t q = first + rd * ngrp;
    const bool have = q < cur_count;
    const Entry e2 = load_entry(q + 2 * ngrp);          // prefetch: entry two rounds ahead,
    const StateRec rec1 = load_rec(e1.state);            // record one round ahead
    uint32_t nf = 0, f0 = 0, rgx = 0;
    uint64_t sp = e0.sp, ep = e0.ep;
    bool emit = false;
    if (have) {
      const uint32_t c = rec0.c_emit & 0xFFu;
      rgx = rec0.regex;
      const uint16_t slot = s_slot[c];
      const uint64_t cfc = s_cf[c];
      if (level == 0) {       // every level-0 element is (0, n): rank(c, 0) = 0, rank(c, n) = the symbol's count
        sp = cfc;
        ep = (c == 255u) ? ix.n : s_cf[c + 1];
        if (slot == kSlotNone) ep = sp;
        else if (slot == kSlotEof) ep = sp + 1;
      } else {
        fs.reqs += backward_step<WIDE, LAYOUT>(ix, c, slot, cfc, lc, sp, ep);
      }
      stepped++;
      fs.reads++;
      if (sp < ep) {                                   // Some((sp1,ep1)), retree.scala:634
        // Glushkov tables: an isLast state emits and has no follows here (:636-641); Thompson / DFA
        // tables may both emit and push (re2.scala:639-649, dfa.scala:270-282)
        emit = (rec0.c_emit >> 8) != 0;
        f0 = rec0.fol_off;
        nf = rec0.fol_cnt;
        fs.pushes += nf;
        fs.emits += emit ? 1u : 0u;
      }
    }
    // ---- compaction.  Results: ballot + one atomic per wave (they are few).  Pushes: a single
    // queue-tail counter cannot take one atomic per wave and round (same-address device atomics run
    // at ~100 per microsecond), so each wave stages its survivors in LDS and reserves queue slots only
    // when the stage fills: one atomic per ~150 elements and coalesced queue writes.
    const bool lead = t == 0;
    const uint32_t lane = __lane_id();
    const unsigned long long em = __builtin_amdgcn_ballot_w64(lead && emit);
    if (em) {
      const uint32_t so = (w + appends++) % kSub;
      unsigned long long rbase = 0;
      if (lane == 0) rbase = atomicAdd(&ctl->res_count[so].v, (unsigned long long)__builtin_popcountll(em));
      rbase = __shfl(rbase, 0, 64);
      if (lead && emit) {
        const unsigned long long at = rbase + __builtin_popcountll(em & ((1ull << lane) - 1ull));
        if (at < seg_cap) {
          fmx_result r;
          r.regex = rgx;
          r.len = level + 1;
          r.sp = sp;
          r.ep = ep;
          res[(uint64_t)so * seg_cap + at] = r;
        } else {
          atomicOr(&ctl->overflow, 2ull);
        }
      }
    }
    const uint32_t nsmall = nf <= kStageSmall ? nf : 0u;
    uint32_t small_total = 0;
    const uint32_t small_off = wave_excl_scan(lead ? nsmall : 0u, small_total);
    if (staged + small_total > kStageCap) flush();
    if (small_total) {
      const uint32_t my_off = staged + __shfl(small_off, lane & ~(uint32_t)(G - 1), 64);
      for (uint32_t j = t; j < nsmall; j += G) {
        // the first kInlineFollows follows ride in the state record
        uint32_t fs;
        if (j < kInlineFollows) fs = j < 2 ? (j == 0 ? rec0.f[0] : rec0.f[1]) : (j == 2 ? rec0.f[2] : rec0.f[3]);
        else fs = nfa.fol[f0 + j];
        stg.state[my_off + j] = fs;
        stg.sp[my_off + j] = sp;
        stg.ep[my_off + j] = ep;
      }
      staged += small_total;
    }
    // long follows lists (a '.' has 253) go straight to the queue
    if (__builtin_amdgcn_ballot_w64(nf > kStageSmall)) {
      const uint32_t nlarge = nf > kStageSmall ? nf : 0u;
      uint32_t large_total = 0;
      const uint32_t large_off = wave_excl_scan(lead ? nlarge : 0u, large_total);
      const uint32_t so = (w + appends++) % kSub;
      const uint64_t out_off = (uint64_t)so * sub_cap;
      unsigned long long qbase = 0;
      if (lane == 0) qbase = atomicAdd(&next_count[so].v, (unsigned long long)large_total);
      qbase = __shfl(qbase, 0, 64);
      const uint32_t my_off = __shfl(large_off, lane & ~(uint32_t)(G - 1), 64);
      for (uint32_t j = t; j < nlarge; j += G) {
        const unsigned long long at = qbase + my_off + j;
        if (at < sub_cap) {
          nxt.state[out_off + at] = nfa.fol[f0 + j];
          nxt.sp[out_off + at] = sp;
          nxt.ep[out_off + at] = ep;
        } else {
          atomicOr(&ctl->overflow, 1ull);
        }
      }
    }
    e0 = e1;
    e1 = e2;
    rec0 = rec1;
  }
  flush();
}

// One level on the whole grid: wave w reads slice w % kSub together with the other waves of that class.
// `j` is the launch's number in its chain: it works on level ctl->level_base + j.
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kFThreads) void k_frontier(DevIndex ix, NfaTables nfa, Queue qa, Queue qb, uint32_t j,
                                                         uint64_t sub_cap, fmx_result *__restrict__ res,
                                                         uint64_t seg_cap, FrontierCtl *__restrict__ ctl,
                                                         unsigned long long *__restrict__ counters) {
  // After a queue overflow the appended count exceeds what was stored: later levels of the chain
  // must not run (they would read past the queue); the host reports FMX_ERR_OVERFLOW.
  if (ctl->overflow & 1ull) return;
  const uint32_t level = ctl->level_base + j;
  if (level >= ctl->max_level) return;
  const Queue &cur = (level & 1u) ? qb : qa;
  const Queue &nxt = (level & 1u) ? qa : qb;
  if (blockIdx.x == 0 && threadIdx.x < kSub) ctl->count[(level + 2) % 3][threadIdx.x].v = 0;
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  __shared__ Stage s_stage[kFThreads / 64];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  const uint32_t w = (blockIdx.x * kFThreads + threadIdx.x) >> 6;      // this wave
  const uint32_t nw = gridDim.x * (kFThreads / 64);
  const uint32_t sub = w % kSub;             // the slice this wave reads
  const uint32_t class_waves = (nw - sub + kSub - 1) / kSub;
  uint32_t appends = 0, stepped = 0;
  FStat fs;
  frontier_slice<WIDE, LAYOUT>(ix, nfa, cur, nxt, level, sub_cap, res, seg_cap, ctl, s_cf, s_slot,
                               s_stage[threadIdx.x >> 6], w, sub, w / kSub, class_waves, ctl->count[level % 3][sub].v,
                               appends, stepped, fs);
  const uint32_t t = threadIdx.x & (Lay<LAYOUT>::G - 1);
  counters_add(counters, t == 0 ? 2ull * stepped : 0ull, t == 0 ? stepped : 0u, 0);
  counters_add_frontier(counters, t == 0 ? fs.reqs : 0u, t == 0 ? fs.pushes : 0u, t == 0 ? fs.emits : 0u,
                        t == 0 ? stepped : 0u, t == 0 ? fs.reads : 0u);
}

// Closes a chain of grid launches: the next chain starts `by` levels further.
__global__ void k_level_advance(FrontierCtl *__restrict__ ctl, uint32_t by) {
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->level_base += by;
}

// The long tail of a match -- levels with a handful of elements -- is bound by launches and host looks,
// not by work.  One persistent workgroup runs those levels back to back: its 16 waves share the 64 slices,
// a workgroup barrier (with agent-scope release/acquire fences: the next level reads what other waves of
// this workgroup appended) separates the levels, and it hands back to the grid kernel when the frontier
// dies, reaches max_level, or outgrows what one workgroup should handle.
struct TailState {
  uint32_t level;      // first level not processed
  uint32_t reason;     // 0 frontier empty, 1 max_level reached, 2 frontier outgrew the tail kernel, 3 queue overflow
  uint32_t pending;    // elements still alive in the kernel's LDS queue at max_level (the global counts read 0 then)
  uint32_t pad;
};
constexpr int kTailThreads = 1024;
constexpr uint64_t kTailMax = 8192;          // elements per level one workgroup keeps; it is entered below half of it

template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kTailThreads) void k_frontier_tail(DevIndex ix, NfaTables nfa, Queue qa, Queue qb,
                                                                 uint32_t level0, uint32_t max_level, uint64_t sub_cap,
                                                                 fmx_result *__restrict__ res, uint64_t seg_cap,
                                                                 FrontierCtl *__restrict__ ctl,
                                                                 unsigned long long *__restrict__ counters,
                                                                 TailState *__restrict__ ts) {
  constexpr int G = Lay<LAYOUT>::G;
  constexpr uint32_t kTiny = kTailThreads / G;            // elements the LDS mode holds: one per lane group
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  __shared__ Stage s_stage[kTailThreads / 64];
  __shared__ uint32_t s_verdict, s_tcnt, s_push;
  __shared__ uint64_t s_prefix[kSub + 1];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  static_assert(kSub == 64, "one slice counter per lane below");
  // LDS mode keeps the frontier in two small queues that live in the (then unused) staging area
  static_assert(sizeof(s_stage) >= 2 * kTiny * 20 + 64, "the LDS queues must fit in the staging area");
  uint8_t *raw = reinterpret_cast<uint8_t *>(&s_stage[0]);
  uint64_t *tq_sp[2] = {reinterpret_cast<uint64_t *>(raw), reinterpret_cast<uint64_t *>(raw) + 2 * kTiny};
  uint64_t *tq_ep[2] = {tq_sp[0] + kTiny, tq_sp[1] + kTiny};
  uint32_t *tq_state[2] = {reinterpret_cast<uint32_t *>(tq_sp[1] + 2 * kTiny), reinterpret_cast<uint32_t *>(tq_sp[1] + 2 * kTiny) + kTiny};
  const LaneConst lc = lane_const<G>();
  const uint32_t t = lc.t;
  const uint32_t w = threadIdx.x >> 6;
  constexpr uint32_t nw = kTailThreads / 64;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t grp = threadIdx.x / G;                   // this lane group's number in the workgroup
  uint32_t appends = 0, stepped = 0;
  FStat fs;
  uint32_t level = level0, reason = 1, pending = 0;
  while (level < max_level) {
    // lane j reads slice j's count (coherent load: other waves' atomics produced it).  Wave 0 decides for the
    // whole workgroup -- the overflow flag can change while a level runs, and every thread must take the
    // same way out of this loop (there is a barrier at its end)
    const unsigned long long mine = __hip_atomic_load(&ctl->count[level % 3][lane].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w == 0) {
      const unsigned long long clamped = mine < sub_cap ? mine : sub_cap;
      const unsigned long long total = wave_sum(clamped);
      unsigned long long incl = clamped;                   // inclusive scan over the 64 lanes
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += y;
      }
      s_prefix[lane] = incl - clamped;
      if (lane == 63) s_prefix[kSub] = incl;
      const unsigned long long ovf = __hip_atomic_load(&ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0)
        s_verdict = (ovf & 1ull) ? 3u : (total == 0 ? 0u : (total > kTailMax ? 2u : (total <= kTiny ? 5u : 4u)));
    }
    __syncthreads();
    const uint32_t verdict = s_verdict;
    if (verdict < 4u) { reason = verdict; break; }
    const Queue &cur = (level & 1u) ? qb : qa;
    const Queue &nxt = (level & 1u) ? qa : qb;
    if (verdict == 4u) {
      // ---- a level through the global queues
      if (threadIdx.x < kSub) __hip_atomic_store(&ctl->count[(level + 2) % 3][threadIdx.x].v, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      frontier_slice<WIDE, LAYOUT, true>(ix, nfa, cur, nxt, level, sub_cap, res, seg_cap, ctl, s_cf, s_slot, s_stage[w], w, 0,
                                         w, nw, s_prefix[kSub], appends, stepped, fs, s_prefix);
      // level boundary.  The appends of this level were made by waves of this workgroup: draining the stores
      // (workgroup-scope release) makes them reach L2; the acquire invalidates this CU's L1, which may still
      // hold lines of the queue buffer from two levels ago.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      level++;
      continue;
    }
    // ---- LDS mode: at most one element per lane group.  The frontier moves into LDS and stays there, level
    // after level, without queue traffic, counters or fences (a level is then: state record, rank blocks,
    // three barriers), until it dies, reaches max_level, or a level's survivors no longer fit.
    {
      const uint32_t total = (uint32_t)s_prefix[kSub];
      if (grp < total && t == 0) {
        uint32_t sl = 0;
#pragma unroll
        for (uint32_t step = kSub / 2; step; step >>= 1)
          if (s_prefix[sl + step] <= grp) sl += step;
        const uint64_t i = (uint64_t)sl * sub_cap + (grp - s_prefix[sl]);
        tq_state[0][grp] = cur.state[i];
        tq_sp[0][grp] = cur.sp[i];
        tq_ep[0][grp] = cur.ep[i];
      }
      // the global counts of this level are consumed; nothing is appended while the frontier lives in LDS
      if (threadIdx.x < kSub)
        for (int sct = 0; sct < 3; sct++)
          __hip_atomic_store(&ctl->count[sct][threadIdx.x].v, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0) { s_tcnt = total; s_push = 0; }
    }
    __syncthreads();
    uint32_t cq = 0;                                       // which LDS queue holds the current level
    bool spilled = false;
    for (;;) {
      const uint32_t tc = s_tcnt;
      if (tc == 0) { reason = 0; break; }
      if (level >= max_level) { reason = 1; pending = tc; break; }
      const bool have = grp < tc;
      uint32_t nf = 0, f0 = 0, rgx = 0;
      uint64_t sp = 0, ep = 0;
      uint32_t inl[kInlineFollows] = {0, 0, 0, 0};
      bool emit = false;
      if (have) {
        const uint32_t state = tq_state[cq][grp];
        sp = tq_sp[cq][grp];
        ep = tq_ep[cq][grp];
        const uint4 *p = reinterpret_cast<const uint4 *>(nfa.st + state);
        const uint4 ra = p[0], rb = p[1];
        const uint32_t c = ra.w & 0xFFu;
        rgx = ra.z;
        const uint16_t slot = s_slot[c];
        const uint64_t cfc = s_cf[c];
        if (level == 0) {
          sp = cfc;
          ep = (c == 255u) ? ix.n : s_cf[c + 1];
          if (slot == kSlotNone) ep = sp;
          else if (slot == kSlotEof) ep = sp + 1;
        } else {
          fs.reqs += backward_step<WIDE, LAYOUT>(ix, c, slot, cfc, lc, sp, ep);
        }
        stepped++;
        if (sp < ep) {
          emit = (ra.w >> 8) != 0;
          f0 = ra.x;
          nf = ra.y;
          fs.emits += emit ? 1u : 0u;
          inl[0] = rb.x; inl[1] = rb.y; inl[2] = rb.z; inl[3] = rb.w;
        }
      }
      const bool lead = t == 0;
      const unsigned long long em = __builtin_amdgcn_ballot_w64(lead && emit);
      if (em) {
        const uint32_t so = (w + appends++) % kSub;
        unsigned long long rbase = 0;
        if (lane == 0) rbase = atomicAdd(&ctl->res_count[so].v, (unsigned long long)__builtin_popcountll(em));
        rbase = __shfl(rbase, 0, 64);
        if (lead && emit) {
          const unsigned long long at = rbase + __builtin_popcountll(em & ((1ull << lane) - 1ull));
          if (at < seg_cap) {
            fmx_result r;
            r.regex = rgx; r.len = level + 1; r.sp = sp; r.ep = ep;
            res[(uint64_t)so * seg_cap + at] = r;
          } else {
            atomicOr(&ctl->overflow, 2ull);
          }
        }
      }
      // reserve room in the next LDS queue
      uint32_t off = 0;
      if (lead && nf) off = atomicAdd(&s_push, nf);
      off = __shfl(off, (int)(lane & ~(uint32_t)(G - 1)), 64);
      __syncthreads();
      const uint32_t tp = s_push;
      if (tp <= kTiny) {
        for (uint32_t j = t; j < nf; j += G) {
          const uint32_t fs = j < kInlineFollows ? (j < 2 ? (j == 0 ? inl[0] : inl[1]) : (j == 2 ? inl[2] : inl[3])) : nfa.fol[f0 + j];
          tq_state[cq ^ 1][off + j] = fs;
          tq_sp[cq ^ 1][off + j] = sp;
          tq_ep[cq ^ 1][off + j] = ep;
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_tcnt = tp; s_push = 0; }
        cq ^= 1;
        level++;
        __syncthreads();
        continue;
      }
      // ---- the survivors no longer fit: append them to the global queue of the next level and leave LDS mode
      {
        uint32_t wave_total = 0;
        const uint32_t woff = wave_excl_scan(lead ? nf : 0u, wave_total);
        if (wave_total) {
          const uint32_t so = (w + appends++) % kSub;
          const uint64_t out_off = (uint64_t)so * sub_cap;
          unsigned long long qbase = 0;
          if (lane == 0) qbase = atomicAdd(&ctl->count[(level + 1) % 3][so].v, (unsigned long long)wave_total);
          qbase = __shfl(qbase, 0, 64);
          const uint32_t my_off = __shfl(woff, (int)(lane & ~(uint32_t)(G - 1)), 64);
          const Queue &nq = (level & 1u) ? qa : qb;
          for (uint32_t j = t; j < nf; j += G) {
            const unsigned long long at = qbase + my_off + j;
            if (at < sub_cap) {
              nq.state[out_off + at] = j < kInlineFollows ? (j < 2 ? (j == 0 ? inl[0] : inl[1]) : (j == 2 ? inl[2] : inl[3])) : nfa.fol[f0 + j];
              nq.sp[out_off + at] = sp;
              nq.ep[out_off + at] = ep;
            } else {
              atomicOr(&ctl->overflow, 1ull);
            }
          }
        }
        level++;
        spilled = true;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        break;
      }
    }
    if (!spilled) break;              // died or reached max_level inside LDS mode
  }
  if (threadIdx.x == 0) { ts->level = level; ts->reason = reason; ts->pending = pending; ctl->level_base = level; }
  counters_add(counters, t == 0 ? 2ull * stepped : 0ull, t == 0 ? stepped : 0u, 0);
  counters_add_frontier(counters, t == 0 ? fs.reqs : 0u, t == 0 ? fs.pushes : 0u, t == 0 ? fs.emits : 0u,
                        t == 0 ? stepped : 0u, t == 0 ? fs.reads : 0u);
}

// result groups the device leaves to the host (k_res_sort)
constexpr uint32_t kSmallGroup = 12;
constexpr uint32_t kBigMax = 16384;
struct BigGroups {
  uint32_t n;
  uint32_t pad;
  uint32_t ent[2 * kBigMax];     // (first result, count) of each group left unsorted
};

namespace {

struct DevMem {
  std::vector<void *> ptrs;
  ~DevMem() { for (void *p : ptrs) (void)hipFree(p); }
  template <class T>
  hipError_t alloc(T **out, size_t count) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
    if (e == hipSuccess) { ptrs.push_back(p); *out = (T *)p; }
    return e;
  }
};

}  // namespace

#define HIP_TRY(call, what)                            \
  do {                                                 \
    hipError_t e__ = (call);                           \
    if (e__ != hipSuccess) return hip_fail(e__, what); \
  } while (0)

// A batch of compiled regexes made resident on one device: concatenated Glushkov tables plus
// the level-0 frontier (root.firsts x (0, 0, n), retree.scala:576).  Reusable across calls.
struct RegexBatch {
  int device = 0;
  size_t k = 0;
  uint64_t n_index = 0;
  uint64_t index_serial = 0;           // the fmx_index this batch was made for (Index::serial): its pointers are inside the
                                       // captured level chain, so no other handle may match against the batch
  size_t n_first = 0;
  std::vector<uint32_t> start_final;   // DFA engines whose start state is final: result (len 0, 0, n)
  DevMem mem;
  // scratch reused across matches of this batch (one match at a time per batch object)
  std::unique_ptr<DevMem> scratch;
  Queue qa{}, qb{};
  fmx_result *d_res = nullptr;        // packed results
  fmx_result *d_res_seg = nullptr;    // kSub result slices the levels append to
  FrontierCtl *d_ctl = nullptr;
  TailState *d_tail = nullptr;
  uint32_t *d_rcnt = nullptr, *d_rstart = nullptr, *d_rfill = nullptr;   // per-regex result counts / offsets
  uint32_t *d_rpart = nullptr;         // chunk totals of the offsets' scan
  BigGroups *d_big = nullptr;
  FrontierCtl *h_ctl = nullptr;        // pinned host copy the chain's last node fills
  hipGraphExec_t chain_exec = nullptr; // one chain of grid levels + advance + counter copy, captured once
  uint32_t chain_len = 0;
  uint32_t matches = 0;                // the chain is captured from a batch's second match on (a one-shot batch
                                       // would pay the capture and never replay it)
  ~RegexBatch() {
    if (chain_exec) (void)hipGraphExecDestroy(chain_exec);
    if (h_ctl) (void)hipHostFree(h_ctl);
  }
  uint64_t qcap = 0;
  size_t rcap = 0;
  NfaTables nfa{};
  uint32_t *d_first_state = nullptr;
  // reference-order mode (ReTree batches only): heap keys, per-regex firsts, the largest fan-out
  uint32_t *d_st_num = nullptr, *d_first_off = nullptr;
  uint32_t max_fanout = 1;
  bool all_retree = true;
};

int regex_batch_create(const Index *h, const Regex *const *res, size_t k, RegexBatch **out) {
  // sizes first, then one pass that fills pre-sized arrays (100 k regexes: 1.3 M states, 2 M follows)
  size_t n_states = 0, n_fol = 0, n_first = 0;
  bool all_retree = true;
  for (size_t r = 0; r < k; r++) {
    const Regex &re = *res[r];
    n_states += re.st_c.size();
    n_first += re.firsts.size();
    all_retree = all_retree && re.engine == 0;
    for (size_t s = 0; s < re.st_c.size(); s++)
      if (!(re.last_stops && re.st_last[s])) n_fol += (size_t)(re.fol_off[s + 1] - re.fol_off[s]);
  }
  std::vector<StateRec> recs(n_states);
  std::vector<uint32_t> fol(n_fol), q_state(n_first), st_num(n_states), first_off(k + 1, 0), start_final;
  uint32_t max_fanout = 1;
  size_t base = 0, fo = 0, qo = 0;
  for (size_t r = 0; r < k; r++) {
    const Regex &re = *res[r];
    for (size_t s = 0; s < re.st_c.size(); s++) {
      StateRec &rec = recs[base + s];
      rec.fol_off = (uint32_t)fo;
      // ReTree: `if (q.state.isLast) ret ::= ... else pqFront ++= follows` -- last states do not expand
      if (!(re.last_stops && re.st_last[s]))
        for (int32_t j = re.fol_off[s]; j < re.fol_off[s + 1]; j++) fol[fo++] = (uint32_t)base + (uint32_t)re.fol[j];
      rec.fol_cnt = (uint32_t)fo - rec.fol_off;
      rec.regex = (uint32_t)r;
      rec.c_emit = (uint32_t)re.st_c[s] | ((uint32_t)(re.st_last[s] ? 1 : 0) << 8);
      for (uint32_t j = 0; j < kInlineFollows; j++) rec.f[j] = j < rec.fol_cnt ? fol[rec.fol_off + j] : 0u;
      st_num[base + s] = (uint32_t)re.st_num[s];
      max_fanout = std::max(max_fanout, rec.fol_cnt);
    }
    for (int32_t f : re.firsts) q_state[qo++] = (uint32_t)base + (uint32_t)f;
    first_off[r + 1] = (uint32_t)qo;
    max_fanout = std::max<uint32_t>(max_fanout, (uint32_t)re.firsts.size());
    if (re.start_is_final) start_final.push_back((uint32_t)r);
    base += re.st_c.size();
  }
  HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
  std::unique_ptr<RegexBatch> b(new RegexBatch());
  b->device = h->device;
  b->k = k;
  b->n_index = h->n;
  b->index_serial = h->serial;
  b->n_first = q_state.size();
  b->start_final = start_final;
  b->max_fanout = max_fanout;
  b->all_retree = all_retree;
  StateRec *d_st = nullptr;
  uint32_t *d_fol = nullptr;
  HIP_TRY(b->mem.alloc(&d_st, recs.size()), "hipMalloc");
  HIP_TRY(b->mem.alloc(&d_fol, fol.size()), "hipMalloc");
  HIP_TRY(b->mem.alloc(&b->d_first_state, q_state.size()), "hipMalloc");
  if (!recs.empty()) HIP_TRY(hipMemcpy(d_st, recs.data(), recs.size() * sizeof(StateRec), hipMemcpyHostToDevice), "H2D");
  if (!fol.empty()) HIP_TRY(hipMemcpy(d_fol, fol.data(), fol.size() * 4, hipMemcpyHostToDevice), "H2D");
  if (!q_state.empty()) HIP_TRY(hipMemcpy(b->d_first_state, q_state.data(), q_state.size() * 4, hipMemcpyHostToDevice), "H2D");
  if (all_retree) {
    HIP_TRY(b->mem.alloc(&b->d_st_num, st_num.size()), "hipMalloc");
    HIP_TRY(b->mem.alloc(&b->d_first_off, first_off.size()), "hipMalloc");
    if (!st_num.empty()) HIP_TRY(hipMemcpy(b->d_st_num, st_num.data(), st_num.size() * 4, hipMemcpyHostToDevice), "H2D");
    HIP_TRY(hipMemcpy(b->d_first_off, first_off.data(), first_off.size() * 4, hipMemcpyHostToDevice), "H2D");
  }
  b->nfa = NfaTables{d_st, d_fol};
  *out = b.release();
  return FMX_OK;
}

// Level-0 queue: states = firsts, sp = 0, ep = n, dealt round-robin over the slices.
__global__ void k_frontier_init(Queue q, const uint32_t *__restrict__ first_state, uint64_t count, uint64_t n,
                                uint64_t sub_cap, uint32_t max_level, FrontierCtl *__restrict__ ctl) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kSub) {
    ctl->count[0][i].v = count > i ? (count - i + kSub - 1) / kSub : 0;
    ctl->count[1][i].v = 0;
    ctl->count[2][i].v = 0;
    ctl->res_count[i].v = 0;
  }
  if (i == 0) { ctl->overflow = 0; ctl->level_base = 0; ctl->max_level = max_level; }
  if (i < count) {
    const uint64_t at = (i % kSub) * sub_cap + i / kSub;
    q.state[at] = first_state[i]; q.sp[at] = 0; q.ep[at] = n;
  }
}

// Results leave the device grouped by regex: count per regex, scan, scatter (three small kernels; ordering
// 50 k results on the host cost half as much as all the levels together).
__global__ __launch_bounds__(256) void k_res_count(const fmx_result *__restrict__ seg, uint64_t seg_cap,
                                                    const FrontierCtl *__restrict__ ctl, uint32_t *__restrict__ rcnt) {
  const uint32_t sl = blockIdx.y;
  const uint64_t mine = min((uint64_t)ctl->res_count[sl].v, seg_cap);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < mine; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&rcnt[seg[(uint64_t)sl * seg_cap + i].regex], 1u);
}

// Exclusive prefix sums of cnt[0..k] into start[0..k] in three small parallel launches: each workgroup scans
// its chunk of 1024 counts (start = sums inside the chunk, part[chunk] = the chunk's total), one workgroup
// scans the chunk totals, and every workgroup adds its chunk's offset.
constexpr uint32_t kScanChunk = 1024;
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t *s_wave /* [16] */, uint32_t &total) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t wtot = 0;
  const uint32_t ex = wave_excl_scan(v, wtot);          // inside the wave
  if (lane == 0) s_wave[wv] = wtot;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (uint32_t j = 0; j < 16; j++) {                   // 16 wave totals: every thread adds them up itself
    const uint32_t x = s_wave[j];
    before += j < wv ? x : 0u;
    all += x;
  }
  __syncthreads();
  total = all;
  return before + ex;
}

__global__ __launch_bounds__(kScanChunk) void k_res_scan_chunks(const uint32_t *__restrict__ cnt, uint32_t n,
                                                                uint32_t *__restrict__ start, uint32_t *__restrict__ part) {
  __shared__ uint32_t s_wave[16];
  const uint32_t i = blockIdx.x * kScanChunk + threadIdx.x;
  uint32_t total = 0;
  const uint32_t ex = block_excl_scan_1024(i < n ? cnt[i] : 0u, s_wave, total);
  if (i < n) start[i] = ex;
  if (threadIdx.x == 0) part[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanChunk) void k_res_scan_parts(uint32_t *__restrict__ part, uint32_t nparts) {
  __shared__ uint32_t s_wave[16];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nparts; base += kScanChunk) {        // one trip up to a million regexes
    const uint32_t i = base + threadIdx.x;
    uint32_t total = 0;
    const uint32_t ex = block_excl_scan_1024(i < nparts ? part[i] : 0u, s_wave, total);
    if (i < nparts) part[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(kScanChunk) void k_res_scan_add(uint32_t *__restrict__ start, uint32_t n,
                                                             const uint32_t *__restrict__ part) {
  const uint32_t i = blockIdx.x * kScanChunk + threadIdx.x;
  if (i < n) start[i] += part[blockIdx.x];
}

__global__ __launch_bounds__(256) void k_res_scatter(const fmx_result *__restrict__ seg, uint64_t seg_cap,
                                                      const FrontierCtl *__restrict__ ctl,
                                                      const uint32_t *__restrict__ start, uint32_t *__restrict__ fill,
                                                      fmx_result *__restrict__ out, uint64_t out_cap) {
  const uint32_t sl = blockIdx.y;
  const uint64_t mine = min((uint64_t)ctl->res_count[sl].v, seg_cap);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < mine; i += (uint64_t)gridDim.x * blockDim.x) {
    const fmx_result r = seg[(uint64_t)sl * seg_cap + i];
    const uint64_t at = (uint64_t)start[r.regex] + atomicAdd(&fill[r.regex], 1u);
    if (at < out_cap) out[at] = r;
  }
}

// Orders each regex's group by (len, sp, ep): one thread per regex, insertion sort for the usual handful of
// results; larger groups are listed for the host.
__global__ __launch_bounds__(256) void k_res_sort(fmx_result *__restrict__ out, const uint32_t *__restrict__ start, uint32_t k,
                                                   BigGroups *__restrict__ big) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= k) return;
  const uint32_t lo = start[r], m = start[r + 1] - lo;
  if (m < 2) return;
  if (m > kSmallGroup) {
    const uint32_t at = atomicAdd(&big->n, 1u);
    if (at < kBigMax) { big->ent[2 * at] = lo; big->ent[2 * at + 1] = m; }
    return;
  }
  auto less = [](const fmx_result &a, const fmx_result &b) {
    if (a.len != b.len) return a.len < b.len;
    if (a.sp != b.sp) return a.sp < b.sp;
    return a.ep < b.ep;
  };
  for (uint32_t i = 1; i < m; i++) {
    const fmx_result x = out[lo + i];
    uint32_t j = i;
    while (j > 0 && less(x, out[lo + j - 1])) { out[lo + j] = out[lo + j - 1]; j--; }
    out[lo + j] = x;
  }
}

int regex_batch_match(const Index *h, RegexBatch *b, const fmx_limits *lim, fmx_result *out, size_t cap,
                      size_t *n_out, uint32_t *per_regex_count) {
  static const bool trace = getenv("FMX_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (trace) fprintf(stderr, "[fmx] regex_batch_match %-18s +%.3f ms\n", what,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  const uint32_t max_steps = (lim && lim->max_steps) ? lim->max_steps : 4096u;
  const uint64_t qcap = (lim && lim->max_frontier) ? lim->max_frontier : (1ull << 22);
  if (b->index_serial != h->serial) { set_error("regex batch was prepared for another index"); return FMX_ERR_ARG; }
  if (per_regex_count) std::fill(per_regex_count, per_regex_count + b->k, 0u);
  *n_out = 0;
  if (b->n_first == 0 && b->start_final.empty()) return FMX_OK;
  if (b->n_first > qcap) { set_error("initial frontier exceeds max_frontier"); return FMX_ERR_OVERFLOW; }
  HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
  // slices: each holds its share of max_frontier plus a quarter of headroom (appends rotate over the slices,
  // so they fill evenly, not exactly); the result segments get 4x their share
  const uint64_t sub_cap = (qcap + kSub - 1) / kSub + qcap / (4 * kSub) + 1024;
  if (!b->scratch || b->qcap != qcap || b->rcap < (cap ? cap : 1)) {
    b->scratch.reset(new DevMem());
    b->qcap = 0;
    if (b->chain_exec) { (void)hipGraphExecDestroy(b->chain_exec); b->chain_exec = nullptr; }   // it holds the old pointers
    if (!b->h_ctl) HIP_TRY(hipHostMalloc((void **)&b->h_ctl, sizeof(FrontierCtl), hipHostMallocDefault), "hipHostMalloc(ctl)");
    const uint64_t seg_cap = (uint64_t)(cap ? cap : 1) / 16 + 1024;
    for (Queue *q : {&b->qa, &b->qb}) {
      HIP_TRY(b->scratch->alloc(&q->state, kSub * sub_cap), "hipMalloc(queue)");
      HIP_TRY(b->scratch->alloc(&q->sp, kSub * sub_cap), "hipMalloc(queue)");
      HIP_TRY(b->scratch->alloc(&q->ep, kSub * sub_cap), "hipMalloc(queue)");
    }
    HIP_TRY(b->scratch->alloc(&b->d_res, cap ? cap : 1), "hipMalloc(results)");
    HIP_TRY(b->scratch->alloc(&b->d_res_seg, kSub * seg_cap), "hipMalloc(result slices)");
    HIP_TRY(b->scratch->alloc(&b->d_ctl, 1), "hipMalloc(ctl)");
    HIP_TRY(b->scratch->alloc(&b->d_tail, 1), "hipMalloc(tail state)");
    HIP_TRY(b->scratch->alloc(&b->d_rcnt, b->k + 1), "hipMalloc(result counts)");
    HIP_TRY(b->scratch->alloc(&b->d_rstart, b->k + 1), "hipMalloc(result offsets)");
    HIP_TRY(b->scratch->alloc(&b->d_rfill, b->k + 1), "hipMalloc(result fill)");
    HIP_TRY(b->scratch->alloc(&b->d_rpart, (b->k + 1) / kScanChunk + 2), "hipMalloc(scan parts)");
    HIP_TRY(b->scratch->alloc(&b->d_big, 1), "hipMalloc(big groups)");
    b->qcap = qcap;
    b->rcap = cap ? cap : 1;
  }
  // Slice capacity of the result buffer: a function of the ALLOCATED size, so that it stays what the captured
  // level chain was recorded with when a later call passes a smaller cap (the scratch is kept then).
  const uint64_t seg_cap = (uint64_t)b->rcap / 16 + 1024;
  const Queue qa = b->qa, qb = b->qb;
  fmx_result *d_res = b->d_res;
  fmx_result *d_res_seg = b->d_res_seg;
  FrontierCtl *d_ctl = b->d_ctl;
  CtxLease lease(h);                 // stream and events from the handle's pool
  if (!lease.c) return FMX_ERR_HIP;
  hipStream_t st = lease.c->stream;
  hipEvent_t e0 = lease.c->ev_a, e1 = lease.c->ev_b;

  mark("setup");
  HIP_TRY(hipEventRecord(e0, st), "hipEventRecord");
  k_frontier_init<<<(int)((std::max<uint64_t>(b->n_first, kSub) + 255) / 256), 256, 0, st>>>(qa, b->d_first_state, b->n_first, h->n, sub_cap, max_steps, d_ctl);
  HIP_TRY(hipGetLastError(), "k_frontier_init");
  // Levels are chained on the stream without host round trips; the host looks at the counters
  // every kChain levels.  A level with an empty queue returns at once.
  static const uint32_t kChain = getenv("FMX_FRONTIER_CHAIN") ? (uint32_t)std::max(1, atoi(getenv("FMX_FRONTIER_CHAIN"))) : 8u;   // levels between host looks
  const int grid_full = h->cu_count * 6;     // what stays resident at 80 vector registers per lane
  static const bool use_tail = !(getenv("FMX_FRONTIER_TAIL") && atoi(getenv("FMX_FRONTIER_TAIL")) == 0);   // A/B switch
  std::unique_ptr<FrontierCtl> ctl_host(new FrontierCtl());
  FrontierCtl &ctl = *ctl_host;
  uint64_t n_res = 0;
  uint32_t level = 0;
  uint64_t launches = 1;
  bool alive = true, truncated = false;
  // One chain = kChain grid levels + k_level_advance + the counters' copy to pinned host memory.  Its kernel
  // arguments do not change from chain to chain (the level comes from ctl->level_base), so it is captured
  // into a hipGraph once per batch and replayed: one launch call per chain instead of kChain + 2.
  static const bool use_graph = !(getenv("FMX_FRONTIER_GRAPH") && atoi(getenv("FMX_FRONTIER_GRAPH")) == 0);
  auto enqueue_chain = [&](hipStream_t s) -> hipError_t {
    for (uint32_t j = 0; j < kChain; j++) {
#define CALL(W, L) k_frontier<W, L><<<grid_full, kFThreads, 0, s>>>(h->dev, b->nfa, qa, qb, j, sub_cap, d_res_seg, seg_cap, d_ctl, h->d_counters)
      FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
    }
    k_level_advance<<<1, 1, 0, s>>>(d_ctl, kChain);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(b->h_ctl, d_ctl, sizeof(FrontierCtl), hipMemcpyDeviceToHost, s);
  };
  b->matches++;
  if (use_graph && b->matches >= 2 && (!b->chain_exec || b->chain_len != kChain)) {
    if (b->chain_exec) { (void)hipGraphExecDestroy(b->chain_exec); b->chain_exec = nullptr; }
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      const hipError_t e1 = enqueue_chain(st);
      const hipError_t e2 = hipStreamEndCapture(st, &g);
      e = e1 != hipSuccess ? e1 : e2;
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&b->chain_exec, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (e != hipSuccess) { (void)hipGetLastError(); b->chain_exec = nullptr; }    // fall back to plain launches
    b->chain_len = kChain;
  }
  // a small batch (a single regex, say) starts in the tail kernel: no grid levels, no look
  bool grid_first = !(use_tail && b->n_first <= kTailMax / 2 && max_steps > 0);
  while (alive) {
    uint64_t next_total = b->n_first;
    if (grid_first) {
      if (b->chain_exec) HIP_TRY(hipGraphLaunch(b->chain_exec, st), "hipGraphLaunch(level chain)");
      else HIP_TRY(enqueue_chain(st), "k_frontier chain");
      launches += kChain + 1;
      HIP_TRY(hipStreamSynchronize(st), "sync(levels)");
      std::memcpy(&ctl, b->h_ctl, sizeof ctl);
      level = std::min<uint64_t>((uint64_t)level + kChain, max_steps);
      if (ctl.overflow & 1ull) { set_error("frontier work queue overflow (raise fmx_limits.max_frontier)"); return FMX_ERR_OVERFLOW; }
      next_total = 0;
      n_res = 0;
      for (uint32_t j = 0; j < kSub; j++) { next_total += ctl.count[level % 3][j].v; n_res += ctl.res_count[j].v; }
      alive = next_total != 0;
      if (trace)
        fprintf(stderr, "[fmx] frontier level %u: next %llu, results %llu, overflow %llu\n", level,
                (unsigned long long)next_total, (unsigned long long)n_res, ctl.overflow);
    }
    grid_first = true;
    if (alive && level < max_steps && next_total <= kTailMax / 2 && use_tail) {
      // nearly empty frontier: one persistent workgroup runs the following levels without launches in between
      TailState tsh{};
#define CALL(W, L) k_frontier_tail<W, L><<<1, kTailThreads, 0, st>>>(h->dev, b->nfa, qa, qb, level, max_steps, sub_cap, d_res_seg, seg_cap, d_ctl, h->d_counters, b->d_tail)
      FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
      HIP_TRY(hipGetLastError(), "k_frontier_tail");
      launches++;
      HIP_TRY(hipMemcpyAsync(&tsh, b->d_tail, sizeof tsh, hipMemcpyDeviceToHost, st), "D2H(tail)");
      HIP_TRY(hipMemcpyAsync(&ctl, d_ctl, sizeof ctl, hipMemcpyDeviceToHost, st), "D2H(ctl)");
      HIP_TRY(hipStreamSynchronize(st), "sync(tail)");
      if ((ctl.overflow & 1ull) || tsh.reason == 3) { set_error("frontier work queue overflow (raise fmx_limits.max_frontier)"); return FMX_ERR_OVERFLOW; }
      level = tsh.level;
      next_total = 0;
      n_res = 0;
      for (uint32_t j = 0; j < kSub; j++) { next_total += ctl.count[level % 3][j].v; n_res += ctl.res_count[j].v; }
      alive = next_total != 0 || tsh.pending != 0;
      if (trace)
        fprintf(stderr, "[fmx] frontier tail kernel stopped at level %u (reason %u): next %llu, results %llu\n", level,
                tsh.reason, (unsigned long long)next_total, (unsigned long long)n_res);
    }
    if (alive && level >= max_steps) { truncated = true; alive = false; }
  }
  const bool grouped = n_res && !(ctl.overflow & 2ull) && n_res <= cap;
  if (grouped) {
    HIP_TRY(hipMemsetAsync(b->d_rcnt, 0, (b->k + 1) * 4, st), "memset(result counts)");
    HIP_TRY(hipMemsetAsync(b->d_rfill, 0, (b->k + 1) * 4, st), "memset(result fill)");
    const dim3 rg(8, kSub);
    k_res_count<<<rg, 256, 0, st>>>(d_res_seg, seg_cap, d_ctl, b->d_rcnt);
    {
      const uint32_t n_scan = (uint32_t)b->k + 1, nparts = (n_scan + kScanChunk - 1) / kScanChunk;     // cnt[k] is 0: start[k] = total
      k_res_scan_chunks<<<nparts, kScanChunk, 0, st>>>(b->d_rcnt, n_scan, b->d_rstart, b->d_rpart);
      k_res_scan_parts<<<1, kScanChunk, 0, st>>>(b->d_rpart, nparts);
      k_res_scan_add<<<nparts, kScanChunk, 0, st>>>(b->d_rstart, n_scan, b->d_rpart);
    }
    k_res_scatter<<<rg, 256, 0, st>>>(d_res_seg, seg_cap, d_ctl, b->d_rstart, b->d_rfill, d_res, (uint64_t)cap);
    HIP_TRY(hipMemsetAsync(b->d_big, 0, 8, st), "memset(big groups)");
    k_res_sort<<<(int)((b->k + 255) / 256), 256, 0, st>>>(d_res, b->d_rstart, (uint32_t)b->k, b->d_big);
    HIP_TRY(hipGetLastError(), "result grouping kernels");
    launches += 6;
  }
  HIP_TRY(hipEventRecord(e1, st), "hipEventRecord");
  HIP_TRY(hipStreamSynchronize(st), "sync");
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  mark("levels done");
  {
    std::lock_guard<std::mutex> lk(h->mu);
    h->last_kernel_ms = ms;
    h->launches += launches;
  }
  struct { unsigned long long res_count; } tot{n_res};
  const size_t extra = b->start_final.size();
  *n_out = (size_t)tot.res_count + extra;
  if ((ctl.overflow & 2ull) || tot.res_count + extra > cap) { set_error("result buffer too small"); return FMX_ERR_OVERFLOW; }
  if (tot.res_count)
    HIP_TRY(hipMemcpy(out, d_res, (size_t)tot.res_count * sizeof(fmx_result), hipMemcpyDeviceToHost), "D2H(results)");
  mark("results copied");
  for (size_t j = 0; j < extra; j++) {           // dfa.scala:270-273 with the start StatePoint(0,0,0,n)
    fmx_result &o = out[tot.res_count + j];
    o.regex = b->start_final[j]; o.len = 0; o.sp = 0; o.ep = h->n;
  }
  tot.res_count += extra;
  if (tot.res_count) {
    // canonical order (regex, len, sp, ep).  The device delivered the frontier's results grouped by regex
    // (any order inside a group); regexes that start in a final DFA state add theirs on the host.
    const size_t nres = (size_t)tot.res_count;
    const size_t ndev = nres - extra;
    auto by_key = [](const fmx_result &a, const fmx_result &b) {
      if (a.len != b.len) return a.len < b.len;
      if (a.sp != b.sp) return a.sp < b.sp;
      return a.ep < b.ep;
    };
    uint32_t nbig = 0;
    if (ndev) HIP_TRY(hipMemcpy(&nbig, &b->d_big->n, 4, hipMemcpyDeviceToHost), "D2H(big groups)");
    if (!extra && nbig <= kBigMax) {
      // the device ordered every group of up to kSmallGroup results; the few larger ones are listed
      if (nbig) {
        std::vector<uint32_t> ent(2 * (size_t)nbig);
        HIP_TRY(hipMemcpy(ent.data(), b->d_big->ent, ent.size() * 4, hipMemcpyDeviceToHost), "D2H(big groups)");
        for (uint32_t g = 0; g < nbig; g++) std::sort(out + ent[2 * g], out + ent[2 * g] + ent[2 * g + 1], by_key);
      }
      if (per_regex_count && ndev)
        HIP_TRY(hipMemcpy(per_regex_count, b->d_rcnt, b->k * 4, hipMemcpyDeviceToHost), "D2H(result counts)");
    } else {
      // host-made results to merge in (or too many large groups to list): bucket everything by regex id
      std::vector<uint32_t> cnt(b->k + 1, 0);
      for (size_t j = 0; j < nres; j++) cnt[out[j].regex]++;
      std::vector<uint32_t> start(b->k + 1, 0);
      for (size_t r = 0; r < b->k; r++) start[r + 1] = start[r] + cnt[r];
      std::vector<fmx_result> tmp(out, out + nres);
      std::vector<uint32_t> fill(start.begin(), start.end() - 1);
      for (size_t j = 0; j < nres; j++) out[fill[tmp[j].regex]++] = tmp[j];
      for (size_t r = 0; r < b->k; r++)
        if (cnt[r] > 1) std::sort(out + start[r], out + start[r + 1], by_key);
      if (per_regex_count)
        for (size_t r = 0; r < b->k; r++) per_regex_count[r] = cnt[r];
    }
  }
  mark("results ordered");
  if (truncated) {
    set_error("frontier still alive after max_steps levels: results hold every match of length <= max_steps");
    return FMX_TRUNCATED;
  }
  return FMX_OK;
}

}  // namespace fmx

using namespace fmx;

extern "C" {

int fmx_regex_compile(const char *re, int line_only, fmx_regex **out) {
  if (!re || !out) { set_error("null argument"); return FMX_ERR_ARG; }
  *out = nullptr;
  try {
    Regex *r = new Regex(compile_regex(re, line_only != 0));
    *out = reinterpret_cast<fmx_regex *>(r);
    return FMX_OK;
  } catch (const RegexError &e) {
    set_error(e.msg);
    return e.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return FMX_ERR_NOMEM;
  }
}

// REParser.createNFA (re2/re2.scala:264-334): `src` is a regex (parsed by re2post) or, with
// src_is_postfix, a postfix string for post2re (:188-205, '.' = concat) as the reference's tests use.
int fmx_nfa_compile(const char *src, int line_only, int src_is_postfix, fmx_regex **out) {
  if (!src || !out) { set_error("null argument"); return FMX_ERR_ARG; }
  *out = nullptr;
  try {
    const std::vector<PostPoint> post = src_is_postfix ? post2re(src) : re2post(src, line_only != 0);
    *out = reinterpret_cast<fmx_regex *>(new Regex(compile_thompson(post, src)));
    return FMX_OK;
  } catch (const RegexError &e) {
    set_error(e.msg);
    return e.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return FMX_ERR_NOMEM;
  }
}

int fmx_dfa_compile(const int32_t *moves, uint32_t nstates, uint32_t nchars, const uint8_t *finish, fmx_regex **out) {
  if (!out) { set_error("null argument"); return FMX_ERR_ARG; }
  *out = nullptr;
  try {
    *out = reinterpret_cast<fmx_regex *>(new Regex(compile_dfa(moves, nstates, nchars, finish)));
    return FMX_OK;
  } catch (const RegexError &e) {
    set_error(e.msg);
    return e.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return FMX_ERR_NOMEM;
  }
}

int fmx_regex_free(fmx_regex *re) {
  delete reinterpret_cast<Regex *>(re);
  return FMX_OK;
}

int fmx_regex_post_string(const char *re, int line_only, char *out, size_t cap) {
  if (!re || !out || !cap) { set_error("null argument"); return FMX_ERR_ARG; }
  try {
    std::string s = re2poststr(re, line_only != 0);
    if (s.size() + 1 > cap) { set_error("output buffer too small"); return FMX_ERR_OVERFLOW; }
    std::copy(s.begin(), s.end(), out);
    out[s.size()] = 0;
    return FMX_OK;
  } catch (const RegexError &e) {
    set_error(e.msg);
    return e.code;
  }
}

int fmx_regex_tables(const fmx_regex *re, uint32_t *n_states, uint8_t *st_c, int32_t *st_num, uint8_t *st_last,
                     int32_t *fol_off, uint32_t *n_follows, int32_t *fol, uint32_t *n_firsts, int32_t *firsts) {
  if (!re) { set_error("null argument"); return FMX_ERR_ARG; }
  const Regex *r = reinterpret_cast<const Regex *>(re);
  if (n_states) *n_states = (uint32_t)r->st_c.size();
  if (n_follows) *n_follows = (uint32_t)r->fol.size();
  if (n_firsts) *n_firsts = (uint32_t)r->firsts.size();
  if (st_c) std::copy(r->st_c.begin(), r->st_c.end(), st_c);
  if (st_num) std::copy(r->st_num.begin(), r->st_num.end(), st_num);
  if (st_last) std::copy(r->st_last.begin(), r->st_last.end(), st_last);
  if (fol_off) std::copy(r->fol_off.begin(), r->fol_off.end(), fol_off);
  if (fol) std::copy(r->fol.begin(), r->fol.end(), fol);
  if (firsts) std::copy(r->firsts.begin(), r->firsts.end(), firsts);
  return FMX_OK;
}

int fmx_regex_batch_create(const fmx_index *idx, fmx_regex *const *res, size_t k, fmx_regex_batch **out) {
  if (!idx || !out || (k && !res)) { set_error("null argument"); return FMX_ERR_ARG; }
  *out = nullptr;
  for (size_t r = 0; r < k; r++)
    if (!res[r]) { set_error("null regex handle"); return FMX_ERR_ARG; }
  RegexBatch *b = nullptr;
  int rc = regex_batch_create(reinterpret_cast<const Index *>(idx), reinterpret_cast<const Regex *const *>(res), k, &b);
  if (rc == FMX_OK) *out = reinterpret_cast<fmx_regex_batch *>(b);
  return rc;
}

int fmx_regex_batch_free(fmx_regex_batch *b) {
  RegexBatch *rb = reinterpret_cast<RegexBatch *>(b);
  if (rb) { (void)hipSetDevice(rb->device); delete rb; }
  return FMX_OK;
}

int fmx_regex_batch_match(const fmx_index *idx, fmx_regex_batch *b, const fmx_limits *lim, fmx_result *out,
                          size_t cap, size_t *n_out, uint32_t *per_regex_count) {
  if (!idx || !b || !n_out || (cap && !out)) { set_error("null argument"); return FMX_ERR_ARG; }
  const Index *h = reinterpret_cast<const Index *>(idx);
  RegexBatch *rb = reinterpret_cast<RegexBatch *>(b);
  if (lim && lim->mode == FMX_MATCH_REFERENCE) {
    if (lim->max_branching == 0) { set_error("max_branching must be positive"); return FMX_ERR_ARG; }
    if (!rb->all_retree) {
      set_error("the reference-order mode replays ReTree._matchSA; Thompson and DFA handles use the frontier mode");
      return FMX_ERR_UNSUPPORTED;
    }
    if (rb->index_serial != h->serial) { set_error("regex batch was prepared for another index"); return FMX_ERR_ARG; }
    const RefTables rt{rb->nfa.st, rb->nfa.fol, rb->d_st_num, rb->d_first_off, rb->d_first_state};
    return regex_match_reference(h, rt, rb->k, rb->max_fanout, lim->max_branching, lim->max_iterations, out, cap, n_out,
                                 per_regex_count, nullptr);
  }
  if (lim && lim->mode != FMX_MATCH_FRONTIER) { set_error("unknown fmx_limits.mode"); return FMX_ERR_ARG; }
  return regex_batch_match(h, rb, lim, out, cap, n_out, per_regex_count);
}

int fmx_regex_match_batch(const fmx_index *idx, fmx_regex *const *res, size_t k, const fmx_limits *lim,
                          fmx_result *out, size_t cap, size_t *n_out, uint32_t *per_regex_count) {
  if (!n_out) { set_error("null argument"); return FMX_ERR_ARG; }
  fmx_regex_batch *b = nullptr;
  int rc = fmx_regex_batch_create(idx, res, k, &b);
  if (rc != FMX_OK) return rc;
  rc = fmx_regex_batch_match(idx, b, lim, out, cap, n_out, per_regex_count);
  fmx_regex_batch_free(b);
  return rc;
}

}  // extern "C"

// fmx_refmatch.hip -- ReTree._matchSA in the reference's own order, limits included.
//
// Reference: re2/retree.scala:618-653.  One scala.collection.mutable.PriorityQueue of
// StatePoint(len, sp, ep, state) ordered by smallest state.num (:562-567); the loop runs while
// the queue is non-empty, shorter than maxBranching, and (maxIterations == 0 or i < maxIterations)
// with i starting at 1 (:622,628).  When those limits bind, which results come out depends on the
// exact pop order, ties included, so this mode replays the queue itself: one regex per lane group
// (quad or octet, fmx_device.h), the group's lane 0 keeps the regex's binary heap in device memory and performs the same
// fixUp / fixDown as the Scala 2.10.0 library (`+=`: append then sift up with `<`; `dequeue`:
// swap root and last, sift down picking the right child only when left < right, stop when
// parent >= child), the popped element is broadcast to the group and stepped with the shared
// rank primitive.  Results carry their discovery number so the host can return them newest
// first, the order of the reference's `ret ::= ...` list.
//
// The frontier kernel (fmx_frontier.hip) is the throughput path; this one exists so that a
// caller who keeps the reference's default limits (1024 / 1000) gets the reference's answer.
#include <fmx.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fmx_device.h"
#include "fmx_host.h"
#include "fmx_nfa.h"

namespace fmx {

constexpr int kRThreads = 256;

struct HeapElem {          // 32 bytes
  uint32_t num;            // CharNode.num: the only key StatePoint.compare looks at (:564)
  uint32_t state;          // global CharNode id
  uint32_t len;
  uint32_t pad;
  uint64_t sp, ep;
};

struct RefResult {
  uint32_t regex, len, seq, pad;
  uint64_t sp, ep;
};

struct RefCtl {
  unsigned long long res_count;
  unsigned long long overflow;
};

// this < that  <=>  this.num > that.num   (StatePoint.compare, :564)
__device__ __forceinline__ bool heap_lt(const HeapElem &x, const HeapElem &y) { return x.num > y.num; }

__device__ void heap_push(HeapElem *a, uint32_t &size0, const HeapElem &e) {
  a[size0] = e;
  uint32_t k = size0;
  while (k > 1) {
    const HeapElem p = a[k / 2];
    if (!heap_lt(p, e)) break;
    a[k] = p;                     // swap(k, k/2) with e travelling up
    k /= 2;
  }
  a[k] = e;
  size0 += 1;
}

__device__ HeapElem heap_pop(HeapElem *a, uint32_t &size0) {
  size0 -= 1;
  const HeapElem top = a[1];
  HeapElem x = a[size0];           // swap(1, size0): the last element goes to the root...
  a[size0] = top;
  const uint32_t n = size0 - 1;    // ...and sifts down inside a[1..n]
  uint32_t k = 1;
  while (n >= 2 * k) {
    uint32_t j = 2 * k;
    HeapElem c = a[j];
    if (j < n) {
      const HeapElem c2 = a[j + 1];
      if (heap_lt(c, c2)) { j += 1; c = c2; }
    }
    if (!heap_lt(x, c)) break;     // as(k) >= as(j)
    a[k] = c;
    k = j;
  }
  if (n >= 1) a[k] = x;
  return top;
}

template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kRThreads) void k_match_ref(DevIndex ix, RefTables rt, uint32_t k_regex,
                                                          HeapElem *__restrict__ heaps, uint32_t heap_cap,
                                                          uint32_t max_branching, uint32_t max_iterations,
                                                          RefResult *__restrict__ res, uint64_t res_cap,
                                                          uint32_t *__restrict__ front_left,
                                                          RefCtl *__restrict__ ctl,
                                                          unsigned long long *__restrict__ counters) {
  __shared__ uint64_t s_cf[256];
  __shared__ uint16_t s_slot[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  __syncthreads();
  constexpr int G = Lay<LAYOUT>::G;
  const LaneConst lc = lane_const<G>();
  const uint32_t t = lc.t;
  const uint32_t lane = __lane_id();
  const uint32_t leader = lane & ~(uint32_t)(G - 1);
  const uint32_t group = (blockIdx.x * kRThreads + threadIdx.x) / G;
  const uint32_t noct = gridDim.x * (kRThreads / G);
  HeapElem *heap = heaps + (size_t)group * heap_cap;
  uint32_t stepped = 0;
  for (uint32_t r = group; r < k_regex; r += noct) {
    uint32_t size0 = 1, nres = 0, it = 1;
    bool bad = false;
    if (t == 0) {
      for (uint32_t f = rt.first_off[r]; f < rt.first_off[r + 1]; f++) {     // pqFront ++= inputStates, :624
        HeapElem e;
        e.state = rt.first[f]; e.num = rt.st_num[e.state]; e.len = 0; e.pad = 0; e.sp = 0; e.ep = ix.n;
        if (size0 + 1 > heap_cap) { bad = true; break; }
        heap_push(heap, size0, e);
      }
    }
    for (;;) {
      // loop condition, :628 (decided by the leader, shared with the group)
      uint32_t go = 0, state = 0, len = 0, splo = 0, sphi = 0, eplo = 0, ephi = 0;
      if (t == 0 && !bad && size0 >= 2 && (size0 - 1) < max_branching && (max_iterations == 0 || it < max_iterations)) {
        const HeapElem q = heap_pop(heap, size0);
        go = 1; state = q.state; len = q.len;
        splo = (uint32_t)q.sp; sphi = (uint32_t)(q.sp >> 32); eplo = (uint32_t)q.ep; ephi = (uint32_t)(q.ep >> 32);
      }
      go = __shfl(go, leader, 64);
      if (!go) break;
      state = __shfl(state, leader, 64);
      len = __shfl(len, leader, 64);
      uint64_t sp = ((uint64_t)__shfl(sphi, leader, 64) << 32) | __shfl(splo, leader, 64);
      uint64_t ep = ((uint64_t)__shfl(ephi, leader, 64) << 32) | __shfl(eplo, leader, 64);
      // sa.getPrevRange(q.sp, q.ep, q.state.c), :633
      const StateRec rec = rt.st[state];
      const uint32_t c = rec_c(rec);
      const uint16_t slot = s_slot[c];
      const uint64_t cfc = s_cf[c];
      backward_step<WIDE, LAYOUT>(ix, c, slot, cfc, lc, sp, ep);
      stepped++;
      if (t == 0 && sp < ep) {
        if (rec_emit(rec)) {                                     // isLast, :636-638
          const unsigned long long at = atomicAdd(&ctl->res_count, 1ull);
          if (at < res_cap) {
            RefResult o;
            o.regex = r; o.len = len + 1; o.seq = nres; o.pad = 0; o.sp = sp; o.ep = ep;
            res[at] = o;
          } else {
            atomicOr(&ctl->overflow, 2ull);
          }
          nres++;
        } else {                                                   // :641
          for (uint32_t f = rec.fol_off; f < rec.fol_off + rec_cnt(rec); f++) {
            HeapElem e;
            e.state = rt.fol[f]; e.num = rt.st_num[e.state]; e.len = len + 1; e.pad = 0; e.sp = sp; e.ep = ep;
            if (size0 + 1 > heap_cap) { bad = true; atomicOr(&ctl->overflow, 1ull); break; }
            heap_push(heap, size0, e);
          }
        }
      }
      it++;
    }
    if (t == 0 && front_left) front_left[r] = size0 - 1;
  }
  counters_add(counters, t == 0 ? 2ull * stepped : 0ull, t == 0 ? stepped : 0u, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// The same replay, one regex per WAVE, for limits that fit LDS (the reference's defaults do).  A regex's pops are
// serial by definition (which results come out depends on their order), so what a launch lasts is (pops of the
// longest regex) x (time per pop), and a pop is a chain of dependent accesses.  What keeps that chain short here:
//  * The heap is an array of 4-byte entries {num:16 | slot:16} in LDS.  Only `num` orders the queue (:564), so the
//    sifts move 4-byte keys; what an element carries (len, interval, its state's follow list) sits in a slab in
//    device memory under its slot number.  Free slots need no list: the entries of the array behind the heap are
//    always a permutation of the unused slot numbers (a pop swaps the popped entry to a[size - 1], just outside the
//    heap; a push takes the slot number it finds at a[size]).  A push is one LDS round trip (every ancestor read at
//    once, one ballot, the shifted entries written back at once); a pop walks down one level per round trip (both
//    children in one read).
//  * An element's step does not depend on WHEN it is popped, only the heap's shape does.  So getPrevRange is
//    evaluated when the element is PUSHED: a pop's follows are stepped together, one lane group each, all their rank
//    blocks in flight at once (the reference pays one dependent rank query per pop), and whether the step came back
//    empty is kept in LDS beside the heap.  Popping an element whose interval is empty -- most pops once the
//    intervals are narrow: a row has one preceding character, the others die -- touches no memory at all:
//    it only counts as an iteration and leaves the heap, exactly as in the reference's loop.
//  * The elements of the last push stay in the registers of the lane groups that stepped them; a pop that takes one
//    of them (the usual case: the search runs depth-first through a regex's positions) reads no slab entry.
//  * An element carries the bytes of its state's first four follows (FolRec::fc), so when it is popped its follows'
//    rank blocks are requested at once, beside the loads of the follows' own records.
//  * Results are written to chunks of 64 slots a wave reserves with one atomic; a second small launch puts them in
//    the reference's order: one wave owns a regex, so result `seq` of a regex with `cnt` results belongs at
//    start[regex] + cnt - 1 - seq (newest first, the order of `ret ::= ...`, :638) -- no sort.
// Elements still queued when a limit ends the loop were stepped for nothing; nothing of them is observable
// (the statistics count pops, like the reference's own getPrevRange calls).
struct RefSlot {           // 32 bytes, written when the element is pushed (only for non-empty intervals)
  uint64_t sp, ep;         // the interval AFTER the element's own step
  uint32_t fc, len;        // FolRec::fc; len = the StatePoint's len (the step makes it len + 1)
  uint32_t fol_off, cnt_c_emit;
};
struct RefCtl2 {
  unsigned long long res_count;      // result slots reserved (chunks of kResChunk)
  unsigned long long overflow;
  unsigned long long valid;          // results written
  unsigned int next;                 // next regex to hand out
  unsigned int pad;
};
constexpr uint32_t kWaveCf = 256 * 8 + 256 * 2;      // bytes of the C[] / slot tables in front of the waves' heaps
constexpr uint32_t kResChunk = 64;
constexpr uint32_t kNoRegex = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t runi(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// LDS traffic of one wave is processed in program order; what must not happen is the compiler moving accesses across
// the points where lanes read what other lanes wrote
__device__ __forceinline__ void lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// a[size0] = key, sifted up as `+=` does (fixUp: while k > 1 && a[k/2] < a[k] swap): one LDS round trip
__device__ __forceinline__ void wave_heap_push(uint32_t *a, uint32_t &size0, uint32_t key, uint32_t lane) {
  const uint32_t k = size0;
  const uint32_t pos = lane < 32u ? k >> lane : 0u;              // lane l: the l-th ancestor of position k
  const bool is_anc = lane >= 1u && pos >= 1u;
  const uint32_t anc = is_anc ? a[pos] : 0u;
  const bool up = is_anc && (anc >> 16) > (key >> 16);            // heap_lt(ancestor, e): the new element moves above it
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(up);
  const uint32_t m = (uint32_t)__builtin_ctzll(~(bal >> 1));     // ancestors passed, from the nearest on
  if (lane >= 1u && lane <= m) a[k >> (lane - 1u)] = anc;
  if (lane == 0u) a[k >> m] = key;
  size0 = k + 1u;
  lds_sync();
}

// dequeue (swap(1, size - 1), fixDown inside a[1 .. size - 2]); returns the popped entry, which now sits at a[size0].
// Everything here is wave-uniform and kept in scalar registers; every lane writes the same value to the same LDS
// address (no exec-mask juggling around a one-lane store: the launch is bound by instruction issue, not by latency).
__device__ __forceinline__ uint32_t wave_heap_pop(uint32_t *a, uint32_t &size0) {
  size0 -= 1u;
  const uint32_t top = runi(a[1]);
  const uint32_t x = runi(a[size0]);
  const uint32_t n = size0 - 1u, xnum = x >> 16;
  uint32_t k = 1u;
  a[size0] = top;
  while (n >= 2u * k) {
    uint32_t j = 2u * k;
    const uint2 cc = *reinterpret_cast<const uint2 *>(a + j);     // both children (j is even)
    uint32_t c = runi(cc.x);
    const uint32_t c2 = runi(cc.y);
    const bool right = j < n && (c >> 16) > (c2 >> 16);           // right child only when left < right
    j += right ? 1u : 0u;
    c = right ? c2 : c;
    if (!(xnum > (c >> 16))) break;                               // a[k] >= a[j]
    a[k] = c;
    k = j;
  }
  if (n >= 1u) a[k] = x;
  lds_sync();
  return top;
}

template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kRThreads) void k_match_ref_wave(DevIndex ix, RefTables rt, uint32_t k_regex,
                                                               RefSlot *__restrict__ slabs, uint32_t heap_cap,
                                                               uint32_t max_branching, uint32_t max_iterations,
                                                               RefResult *__restrict__ res, uint64_t res_cap,
                                                               uint32_t *__restrict__ rcnt, uint32_t *__restrict__ front_left,
                                                               RefCtl2 *__restrict__ ctl,
                                                               unsigned long long *__restrict__ counters) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  uint64_t *s_cf = reinterpret_cast<uint64_t *>(s_raw);
  uint16_t *s_slot = reinterpret_cast<uint16_t *>(s_raw + 256 * 8);
  for (int c = threadIdx.x; c < 256; c += blockDim.x) { s_cf[c] = ix.cf[c]; s_slot[c] = ix.slot[c]; }
  constexpr int G = Lay<LAYOUT>::G;
  constexpr uint32_t NG = 64 / G;                 // lane groups of a wave = follows stepped at once
  const LaneConst lc = lane_const<G>();
  const uint32_t lane = __lane_id();
  const uint32_t wv = runi(threadIdx.x >> 6);
  const uint32_t q = lane / G;                    // this lane's group
  const uint32_t wave_bytes = (heap_cap * 5u + 15u) & ~15u;
  uint32_t *a = reinterpret_cast<uint32_t *>(s_raw + kWaveCf + (size_t)wv * wave_bytes);   // the heap, 1-based
  uint8_t *alive = reinterpret_cast<uint8_t *>(a + heap_cap);                               // by slot number
  RefSlot *slab = slabs + (size_t)(blockIdx.x * (blockDim.x >> 6) + wv) * heap_cap;
  for (uint32_t i = lane; i < heap_cap; i += 64u) a[i] = i;       // every slot number once
  __syncthreads();
  uint32_t stepped = 0;
  uint32_t n_reqs = 0, n_push = 0, n_live_push = 0, n_live_pop = 0, n_res = 0;      // per lane / wave-uniform tallies for fmx_stats
  unsigned long long res_at = 0, res_end = 0;     // the wave's current chunk of result slots (wave-uniform)
  // the elements of the last push, one per lane group: interval after their step, push record, slot (h_ok: non-empty)
  uint64_t h_sp = 0, h_ep = 0;
  uint4 h_fr = make_uint4(0, 0, 0, 0);
  uint32_t h_slot = 0, h_len = 0;
  bool h_ok = false;
  for (;;) {
    uint32_t r = 0;
    if (lane == 0u) r = atomicAdd(&ctl->next, 1u);
    r = runi(r);
    if (r >= k_regex) break;
    uint32_t size0 = 1u, nres = 0u, it = 1u;
    bool bad = false;
    // pqFront ++= / += of `cnt` elements whose records are recs[0 .. cnt): each is stepped from the parent's interval
    // right away (lane group g steps element j0 + g), then pushed in order.  pfc = the bytes of the first four
    // (FolRec::fc of the parent) when known: their rank blocks are then requested beside their records.
    auto push_all = [&](const FolRec *recs, uint32_t cnt, uint64_t psp, uint64_t pep, uint32_t clen, uint32_t pfc, bool have_fc) {
      for (uint32_t j0 = 0; j0 < cnt && !bad; j0 += NG) {
        const uint32_t nb = cnt - j0 < NG ? cnt - j0 : NG;
        if (size0 + nb > heap_cap) { bad = true; if (lane == 0u) atomicOr(&ctl->overflow, 1ull); break; }
        const bool act = q < nb;
        uint4 fr = make_uint4(0, 0, 0, 0);
        if (act) fr = *reinterpret_cast<const uint4 *>(recs + j0 + q);
        const uint32_t slotno = act ? (a[size0 + q] & 0xFFFFu) : 0u;      // the free slot this push will take
        uint64_t sp = psp, ep = pep;
        if (have_fc && j0 == 0u && nb <= 4u) {          // wave-uniform: the bytes come from the parent, the records ride along
          if (act) {
            const uint32_t c = (pfc >> (8u * q)) & 0xFFu;
            const uint32_t rq = backward_step<WIDE, LAYOUT>(ix, c, s_slot[c], s_cf[c], lc, sp, ep);
            if (lc.t == 0u) n_reqs += rq;
          }
        } else if (act) {
          const uint32_t c = (fr.z >> 16) & 0xFFu;
          const uint32_t rq = backward_step<WIDE, LAYOUT>(ix, c, s_slot[c], s_cf[c], lc, sp, ep);
          if (lc.t == 0u) n_reqs += rq;
        }
        const bool live = act && sp < ep;
        n_push += nb;
        if (act && lc.t == 0u) {
          n_live_push += live ? 1u : 0u;
          alive[slotno] = live ? 1 : 0;
          if (live) {
            uint4 *d = reinterpret_cast<uint4 *>(slab + slotno);
            d[0] = make_uint4((uint32_t)sp, (uint32_t)(sp >> 32), (uint32_t)ep, (uint32_t)(ep >> 32));
            d[1] = make_uint4(fr.x, clen, fr.y, fr.z);
          }
        }
        h_sp = sp; h_ep = ep; h_fr = fr; h_slot = slotno; h_ok = live; h_len = clen;
        lds_sync();
        for (uint32_t g = 0; g < nb; g++) {
          const uint32_t num = (uint32_t)__builtin_amdgcn_readlane((int)fr.w, (int)(g * G));
          const uint32_t sn = (uint32_t)__builtin_amdgcn_readlane((int)slotno, (int)(g * G));
          wave_heap_push(a, size0, (num << 16) | sn, lane);
        }
      }
    };
    {
      const uint32_t f0 = rt.first_off[r], f1 = rt.first_off[r + 1];
      push_all(rt.first_rec + f0, f1 - f0, 0ull, ix.n, 0u, 0u, false);      // pqFront ++= inputStates, :624
    }
    // loop condition, :628
    while (!bad && size0 >= 2u && (size0 - 1u) < max_branching && (max_iterations == 0u || it < max_iterations)) {
      const uint32_t e = wave_heap_pop(a, size0);
      const uint32_t sn = e & 0xFFFFu;
      stepped++;
      it++;
#ifdef FMX_REF_PRIO
      if (it == FMX_REF_PRIO) __builtin_amdgcn_s_setprio(3);               // a long regex is the launch's critical path
#endif
      if (!runi((uint32_t)alive[sn])) continue;                             // None, :634: nothing to do
      n_live_pop++;
      uint64_t psp, pep;
      uint32_t len, fol_off, cce, fc;
      const unsigned long long held = __builtin_amdgcn_ballot_w64(h_ok && h_slot == sn);
      if (held) {                                                            // still in the registers of the group that stepped it
        const int src = (int)__builtin_ctzll(held);
        psp = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(h_sp >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)h_sp, src);
        pep = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(h_ep >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)h_ep, src);
        fc = (uint32_t)__builtin_amdgcn_readlane((int)h_fr.x, src);
        fol_off = (uint32_t)__builtin_amdgcn_readlane((int)h_fr.y, src);
        cce = (uint32_t)__builtin_amdgcn_readlane((int)h_fr.z, src);
        len = h_len;
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the wave's own slab stores have landed
        const uint4 *sp4 = reinterpret_cast<const uint4 *>(slab + sn);
        const uint4 p0 = sp4[0], p1 = sp4[1];
        psp = ((uint64_t)runi(p0.y) << 32) | runi(p0.x);
        pep = ((uint64_t)runi(p0.w) << 32) | runi(p0.z);
        fc = runi(p1.x); len = runi(p1.y); fol_off = runi(p1.z); cce = runi(p1.w);
      }
      if ((cce >> 24) & 1u) {                                                // isLast, :636-638
        if (res_at == res_end) {
          unsigned long long base = 0;
          if (lane == 0u) base = atomicAdd(&ctl->res_count, (unsigned long long)kResChunk);
          res_at = ((unsigned long long)runi((uint32_t)(base >> 32)) << 32) | runi((uint32_t)base);
          res_end = res_at + kResChunk;
        }
        if (lane == 0u) {
          if (res_at < res_cap) {
            RefResult o;
            o.regex = r; o.len = len + 1u; o.seq = nres; o.pad = 0; o.sp = psp; o.ep = pep;
            res[res_at] = o;
          } else {
            atomicOr(&ctl->overflow, 2ull);
          }
        }
        res_at++;
        nres++;
        n_res++;
      } else {                                                               // :641
        push_all(rt.fol_rec + fol_off, cce & 0xFFFFu, psp, pep, len + 1u, fc, true);
      }
    }
    if (lane == 0u) {
      rcnt[r] = nres;
      if (front_left) front_left[r] = size0 - 1u;
    }
    h_ok = false;
#ifdef FMX_REF_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  }
  // the rest of the wave's last chunk holds no results
  for (unsigned long long i = res_at + lane; i < res_end; i += 64u)
    if (i < res_cap) res[i].regex = kNoRegex;
  if (lane == 0u && n_res) atomicAdd(&ctl->valid, (unsigned long long)n_res);
  counters_add(counters, lane == 0u ? 2ull * stepped : 0ull, lane == 0u ? stepped : 0u, 0);
  // the frontier counters of fmx_stats, read for this kernel as: rank-line requests | slots written (elements pushed
  // with a non-empty interval) | results | elements stepped at push time | slots read (non-empty elements popped) |
  // push records loaded
  counters_add_frontier(counters, n_reqs, n_live_push, lane == 0u ? n_res : 0u, lane == 0u ? n_push : 0u,
                        lane == 0u ? n_live_pop : 0u, lane == 0u ? n_push : 0u);
}

// Exclusive prefix sums of the per-regex result counts (cnt[k] reads as 0: start[k] = the total) in two small
// launches: every workgroup scans its chunk of 1024 counts and leaves the chunk's total, then adds up the totals of the
// chunks before its own (a hundred values for 100 k regexes).  (One workgroup walking all the chunks took 114 us.)
__global__ __launch_bounds__(1024) void k_ref_scan_chunks(const uint32_t *__restrict__ cnt, uint32_t k, uint32_t *__restrict__ start,
                                                           uint32_t *__restrict__ part) {
  __shared__ uint32_t s_wave[16];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  const uint32_t v = i < k ? cnt[i] : 0u;
  uint32_t x = v;                                   // inclusive scan inside the wave
  for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if ((int)lane >= d) x += y; }
  if (lane == 63u) s_wave[wv] = x;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (uint32_t j = 0; j < 16; j++) { const uint32_t t = s_wave[j]; before += j < wv ? t : 0u; all += t; }
  if (i <= k) start[i] = before + x - v;
  if (threadIdx.x == 0) part[blockIdx.x] = all;
}
__global__ __launch_bounds__(1024) void k_ref_scan_add(uint32_t *__restrict__ start, uint32_t k, const uint32_t *__restrict__ part) {
  __shared__ uint32_t s_sum[16];
  uint32_t mine = 0;
  for (uint32_t q = threadIdx.x; q < blockIdx.x; q += 1024u) mine += part[q];
  for (int d = 1; d < 64; d <<= 1) mine += (uint32_t)__shfl_xor((int)mine, d, 64);
  if ((threadIdx.x & 63u) == 0) s_sum[threadIdx.x >> 6] = mine;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t j = 0; j < 16; j++) before += s_sum[j];
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  if (i <= k) start[i] += before;
}

// raw result i of regex r with sequence number seq goes to start[r] + cnt[r] - 1 - seq
__global__ __launch_bounds__(256) void k_ref_place(const RefResult *__restrict__ raw, const RefCtl2 *__restrict__ ctl, uint64_t res_cap,
                                                    const uint32_t *__restrict__ start, const uint32_t *__restrict__ cnt,
                                                    fmx_result *__restrict__ out, uint64_t out_cap) {
  const uint64_t total = ctl->res_count < res_cap ? ctl->res_count : res_cap;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const RefResult r = raw[i];
    if (r.regex == kNoRegex) continue;
    const uint64_t at = (uint64_t)start[r.regex] + (cnt[r.regex] - 1u - r.seq);
    if (at < out_cap) { fmx_result o; o.regex = r.regex; o.len = r.len; o.sp = r.sp; o.ep = r.ep; out[at] = o; }
  }
}

#define HIP_TRY(call, what)                            \
  do {                                                 \
    hipError_t e__ = (call);                           \
    if (e__ != hipSuccess) return hip_fail(e__, what); \
  } while (0)


// results per regex, newest first: the order of the reference's prepended list (:638)
static void deliver_results(std::vector<RefResult> &tmp, fmx_result *out, uint32_t *per_regex_count) {
  std::sort(tmp.begin(), tmp.end(), [](const RefResult &a, const RefResult &b) {
    if (a.regex != b.regex) return a.regex < b.regex;
    return a.seq > b.seq;
  });
  for (size_t j = 0; j < tmp.size(); j++) {
    out[j].regex = tmp[j].regex; out[j].len = tmp[j].len; out[j].sp = tmp[j].sp; out[j].ep = tmp[j].ep;
    if (per_regex_count) per_regex_count[tmp[j].regex]++;
  }
}

static int match_reference_wave(const Index *h, const RefTables &rt, size_t k, uint32_t heap_cap, uint32_t max_branching,
                                uint32_t max_iterations, fmx_result *out, size_t cap, size_t *n_out,
                                uint32_t *per_regex_count, uint32_t *front_left) {
  const size_t wave_bytes = ((size_t)heap_cap * 5 + 15) & ~(size_t)15;
  // waves per workgroup: as many of 4 as fit 64 KB of LDS; workgroups per CU: what LDS (160 KB) and the wave slots allow
  uint32_t wpw = 4;
  while (wpw > 1 && kWaveCf + wpw * wave_bytes > (64u << 10)) wpw >>= 1;
  const size_t lds = kWaveCf + wpw * wave_bytes;
  uint32_t wg_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160u << 10) / lds, 32 / wpw));
  if (const char *e = getenv("FMX_REF_WGS")) wg_per_cu = (uint32_t)std::max(1, std::min<int>(atoi(e), (int)wg_per_cu));      // A/B runs
  uint64_t grid = std::min<uint64_t>((k + wpw - 1) / wpw, (uint64_t)h->cu_count * wg_per_cu);
  // bound the slab arena (32 B x heap_cap per wave) to ~2 GiB
  grid = std::max<uint64_t>(1, std::min<uint64_t>(grid, (2ull << 30) / ((uint64_t)heap_cap * sizeof(RefSlot) * wpw)));
  CtxLease lease(h);
  if (!lease.c) return FMX_ERR_HIP;
  hipStream_t st = lease.c->stream;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t waves = (size_t)grid * wpw;
  // raw results: the caller's capacity plus one chunk of slack per wave (a wave reserves result slots 64 at a time)
  const size_t raw_cap = (cap ? cap : 1) + waves * kResChunk;
  const size_t o_ctl = 0, o_slab = up(sizeof(RefCtl2)), o_raw = o_slab + up(waves * heap_cap * sizeof(RefSlot)),
               o_out = o_raw + up(raw_cap * sizeof(RefResult)), o_cnt = o_out + up((cap ? cap : 1) * sizeof(fmx_result)),
               o_start = o_cnt + up((k + 1) * 4), o_part = o_start + up((k + 1) * 4), o_left = o_part + up(((k + 1) / 1024 + 2) * 4),
               total = o_left + up(k * 4);
  void *arena_v = nullptr;
  HIP_TRY(ctx_scratch(lease.c, 0, total, &arena_v), "hipMalloc(reference-order arena)");
  uint8_t *arena = static_cast<uint8_t *>(arena_v);
  HIP_TRY(hipMemsetAsync(arena + o_ctl, 0, sizeof(RefCtl2), st), "memset(ctl)");
  RefSlot *d_slab = reinterpret_cast<RefSlot *>(arena + o_slab);
  RefResult *d_raw = reinterpret_cast<RefResult *>(arena + o_raw);
  fmx_result *d_out = reinterpret_cast<fmx_result *>(arena + o_out);
  uint32_t *d_cnt = reinterpret_cast<uint32_t *>(arena + o_cnt), *d_start = reinterpret_cast<uint32_t *>(arena + o_start),
           *d_part = reinterpret_cast<uint32_t *>(arena + o_part);
  RefCtl2 *d_ctl = reinterpret_cast<RefCtl2 *>(arena + o_ctl);
  uint32_t *d_left = front_left ? reinterpret_cast<uint32_t *>(arena + o_left) : nullptr;
  struct HostCtl { RefCtl2 ctl; };
  void *pin_v = nullptr;
  if (lease.c->pin_cap < sizeof(RefCtl2)) {
    if (lease.c->pin) { (void)hipHostFree(lease.c->pin); lease.c->pin = nullptr; lease.c->pin_cap = 0; }
    HIP_TRY(hipHostMalloc(&lease.c->pin, 4096, hipHostMallocDefault), "hipHostMalloc");
    lease.c->pin_cap = 4096;
  }
  pin_v = lease.c->pin;
  RefCtl2 *h_ctl = static_cast<RefCtl2 *>(pin_v);
  hipEvent_t e0 = lease.c->ev_a, e1 = lease.c->ev_b;
  HIP_TRY(hipEventRecord(e0, st), "hipEventRecord");
#define CALL(W, L)                                                                                                          \
  k_match_ref_wave<W, L><<<(int)grid, (int)(wpw * 64), lds, st>>>(h->dev, rt, (uint32_t)k, d_slab, heap_cap, max_branching, \
                                                                  max_iterations, d_raw, (uint64_t)raw_cap, d_cnt, d_left, d_ctl, h->d_counters)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  HIP_TRY(hipGetLastError(), "k_match_ref_wave");
  const int nparts = (int)((k + 1 + 1023) / 1024);
  k_ref_scan_chunks<<<nparts, 1024, 0, st>>>(d_cnt, (uint32_t)k, d_start, d_part);
  k_ref_scan_add<<<nparts, 1024, 0, st>>>(d_start, (uint32_t)k, d_part);
  k_ref_place<<<256, 256, 0, st>>>(d_raw, d_ctl, (uint64_t)raw_cap, d_start, d_cnt, d_out, (uint64_t)cap);
  HIP_TRY(hipGetLastError(), "k_ref_scan / k_ref_place");
  HIP_TRY(hipEventRecord(e1, st), "hipEventRecord");
  HIP_TRY(hipMemcpyAsync(h_ctl, d_ctl, sizeof(RefCtl2), hipMemcpyDeviceToHost, st), "D2H(ctl)");
  if (per_regex_count) HIP_TRY(hipMemcpyAsync(per_regex_count, d_cnt, k * 4, hipMemcpyDeviceToHost, st), "D2H(result counts)");
  if (front_left) HIP_TRY(hipMemcpyAsync(front_left, d_left, k * 4, hipMemcpyDeviceToHost, st), "D2H(front_left)");
  HIP_TRY(hipStreamSynchronize(st), "sync(k_match_ref_wave)");
  const RefCtl2 ctl = *h_ctl;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  {
    std::lock_guard<std::mutex> lk(h->mu);
    h->last_kernel_ms = ms;
    h->launches += 4;
  }
  if (ctl.overflow & 1ull) { set_error("reference-order heap overflow (internal bound)"); return FMX_ERR_OVERFLOW; }
  *n_out = (size_t)ctl.valid;
  if (ctl.valid > cap || (ctl.overflow & 2ull)) { set_error("result buffer too small"); return FMX_ERR_OVERFLOW; }
  if (ctl.valid) {
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)ctl.valid * sizeof(fmx_result), hipMemcpyDeviceToHost, st), "D2H(results)");
    HIP_TRY(hipStreamSynchronize(st), "sync(results)");
  }
  return FMX_OK;
}

int regex_match_reference(const Index *h, const RefTables &rt, size_t k, uint32_t max_fanout, uint32_t max_branching,
                          uint32_t max_iterations, fmx_result *out, size_t cap, size_t *n_out,
                          uint32_t *per_regex_count, uint32_t *front_left) {
  if (per_regex_count) std::fill(per_regex_count, per_regex_count + k, 0u);
  *n_out = 0;
  if (!k) return FMX_OK;
  // the queue is checked against maxBranching before a pop, then grows by at most one fan-out
  const uint64_t heap_cap64 = (uint64_t)max_branching + max_fanout + 2;
  if (heap_cap64 > (1u << 22)) { set_error("max_branching too large for the reference-order mode"); return FMX_ERR_ARG; }
  const uint32_t heap_cap = (uint32_t)heap_cap64;
  HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
  // The wave-per-regex kernel when a regex's heap fits LDS (FMX_REFMATCH=group keeps the kernel above)
  const char *which = getenv("FMX_REFMATCH");
  const bool force_group = which && std::strcmp(which, "group") == 0;
  const size_t wave_bytes = ((size_t)heap_cap * 5 + 15) & ~(size_t)15;
  if (!force_group && heap_cap <= 0xFFFFu && rt.max_num <= 0xFFFFu && rt.fol_rec && kWaveCf + wave_bytes <= (64u << 10))
    return match_reference_wave(h, rt, k, heap_cap, max_branching, max_iterations, out, cap, n_out, per_regex_count, front_left);
  const uint64_t per_wg = kRThreads / (h->layout == kLayoutBytes ? Lay<kLayoutBytes>::G : Lay<kLayoutOneHot>::G);   // regexes per workgroup
  uint64_t want = (k + per_wg - 1) / per_wg;
  uint64_t gcap = (uint64_t)h->cu_count * 8;
  // bound the heap arena (32 B x heap_cap per lane group) to ~4 GiB
  const uint64_t arena_groups = std::max<uint64_t>(64, (4ull << 30) / ((uint64_t)heap_cap * sizeof(HeapElem)));
  gcap = std::min<uint64_t>(gcap, std::max<uint64_t>(1, arena_groups / per_wg));
  const int grid = (int)std::min(want, gcap);
  // One device arena from the handle's call context: [ctl | heaps | results | front_left]; the tables are
  // the resident batch's.
  CtxLease lease(h);
  if (!lease.c) return FMX_ERR_HIP;
  hipStream_t st = lease.c->stream;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t groups = (size_t)grid * per_wg;
  const size_t o_ctl = 0, o_heaps = up(sizeof(RefCtl)), o_res = o_heaps + up(groups * heap_cap * sizeof(HeapElem)),
               o_left = o_res + up((cap ? cap : 1) * sizeof(RefResult)), total = o_left + up(k * 4);
  void *arena_v = nullptr;
  HIP_TRY(ctx_scratch(lease.c, 0, total, &arena_v), "hipMalloc(reference-order arena)");
  uint8_t *arena = static_cast<uint8_t *>(arena_v);
  HIP_TRY(hipMemsetAsync(arena + o_ctl, 0, sizeof(RefCtl), st), "memset(ctl)");
  HeapElem *d_heaps = reinterpret_cast<HeapElem *>(arena + o_heaps);
  RefResult *d_res = reinterpret_cast<RefResult *>(arena + o_res);
  RefCtl *d_ctl = reinterpret_cast<RefCtl *>(arena + o_ctl);
  uint32_t *d_left = front_left ? reinterpret_cast<uint32_t *>(arena + o_left) : nullptr;
  hipEvent_t e0 = lease.c->ev_a, e1 = lease.c->ev_b;
  HIP_TRY(hipEventRecord(e0, st), "hipEventRecord");
#define CALL(W, L)                                                                                          \
  k_match_ref<W, L><<<grid, kRThreads, 0, st>>>(h->dev, rt, (uint32_t)k, d_heaps, heap_cap, max_branching, max_iterations, \
                                                d_res, (uint64_t)cap, d_left, d_ctl, h->d_counters)
  FMX_LAYOUT_DISPATCH(h, CALL);
#undef CALL
  HIP_TRY(hipGetLastError(), "k_match_ref");
  HIP_TRY(hipEventRecord(e1, st), "hipEventRecord");
  HIP_TRY(hipStreamSynchronize(st), "sync(k_match_ref)");
  RefCtl ctl{};
  HIP_TRY(hipMemcpyAsync(&ctl, d_ctl, sizeof ctl, hipMemcpyDeviceToHost, st), "D2H(ctl)");
  HIP_TRY(hipStreamSynchronize(st), "sync(ctl)");
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  {
    std::lock_guard<std::mutex> lk(h->mu);
    h->last_kernel_ms = ms;
    h->launches += 1;
  }
  if (ctl.overflow & 1ull) { set_error("reference-order heap overflow (internal bound)"); return FMX_ERR_OVERFLOW; }
  *n_out = (size_t)ctl.res_count;
  if (ctl.res_count > cap) { set_error("result buffer too small"); return FMX_ERR_OVERFLOW; }
  if (front_left) {
    HIP_TRY(hipMemcpyAsync(front_left, d_left, k * 4, hipMemcpyDeviceToHost, st), "D2H(front_left)");
    HIP_TRY(hipStreamSynchronize(st), "sync(front_left)");
  }
  if (ctl.res_count) {
    std::vector<RefResult> tmp((size_t)ctl.res_count);
    HIP_TRY(hipMemcpyAsync(tmp.data(), d_res, tmp.size() * sizeof(RefResult), hipMemcpyDeviceToHost, st), "D2H(results)");
    HIP_TRY(hipStreamSynchronize(st), "sync(results)");
    deliver_results(tmp, out, per_regex_count);
  }
  return FMX_OK;
}

}  // namespace fmx

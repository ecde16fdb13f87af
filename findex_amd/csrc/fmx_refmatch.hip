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

#define HIP_TRY(call, what)                            \
  do {                                                 \
    hipError_t e__ = (call);                           \
    if (e__ != hipSuccess) return hip_fail(e__, what); \
  } while (0)


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
    // per regex, newest first: the order of the reference's prepended list (:638)
    std::sort(tmp.begin(), tmp.end(), [](const RefResult &a, const RefResult &b) {
      if (a.regex != b.regex) return a.regex < b.regex;
      return a.seq > b.seq;
    });
    for (size_t j = 0; j < tmp.size(); j++) {
      out[j].regex = tmp[j].regex; out[j].len = tmp[j].len; out[j].sp = tmp[j].sp; out[j].ep = tmp[j].ep;
      if (per_regex_count) per_regex_count[tmp[j].regex]++;
    }
  }
  return FMX_OK;
}

}  // namespace fmx

// fmx_frontier.hip -- K5: Glushkov SA-interval frontier expansion for a batch of regexes.
//
// Reference: ReTree._matchSA, re2/retree.scala:618-653.  There one priority queue of
// StatePoint(len, sp, ep, state) is popped serially; each pop is one getPrevRange; a non-empty
// range either emits SAResult (isLast) or pushes one StatePoint per entry of state.follows.
// Every frontier element is independent of the others, so the order in which they are stepped does
// not change the result multiset (whenever the reference's maxBranching / maxIterations do not bind),
// and the device is free to choose the order that costs least memory traffic:
//
//   * A LANE holds an element in registers -- state, length, interval and the state's 32-byte record -- and
//     follows it: after a step that survives, the lane keeps the first follow for itself (same sp/ep, len + 1) and
//     goes on -- a literal stretch of a regex is walked like a literal pattern in k_search4, with no queue round
//     trip per character.  The state record carries the bytes of its first follows, so the next step's rank blocks
//     are requested together with the next state's record: one memory latency per step, not two.  All of an
//     element's bookkeeping runs once, in its lane (64 elements per wave).
//   * The rank queries themselves want a lane group (a quad in the one-hot layout, an octet in the bytes layout:
//     16 bytes of the block per lane, DPP reductions).  Each round hands the 64 intervals through a wave-private
//     exchange area in LDS to the lane groups in G sub-rounds of 64/G elements and takes the stepped intervals back
//     the same way.  A narrowed interval lies inside one block: one line per element, the rare second block in a
//     second trip.
//   * The other follows go to the wave's own POOL in LDS (256 entries, newest first out: depth-first, so the live
//     set stays small; a nearly empty pool is emptied oldest first); lanes whose element died take their next one
//     from there.
//   * What a wave cannot hold goes to a sliced work queue in HBM whose entries other waves may take DURING THE SAME
//     LAUNCH (tagged 8-byte granules, agent-scope stores and loads; below), and what was queued before a launch is
//     dealt out in fixed shares.  No wave ever waits for another: one that finds nothing ends, every wave ends
//     after at most `max_rounds` rounds (handing over what it still holds), and launches repeat until the queue is
//     empty.  A batch like C4 is one launch (round 1 ran one launch per match length: 64+ launches, every element
//     through HBM queues at every level; profiles/r02_c4_analysis.md has the history).
#include <fmx.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "fmx_device.h"
#include "fmx_host.h"
#include "fmx_hostpar.h"
#include "fmx_nfa.h"
#include "fmx_regex.h"

namespace fmx {

#ifndef FMX_FTHREADS
#define FMX_FTHREADS 256
#endif
constexpr int kFThreads = FMX_FTHREADS;      // threads of a frontier workgroup
constexpr int kFWaves = kFThreads / 64;       // waves of a workgroup (they share one mailbox)

// The work queue in HBM.  An entry is three 8-byte GRANULES, each written by one aligned agent-scope (write-through)
// store and carrying the tag of its buffer's current generation in its top 16 bits -- the data is its own "ready"
// flag (cdna_hip_programming.md, Guideline 16, form R2), so a wave may take entries that another wave appended
// earlier IN THE SAME LAUNCH: it re-reads a granule until the tag matches.
//   g0 = tag:16 | byte:8 | sp:40        g1 = tag:16 | 0:8 | ep:40        g2 = tag:16 | len:16 | state:32
struct FlowQueue {
  unsigned long long *g0, *g1, *g2;    // kSub slices x 2 buffers x sub_cap entries each
};
// A call's START elements (states = the regexes' firsts, length 0, every row) are not queued: the first launch of a
// call makes them up where it would have read them from slice `s`, position `p` of the queue -- elem[s * cap + p] =
// state | its byte << 40, laid out slice by slice when the batch was made resident (round 3 wrote them to the queue
// in a launch of their own: 12 us of a 0.4 ms call).
struct StartSrc {                     // (kept in FrontierCtl and read there when a wave takes a batch: no registers held over the rounds)
  const unsigned long long *elem;
  unsigned long long cap;             // entries per slice in `elem`
  unsigned long long ep;              // what a start element carries as its interval's end: n, or 0 = the empty k-mer code
};
constexpr uint32_t kMaxLen = 0xFFFFu;
constexpr uint64_t kMaxRows = 1ull << 40;     // sp / ep fields of a granule

// The queue and the result buffer are cut into kSub slices with their own counters, every slice on its own
// 128-byte line: a single tail cannot take the appends of a whole launch (same-address device atomics complete at
// ~100 per microsecond).  A wave appends to slice (wave + number of its earlier appends) % kSub, so slices stay
// balanced even when one wave produces everything.
// Every slice has TWO linear buffers.  Appends go to buffer `wsel`; takers empty the other one first.  Between
// launches (k_frontier_advance) a buffer that has been emptied is rewound -- tail = head = 0, next tag -- and
// becomes the slice's write buffer, so the memory a search needs follows the frontier's width, not its total work.
constexpr uint32_t kSub = 64;
struct alignas(128) SliceCtl {
  unsigned long long tail[2];      // entries appended (agent-scope atomic adds)
  unsigned long long head[2];      // entries taken (atomic add on the buffer that is not written; CAS on the other)
  uint32_t tag[2];                 // generation tag of each buffer: 1..65535, 0 = never written
  uint32_t wsel;                   // the buffer this launch appends to
  uint32_t pad_[21];
};
struct alignas(128) PaddedCount {
  unsigned long long v;
  unsigned long long pad[15];
};
__host__ __device__ inline uint32_t next_tag(uint32_t t) { return t % 0xFFFFu + 1u; }

struct FrontierCtl {     // device-resident counters
  SliceCtl q[kSub];
  PaddedCount res_count[kSub];
  unsigned long long overflow;     // bit 0: queue, bit 1: results, bit 2: an appended entry never became readable
  unsigned long long truncated;    // some element was not expanded because its follows would have len >= max_len
  uint32_t max_len;
  uint32_t deep_len;               // elements of looping states at least this long jump the wave's queue (express pool)
  uint32_t fresh;                  // this chain of launches begins a call (k_frontier_reset): its first launch makes up the start elements
  unsigned long long left;         // entries queued when the launch began (k_frontier_reset / k_frontier_advance): 0 = nothing to do
  StartSrc start;                  // this call's start elements (k_frontier_reset)
};
struct FrontierSummary { // what the host reads after a chain of launches (k_frontier_advance)
  unsigned long long left;         // entries still queued
  unsigned long long results;
  unsigned long long overflow;
  unsigned long long truncated;
};

// Agent-scope relaxed accesses to words other workgroups write during the launch: global_load / global_store with
// sc1 (they bypass the CU's L1 and are written through; plain accesses could be served from a stale line).
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
  return __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store((gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FStat {           // wave-uniform sums for the frontier counters (fmx_device.h, slots 3..7)
  uint32_t reqs = 0, writes = 0, emits = 0, reads = 0, recs = 0, ktl = 0;
};

// Exclusive prefix sum over the 64 lanes with DPP adds only (row shifts inside each row of 16, then the row
// totals carried over with row_bcast:15 / row_bcast:31): six VALU instructions, no LDS crossbar round trips
// (the __shfl_up form of round 1 cost seven dependent ds_bpermute per scan).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t x) {
  return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
  uint32_t x = v;
  x = dpp_add<0x111, 0xF>(x);     // row_shr:1
  x = dpp_add<0x112, 0xF>(x);     // row_shr:2
  x = dpp_add<0x114, 0xF>(x);     // row_shr:4
  x = dpp_add<0x118, 0xF>(x);     // row_shr:8
  x = dpp_add<0x142, 0xA>(x);     // row_bcast:15 into rows 1 and 3
  x = dpp_add<0x143, 0xC>(x);     // row_bcast:31 into rows 2 and 3
  total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  return x - v;
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
  return ((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}

constexpr uint32_t kPoolSmall = 6;      // follow lists with up to this many pushed entries go through the pool
#ifdef FMX_WAVELOG
// Diagnostic build only (tools/wave_timeline.py): per launch and wave {start, first round, end, rounds} in the
// constant 100 MHz clock, so that the occupancy of the wave slots over a launch can be drawn.
constexpr uint32_t kLogPasses = 16, kLogWaves = 1u << 15;
__device__ unsigned long long g_wavelog[kLogPasses][kLogWaves][8];
// launch 0, per wave and round (first 128): elements held | pool entries << 8 | deepest held length << 20 | narrow << 28 | express << 29
__device__ unsigned int g_wavetrace[kLogWaves][128];
// with -DFMX_PHASELOG=<round>: cycles (s_memtime) a wave spends in the four phases of its rounds from that round on
// (take | stage + issue | wait + ranks | bookkeeping), rounds counted, in g_phaselog[wave][0..4] of launch 0
__device__ unsigned long long g_phaselog[kLogWaves][8];
#endif
__device__ __forceinline__ void pool_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// One launch.  Every wave works until it finds nothing more to take (or its round budget ends): it takes entries
// from the queue, follows them, keeps what they branch into in its pool, and hands pool entries beyond kPoolKeep
// back to the queue, where waves that have run dry find them during the same launch.  No wave ever waits for
// another one: a wave that finds the queue empty a few polls in a row simply ends, and whatever is appended after
// that is the next launch's input.
#ifndef FMX_FWAVES
#define FMX_FWAVES 4
#endif
#ifndef FMX_FBATCH
#define FMX_FBATCH 4
#endif
#ifndef FMX_TAKE_OLDEST
#define FMX_TAKE_OLDEST 32
#endif
constexpr uint32_t kTakeOldest = FMX_TAKE_OLDEST;   // pool sizes up to which idle lanes take the oldest entries
#ifndef FMX_POOL_KEEP
#define FMX_POOL_KEEP 192
#endif
#ifndef FMX_GRAB
#define FMX_GRAB 64
#endif
constexpr uint32_t kPoolKeep = FMX_POOL_KEEP;   // pool entries a wave keeps for itself; the older ones go to the queue
constexpr uint32_t kGrab = FMX_GRAB;            // entries taken from the queue at a time
#ifndef FMX_IDLE_LOOKS
#define FMX_IDLE_LOOKS 3
#endif
#ifndef FMX_NARROW
#define FMX_NARROW 1
#endif
#ifndef FMX_DEEP_PRIO
#define FMX_DEEP_PRIO 1
#endif

#ifndef FMX_DEEP_SLACK
#define FMX_DEEP_SLACK 4u
#endif
constexpr uint32_t kIdleLooks = FMX_IDLE_LOOKS; // looks that find nothing before a wave without work ends
constexpr uint32_t kTagLimit = 60000;          // host-side bound on a buffer's generation tag before everything is zeroed
constexpr uint32_t kTagSpins = 1u << 16;        // re-reads of a reserved entry before the wave gives up (an error)
constexpr uint32_t kPool64 = 256, kPool64Mask = kPool64 - 1;
struct Pool64 {          // 24 bytes per entry, 6 KiB per wave
  uint32_t state[kPool64];
  uint32_t meta[kPool64];
  uint64_t sp[kPool64];
  uint64_t ep[kPool64];
};
constexpr uint32_t kRes64 = 64;
struct ResStage64 {
  fmx_result r[kRes64];
};
// The EXPRESS pool of a wave: follows pushed deep into a match (FrontierCtl::deep_len: past the length at which an
// interval has narrowed to a row).  Few elements get there, and those that keep branching there (a starred class that
// keeps matching) are links of a chain of dependent steps that may run to the longest match explored -- the launch's
// critical path -- while the newest-first pool serves whatever was pushed last: under the push traffic of a wave's
// busy phase a chain's entries sank at length 9 and were only stepped again when the pool had drained (the last waves
// of a launch did 30 rounds of bulk work at lengths <= 8, then one level of one chain per round up to length 64:
// profiles/r03_c4_wave_trace_before.txt).  Idle lanes take express entries before anything else.
#ifndef FMX_EXPRESS
#define FMX_EXPRESS 0      // measured on C4 (profiles/r03_c4_k5_experiments.md): no gain -- a chain starts late because its ANCESTOR at
                           // length <= 8 is one of a wave's few thousand bulk elements, not because its deep entries wait
#endif
constexpr uint32_t kXp = FMX_EXPRESS;
struct Express {
  uint32_t state[kXp ? kXp : 1];
  uint32_t meta[kXp ? kXp : 1];
  uint64_t sp[kXp ? kXp : 1];
  uint64_t ep[kXp ? kXp : 1];
};
struct Xchg {            // a round's intervals on their way to the lane groups and back, by element (= lane) number
  uint64_t sp[64];
  uint64_t ep[64];
  uint32_t key[64];      // slot | byte << 16, 0xFFFFFFFF: this element has no rank query this round
};
constexpr uint32_t kNoQuery = 0xFFFFFFFFu;
// A workgroup's MAILBOX: a wave with a long backlog leaves some of its oldest pool entries here for the three other
// waves of its workgroup, which look here first when they run dry -- the launch waits for its longest wave, and this
// shares a long backlog at the price of LDS accesses (handing over through the HBM queue costs more than it
// balances).  Guarded by a lock that one lane takes with an LDS compare-and-swap; the holder only copies entries.
// Every wave empties the mailbox before it ends, so nothing is left behind.
#ifndef FMX_MAIL
#define FMX_MAIL 80
#endif
#ifndef FMX_MAIL_KEEP
#define FMX_MAIL_KEEP 128
#endif
#ifndef FMX_MAIL_GIVE
#define FMX_MAIL_GIVE 32
#endif
constexpr uint32_t kMail = FMX_MAIL, kMailKeep = FMX_MAIL_KEEP, kMailGive = FMX_MAIL_GIVE;
struct Mailbox {
  uint32_t lock;           // 0 free, 1 held
  uint32_t n;              // entries held (a stack)
  uint32_t state[kMail ? kMail : 1];
  uint32_t meta[kMail ? kMail : 1];
  uint64_t sp[kMail ? kMail : 1];
  uint64_t ep[kMail ? kMail : 1];
};

template <bool WIDE, uint32_t LAYOUT>
__device__ __forceinline__ void frontier_pass(const DevIndex &ix, const KTab &kt, const NfaTables &nfa, const FlowQueue &fq, uint32_t j,
                                              uint32_t max_rounds, uint64_t sub_cap, fmx_result *__restrict__ res,
                                              uint64_t seg_cap, FrontierCtl *__restrict__ ctl, uint32_t *__restrict__ rcnt,
                                              unsigned long long *__restrict__ counters) {
  constexpr int G = Lay<LAYOUT>::G;
  constexpr uint32_t EPS = 64 / G;           // elements per sub-round
  constexpr int B = LAYOUT == kLayoutBytes ? 2 : FMX_FBATCH;     // sub-rounds whose lines are requested before any is consumed
#ifdef FMX_WAVELOG
  const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long wl_t1 = 0;
#endif
  const uint32_t lane = __lane_id();
  const uint32_t w = uni((blockIdx.x * kFThreads + threadIdx.x) >> 6);      // this wave (kept in scalar registers)
  // After a queue overflow entries are missing: later launches of the chain do nothing (the host reports it).
  if (ctl->left == 0 || (ctl->overflow & 1ull)) return;      // uniform over the grid
  // which buffer of a slice is written, and the buffers' tags, do not change during a launch
  __shared__ uint32_t s_wsel[kSub], s_tags[kSub];            // tags: tag[0] | tag[1] << 16
  if (threadIdx.x < kSub) {
    const SliceCtl &q = ctl->q[threadIdx.x];
    s_wsel[threadIdx.x] = q.wsel;
    s_tags[threadIdx.x] = q.tag[0] | (q.tag[1] << 16);
  }
  __shared__ Mailbox s_mail;
  __shared__ uint16_t s_slot[256];
  __shared__ Pool64 s_pool[kFThreads / 64];
  __shared__ ResStage64 s_res[kFThreads / 64];
  __shared__ Xchg s_xc[kFThreads / 64];
  __shared__ Express s_xp[kFThreads / 64];
  __shared__ const uint4 *s_lvl[16];         // the k-mer table's levels (picked by an element's length at run time)
  for (int c = threadIdx.x; c < 256; c += blockDim.x) s_slot[c] = ix.slot[c];
  if (threadIdx.x == 0) { s_mail.lock = 0; s_mail.n = 0; }
  if (threadIdx.x < 16) s_lvl[threadIdx.x] = kt.k ? kt.level_dev[threadIdx.x] : nullptr;
  __syncthreads();
  Pool64 &pl = s_pool[threadIdx.x >> 6];
  ResStage64 &rs = s_res[threadIdx.x >> 6];
  Xchg &xc = s_xc[threadIdx.x >> 6];
  Express &xp = s_xp[threadIdx.x >> 6];
  uint32_t xn = 0;                           // entries in the express pool (wave-uniform)
  const uint32_t deep_len = ctl->deep_len;
  uint32_t rs_n = 0;                         // wave-uniform
  const LaneConst lc = lane_const<G>();
  const uint32_t grp = lane / G;             // this lane's group: it serves element r * EPS + grp in sub-round r
  const uint32_t max_len = ctl->max_len;
  uint32_t pb = 0, pn = 0, appends = 0, grabs = 0, rounds = 0, idle_looks = 0;     // wave-uniform
  // This wave's share of what the launch found queued: slice w % kSub's buffer that is not written now is dealt out
  // evenly to the waves of that class, one contiguous chunk each, read without atomics (nobody else touches it).
  const uint32_t nw = gridDim.x * (kFThreads / 64);
  const uint32_t sub = w % kSub, part = w / kSub;
  const uint32_t class_waves = (nw - sub + kSub - 1) / kSub;
  const uint32_t in_buf = 1u - s_wsel[sub];
  const uint64_t in_off = ((uint64_t)sub * 2 + in_buf) * sub_cap;
  uint64_t a_next, a_end;
  {
    const SliceCtl &q = ctl->q[sub];
    const uint64_t hd = q.head[in_buf], tl = q.tail[in_buf] < sub_cap ? q.tail[in_buf] : sub_cap;
    const uint64_t cnt = tl > hd ? tl - hd : 0, chunk = (cnt + class_waves - 1) / class_waves;
    a_next = hd + (uint64_t)part * chunk;
    a_end = a_next + chunk < tl ? a_next + chunk : tl;
    if (a_next > a_end) a_next = a_end;
    a_next = uni64(a_next);
    a_end = uni64(a_end);
  }
  bool prio_up = false;                      // wave-uniform: running at raised issue priority (FMX_DEEP_PRIO)
  bool have = false;                         // the element this lane holds
  uint32_t state = 0, meta = 0;
  uint64_t sp = 0, ep = 0;
  // the held element's state record; it stays while the element walks a literal stretch (meta bits 24..31)
  uint4 ra = make_uint4(0, 0, 0, 0);         // fol_off, cnt_c_emit, f[0], f[1]
  uint4 rb = make_uint4(0, 0, 0, 0);         // f[2], f[3], fc, regex
  uint32_t n_reqs = 0, n_recs = 0, n_ktl = 0, n_jl = 0, n_writes = 0, n_emits = 0, n_reads = 0, stepped = 0, trunc = 0;

  // room for `cnt` entries in some slice's write buffer: slice, buffer, tag and first index (wave-uniform)
  struct Slot { uint64_t first; uint32_t tag; unsigned long long at; };
  auto reserve = [&](uint32_t cnt) -> Slot {
    const uint32_t so = (w + appends++) % kSub;
    const uint32_t buf = s_wsel[so];
    Slot s;
    s.tag = (s_tags[so] >> (16u * buf)) & 0xFFFFu;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&ctl->q[so].tail[buf], (unsigned long long)cnt);
    s.at = __shfl(base, 0, 64);
    s.first = ((uint64_t)so * 2 + buf) * sub_cap;
    return s;
  };
  auto put = [&](const Slot &s, unsigned long long at, uint32_t st, uint32_t mt, uint64_t esp, uint64_t eep) {
    if (at < sub_cap) {
      const unsigned long long tg = (unsigned long long)s.tag << 48;
      const uint64_t i = s.first + at;
      st_agent(fq.g0 + i, tg | ((unsigned long long)((mt >> 16) & 0xFFu) << 40) | esp);
      st_agent(fq.g1 + i, tg | eep);
      st_agent(fq.g2 + i, tg | ((unsigned long long)(mt & 0xFFFFu) << 32) | st);
    } else {
      atomicOr(&ctl->overflow, 1ull);
    }
  };
  // the `cnt` oldest pool entries go to the queue
  auto spill = [&](uint32_t cnt) {
    if (!cnt) return;
    pool_sync();
    const Slot s = reserve(cnt);
    for (uint32_t i = lane; i < cnt; i += 64) {
      const uint32_t idx = (pb + i) & kPool64Mask;
      put(s, s.at + i, pl.state[idx], pl.meta[idx], pl.sp[idx], pl.ep[idx]);
    }
    pool_sync();
    pb = uni((pb + cnt) & kPool64Mask);
    pn = uni(pn - cnt);
    n_writes += cnt;
  };
  auto to_pool = [&](uint32_t cnt, unsigned long long e0, unsigned long long e1, unsigned long long e2) {
    if (lane < cnt) {
      const uint32_t idx = (pb + pn + lane) & kPool64Mask;
      pl.state[idx] = (uint32_t)e2;
      pl.meta[idx] = (uint32_t)((e2 >> 32) & 0xFFFFu) | ((uint32_t)((e0 >> 40) & 0xFFu) << 16);
      pl.sp[idx] = e0 & (kMaxRows - 1);
      pl.ep[idx] = e1 & (kMaxRows - 1);
    }
    pn = uni(pn + cnt);
    n_reads += cnt;
    pool_sync();
  };
  // The next entries of the wave's share enter the pool (wave-uniform).  They were written before this launch began.
  auto take_batch = [&]() -> bool {
    if (a_next >= a_end) return false;
    const uint64_t left = a_end - a_next;
    uint32_t room = kPool64 - 64u - pn;      // the push phase wants 64 free entries afterwards
    if (room > 64u) room = 64u;
    const uint32_t cnt = left < room ? (uint32_t)left : room;
    unsigned long long e0 = 0, e1 = 0, e2 = 0;
    if (lane < cnt) {
      if (j == 0 && ctl->fresh != 0) {                 // uniform over the grid: a call's first launch, its share is start elements
        const StartSrc ss = ctl->start;
        const unsigned long long se = ss.elem[(uint64_t)sub * ss.cap + a_next + lane];
        e0 = se & (0xFFull << 40);                       // the state's byte; sp = 0
        e1 = ss.ep;
        e2 = se & 0xFFFFFFFFull;                         // len = 0
      } else {
        const uint64_t i = in_off + a_next + lane;
        e0 = fq.g0[i]; e1 = fq.g1[i]; e2 = fq.g2[i];
      }
    }
    a_next += cnt;
    to_pool(cnt, e0, e1, e2);
    return true;
  };
  // A wave with nothing left looks for entries that other waves appended during THIS launch (wave-uniform; returns
  // how many it took): four slices per look, one round trip -- lanes 0..3 read the written buffer's tail and head of
  // one slice each -- then a compare-and-swap on the first head that has something below its tail, and the reserved
  // entries are re-read until they carry the buffer's tag (the wave that reserved them writes them at once).
  auto steal = [&]() -> uint32_t {
    const uint32_t s4 = (w + 4u * grabs++) & 63u;
    unsigned long long tail_w = 0, head_w = 0;
    if (lane < 4) {
      const uint32_t sl = (s4 + lane) & 63u;
      SliceCtl *const q = ctl->q + sl;
      const uint32_t ws = s_wsel[sl];
      tail_w = ld_agent(&q->tail[ws]);
      head_w = ld_agent(&q->head[ws]);
      if (tail_w > sub_cap) tail_w = sub_cap;          // appends past the capacity were dropped (overflow is flagged)
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(tail_w > head_w);
    if (!m) return 0;
    const uint32_t src = (uint32_t)__builtin_ctzll(m);
    const uint32_t s = (s4 + src) & 63u, ws = s_wsel[s];
    tail_w = __shfl(tail_w, (int)src, 64);
    head_w = __shfl(head_w, (int)src, 64);
    const uint32_t cnt = tail_w - head_w < kGrab ? (uint32_t)(tail_w - head_w) : kGrab;
    unsigned long long old = 0;
    if (lane == 0) old = atomicCAS(&ctl->q[s].head[ws], head_w, head_w + cnt);
    if (uni64(old) != head_w) return 0;                // another wave was faster
    const unsigned long long tag = (s_tags[s] >> (16u * ws)) & 0xFFFFu;
    const uint64_t i = ((uint64_t)s * 2 + ws) * sub_cap + head_w + lane;
    unsigned long long e0 = 0, e1 = 0, e2 = 0;
    bool ok = true;
    for (uint32_t spins = 0;; spins++) {
      if (lane < cnt) {
        e0 = ld_agent(fq.g0 + i); e1 = ld_agent(fq.g1 + i); e2 = ld_agent(fq.g2 + i);
        ok = (e0 >> 48) == tag && (e1 >> 48) == tag && (e2 >> 48) == tag;
      }
      if (!__builtin_amdgcn_ballot_w64(!ok)) break;
      if (spins >= kTagSpins) { atomicOr(&ctl->overflow, 4ull); return 0; }
      __builtin_amdgcn_s_sleep(2);
    }
    to_pool(cnt, e0, e1, e2);
    return cnt;
  };
  // staged results go to the result slices: one returning atomic per flush, not per round with a result
  auto flush_results = [&]() {
    if (!rs_n) return;
    pool_sync();
    const uint32_t so = (w + appends++) % kSub;
    unsigned long long rbase = 0;
    if (lane == 0) rbase = atomicAdd(&ctl->res_count[so].v, (unsigned long long)rs_n);
    rbase = __shfl(rbase, 0, 64);
    if (lane < rs_n) {
      const unsigned long long at = rbase + lane;
      if (at < seg_cap) { res[(uint64_t)so * seg_cap + at] = rs.r[lane]; atomicAdd(&rcnt[rs.r[lane].regex], 1u); }   // per-regex counts: the grouping starts from them
      else atomicOr(&ctl->overflow, 2ull);
    }
    pool_sync();
    n_emits += rs_n;
    rs_n = 0;
  };

  Mailbox &mb = s_mail;
  auto mail_lock = [&]() {
    if (lane == 0)
      while (atomicCAS(&mb.lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  };
  auto mail_unlock = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) atomicExch(&mb.lock, 0u);
  };
  // up to `room` mailbox entries enter the pool (wave-uniform; returns how many)
  auto mail_take = [&](uint32_t room) -> uint32_t {
    if (!kMail || !room) return 0;
    pool_sync();
    mail_lock();
    const uint32_t have_n = uni(*(volatile uint32_t *)&mb.n);
    const uint32_t cnt = have_n < room ? have_n : room;
    if (lane < cnt) {
      const uint32_t src = have_n - cnt + lane, idx = (pb + pn + lane) & kPool64Mask;
      pl.state[idx] = mb.state[src]; pl.meta[idx] = mb.meta[src]; pl.sp[idx] = mb.sp[src]; pl.ep[idx] = mb.ep[src];
    }
    if (lane == 0 && cnt) *(volatile uint32_t *)&mb.n = have_n - cnt;
    mail_unlock();
    pn = uni(pn + cnt);
    pool_sync();
    return cnt;
  };
  // up to kMailGive of the pool's oldest entries go to the mailbox (wave-uniform)
  auto mail_give = [&]() {
    pool_sync();
    mail_lock();
    const uint32_t have_n = uni(*(volatile uint32_t *)&mb.n);
    uint32_t cnt = kMail - have_n < kMailGive ? kMail - have_n : kMailGive;
    if (cnt > pn) cnt = pn;
    if (lane < cnt) {
      const uint32_t dst = have_n + lane, idx = (pb + lane) & kPool64Mask;
      mb.state[dst] = pl.state[idx]; mb.meta[dst] = pl.meta[idx]; mb.sp[dst] = pl.sp[idx]; mb.ep[dst] = pl.ep[idx];
    }
    if (lane == 0 && cnt) *(volatile uint32_t *)&mb.n = have_n + cnt;
    mail_unlock();
    pb = uni((pb + cnt) & kPool64Mask);
    pn = uni(pn - cnt);
    pool_sync();
  };
  auto mail_n = [&]() -> uint32_t { return kMail ? uni(*(volatile uint32_t *)&mb.n) : 0u; };

#ifdef FMX_WAVELOG
  wl_t1 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef FMX_PHASELOG
  unsigned long long ph_t = 0, ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph_n = 0;
#define PH_MARK(acc) do { if (rounds >= FMX_PHASELOG) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - ph_t; ph_t = now_; } } while (0)
#else
#define PH_MARK(acc) do {} while (0)
#endif
  for (;;) {
#ifdef FMX_PHASELOG
    ph_t = __builtin_amdgcn_s_memtime();
#endif
    // ---- NARROW rounds.  A wave that holds no more elements than it has lane groups (the thin end of a launch: a
    // few chains of dependent steps, which are the launch's critical path) keeps them in the groups' first lanes and
    // steps each in its own group, like k_search4 does: the interval goes to the group's lanes by DPP, the ranks come
    // back the same way -- no exchange area, no LDS round trips around the memory latency.
    if (kXp && xn) {                         // express entries first, newest first
      const unsigned long long idle = __builtin_amdgcn_ballot_w64(!have);
      if (idle) {
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
        const uint32_t take = n_idle < xn ? n_idle : xn;
        const uint32_t rank = (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
        if (!have && rank < take) {
          const uint32_t idx = xn - 1u - rank;
          state = xp.state[idx]; meta = xp.meta[idx]; sp = xp.sp[idx]; ep = xp.ep[idx];
          have = true;
        }
        xn = uni(xn - take);
        pool_sync();
      }
    }
    bool narrow = false;
#if FMX_NARROW
    {
      const unsigned long long havem = __builtin_amdgcn_ballot_w64(have);
      const uint32_t n_have = (uint32_t)__builtin_popcountll(havem);
      narrow = n_have + pn + xn <= (uint32_t)(64 / G) && n_have + pn != 0u && xn == 0u && a_next >= a_end;
      if (narrow) {
        constexpr unsigned long long kLeaders = G == 4 ? 0x1111111111111111ull : 0x0101010101010101ull;
        const unsigned long long stray = havem & ~kLeaders;       // elements held by other lanes go through the pool
        if (stray) {
          if (have && lc.t != 0u) {
            const uint32_t idx = (pb + pn + (uint32_t)__builtin_popcountll(stray & ((1ull << lane) - 1ull))) & kPool64Mask;
            pl.state[idx] = state; pl.meta[idx] = meta & 0x00FFFFFFu; pl.sp[idx] = sp; pl.ep[idx] = ep;   // the stretch context stays behind
            have = false;
          }
          pn = uni(pn + (uint32_t)__builtin_popcountll(stray));
          pool_sync();
        }
        const unsigned long long idle_l = ~__builtin_amdgcn_ballot_w64(have) & kLeaders;
        if (idle_l && pn) {
          const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle_l);
          const uint32_t take = n_idle < pn ? n_idle : pn;
          const uint32_t rank = (uint32_t)__builtin_popcountll(idle_l & ((1ull << lane) - 1ull));
          if (!have && lc.t == 0u && rank < take) {
            const uint32_t idx = (pb + rank) & kPool64Mask;       // a small pool: oldest first
            state = pl.state[idx]; meta = pl.meta[idx]; sp = pl.sp[idx]; ep = pl.ep[idx];
            have = true;
          }
          pb = uni((pb + take) & kPool64Mask);
          pn = uni(pn - take);
          pool_sync();
        }
      }
    }
#endif
    // ---- lanes without an element take the newest pool entries; a pool that cannot feed them is refilled from the
    // wave's share, and a wave with nothing left at all looks for entries appended during this launch
    if (!narrow) {
      const unsigned long long idle = __builtin_amdgcn_ballot_w64(!have);
      if (idle) {
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
        // what a sibling wave left for this one -- looked for only by a wave that cannot fill half its lanes (every look
        // is an LDS round trip on the round's critical path)
        if (pn < n_idle && n_idle > 32u && mail_n()) mail_take(64u);
        if (pn < n_idle && !take_batch() && pn == 0 && n_idle == 64u) steal();
        if (pn) {
          const uint32_t take = n_idle < pn ? n_idle : pn;
          const uint32_t rank = (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
          // A small pool is emptied from its OLD end: the oldest entries are the shallowest, i.e. the ones with the
          // longest way still to go, and a wave's last rounds are the launch's critical path.  A large pool is emptied
          // from its new end (depth first), which keeps it from growing.
          const bool oldest = pn <= kTakeOldest;
          if (!have && rank < take) {
            const uint32_t idx = (oldest ? pb + rank : pb + pn - 1 - rank) & kPool64Mask;
            state = pl.state[idx]; meta = pl.meta[idx]; sp = pl.sp[idx]; ep = pl.ep[idx];
            have = true;
          }
          if (oldest) pb = uni((pb + take) & kPool64Mask);
          pn = uni(pn - take);
          pool_sync();
        }
      }
    }
    if (!__builtin_amdgcn_ballot_w64(have)) {      // nothing held, pool empty, share used up, nothing found
      if (++idle_looks > kIdleLooks) break;
      for (uint32_t z = 0; z < (1u << (idle_looks < 3u ? idle_looks : 3u)); z++) __builtin_amdgcn_s_sleep(16);      // 1, 2, 4 .. x ~0.5 us: another wave may be about to hand work over
      continue;
    }
    idle_looks = 0;
#ifdef FMX_WAVELOG
    if (j == 0 && w < kLogWaves && rounds < 128u) {
      const uint32_t nh = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(have));
      uint32_t ml = have ? (meta & 0xFFFFu) : 0u;
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)ml, d, 64); ml = o > ml ? o : ml; }
      if (lane == 0) g_wavetrace[w][rounds] = nh | (pn << 8) | ((ml > 255u ? 255u : ml) << 20) | ((narrow ? 1u : 0u) << 28) | ((xn > 7u ? 7u : xn) << 29);
    }
#endif
#if FMX_DEEP_PRIO
    // A wave that holds an element which has advanced in (nearly) every round since the launch began is on the
    // launch's critical path -- a chain of dependent steps as long as the deepest match explored: it gets the SIMD's
    // issue slots ahead of its neighbours, whose backlog is throughput work.
    {
      const uint32_t floor_len = (rounds > 16u ? rounds : 16u);
      const bool deep_now = __builtin_amdgcn_ballot_w64(have && (meta & 0xFFFFu) + FMX_DEEP_SLACK >= floor_len) != 0ull;
      if (deep_now != prio_up) {
        if (deep_now) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
        prio_up = deep_now;
      }
    }
#endif
    PH_MARK(ph0);
    // ---- every element's state record and its step's lines are requested together
    const uint32_t run = meta >> 24;          // > 0: inside a literal stretch whose bytes the held record carries
    const uint32_t c = (meta >> 16) & 0xFFu, len = meta & 0xFFFFu;
    const uint16_t slot = s_slot[c];
    const uint64_t cfc = ix.cf[c];            // C[c]: needed only when the ranks are back, so its load rides along
    if (have && run == 0) {
      const uint4 *rp = reinterpret_cast<const uint4 *>(nfa.st + state);
      ra = rp[0];
      rb = rp[1];
      n_recs++;
    }
    // An element shorter than the k-mer table's K carries its k-mer code instead of an interval (ep == 0 marks it,
    // sp = the code): its step is a lookup in the next level of the table.
    bool cm = have && kt.k != 0 && ep == 0;
    uint32_t ncode = 0;
    bool from_tab = false;
    uint4 ent = make_uint4(0, 0, 0, 0);
    if (cm) {
      if (slot < kSlotEof) {
        ncode = (uint32_t)sp * kt.sigma + slot;
        ent = s_lvl[len][ncode];
        from_tab = true;
        n_ktl++;
      } else if (slot == kSlotNone) {
        sp = 0; ep = 0;                                 // a byte that does not occur: the interval is empty
      } else {                                          // the EOF symbol: leave the table, step on the interval itself
        if (len == 0) { sp = 0; ep = ix.n; }
        else {
          const uint4 e0 = s_lvl[len - 1][sp];
          sp = (((uint64_t)e0.y << 32) | e0.x) & ((1ull << 56) - 1);
          ep = ((uint64_t)e0.w << 32) | e0.z;
        }
        cm = false;
      }
    }
    const bool ranked = have && !cm;
    // a rank query is needed unless the symbol is absent / EOF or the element is a start element ...
    const bool step_q = ranked && len != 0 && slot < kSlotEof;
    // ... or holds one row and the handle has a row table: [r, r + 1) steps to [LF r, LF r + 1) if the state's byte is
    // BWT'[r], to nothing otherwise -- one 8-byte load by the element's own lane, no lane group, no exchange slot
    const bool row_q = kt.row1 != nullptr && step_q && (ep - sp) == 1;
    unsigned long long r1e = 0;
    if (row_q) { r1e = kt.row1[sp]; n_jl++; }
    const bool query = step_q && !row_q;
    // The elements that have a query take the exchange slots 0, 1, 2 .. in lane order, so the lane groups serve
    // ceil(queries / (64/G)) sub-rounds, not all G: in a launch's thin end, and in rounds where most elements step
    // through the k-mer table, most sub-rounds have nothing to do and are skipped.
    const unsigned long long qmask = __builtin_amdgcn_ballot_w64(query);
    const uint32_t n_query = (uint32_t)__builtin_popcountll(qmask);
    const uint32_t xslot = (uint32_t)__builtin_popcountll(qmask & ((1ull << lane) - 1ull));      // this lane's slot, if it has a query
    uint64_t nsp = 0, nep = 0;                         // narrow rounds: the group's ranks
    if (narrow) {
      PH_MARK(ph1);
      // the first lane's interval and symbol to its whole group, then the step where the data is
      const uint32_t qk = group_bcast<G, 0>(query ? ((uint32_t)slot | (c << 16)) : kNoQuery);
      uint64_t gsp = group_bcast64<G, 0>(sp), gep = group_bcast64<G, 0>(ep);
      if (qk != kNoQuery) {
        const uint32_t rq = backward_step<WIDE, LAYOUT>(ix, qk >> 16, (uint16_t)(qk & 0xFFFFu), 0ull, lc, gsp, gep);
        if (lc.t == 0) n_reqs += rq;
        nsp = gsp;
        nep = gep;
      }
    } else {
    if (query) {
      xc.sp[xslot] = sp;
      xc.ep[xslot] = ep;
      xc.key[xslot] = (uint32_t)slot | (c << 16);
    }
    if (lane >= n_query) xc.key[lane] = kNoQuery;      // the slots behind the last query
    pool_sync();
    PH_MARK(ph1);
    if constexpr (LAYOUT == kLayoutBytes) {
      // Bytes layout: a rank query is the 128-position block (16 bytes per lane of the octet) plus its checkpoint.
      // As below, sp and ep of a narrowed interval lie in one block: one request pair per element, both ranks from
      // it, a second trip for the elements whose ep lies in the next block.
      constexpr uint32_t kTwo = 1u << 30, kPending = 1u << 31;
      constexpr int BB = 2;
      bool any_two = false;
#pragma unroll
      for (int r0 = 0; r0 < G; r0 += BB) {
        ByteRankReq q[BB];
        uint32_t key[BB], m[BB];      // m: ep's in-block offset | kTwo
#pragma unroll
        for (int b = 0; b < BB; b++) {
          const uint32_t e = (uint32_t)(r0 + b) * EPS + grp;
          key[b] = xc.key[e];
          m[b] = 0;
          q[b].w = make_uint4(0, 0, 0, 0); q[b].chk = 0; q[b].sup = 0; q[b].rem = 0;
          if (key[b] != kNoQuery) {
            const uint64_t esp = xc.sp[e], eep = xc.ep[e];
            q[b] = byte_rank_issue(ix, (uint16_t)(key[b] & 0xFFFFu), esp, lc);
            const bool two = (eep >> 7) != (esp >> 7);
            m[b] = ((uint32_t)eep & 127u) | (two ? kTwo : 0u);
            if (lc.t == 0) n_reqs += two ? 4u : 2u;
          }
        }
#pragma unroll
        for (int b = 0; b < BB; b++) {
          if (key[b] != kNoQuery) {
            const uint32_t e = (uint32_t)(r0 + b) * EPS + grp;
            const uint32_t ec = (key[b] >> 16) & 0xFFu;
            const bool two = (m[b] & kTwo) != 0;
            const uint64_t r1 = byte_rank_finish(q[b], ec, lc);
            q[b].rem = m[b] & 127u;
            const uint64_t r2 = byte_rank_finish(q[b], ec, lc);
            if (lc.t == 0) {
              xc.sp[e] = r1;
              if (two) xc.key[e] |= kPending;      // ep's own block is still to be read
              else xc.ep[e] = r2;
            }
            any_two |= two;
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(any_two)) {
        pool_sync();
#pragma unroll 1
        for (int r = 0; r < G; r++) {
          const uint32_t e = (uint32_t)r * EPS + grp;
          const uint32_t key = xc.key[e];
          if (key != kNoQuery && (key & kPending)) {
            const ByteRankReq q2 = byte_rank_issue(ix, (uint16_t)(key & 0xFFFFu), xc.ep[e], lc);
            const uint64_t r2 = byte_rank_finish(q2, (key >> 16) & 0xFFu, lc);
            if (lc.t == 0) xc.ep[e] = r2;
          }
        }
      }
    } else {
      // One-hot layout.  An interval that has been narrowed by a few steps lies inside one 448-position block, so the
      // common case is ONE line per element: B sub-rounds request their sp-block together (5 registers each: the
      // line and the two in-block boundaries), both ranks come from it, and the few elements whose ep falls into
      // another block are finished in a second trip afterwards (their ep stays in the exchange area meanwhile).
      constexpr uint32_t kTwo = 1u << 30, kActive = 1u << 31, kPending = 1u << 31;
      bool any_two = false;
#pragma unroll
      for (int r0 = 0; r0 < G; r0 += B) {
        uint4 w[B];
        uint32_t m[B];      // m1 | m2 << 9 | kTwo | kActive
#pragma unroll
        for (int b = 0; b < B; b++) {
          const uint32_t e = (uint32_t)(r0 + b) * EPS + grp;
          const uint32_t key = xc.key[e];
          m[b] = 0;
          w[b] = make_uint4(0, 0, 0, 0);
          if (key != kNoQuery) {
            const uint64_t esp = xc.sp[e], eep = xc.ep[e];
            uint32_t b1, m1, b2, m2;
            split448(esp, b1, m1);
            split448(eep, b2, m2);
            w[b] = load_line16(block_addr(ix, key & 0xFFFFu, b1, lc));
            m[b] = m1 | (m2 << 9) | kActive | (b2 != b1 ? kTwo : 0u);
            if (lc.t == 0) n_reqs += b2 != b1 ? 2u : 1u;
          }
        }
#pragma unroll
        for (int b = 0; b < B; b++) {
          if (m[b] & kActive) {
            const uint32_t e = (uint32_t)(r0 + b) * EPS + grp;
            const bool two = (m[b] & kTwo) != 0;
            const uint64_t r1 = rank_finish<WIDE>(w[b], m[b] & 0x1FFu, lc);
            const uint64_t r2 = rank_finish<WIDE>(w[b], (m[b] >> 9) & 0x1FFu, lc);
            if (lc.t == 0) {
              xc.sp[e] = r1;
              if (two) xc.key[e] |= kPending;      // ep's own block is still to be read
              else xc.ep[e] = r2;
            }
            any_two |= two;
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(any_two)) {
        pool_sync();
        uint4 w2[G];
        uint32_t m2s[G];
#pragma unroll
        for (int r = 0; r < G; r++) {
          const uint32_t e = (uint32_t)r * EPS + grp;
          const uint32_t key = xc.key[e];
          m2s[r] = 0xFFFFFFFFu;
          w2[r] = make_uint4(0, 0, 0, 0);
          if (key != kNoQuery && (key & kPending)) {
            uint32_t b2, m2;
            split448(xc.ep[e], b2, m2);
            w2[r] = load_line16(block_addr(ix, key & 0xFFFFu, b2, lc));
            m2s[r] = m2;
          }
        }
#pragma unroll
        for (int r = 0; r < G; r++) {
          if (m2s[r] != 0xFFFFFFFFu) {
            const uint64_t r2 = rank_finish<WIDE>(w2[r], m2s[r], lc);
            if (lc.t == 0) xc.ep[(uint32_t)r * EPS + grp] = r2;
          }
        }
      }
    }
    pool_sync();
    }      // !narrow
    PH_MARK(ph2);
    // the table entries are not needed before this point: keeps their loads in flight beside the rank lines'
    asm volatile("" : "+v"(ent.x), "+v"(ent.y), "+v"(ent.z), "+v"(ent.w));
    asm volatile("" : "+v"(r1e));
    if (have) {
      if (from_tab) {
        sp = (((uint64_t)ent.y << 32) | ent.x) & ((1ull << 56) - 1);
        ep = ((uint64_t)ent.w << 32) | ent.z;
      } else if (ranked) {
        if (len == 0) {       // every start element is (0, n): rank(c, 0) = 0, rank(c, n) = the symbol's count
          sp = cfc;
          ep = (c == 255u) ? ix.n : ix.cf[c + 1];
          if (slot == kSlotNone) ep = sp;
          else if (slot == kSlotEof) ep = sp + 1;
        } else if (slot >= kSlotEof) {                  // absent symbol, or the EOF symbol 0
          const uint64_t r1 = (slot == kSlotEof && sp > ix.eof) ? 1 : 0;
          const uint64_t r2 = (slot == kSlotEof && ep > ix.eof) ? 1 : 0;
          sp = cfc + r1;
          ep = cfc + r2;
        } else if (row_q) {
          sp = r1e & ((1ull << 40) - 1);
          ep = sp + (((uint32_t)(r1e >> 40) & 0xFFu) == c ? 1u : 0u);
        } else if (narrow) {
          sp = cfc + nsp;
          ep = cfc + nep;
        } else {
          sp = cfc + xc.sp[xslot];
          ep = cfc + xc.ep[xslot];
        }
      }
      stepped++;
    }
    const uint32_t cce = ra.y;
    uint32_t nf = 0, len1 = 0;
    bool emit = false;
    if (have && sp < ep) {                             // Some((sp1,ep1)), retree.scala:634
      emit = run == 0 && ((cce >> 24) & 1u) != 0;
      nf = run ? 1u : (cce & 0xFFFFu);
      len1 = len + 1;
      if (nf && len1 >= max_len) { nf = 0; trunc = 1; }
    }
    // what the element and its follows carry on: the code while they are still inside the table, else the interval
    const bool keep_code = from_tab && len1 < kt.k;
    const uint64_t ssp = keep_code ? (uint64_t)ncode : sp, sep = keep_code ? 0ull : ep;
    // ---- results are staged in LDS
    {
      const unsigned long long em = __builtin_amdgcn_ballot_w64(emit);
      if (em) {
        const uint32_t cnt = (uint32_t)__builtin_popcountll(em);
        if (rs_n + cnt > kRes64) flush_results();
        if (emit) {
          fmx_result r;
          r.regex = rb.w;
          r.len = len + 1;
          r.sp = sp;
          r.ep = ep;
          rs.r[rs_n + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = r;
        }
        rs_n = uni(rs_n + cnt);
      }
    }
    // ---- the first follow stays with the lane, the others go to the pool (short lists) or straight to the
    // output queue (a '.' has 253)
    const uint32_t npush = nf ? nf - 1 : 0u;
    {
      uint32_t nsmall = npush <= kPoolSmall ? npush : 0u;
      bool demoted = false;
      if (kXp) {
        // follows pushed deep into a match go to the express pool while it has room
        const uint32_t nx = (nsmall != 0u && len1 >= deep_len) ? nsmall : 0u;
        if (__builtin_amdgcn_ballot_w64(nx != 0u)) {
          uint32_t x_total = 0;
          const uint32_t x_off = wave_excl_scan(nx, x_total);
          if (xn + x_total <= kXp) {
            for (uint32_t q = 0; q < nx; q++) {
              const uint32_t fj = q + 1;
              const uint32_t fst = fj < kInlineFollows ? (fj == 1 ? ra.w : (fj == 2 ? rb.x : rb.y)) : nfa.fol[ra.x + fj];
              const uint32_t fch = fj < kInlineFollows ? ((rb.z >> (8u * fj)) & 0xFFu) : (uint32_t)nfa.fol_c[ra.x + fj];
              const uint32_t idx = xn + x_off + q;
              xp.state[idx] = fst;
              xp.meta[idx] = len1 | (fch << 16);
              xp.sp[idx] = ssp;
              xp.ep[idx] = sep;
            }
            xn = uni(xn + x_total);
            if (nx) nsmall = 0u;
            pool_sync();
          }
        }
      }
      const bool via_express = kXp && npush != 0u && nsmall == 0u && npush <= kPoolSmall;
      if (__builtin_amdgcn_ballot_w64(nsmall != 0)) {
        uint32_t small_total = 0;
        uint32_t my_off = wave_excl_scan(nsmall, small_total);
        if (small_total > 128u) {      // more than the pool can take in one round: lists of 3 and more go the long way
          demoted = nsmall > 2u;
          if (demoted) nsmall = 0;
          my_off = wave_excl_scan(nsmall, small_total);
        }
        if (pn + small_total + 64u > kPool64) {
          const uint32_t need = pn + small_total + 64u - kPool64;
          const uint32_t half = pn < 64u ? pn : 64u;
          const uint32_t out = need > half ? need : half;
          spill(out < pn ? out : pn);
        }
        for (uint32_t q = 0; q < nsmall; q++) {
          const uint32_t fj = q + 1;
          const uint32_t fst = fj < kInlineFollows ? (fj == 1 ? ra.w : (fj == 2 ? rb.x : rb.y)) : nfa.fol[ra.x + fj];
          const uint32_t fch = fj < kInlineFollows ? ((rb.z >> (8u * fj)) & 0xFFu) : (uint32_t)nfa.fol_c[ra.x + fj];
          const uint32_t idx = (pb + pn + my_off + q) & kPool64Mask;
          pl.state[idx] = fst;
          pl.meta[idx] = len1 | (fch << 16);
          pl.sp[idx] = ssp;
          pl.ep[idx] = sep;
        }
        pn = uni(pn + small_total);
        pool_sync();
        // what the wave cannot work off soon goes to the queue, oldest (shallowest) first: waves that ran dry take it
        // what the wave cannot work off soon goes to the queue, oldest (shallowest) first: waves that ran dry take it
        if (pn > kPoolKeep) spill(pn - kPoolKeep / 2 < 64u ? pn - kPoolKeep / 2 : 64u);
        // a backlog is shared with the workgroup's other waves while the mailbox is nearly empty (nobody takes: it
        // fills once and stays)
        if (kMail && pn > kMailKeep && mail_n() <= kMail - kMailGive) mail_give();
      }
      unsigned long long big = __builtin_amdgcn_ballot_w64((npush > kPoolSmall || demoted) && !via_express);
      while (big) {                                     // one long list at a time, written by the whole wave
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t nl = (uint32_t)__builtin_amdgcn_readlane((int)npush, src);
        const uint32_t fo = (uint32_t)__builtin_amdgcn_readlane((int)ra.x, src);
        const uint32_t l1 = (uint32_t)__builtin_amdgcn_readlane((int)len1, src);
        const uint64_t xsp = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(ssp >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)ssp, src);
        const uint64_t xep = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(sep >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)sep, src);
        const Slot sl = reserve(nl);
        for (uint32_t q = lane; q < nl; q += 64)
          put(sl, sl.at + q, nfa.fol[fo + q + 1], l1 | ((uint32_t)nfa.fol_c[fo + q + 1] << 16), xsp, xep);
        n_writes += nl;
      }
    }
    if (have) {
      if (nf) {
        sp = ssp;
        ep = sep;
        // on a literal stretch the next state is state + 1 and its byte comes from the held record:
        // rr[k - 1] with k = states of the stretch still ahead (the chain length when it was just entered)
        const uint32_t k = run ? run : (nf == 1 ? (cce >> 25) & 0xFu : 0u);
        if (k) {
          const uint32_t word = k <= 4u ? ra.w : (k <= 8u ? rb.x : rb.y);
          const uint32_t ch = (word >> (8u * ((k - 1u) & 3u))) & 0xFFu;
          state = run ? state + 1u : ra.z;
          meta = len1 | (ch << 16) | ((k - 1u) << 24);
        } else {
          state = ra.z;
          meta = len1 | ((rb.z & 0xFFu) << 16);
        }
      } else {
        have = false;
      }
    }
    PH_MARK(ph3);
#ifdef FMX_PHASELOG
    if (rounds >= FMX_PHASELOG) ph_n++;
#endif
    if (++rounds >= max_rounds) {
      // ---- out of rounds: everything this wave still holds goes to the output queue
      const unsigned long long held = __builtin_amdgcn_ballot_w64(have);
      if (held) {
        if (have) {
          const uint32_t idx = (pb + pn + (uint32_t)__builtin_popcountll(held & ((1ull << lane) - 1ull))) & kPool64Mask;
          pl.state[idx] = state; pl.meta[idx] = meta & 0x00FFFFFFu; pl.sp[idx] = sp; pl.ep[idx] = ep;   // the stretch context stays behind
        }
        pn = uni(pn + (uint32_t)__builtin_popcountll(held));
      }
      if (kXp && xn) {                       // express entries leave through the pool
        if (pn + xn + 64u > kPool64) spill(pn < 64u ? pn : 64u);
        if (lane < xn) {
          const uint32_t idx = (pb + pn + lane) & kPool64Mask;
          pl.state[idx] = xp.state[lane]; pl.meta[idx] = xp.meta[lane]; pl.sp[idx] = xp.sp[lane]; pl.ep[idx] = xp.ep[lane];
        }
        pn = uni(pn + xn);
        xn = 0;
        pool_sync();
      }
      while (pn) spill(pn < 64u ? pn : 64u);
      while (take_batch()) spill(pn);        // carried over, not consumed here
      while (mail_n()) { mail_take(64u); spill(pn); }      // nothing stays in the mailbox when its waves are gone
      break;
    }
  }
  flush_results();
#ifdef FMX_PHASELOG
  if (lane == 0 && j == 0 && w < kLogWaves) { unsigned long long *e = g_phaselog[w]; e[0] = ph0; e[1] = ph1; e[2] = ph2; e[3] = ph3; e[4] = ph_n; }
#endif
#ifdef FMX_WAVELOG
  {
    const unsigned long long st_all = wave_sum((unsigned long long)stepped);
    if (lane == 0 && j < kLogPasses && w < kLogWaves) {
      unsigned long long *e = g_wavelog[j][w];
      e[0] = wl_t0; e[1] = wl_t1; e[2] = __builtin_amdgcn_s_memrealtime(); e[3] = rounds | (st_all << 32);
      e[4] = (unsigned long long)n_writes | ((unsigned long long)n_reads << 32);      // queue entries appended | read (lane 0's counts)
      e[5] = (unsigned long long)appends | ((unsigned long long)grabs << 32);         // reservations in the queue / result slices | looks for others' entries
    }
  }
#endif
  if (trunc) atomicOr(&ctl->truncated, 1ull);
  counters_add(counters, 2ull * stepped, stepped, 0);
  counters_add_frontier(counters, n_reqs, lane == 0 ? n_writes : 0u, lane == 0 ? n_emits : 0u, stepped,
                        lane == 0 ? n_reads : 0u, n_recs);
  {
    const unsigned long long lookups = wave_sum((unsigned long long)n_ktl);
    if (lane == 0 && lookups) atomicAdd(counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride + 9, lookups);
    const unsigned long long rows = wave_sum((unsigned long long)n_jl);
    if (lane == 0 && rows) atomicAdd(counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride + 11, rows);
  }
}

// 128 registers: 4 waves per SIMD = 256 elements in flight per SIMD.  `j` = the launch's number in its chain (diagnostics).
template <bool WIDE, uint32_t LAYOUT>
__global__ __launch_bounds__(kFThreads, FMX_FWAVES) void k_frontier(DevIndex ix, KTab kt, NfaTables nfa, FlowQueue fq, uint32_t j,
                                                            uint32_t max_rounds, uint64_t sub_cap,
                                                            fmx_result *__restrict__ res, uint64_t seg_cap,
                                                            FrontierCtl *__restrict__ ctl, uint32_t *__restrict__ rcnt,
                                                            unsigned long long *__restrict__ counters) {
  frontier_pass<WIDE, LAYOUT>(ix, kt, nfa, fq, j, max_rounds, sub_cap, res, seg_cap, ctl, rcnt, counters);
}

// result groups the device leaves to the host (k_res_sort)
constexpr uint32_t kSmallGroup = 12;
constexpr uint32_t kBigMax = 16384;
constexpr uint32_t kMidGroup = 1024;   // groups up to this size are ordered by a workgroup in LDS (k_res_sort's second phase)
struct BigGroups {
  uint32_t n;                    // groups of more than kMidGroup results: left to the host
  uint32_t done;                 // workgroups of k_res_sort that have finished (the last one reports the totals)
  uint32_t total;                // results of the call (k_res_export reads it)
  uint32_t pad_;
  uint32_t ent[2 * kBigMax];     // (first result, count) of each such group
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

// What the host needs to fetch the grouped results: how many there are and how many groups were left unsorted
// (written to pinned host memory by the grouping's last kernel, k_res_export).
struct GroupTotals {
  uint32_t n_results;
  uint32_t n_big;
};
// Where the grouped results go when the caller's buffers are page-locked (fmx_host_alloc): the device writes them
// there itself, behind the grouping and before the host's one synchronisation.  The struct lives in pinned host
// memory; the host fills it before it starts the launches (null pointers: the host copies after the synchronisation).
struct ExportDst {
  fmx_result *out;
  unsigned long long cap;
  uint32_t *per;
  // and what a call's first chain of launches starts from (k_frontier_reset reads it): the arguments of a captured
  // graph are fixed, the call's own values travel through this page-locked struct
  uint32_t max_len;
  uint32_t fresh;        // 1: this chain begins a call (reset the queue, write the start elements); 0: it continues one
  uint32_t direct;       // 1: `out` is device memory (fmx_regex_batch_match_dev): the grouping kernels scatter and order the
                         // results right there, no export copy of them
  uint32_t deep_len;     // FrontierCtl::deep_len of this call
};
// where the grouping works: the batch's own result buffer, or straight in the caller's device memory
__device__ __forceinline__ fmx_result *group_out(const ExportDst *dst, fmx_result *own, uint64_t own_cap, uint64_t &cap) {
  const ExportDst d = *dst;
  if (d.direct) { cap = d.cap < own_cap ? d.cap : own_cap; return d.out; }
  cap = own_cap;
  return own;
}
// A batch of compiled regexes made resident on one device: concatenated Glushkov tables plus
// the level-0 frontier (root.firsts x (0, 0, n), retree.scala:576).  Reusable across calls.
struct RegexBatch {
  int device = 0;
  size_t k = 0;
  uint64_t n_index = 0;
  uint64_t index_serial = 0;           // the fmx_index this batch was made for (Index::serial): its pointers are inside the
                                       // captured level chain, so no other handle may match against the batch
  size_t n_first = 0, n_states = 0, n_fol = 0;
  std::vector<uint32_t> start_final;   // DFA engines whose start state is final: result (len 0, 0, n)
  DevMem mem;
  // scratch reused across matches of this batch (one match at a time per batch object)
  std::unique_ptr<DevMem> scratch;
  FlowQueue fq{};
  uint32_t tag_bound = 0;              // upper bound of the buffers' generation tags (they wrap at 65535: the queue is zeroed before)
  fmx_result *d_res = nullptr;        // packed results
  fmx_result *d_res_seg = nullptr;    // kSub result slices the levels append to
  FrontierCtl *d_ctl = nullptr;
  uint32_t *d_rcnt = nullptr, *d_rstart = nullptr, *d_rfill = nullptr;   // per-regex result counts / offsets
  // Round 5: a call that ends normally leaves the batch READY for the next one -- k_res_sort's last workgroup rewinds the
  // queue's slices (what k_frontier_reset's first wave did), and the scan of the per-regex counts zeroes the OTHER of two
  // count arrays, which the next call counts into -- so that a call with the same limits starts with the frontier launch
  // (one launch and ~6 us less per call; C4text: a twentieth of the call).  pre_* = what the batch was left ready for.
  uint32_t *d_rcnt2[2] = {nullptr, nullptr};
  uint32_t rc_sel = 0;
  bool pre_ok = false;
  uint32_t pre_max_len = 0, pre_deep = 0;
  size_t pre_count = 0;
  StartSrc pre_ss{nullptr, 0, 0};
  uint32_t *d_rpart = nullptr;         // chunk totals of the offsets' scan
  BigGroups *d_big = nullptr;
  FrontierSummary *h_sum = nullptr;    // pinned: what a chain reports (written by k_frontier_advance)
  ExportDst *h_dst = nullptr;          // pinned: where k_res_export writes (set per call)
  // one chain of launches + the grouping, captured once per (grid, delivery): [0] the full grid, [1] the small one;
  // [.][1] = results delivered in the caller's device memory (no export launch)
  hipGraphExec_t chain_exec[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  GroupTotals *h_tot = nullptr;        // pinned: the grouping's totals, written by k_res_sort's last workgroup
  uint32_t chain_len = 0, chain_rounds = 0;
  uint32_t matches = 0;                // the chain is captured from a batch's second match on (a one-shot batch
                                       // would pay the capture and never replay it)
  ~RegexBatch() {
    drop_graphs();
    if (h_tot) (void)hipHostFree(h_tot);
    if (h_sum) (void)hipHostFree(h_sum);
    if (h_dst) (void)hipHostFree(h_dst);
  }
  void drop_graphs() {
    for (auto &row : chain_exec)
      for (hipGraphExec_t &g : row)
        if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
  }
  uint64_t qcap = 0;
  size_t rcap = 0;
  NfaTables nfa{};
  uint32_t *d_first_state = nullptr;   // root.firsts of every regex, regex by regex
  unsigned long long *d_start_elem = nullptr;   // the frontier kernel's start elements (StartSrc), balanced over the waves, slice by slice
  uint64_t start_cap = 0;              // entries per slice there
  // reference-order mode (ReTree batches only): heap keys, per-regex firsts, the largest fan-out
  uint32_t *d_st_num = nullptr, *d_first_off = nullptr;
  FolRec *d_fol_rec = nullptr, *d_first_rec = nullptr;
  uint32_t max_fanout = 1, max_num = 0;
  bool all_retree = true;
};

// Copies go through the caller's own (non-blocking) stream and wait for it: a plain hipMemcpy runs on the legacy
// stream, which implicitly waits for every blocking stream -- an error while another host thread is capturing a graph.
static hipError_t copy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  return e == hipSuccess ? hipStreamSynchronize(st) : e;
}

// Expected number of frontier elements a regex makes on an index of n rows over sigma symbols whose BWT looks random:
// an element at depth d (characters matched so far) holds an interval of about n / sigma^d rows; its step survives
// with probability min(1, rows / sigma), and a surviving element pushes its follows.  Summed over depths until the
// expectation has died away.  Used to deal the start elements to the waves so that every wave gets about the same
// amount of work (a starred class next to the regex's end is a few hundred elements, one next to its beginning a
// handful), and to cut a batch into slices for several GPUs.
static double frontier_work_estimate(const Regex &re, double n, double sigma, std::vector<double> &cur, std::vector<double> &nxt) {
  const size_t ns = re.st_c.size();
  if (!ns || re.firsts.empty()) return 1.0;
  cur.assign(ns, 0.0);
  for (int32_t f : re.firsts) cur[(size_t)f] += 1.0;
  double work = 0.0, rows = n;
  if (sigma < 2.0) sigma = 2.0;
  for (int depth = 0; depth < 48; depth++) {
    const double p = std::min(1.0, rows / sigma);      // the step at this depth survives
    rows = std::max(1.0, rows / sigma);
    nxt.assign(ns, 0.0);
    double level = 0.0, alive = 0.0;
    for (size_t s = 0; s < ns; s++) {
      const double c = cur[s];
      if (c == 0.0) continue;
      level += c;
      if (re.last_stops && re.st_last[s]) continue;
      const double live = c * p;
      for (int32_t j = re.fol_off[s]; j < re.fol_off[s + 1]; j++) { nxt[(size_t)re.fol[j]] += live; alive += live; }
    }
    work += level;
    if (alive < 1e-3) break;
    if (work > 1e9) break;
    cur.swap(nxt);
  }
  return work;
}

// The order of a batch's start elements.  Element i belongs to slice i % kSub (StartSrc), and a launch hands slice s's entries to the waves s, s + kSub, s + 2 kSub .. in contiguous chunks
// (frontier_pass): which wave gets element i is a function of i, the element count and the number of waves.  The
// elements are sorted by expected work and dealt to the waves in serpentine passes (heaviest first), so the waves'
// totals come out even; the element order itself carries no meaning (results are grouped by regex afterwards).
static void balanced_start_order(const std::vector<double> &work, size_t waves, std::vector<uint32_t> &perm /* position -> element */) {
  const size_t count = work.size();
  perm.resize(count);
  std::vector<uint32_t> by_work(count);
  for (size_t i = 0; i < count; i++) by_work[i] = (uint32_t)i;
  std::stable_sort(by_work.begin(), by_work.end(), [&](uint32_t a, uint32_t b) { return work[a] > work[b]; });
  if (waves < kSub) waves = kSub;
  std::vector<std::vector<uint32_t>> pos(waves);
  for (size_t i = 0; i < count; i++) {
    const size_t s = i % kSub, j = i / kSub;
    const size_t cnt_s = count > s ? (count - s + kSub - 1) / kSub : 0;
    const size_t class_waves = (waves - s + kSub - 1) / kSub;
    const size_t chunk = (cnt_s + class_waves - 1) / class_waves;
    size_t w = s + kSub * (chunk ? j / chunk : 0);
    if (w >= waves) w = s;
    pos[w].push_back((uint32_t)i);
  }
  size_t next = 0;
  for (size_t pass = 0; next < count; pass++) {
    for (size_t q = 0; q < waves; q++) {
      const size_t w = (pass & 1u) ? waves - 1 - q : q;
      if (pass < pos[w].size()) perm[pos[w][pass]] = by_work[next++];
    }
  }
}

int regex_batch_create(const Index *h, const Regex *const *res, size_t k, RegexBatch **out) {
  // Sizes first (per regex, then one prefix sum), then every regex fills its own stretch of the pre-sized arrays:
  // both passes run on all host cores (100 k regexes: 1.3 M states, 2 M follows).
  std::vector<size_t> st_base(k + 1, 0), fol_base(k + 1, 0), first_base(k + 1, 0);
  std::atomic<int> not_retree{0}, too_many{0};
  parallel_for(k, 1024, [&](size_t a, size_t b) {
    for (size_t r = a; r < b; r++) {
      const Regex &re = *res[r];
      size_t nf = 0;
      for (size_t s = 0; s < re.st_c.size(); s++)
        if (!(re.last_stops && re.st_last[s])) {
          const size_t cnt = (size_t)(re.fol_off[s + 1] - re.fol_off[s]);
          if (cnt > kMaxFollows) too_many.store(1);
          nf += cnt;
        }
      st_base[r + 1] = re.st_c.size();
      fol_base[r + 1] = nf;
      first_base[r + 1] = re.firsts.size();
      if (re.engine != 0) not_retree.store(1);
    }
  });
  if (too_many.load()) { set_error("a state has more than 65535 follows"); return FMX_ERR_UNSUPPORTED; }
  for (size_t r = 0; r < k; r++) { st_base[r + 1] += st_base[r]; fol_base[r + 1] += fol_base[r]; first_base[r + 1] += first_base[r]; }
  const size_t n_states = st_base[k], n_fol = fol_base[k], n_first = first_base[k];
  if (n_states >= (1ull << 32) || n_fol >= (1ull << 32)) { set_error("regex batch too large (2^32 states or follows)"); return FMX_ERR_UNSUPPORTED; }
  const bool all_retree = not_retree.load() == 0;
  std::vector<StateRec> recs(n_states);
  std::vector<uint32_t> fol(n_fol), q_state(n_first), st_num(n_states), first_off(k + 1, 0), start_final;
  std::vector<uint8_t> fol_c(n_fol);
  std::vector<uint32_t> fanout(k, 1);
  std::vector<double> elem_work(n_first, 1.0);
  const double est_n = (double)h->n, est_sigma = (double)std::max<uint32_t>(h->nslots, 2u);
  parallel_for(k, 1024, [&](size_t ra, size_t rb) {
    std::vector<double> dp_a, dp_b;
    for (size_t r = ra; r < rb; r++) {
      const Regex &re = *res[r];
      const size_t base = st_base[r];
      if (!re.firsts.empty()) {
        const double w = frontier_work_estimate(re, est_n, est_sigma, dp_a, dp_b) / (double)re.firsts.size();
        for (size_t f = 0; f < re.firsts.size(); f++) elem_work[first_base[r] + f] = w;
      }
      size_t fo = fol_base[r], qo = first_base[r];
      uint32_t max_fanout = 1;
      for (size_t s = 0; s < re.st_c.size(); s++) {
        StateRec &rec = recs[base + s];
        rec.fol_off = (uint32_t)fo;
        // ReTree: `if (q.state.isLast) ret ::= ... else pqFront ++= follows` -- last states do not expand
        if (!(re.last_stops && re.st_last[s]))
          for (int32_t j = re.fol_off[s]; j < re.fol_off[s + 1]; j++) {
            fol_c[fo] = re.st_c[(size_t)re.fol[j]];
            fol[fo++] = (uint32_t)base + (uint32_t)re.fol[j];
          }
        const uint32_t cnt = (uint32_t)fo - rec.fol_off;
        rec.cnt_c_emit = cnt | ((uint32_t)re.st_c[s] << 16) | ((uint32_t)(re.st_last[s] ? 1 : 0) << 24);
        rec.regex = (uint32_t)r;
        rec.fc = 0;
        for (uint32_t j = 0; j < kInlineFollows; j++) {
          rec.f[j] = j < cnt ? fol[rec.fol_off + j] : 0u;
          if (j < cnt) rec.fc |= (uint32_t)fol_c[rec.fol_off + j] << (8 * j);
        }
        st_num[base + s] = (uint32_t)re.st_num[s];
        max_fanout = std::max(max_fanout, cnt);
      }
      // literal stretches (fmx_nfa.h): chain lengths from the regex's last state backwards, then the bytes
      {
        const size_t ns = re.st_c.size();
        auto single = [&](size_t s) {
          const StateRec &rec = recs[base + s];
          return rec_cnt(rec) == 1 && !rec_emit(rec) && s + 1 < ns && rec.f[0] == (uint32_t)(base + s + 1);
        };
        uint32_t next_chain = 0;
        for (size_t s = ns; s-- > 0;) {
          const uint32_t chain = single(s) ? std::min<uint32_t>(kMaxChain, 1 + next_chain) : 0;
          next_chain = chain;
          if (!chain) continue;
          StateRec &rec = recs[base + s];
          rec.cnt_c_emit |= chain << 25;
          uint8_t rr[kMaxChain] = {0};
          for (uint32_t j = 0; j < chain; j++) rr[chain - 1 - j] = re.st_c[s + 1 + j];
          std::memcpy(&rec.f[1], rr, kMaxChain);
        }
      }
      for (int32_t f : re.firsts) q_state[qo++] = (uint32_t)base + (uint32_t)f;
      first_off[r + 1] = (uint32_t)qo;
      fanout[r] = std::max<uint32_t>(max_fanout, (uint32_t)re.firsts.size());
    }
  });
  uint32_t max_fanout = 1;
  for (size_t r = 0; r < k; r++) {
    max_fanout = std::max(max_fanout, fanout[r]);
    if (res[r]->start_is_final) start_final.push_back((uint32_t)r);
  }
  HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
  CtxLease lease(h);
  if (!lease.c) return FMX_ERR_HIP;
  hipStream_t st = lease.c->stream;
  std::unique_ptr<RegexBatch> b(new RegexBatch());
  b->device = h->device;
  b->k = k;
  b->n_index = h->n;
  b->index_serial = h->serial;
  b->n_first = q_state.size();
  b->n_states = n_states;
  b->n_fol = n_fol;
  b->start_final = start_final;
  b->max_fanout = max_fanout;
  b->all_retree = all_retree;
  StateRec *d_st = nullptr;
  uint32_t *d_fol = nullptr;
  uint8_t *d_fol_c = nullptr;
  HIP_TRY(b->mem.alloc(&d_st, recs.size()), "hipMalloc");
  HIP_TRY(b->mem.alloc(&d_fol, fol.size()), "hipMalloc");
  HIP_TRY(b->mem.alloc(&d_fol_c, fol_c.size() + 4), "hipMalloc");
  if (!fol_c.empty()) HIP_TRY(copy_sync(d_fol_c, fol_c.data(), fol_c.size(), hipMemcpyHostToDevice, st), "H2D");
  HIP_TRY(b->mem.alloc(&b->d_first_state, q_state.size()), "hipMalloc");
  if (!recs.empty()) HIP_TRY(copy_sync(d_st, recs.data(), recs.size() * sizeof(StateRec), hipMemcpyHostToDevice, st), "H2D");
  if (!fol.empty()) HIP_TRY(copy_sync(d_fol, fol.data(), fol.size() * 4, hipMemcpyHostToDevice, st), "H2D");
  if (!q_state.empty()) HIP_TRY(copy_sync(b->d_first_state, q_state.data(), q_state.size() * 4, hipMemcpyHostToDevice, st), "H2D");
  {   // the frontier kernel's start elements, in the order that balances the waves (the full grid's wave count)
    static const bool balance = !(getenv("FMX_FRONTIER_BALANCE") && atoi(getenv("FMX_FRONTIER_BALANCE")) == 0);      // A/B runs
    std::vector<uint32_t> perm, q_perm(q_state.size());
    if (balance && !q_state.empty()) {
      balanced_start_order(elem_work, (size_t)std::max(1, h->cu_count * FMX_FWAVES * 4 / kFWaves) * kFWaves, perm);
      for (size_t i = 0; i < q_state.size(); i++) q_perm[i] = q_state[perm[i]];
    } else {
      q_perm = q_state;
    }
    // element i belongs to slice i % kSub, position i / kSub (what balanced_start_order assumed)
    b->start_cap = (q_perm.size() + kSub - 1) / kSub;
    std::vector<unsigned long long> elem((size_t)b->start_cap * kSub, 0ull);
    for (size_t i = 0; i < q_perm.size(); i++)
      elem[(i % kSub) * b->start_cap + i / kSub] = (unsigned long long)q_perm[i] | ((unsigned long long)rec_c(recs[q_perm[i]]) << 40);
    HIP_TRY(b->mem.alloc(&b->d_start_elem, elem.size()), "hipMalloc");
    if (!elem.empty()) HIP_TRY(copy_sync(b->d_start_elem, elem.data(), elem.size() * 8, hipMemcpyHostToDevice, st), "H2D");
  }
  if (all_retree) {
    HIP_TRY(b->mem.alloc(&b->d_st_num, st_num.size()), "hipMalloc");
    HIP_TRY(b->mem.alloc(&b->d_first_off, first_off.size()), "hipMalloc");
    if (!st_num.empty()) HIP_TRY(copy_sync(b->d_st_num, st_num.data(), st_num.size() * 4, hipMemcpyHostToDevice, st), "H2D");
    HIP_TRY(copy_sync(b->d_first_off, first_off.data(), first_off.size() * 4, hipMemcpyHostToDevice, st), "H2D");
    // the reference-order kernel's push records (fmx_nfa.h)
    std::vector<FolRec> fr(fol.size()), qr(q_state.size());
    auto rec_of = [&](uint32_t sid) { return FolRec{recs[sid].fc, recs[sid].fol_off, recs[sid].cnt_c_emit & 0x01FFFFFFu, st_num[sid]}; };
    parallel_for(fol.size(), 1 << 16, [&](size_t a, size_t e) { for (size_t i = a; i < e; i++) fr[i] = rec_of(fol[i]); });
    for (size_t i = 0; i < q_state.size(); i++) qr[i] = rec_of(q_state[i]);
    uint32_t mx = 0;
    for (uint32_t v : st_num) mx = std::max(mx, v);
    b->max_num = mx;
    HIP_TRY(b->mem.alloc(&b->d_fol_rec, fr.size()), "hipMalloc");
    HIP_TRY(b->mem.alloc(&b->d_first_rec, qr.size()), "hipMalloc");
    if (!fr.empty()) HIP_TRY(copy_sync(b->d_fol_rec, fr.data(), fr.size() * sizeof(FolRec), hipMemcpyHostToDevice, st), "H2D");
    if (!qr.empty()) HIP_TRY(copy_sync(b->d_first_rec, qr.data(), qr.size() * sizeof(FolRec), hipMemcpyHostToDevice, st), "H2D");
  }
  b->nfa = NfaTables{d_st, d_fol, d_fol_c};
  *out = b.release();
  return FMX_OK;
}

// A call starts from rewound buffers under fresh tags (workgroup 0's first wave, lane = slice): buffer 0 of every
// slice stands for the slice's share of the start elements (which the first launch makes up, StartSrc), buffer 1 is
// the first one written.  The whole grid clears the per-regex result counts.
__global__ __launch_bounds__(256) void k_frontier_reset(FrontierCtl *__restrict__ ctl, uint64_t count, const uint32_t *__restrict__ call /* {max_len, fresh, direct, deep_len}, pinned host */,
                                                         uint32_t *__restrict__ rcnt, uint32_t k, StartSrc start) {
  const uint32_t fresh = call[1];
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const uint32_t i = threadIdx.x;
    const uint32_t max_len = call[0], deep_len = call[3];
    if (i == 0) { ctl->fresh = fresh; ctl->deep_len = deep_len; ctl->start = start; }
    if (fresh) {
      SliceCtl &q = ctl->q[i];
      q.tail[0] = count > i ? (count - i + kSub - 1) / kSub : 0;
      q.tail[1] = 0; q.head[0] = 0; q.head[1] = 0;
      q.tag[0] = next_tag(q.tag[0]); q.tag[1] = next_tag(q.tag[1]);
      q.wsel = 1;
      ctl->res_count[i].v = 0;
      if (i == 0) { ctl->overflow = 0; ctl->truncated = 0; ctl->max_len = max_len; ctl->left = count; }
    }
  }
  if (!fresh) return;                      // a chain that continues a call
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= k; r += (uint64_t)gridDim.x * blockDim.x) rcnt[r] = 0;      // the frontier kernel counts results per regex
}

// Results leave the device grouped by regex: the frontier kernel counts them per regex where it flushes them; then
// offsets (a scan), scatter, order -- THREE launches behind the frontier's (round 3 had seven: the start elements' own
// launch, scan in two, scatter, two sorts, export; at ~5 us each they were a fifth of a C4 call and over half of a
// call on a real text, where the frontier dies early).
// Exclusive prefix sums of cnt[0..k] in two parts: start[i] = the sum inside i's chunk of 1024 counts, part[chunk] = the
// chunk's total.  The consumers add the chunks before theirs themselves (a hundred values for 100 k regexes, summed
// up in LDS by every workgroup: part_prefix) -- no launch for the scan of the chunk totals, none to add them.
constexpr uint32_t kScanChunk = 1024;
constexpr uint32_t kMaxPartsLds = 2048;      // chunk totals a consumer sums up itself (2 M regexes); beyond: k_res_scan_add
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

// Between launches (workgroup 0's first wave, lane = slice): a buffer that has been emptied is rewound under its next
// tag; when that is the buffer the slice was emptying and the written one holds entries, the two swap roles.  Also
// sums up what the host wants to know after a chain.  Behind a chain's LAST launch the same grid starts the grouping:
// every workgroup scans its chunk of the per-regex counts (n != 0).
__global__ __launch_bounds__(kScanChunk) void k_frontier_advance(FrontierCtl *__restrict__ ctl, uint64_t sub_cap, FrontierSummary *__restrict__ sum,
                                                                 const uint32_t *__restrict__ cnt, uint32_t n, uint32_t *__restrict__ start,
                                                                 uint32_t *__restrict__ part, uint32_t *__restrict__ fill, BigGroups *__restrict__ big,
                                                                 uint32_t *__restrict__ zero_next /* the count array the NEXT call counts into, or null */) {
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    SliceCtl &q = ctl->q[threadIdx.x];
    const uint32_t wr = q.wsel, ot = 1u - wr;
    // the launch has worked off (or carried over) everything in the buffer it was not writing
    const unsigned long long tl = q.tail[wr] < sub_cap ? q.tail[wr] : sub_cap;
    const unsigned long long lw = tl > q.head[wr] ? tl - q.head[wr] : 0ull;
    if (q.tail[ot] | q.head[ot]) { q.tail[ot] = 0; q.head[ot] = 0; q.tag[ot] = next_tag(q.tag[ot]); }
    if (lw) q.wsel = ot;
    else if (q.tail[wr] | q.head[wr]) { q.tail[wr] = 0; q.head[wr] = 0; q.tag[wr] = next_tag(q.tag[wr]); }
    const unsigned long long left = wave_sum(lw), results = wave_sum(ctl->res_count[threadIdx.x].v);
    if (threadIdx.x == 0) { ctl->left = left; sum->left = left; sum->results = results; sum->overflow = ctl->overflow; sum->truncated = ctl->truncated; }
  }
  if (!n) return;                                       // uniform over the grid
  __shared__ uint32_t s_wave[16];
  const uint32_t i = blockIdx.x * kScanChunk + threadIdx.x;
  uint32_t total = 0;
  const uint32_t ex = block_excl_scan_1024(i < n ? cnt[i] : 0u, s_wave, total);
  if (i < n) { start[i] = ex; fill[i] = 0; if (zero_next) zero_next[i] = 0; }      // the scatter's cursors start from zero in every grouping
  if (threadIdx.x == 0) part[blockIdx.x] = total;
  if (blockIdx.x == 0 && threadIdx.x == 0) { big->n = 0; big->done = 0; big->total = 0; }
}

// More chunks than a consumer sums up in LDS: their totals are added here, once, and the consumers get part = null.
__global__ __launch_bounds__(kScanChunk) void k_res_scan_add(uint32_t *__restrict__ start, uint32_t n,
                                                             const uint32_t *__restrict__ part) {
  __shared__ unsigned long long s_sum[kScanChunk / 64];
  unsigned long long mine = 0;
  for (uint32_t q = threadIdx.x; q < blockIdx.x; q += kScanChunk) mine += part[q];
  mine = wave_sum(mine);
  if ((threadIdx.x & 63u) == 0) s_sum[threadIdx.x >> 6] = mine;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t j = 0; j < kScanChunk / 64; j++) before += (uint32_t)s_sum[j];
  const uint32_t i = blockIdx.x * kScanChunk + threadIdx.x;
  if (i < n) start[i] += before;
}

// s_pre[c] = part[0] + .. + part[c - 1] for c = 0 .. nparts, by a workgroup of 256 threads (ends with a barrier).
// part == null: the offsets are absolute already (k_res_scan_add ran), s_pre is all zeros.
struct PartPrefix {
  uint32_t pre[kMaxPartsLds + 8];
  uint32_t wave[4];
};
__device__ __forceinline__ void part_prefix(const uint32_t *__restrict__ part, uint32_t nparts, PartPrefix &pp) {
  const uint32_t per = (nparts + 256u) / 256u;          // entries per thread, covering 0 .. nparts
  const uint32_t lo = threadIdx.x * per;
  uint32_t mine = 0;
  if (part)
    for (uint32_t q = lo; q < lo + per && q < nparts; q++) mine += part[q];
  uint32_t wtot = 0;
  uint32_t ex = wave_excl_scan(mine, wtot);
  if ((threadIdx.x & 63u) == 0) pp.wave[threadIdx.x >> 6] = wtot;
  __syncthreads();
  for (uint32_t j = 0; j < (threadIdx.x >> 6); j++) ex += pp.wave[j];
  for (uint32_t q = lo; q < lo + per && q <= nparts; q++) {
    pp.pre[q] = ex;
    if (part && q < nparts) ex += part[q];
  }
  __syncthreads();
}
__device__ __forceinline__ uint32_t offset_of(const uint32_t *__restrict__ start, const PartPrefix &pp, uint32_t r) {
  return start[r] + pp.pre[r / kScanChunk];
}

__global__ __launch_bounds__(256) void k_res_scatter(const fmx_result *__restrict__ seg, uint64_t seg_cap,
                                                      const FrontierCtl *__restrict__ ctl,
                                                      const uint32_t *__restrict__ start, const uint32_t *__restrict__ part, uint32_t nparts,
                                                      uint32_t *__restrict__ fill,
                                                      fmx_result *__restrict__ own, uint64_t own_cap, const ExportDst *__restrict__ dst) {
  __shared__ PartPrefix s_pp;
  uint64_t out_cap;
  fmx_result *out = group_out(dst, own, own_cap, out_cap);
  const uint32_t sl = blockIdx.y;
  const uint64_t mine = min((uint64_t)ctl->res_count[sl].v, seg_cap);
  if ((uint64_t)blockIdx.x * blockDim.x >= mine) return;             // uniform over the workgroup: nothing of this slice is ours
  part_prefix(part, nparts, s_pp);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < mine; i += (uint64_t)gridDim.x * blockDim.x) {
    const fmx_result r = seg[(uint64_t)sl * seg_cap + i];
    const uint64_t at = (uint64_t)offset_of(start, s_pp, r.regex) + atomicAdd(&fill[r.regex], 1u);
    if (at < out_cap) out[at] = r;
  }
}

// Orders each regex's group by (len, sp, ep).  The results of a workgroup's 256 regexes are one contiguous stretch of
// `out`; it is read into LDS (it nearly always fits), and EVERY RESULT FINDS ITS OWN PLACE: a thread takes a result,
// counts the results of the same group that sort before it (ties by position) -- independent LDS reads, no chain of
// dependent accesses -- and writes it to that place of `out` (groups of up to 64; C4's 100 k regexes: 11 000 groups of
// 2 .. 23 results, 1.2 ms of single-thread insertion sorts in round 3's kernel).  Larger groups, up to 1024 results, are
// ordered one after the other by a bitonic sort in LDS with all the workgroup's threads; groups beyond that are listed
// for the host.  The workgroup that finishes last reports the call's totals to the host's page-locked GroupTotals;
// with the results delivered in device memory (ExportDst::direct) every workgroup also copies its regexes' counts
// out, and nothing is left for k_res_export to do: it is not launched then.
constexpr uint32_t kRankGroup = 64;
__global__ __launch_bounds__(256) void k_res_sort(fmx_result *__restrict__ own, uint64_t own_cap, const uint32_t *__restrict__ start,
                                                   const uint32_t *__restrict__ part, uint32_t nparts, uint32_t k,
                                                   const uint32_t *__restrict__ rcnt, BigGroups *__restrict__ big,
                                                   const ExportDst *__restrict__ dst, GroupTotals *__restrict__ tot /* pinned host */,
                                                   FrontierCtl *__restrict__ ctl, uint64_t count, StartSrc start_src, uint32_t pre_next) {
  __shared__ fmx_result s_r[kMidGroup];
  __shared__ uint32_t s_last;
  __shared__ PartPrefix s_pp;
  __shared__ uint32_t s_off[257];             // offsets of the workgroup's regexes (and of the one behind them)
  __shared__ uint32_t s_mid[2 * 256], s_nmid;
  if (threadIdx.x == 0) s_nmid = 0;
  part_prefix(part, nparts, s_pp);
  const ExportDst d = *dst;
  uint64_t out_cap;
  fmx_result *out = group_out(dst, own, own_cap, out_cap);
  const uint32_t r0 = blockIdx.x * blockDim.x, r = r0 + threadIdx.x;
  const uint32_t r_end = r0 + blockDim.x < k ? r0 + blockDim.x : k, nr = r_end - r0;
  for (uint32_t t = threadIdx.x; t <= nr; t += blockDim.x) s_off[t] = offset_of(start, s_pp, r0 + t);
  const uint32_t total = offset_of(start, s_pp, k);
  __syncthreads();
  const uint32_t base = s_off[0], span = s_off[nr] - base;      // uniform over the workgroup
  auto less = [](const fmx_result &a, const fmx_result &b) {
    if (a.len != b.len) return a.len < b.len;
    if (a.sp != b.sp) return a.sp < b.sp;
    return a.ep < b.ep;
  };
  // more results than the buffer holds: the call fails with FMX_ERR_OVERFLOW, nothing to order
  if ((uint64_t)base + span <= out_cap) {
    const bool staged = span <= kMidGroup;
    if (staged) {
      for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) s_r[i] = out[base + i];
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) {
        const fmx_result x = s_r[i];
        const uint32_t t = x.regex - r0;
        if (t >= nr) continue;                 // (cannot happen: the stretch holds this workgroup's regexes only)
        const uint32_t lo = s_off[t] - base, m = s_off[t + 1] - s_off[t];
        if (m < 2) continue;
        if (m <= kRankGroup) {
          uint32_t rank = 0;
          for (uint32_t j = 0; j < m; j++) {
            const fmx_result y = s_r[lo + j];
            rank += (less(y, x) || (!less(x, y) && lo + j < i)) ? 1u : 0u;
          }
          if (lo + rank != i) out[base + lo + rank] = x;
        } else if (i == lo) {                  // the group's first result lists it (m <= span <= kMidGroup)
          const uint32_t at = atomicAdd(&s_nmid, 1u);
          s_mid[2 * at] = base + lo;
          s_mid[2 * at + 1] = m;
        }
      }
    } else if (r < r_end) {
      // a stretch that does not fit (a group of hundreds of results among the 256): one thread per regex, insertion
      // sort in `out` for the small groups, the others listed
      const uint32_t lo = s_off[threadIdx.x], m = s_off[threadIdx.x + 1] - lo;
      if (m > kMidGroup) {
        const uint32_t at = atomicAdd(&big->n, 1u);
        if (at < kBigMax) { big->ent[2 * at] = lo; big->ent[2 * at + 1] = m; }
      } else if (m > kSmallGroup) {
        const uint32_t at = atomicAdd(&s_nmid, 1u);
        s_mid[2 * at] = lo;
        s_mid[2 * at + 1] = m;
      } else if (m >= 2) {
        fmx_result *g = out + lo;
        for (uint32_t i = 1; i < m; i++) {
          const fmx_result x = g[i];
          uint32_t j = i;
          while (j > 0 && less(x, g[j - 1])) { g[j] = g[j - 1]; j--; }
          g[j] = x;
        }
      }
    }
    __syncthreads();
    const uint32_t nmid = s_nmid;               // the listed groups still lie in `out` as the scatter left them
    for (uint32_t g = 0; g < nmid; g++) {
      const uint32_t lo = s_mid[2 * g], m = s_mid[2 * g + 1];
      uint32_t p2 = 16;
      while (p2 < m) p2 <<= 1;
      for (uint32_t i = threadIdx.x; i < p2; i += blockDim.x) {
        fmx_result x;
        if (i < m) x = out[lo + i];
        else { x.regex = 0; x.len = 0xFFFFFFFFu; x.sp = ~0ull; x.ep = ~0ull; }      // padding sorts last
        s_r[i] = x;
      }
      __syncthreads();
      for (uint32_t size = 2; size <= p2; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
          for (uint32_t i = threadIdx.x; i < p2 / 2; i += blockDim.x) {
            const uint32_t a = 2 * i - (i & (stride - 1));          // lower index of the pair
            const uint32_t bidx = a + stride;
            const bool up = (a & size) == 0;
            const fmx_result x = s_r[a], y = s_r[bidx];
            if (less(y, x) == up) { s_r[a] = y; s_r[bidx] = x; }
          }
          __syncthreads();
        }
      for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) out[lo + i] = s_r[i];
      __syncthreads();
    }
  }
  if (d.direct && d.per && r < k) d.per[r] = rcnt[r];
  // No fence here (an agent-scope release writes back the XCD's whole L2: 30 us per launch when every workgroup
  // asks for one).  What the last workgroup reads of the others is big->n, a device atomic whose returned value each
  // of their threads has already used, so it has been performed before that thread reaches the barrier below; the
  // lists and results themselves are read by the host or the next launch, behind the kernel's end.
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool last = atomicAdd(&big->done, 1u) == gridDim.x - 1u;
    s_last = last ? 1u : 0u;
    if (last) {
      big->total = total;
      tot->n_results = total;
      tot->n_big = atomicAdd(&big->n, 0u);
    }
  }
  if (pre_next) {                          // uniform over the grid
    // The call is over when its last launch left nothing queued: the workgroup that finishes last then leaves the batch
    // READY for a next call with the same limits -- the slices rewound under fresh tags, buffer 0 standing for the start
    // elements, exactly what k_frontier_reset's first wave does -- so that such a call begins with the frontier launch
    // itself (the host skips the reset launch: regex_batch_match, pre_ok).  Nothing of this grouping reads the slices
    // any more (the scatter is done); a call that is NOT over (left != 0) keeps its state and resets as before.
    __syncthreads();
    if (s_last && threadIdx.x < kSub && ctl->left == 0 && ctl->overflow == 0) {
      const uint32_t i = threadIdx.x;
      SliceCtl &q = ctl->q[i];
      q.tail[0] = count > i ? (count - i + kSub - 1) / kSub : 0;
      q.tail[1] = 0; q.head[0] = 0; q.head[1] = 0;
      q.tag[0] = next_tag(q.tag[0]); q.tag[1] = next_tag(q.tag[1]);
      q.wsel = 1;
      ctl->res_count[i].v = 0;
      if (i == 0) { ctl->fresh = 1; ctl->deep_len = d.deep_len; ctl->start = start_src; ctl->truncated = 0; ctl->max_len = d.max_len; ctl->left = count; }
    }
  }
}

// Results delivered to the HOST's page-locked buffers (or, in an A/B run, to device memory without the direct form):
// the ordered results and the per-regex counts are copied there, behind the grouping and before the host's one
// synchronisation.  Not launched when the grouping worked in the caller's device memory itself.
__global__ __launch_bounds__(256) void k_res_export(const fmx_result *__restrict__ res, uint32_t k,
                                                     const uint32_t *__restrict__ rcnt, const BigGroups *__restrict__ big,
                                                     const ExportDst *__restrict__ dst) {
  const ExportDst d = *dst;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
  if (d.out && !d.direct) {
    const uint64_t n = big->total < d.cap ? big->total : d.cap;
    // 24-byte results as 16-byte words (both buffers are 16-byte aligned: hipMalloc / page-locked memory), the odd
    // eight bytes at the end on their own
    const uint64_t words = 3 * n;                                   // 8-byte words
    const uint4 *src = reinterpret_cast<const uint4 *>(res);
    uint4 *out = reinterpret_cast<uint4 *>(d.out);
    const bool aligned = (reinterpret_cast<uintptr_t>(d.out) & 15u) == 0;
    if (aligned) {
      for (uint64_t i = tid; i < words / 2; i += nth) out[i] = src[i];
      if ((words & 1u) && tid == 0) reinterpret_cast<uint2 *>(d.out)[words - 1] = reinterpret_cast<const uint2 *>(res)[words - 1];
    } else {
      for (uint64_t i = tid; i < words; i += nth) reinterpret_cast<uint2 *>(d.out)[i] = reinterpret_cast<const uint2 *>(res)[i];
    }
  }
  if (d.per && !d.direct)
    for (uint64_t i = tid; i < k; i += nth) d.per[i] = rcnt[i];
}

// `dev`: out / per_regex_count are DEVICE pointers -- the results stay in HBM (the export kernel's copy is then on
// the device); the rare cases that need the host (groups of more than 1024 results, results the host adds for final
// DFA start states) are staged through host memory and written back.
int regex_batch_match(const Index *h, RegexBatch *b, const fmx_limits *lim, fmx_result *out, size_t cap,
                      size_t *n_out, uint32_t *per_regex_count, bool dev = false) {
  static const bool trace = getenv("FMX_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (trace) fprintf(stderr, "[fmx] regex_batch_match %-18s +%.3f ms\n", what,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  const uint32_t max_steps = std::min<uint32_t>((lim && lim->max_steps) ? lim->max_steps : 4096u, kMaxLen);
  const uint64_t qcap = (lim && lim->max_frontier) ? lim->max_frontier : (1ull << 22);
  if (b->index_serial != h->serial) { set_error("regex batch was prepared for another index"); return FMX_ERR_ARG; }
  if (per_regex_count && !dev) std::fill(per_regex_count, per_regex_count + b->k, 0u);
  *n_out = 0;
  if (b->n_first == 0 && b->start_final.empty()) {
    if (per_regex_count && dev && b->k) {
      HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
      HIP_TRY(hipMemset(per_regex_count, 0, b->k * 4), "hipMemset(result counts)");
    }
    return FMX_OK;
  }
  if (b->n_first > qcap) { set_error("initial frontier exceeds max_frontier"); return FMX_ERR_OVERFLOW; }
  HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
  // slices: each holds its share of max_frontier plus a quarter of headroom (appends rotate over the slices,
  // so they fill evenly, not exactly); the result segments get 4x their share
  const uint64_t sub_cap = (qcap + kSub - 1) / kSub + qcap / (4 * kSub) + 1024;
  if (!b->scratch || b->qcap != qcap || b->rcap < (cap ? cap : 1)) {
    b->scratch.reset(new DevMem());
    b->qcap = 0;
    b->drop_graphs();                  // they hold the old pointers
    if (!b->h_tot) HIP_TRY(hipHostMalloc((void **)&b->h_tot, sizeof(GroupTotals), hipHostMallocDefault), "hipHostMalloc(totals)");
    if (!b->h_sum) HIP_TRY(hipHostMalloc((void **)&b->h_sum, sizeof(FrontierSummary), hipHostMallocDefault), "hipHostMalloc(summary)");
    if (!b->h_dst) HIP_TRY(hipHostMalloc((void **)&b->h_dst, sizeof(ExportDst), hipHostMallocDefault), "hipHostMalloc(export)");
    const uint64_t seg_cap = (uint64_t)(cap ? cap : 1) / 16 + 1024;
    for (unsigned long long **g : {&b->fq.g0, &b->fq.g1, &b->fq.g2}) {
      HIP_TRY(b->scratch->alloc(g, 2 * kSub * sub_cap), "hipMalloc(queue)");
    }
    HIP_TRY(b->scratch->alloc(&b->d_res, cap ? cap : 1), "hipMalloc(results)");
    HIP_TRY(b->scratch->alloc(&b->d_res_seg, kSub * seg_cap), "hipMalloc(result slices)");
    HIP_TRY(b->scratch->alloc(&b->d_ctl, 1), "hipMalloc(ctl)");
    b->tag_bound = ~0u;                // new memory: zeroed below (tag 0 = never written)
    {   // counts, fill cursors and the big-group list in one block: one memset clears what a grouping starts from
      uint32_t *blk = nullptr;
      HIP_TRY(b->scratch->alloc(&blk, 3 * (b->k + 1) + (sizeof(BigGroups) + 3) / 4), "hipMalloc(result counts)");
      b->d_rcnt2[0] = blk;
      b->d_rcnt2[1] = blk + 2 * (b->k + 1);
      b->d_rcnt = blk;
      b->rc_sel = 0;
      b->pre_ok = false;
      b->d_rfill = blk + (b->k + 1);
      b->d_big = reinterpret_cast<BigGroups *>(blk + 3 * (b->k + 1));
    }
    HIP_TRY(b->scratch->alloc(&b->d_rstart, b->k + 1), "hipMalloc(result offsets)");
    HIP_TRY(b->scratch->alloc(&b->d_rpart, (b->k + 1) / kScanChunk + 2), "hipMalloc(scan parts)");
    b->qcap = qcap;
    b->rcap = cap ? cap : 1;
  }
  // Slice capacity of the result buffer: a function of the ALLOCATED size, so that it stays what the captured
  // launch chain was recorded with when a later call passes a smaller cap (the scratch is kept then).
  const uint64_t seg_cap = (uint64_t)b->rcap / 16 + 1024;
  const FlowQueue fq = b->fq;
  fmx_result *d_res = b->d_res;
  fmx_result *d_res_seg = b->d_res_seg;
  FrontierCtl *d_ctl = b->d_ctl;
  CtxLease lease(h);                 // stream and events from the handle's pool
  if (!lease.c) return FMX_ERR_HIP;
  hipStream_t st = lease.c->stream;
  hipEvent_t e0 = lease.c->ev_a, e1 = lease.c->ev_b;

  KTab kt;
  HIP_TRY(ktab_get(h, st, &kt), "k-mer table");
  HIP_TRY(row1_get(h, st, &kt.row1), "row table");
  // Generation tags are 16 bits wide; a buffer's tag advances at most once per launch.  Long before a tag can come
  // round to a value that an old entry still carries, the queue and the tags are zeroed (a 100 MB memset every few
  // thousand calls).
  // (FMX_FRONTIER_TAG_LIMIT: a test makes the tags wrap within a few calls)
  static const uint32_t tag_limit = getenv("FMX_FRONTIER_TAG_LIMIT") ? (uint32_t)std::max(1, atoi(getenv("FMX_FRONTIER_TAG_LIMIT"))) : kTagLimit;
  if (b->tag_bound >= tag_limit) {
    for (unsigned long long *g : {fq.g0, fq.g1, fq.g2}) HIP_TRY(hipMemsetAsync(g, 0, 2 * kSub * sub_cap * 8, st), "hipMemset(queue)");
    HIP_TRY(hipMemsetAsync(d_ctl, 0, sizeof(FrontierCtl), st), "hipMemset(ctl)");
    b->tag_bound = 0;
    b->pre_ok = false;                 // (whatever the last call left ready is gone: this call resets by launch)
  }
  b->tag_bound++;
  mark("setup");
  HIP_TRY(hipEventRecord(e0, st), "hipEventRecord");
  // Launches are chained on the stream without host round trips; the host looks at the summary after every
  // chain.  A launch that finds the queue empty returns at once.
  // one launch per chain on the full grid: a batch like C4 is done by one, and a second launch that finds nothing costs
  // ~8 us with its advance kernel (0.4170 -> 0.4109 ms per call); a search that needs more pays a host look per launch
  static const uint32_t kChain = getenv("FMX_FRONTIER_CHAIN") ? (uint32_t)std::max(1, atoi(getenv("FMX_FRONTIER_CHAIN"))) : 1u;
  // rounds a wave works at most in one launch (what it still holds then goes to the queue): the bound that makes
  // every wave end.  C4 is done in one launch of ~50 rounds per wave (the longest wave: 113); measured 64 / 96 / 128 /
  // 256: 0.565 / 0.548 / 0.527 / 0.534 ms
  static const uint32_t kRounds = getenv("FMX_FRONTIER_ROUNDS") ? (uint32_t)std::max(1, atoi(getenv("FMX_FRONTIER_ROUNDS"))) : 128u;
  // workgroups per CU in the full grid: what is resident at once (FMX_FWAVES waves per SIMD) -- a launch lasts as
  // long as the search does, so a second generation of workgroups would find nothing (measured 3 / 4 / 6: 0.567 /
  // 0.712 / 0.664 ms)
  static const int per_cu = getenv("FMX_FRONTIER_WGS") ? std::max(1, atoi(getenv("FMX_FRONTIER_WGS"))) : FMX_FWAVES;
  const int grid_full = std::max(1, h->cu_count * per_cu * 4 / kFWaves);      // per_cu counts 256-thread units
  const uint64_t per_wg = (uint64_t)kFThreads;                    // elements a workgroup holds at once
  FrontierSummary sum{};
  uint64_t n_res = 0;
  uint32_t pass = 0;
  uint64_t launches = 1;
  bool alive = true, truncated = false;
  const StartSrc ss{b->d_start_elem, b->start_cap, kt.k ? 0 : h->n};
  // `pre_next`: this call leaves the batch ready for the next one (not with captured graphs: their arguments are fixed, and
  // the two count arrays alternate).  `pre_now`: the LAST call left it ready for exactly this one -- no reset launch.
  static const bool pre_off = (getenv("FMX_FRONTIER_GRAPH") && atoi(getenv("FMX_FRONTIER_GRAPH")) != 0) ||
                              (getenv("FMX_FRONTIER_PRERESET") && atoi(getenv("FMX_FRONTIER_PRERESET")) == 0);
  const bool pre_next = !pre_off;
  const uint32_t deep_len_now = [&] {
    const double sig = (double)std::max<uint32_t>(h->nslots, 2u);
    static const int deep_extra = getenv("FMX_FRONTIER_DEEP") ? atoi(getenv("FMX_FRONTIER_DEEP")) : 2;      // A/B runs
    return (uint32_t)std::max(1.0, std::ceil(std::log((double)h->n + 1.0) / std::log(sig)) + deep_extra);
  }();
  if (pre_next) {
    if (b->pre_ok) b->rc_sel ^= 1u;          // the array the last call's scan zeroed
    b->d_rcnt = b->d_rcnt2[b->rc_sel];
  } else {
    b->rc_sel = 0;
    b->d_rcnt = b->d_rcnt2[0];
  }
  const bool pre_now = pre_next && b->pre_ok && b->pre_max_len == max_steps && b->pre_deep == deep_len_now && b->pre_count == b->n_first &&
                       b->pre_ss.elem == ss.elem && b->pre_ss.cap == ss.cap && b->pre_ss.ep == ss.ep;
  b->pre_ok = false;                         // until this call has ended normally
  const uint32_t n_scan = (uint32_t)b->k + 1, nparts = (n_scan + kScanChunk - 1) / kScanChunk;     // cnt[k] is 0: its offset = the total
  const bool parts_in_lds = nparts <= kMaxPartsLds;      // the consumers of the offsets add the chunk totals up themselves
  auto launch_pass = [&](hipStream_t s, int grid, uint32_t j, uint32_t rounds) {
    if (h->layout == kLayoutBytes)
      k_frontier<true, kLayoutBytes><<<grid, kFThreads, 0, s>>>(h->dev, kt, b->nfa, fq, j, rounds, sub_cap, d_res_seg, seg_cap, d_ctl, b->d_rcnt, h->d_counters);
    else if (h->n > (1ull << 32))
      k_frontier<true, kLayoutOneHot><<<grid, kFThreads, 0, s>>>(h->dev, kt, b->nfa, fq, j, rounds, sub_cap, d_res_seg, seg_cap, d_ctl, b->d_rcnt, h->d_counters);
    else
      k_frontier<false, kLayoutOneHot><<<grid, kFThreads, 0, s>>>(h->dev, kt, b->nfa, fq, j, rounds, sub_cap, d_res_seg, seg_cap, d_ctl, b->d_rcnt, h->d_counters);
  };
  // One chain = reset + kChain x (launch + k_frontier_advance), the grouping behind it.  Two grids: the full one for a
  // batch's wide phase, and a small one (64 workgroups: lanes for 16384 elements) for a single regex or the thin end of
  // a batch, whose launches cost a fraction of the full grid's when most of them find nothing to do.
  // The kernel arguments do not change from chain to chain, so the whole call can be captured into a hipGraph once per
  // batch and replayed (FMX_FRONTIER_GRAPH=1).  While a call was twelve kernels that paid; at five, the ~18 us between
  // hipGraphLaunch and the first kernel's start are more than five plain launches cost, which the host enqueues while
  // the first ones run: C4 0.352 -> 0.342 ms, C4text 0.119 -> 0.110 ms, one 24-character literal 71 -> 70 us per call.
  // Plain launches are the default.
  static const bool use_graph = getenv("FMX_FRONTIER_GRAPH") && atoi(getenv("FMX_FRONTIER_GRAPH")) != 0;
  // experiments: rounds per launch of a chain as a comma list (the last entry repeats)
  static const std::vector<uint32_t> plan = [] {
    std::vector<uint32_t> v;
    if (const char *e = getenv("FMX_FRONTIER_PLAN"))
      for (const char *p = e; *p;) { v.push_back((uint32_t)std::max(1l, strtol(p, const_cast<char **>(&p), 10))); if (*p == ',') p++; else break; }
    return v;
  }();
  const int grid_small = std::max(1, 256 / kFWaves);                           // 256 waves
  // the small grid's chain is short: a launch that finds nothing to do still costs ~3 us
  // The small grid serves a single regex or the thin end of a batch: there a search is a few elements that grow
  // into a tree, and what spreads it over the waves is the hand-over at the end of a launch -- short launches, more
  // of them (a[ab]*c on 2 M rows: 302 us per call with 128-round launches, 250 with 32; a 24-character literal: 87 / 95 us)
  static const uint32_t kChainSmall = getenv("FMX_FRONTIER_CHAIN_SMALL") ? (uint32_t)std::max(1, atoi(getenv("FMX_FRONTIER_CHAIN_SMALL"))) : 2u;
  static const uint32_t kRoundsSmall = getenv("FMX_FRONTIER_ROUNDS_SMALL") ? (uint32_t)std::max(1, atoi(getenv("FMX_FRONTIER_ROUNDS_SMALL"))) : 32u;
  auto enqueue_chain = [&](hipStream_t s, int grid) -> hipError_t {
    const uint32_t len = grid == grid_small ? kChainSmall : kChain;
    // a call's first chain begins with the reset (h_dst->fresh; it returns at once otherwise): one graph launch per call
    if (!(pre_now && b->h_dst->fresh))
      k_frontier_reset<<<(int)std::min<size_t>((b->k + 256) / 256, 256), 256, 0, s>>>(d_ctl, b->n_first, &b->h_dst->max_len, b->d_rcnt, (uint32_t)b->k, ss);
    for (uint32_t j = 0; j < len; j++) {
      launch_pass(s, grid, j, grid == grid_small ? kRoundsSmall : (plan.empty() ? kRounds : plan[std::min<size_t>(j, plan.size() - 1)]));
      // the summary goes straight to pinned host memory; behind the chain's last launch the same grid scans the result counts
      if (j + 1 < len) k_frontier_advance<<<1, kScanChunk, 0, s>>>(d_ctl, sub_cap, b->h_sum, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr);
      else k_frontier_advance<<<nparts, kScanChunk, 0, s>>>(d_ctl, sub_cap, b->h_sum, b->d_rcnt, n_scan, b->d_rstart, b->d_rpart, b->d_rfill, b->d_big, pre_next ? b->d_rcnt2[b->rc_sel ^ 1u] : nullptr);
    }
    return hipGetLastError();
  };
  // page-locked caller buffers are written by the device itself (k_res_export, the last launch of the grouping)
  auto pinned = [](const void *p) {
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
  };
  const bool export_out = cap && (dev || (pinned(out) && pinned(out + (cap - 1))));
  const bool export_per = per_regex_count && b->k && (dev || (pinned(per_regex_count) && pinned(per_regex_count + (b->k - 1))));
  b->h_dst->out = export_out ? out : nullptr;
  b->h_dst->cap = cap;
  b->h_dst->per = export_per ? per_regex_count : nullptr;
  b->h_dst->max_len = max_steps;
  b->h_dst->fresh = 1;
  // ceil(log_sigma n) steps narrow an interval to a single row; what still branches two steps later is rare on any
  // index (a row has one preceding character) and is what the longest chains of dependent steps are made of
  b->h_dst->deep_len = deep_len_now;
  static const bool no_direct = getenv("FMX_EXPORT_DIRECT") && atoi(getenv("FMX_EXPORT_DIRECT")) == 0;      // A/B runs
  b->h_dst->direct = (dev && export_out && !no_direct) ? 1u : 0u;
  const bool direct = b->h_dst->direct != 0;
  b->matches++;
  uint64_t total = b->n_first;               // elements queued for the next launch
  const uint64_t kSmallTotal = (uint64_t)grid_small * per_wg;
  if (b->chain_len != kChain || b->chain_rounds != kRounds) {
    b->drop_graphs();
    b->chain_len = kChain;
    b->chain_rounds = kRounds;
  }
  // The results leave the device grouped by regex (count, scan, scatter, order small groups, totals to pinned host
  // memory).  These launches are enqueued right behind every chain, before the host knows whether the search is
  // over: when it is -- the usual case -- the grouped results are ready at the same synchronisation; when it is
  // not, the grouping is simply done again behind the next chain.  All arguments are fixed for the life of the
  // scratch, so this is a captured graph as well.
  const size_t rcap = b->rcap;
  auto enqueue_group = [&](hipStream_t s, bool in_place) -> hipError_t {
    const dim3 rg(8, kSub);
    if (!parts_in_lds) k_res_scan_add<<<nparts, kScanChunk, 0, s>>>(b->d_rstart, n_scan, b->d_rpart);
    const uint32_t *part = parts_in_lds ? b->d_rpart : nullptr;
    k_res_scatter<<<rg, 256, 0, s>>>(d_res_seg, seg_cap, d_ctl, b->d_rstart, part, nparts, b->d_rfill, d_res, (uint64_t)rcap, b->h_dst);
    k_res_sort<<<(int)((b->k + 255) / 256), 256, 0, s>>>(d_res, (uint64_t)rcap, b->d_rstart, part, nparts, (uint32_t)b->k, b->d_rcnt, b->d_big, b->h_dst, b->h_tot, d_ctl, (uint64_t)b->n_first, ss, pre_next ? 1u : 0u);
    if (!in_place) k_res_export<<<256, 256, 0, s>>>(d_res, (uint32_t)b->k, b->d_rcnt, b->d_big, b->h_dst);
    return hipGetLastError();
  };
  auto capture = [&](hipGraphExec_t *exec, int grid, bool in_place) {     // one graph: the chain of launches, then the grouping
    // one capture at a time in the process: concurrent captures from several host threads (the multi-device entry
    // point matches its slices in parallel) invalidated each other on ROCm 7.2
    static std::mutex capture_mu;
    std::lock_guard<std::mutex> lk(capture_mu);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    if (e == hipSuccess) {
      hipError_t e1 = enqueue_chain(st, grid);
      if (e1 == hipSuccess) e1 = enqueue_group(st, in_place);
      const hipError_t e2 = hipStreamEndCapture(st, &g);
      e = e1 != hipSuccess ? e1 : e2;
    }
    if (e == hipSuccess) e = hipGraphInstantiate(exec, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (e != hipSuccess) { (void)hipGetLastError(); *exec = nullptr; }    // fall back to plain launches
  };
  while (alive) {
    const bool small = total <= kSmallTotal;
    hipGraphExec_t *exec = &b->chain_exec[small ? 1 : 0][direct ? 1 : 0];
    const int grid = small ? grid_small : grid_full;
    // the graphs are captured from a batch's second match on (a one-shot batch would pay the capture and never replay it)
    if (use_graph && b->matches >= 2 && !*exec) capture(exec, grid, direct);
    if (*exec) HIP_TRY(hipGraphLaunch(*exec, st), "hipGraphLaunch(launch chain + grouping)");
    else {
      HIP_TRY(enqueue_chain(st, grid), "k_frontier chain");
      HIP_TRY(enqueue_group(st, direct), "result grouping kernels");
    }
    HIP_TRY(hipEventRecord(e1, st), "hipEventRecord");
    const uint32_t done = small ? kChainSmall : kChain;
    launches += 2 * done + (direct ? 3 : 4);
    b->tag_bound += done;
    HIP_TRY(hipStreamSynchronize(st), "sync(passes)");
    b->h_dst->fresh = 0;               // further chains of this call continue the search
    sum = *b->h_sum;
    pass += done;
    if (sum.overflow & 4ull) { set_error("frontier work queue: an appended entry never became readable"); return FMX_ERR_HIP; }
    if (sum.overflow & 1ull) { set_error("frontier work queue overflow (raise fmx_limits.max_frontier)"); return FMX_ERR_OVERFLOW; }
    total = sum.left;
    n_res = sum.results;
    alive = total != 0;
    truncated = sum.truncated != 0;
    if (trace)
      fprintf(stderr, "[fmx] frontier after launch %u: queue %llu, results %llu, overflow %llu\n", pass,
              (unsigned long long)total, (unsigned long long)n_res, sum.overflow);
  }
  if (pre_next && sum.overflow == 0) {       // the device saw the same (left == 0, no overflow) and left the batch ready: k_res_sort
    b->pre_ok = true;
    b->pre_max_len = max_steps;
    b->pre_deep = deep_len_now;
    b->pre_count = b->n_first;
    b->pre_ss = ss;
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  mark("passes done");
  {
    std::lock_guard<std::mutex> lk(h->mu);
    h->last_kernel_ms = ms;
    h->launches += launches;
  }
  const uint32_t nbig = b->h_tot->n_big;
  struct { unsigned long long res_count; } tot{n_res};
  const size_t extra = b->start_final.size();
  *n_out = (size_t)tot.res_count + extra;
  if ((sum.overflow & 2ull) || tot.res_count + extra > cap) { set_error("result buffer too small"); return FMX_ERR_OVERFLOW; }
  // device-resident results that the host has to touch after all
  fmx_result *const dev_out = out;
  uint32_t *const dev_per = per_regex_count;
  std::vector<fmx_result> stage_out;
  std::vector<uint32_t> stage_per;
  bool staged = false;
  if (dev) {
    if (!extra && !nbig) {
      mark("results on the device");
      if (truncated) {
        set_error("some matches run past max_steps: results hold every match of length <= max_steps");
        return FMX_TRUNCATED;
      }
      return FMX_OK;
    }
    staged = true;
    stage_out.resize((size_t)tot.res_count + extra);
    out = stage_out.data();
    if (per_regex_count) { stage_per.assign(b->k, 0u); per_regex_count = stage_per.data(); }
  }
  if (tot.res_count && (!export_out || staged))
    HIP_TRY(copy_sync(out, direct ? dev_out : d_res, (size_t)tot.res_count * sizeof(fmx_result), hipMemcpyDeviceToHost, st), "D2H(results)");
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
    if (!extra && nbig <= kBigMax) {
      // the device ordered every group of up to kSmallGroup results; the few larger ones are listed
      if (nbig) {
        std::vector<uint32_t> ent(2 * (size_t)nbig);
        HIP_TRY(copy_sync(ent.data(), b->d_big->ent, ent.size() * 4, hipMemcpyDeviceToHost, st), "D2H(big groups)");
        for (uint32_t g = 0; g < nbig; g++)
          if (ent[2 * g + 1]) std::sort(out + ent[2 * g], out + ent[2 * g] + ent[2 * g + 1], by_key);
        if (trace) {
          size_t tot_big = 0;
          for (uint32_t g = 0; g < nbig; g++) tot_big += ent[2 * g + 1];
          fprintf(stderr, "[fmx] %u large result groups (%zu results) ordered on the host\n", nbig, tot_big);
        }
      }
      mark("large groups");
      if (per_regex_count && ndev && (!export_per || staged))
        HIP_TRY(copy_sync(per_regex_count, b->d_rcnt, b->k * 4, hipMemcpyDeviceToHost, st), "D2H(result counts)");
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
  if (staged) {
    HIP_TRY(copy_sync(dev_out, out, (size_t)tot.res_count * sizeof(fmx_result), hipMemcpyHostToDevice, st), "H2D(results)");
    if (dev_per) HIP_TRY(copy_sync(dev_per, per_regex_count, b->k * 4, hipMemcpyHostToDevice, st), "H2D(result counts)");
  }
  mark("results ordered");
  if (truncated) {
    set_error("some matches run past max_steps: results hold every match of length <= max_steps");
    return FMX_TRUNCATED;
  }
  return FMX_OK;
}

}  // namespace fmx

using namespace fmx;

extern "C" {
#ifdef FMX_WAVELOG
int fmx_debug_phaselog(void *out, size_t bytes) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phaselog), std::min(bytes, sizeof g_phaselog)) == hipSuccess ? FMX_OK : FMX_ERR_HIP;
}
int fmx_debug_wavetrace(void *out, size_t bytes) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wavetrace), std::min(bytes, sizeof g_wavetrace)) == hipSuccess ? FMX_OK : FMX_ERR_HIP;
}
int fmx_debug_wavelog(void *out, size_t bytes, int clear) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wavelog), std::min(bytes, sizeof g_wavelog)) != hipSuccess) return FMX_ERR_HIP;
  if (clear) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wavelog)) != hipSuccess || hipMemset(p, 0, sizeof g_wavelog) != hipSuccess) return FMX_ERR_HIP;
  }
  return FMX_OK;
}
#endif

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

// The batched front-end: REParser.re2post + ReTree.apply are independent per regex, so a batch is compiled on all the
// host cores the process may use (fmx_hostpar.h).  status[i] = FMX_OK / FMX_ERR_SYNTAX / FMX_ERR_MATCH as
// fmx_regex_compile would return for res[i]; out[i] = its handle or NULL.
int fmx_regex_compile_batch(const char *const *res, size_t k, int line_only, fmx_regex **out, int *status) {
  if ((k && (!res || !out))) { set_error("null argument"); return FMX_ERR_ARG; }
  for (size_t i = 0; i < k; i++) {
    out[i] = nullptr;
    if (!res[i]) { set_error("null regex string"); return FMX_ERR_ARG; }
  }
  std::atomic<size_t> first_bad{k};
  std::atomic<int> nomem{0};
  auto compile_range = [&](size_t a, size_t b) {
    for (size_t i = a; i < b; i++) {
      int rc = FMX_OK;
      try {
        out[i] = reinterpret_cast<fmx_regex *>(new Regex(compile_regex(res[i], line_only != 0)));
      } catch (const RegexError &e) {
        rc = e.code;
      } catch (...) {       // bad_alloc, length_error ..: nothing may leave a worker thread (std::terminate) or this extern "C" function
        rc = FMX_ERR_NOMEM;
        nomem.store(1);
      }
      if (status) status[i] = rc;
      if (rc != FMX_OK) {
        size_t cur = first_bad.load();
        while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
      }
    }
  };
  parallel_for(k, 256, compile_range);      // (starts as many threads as it can get and never throws: fmx_hostpar.cpp)
  if (nomem.load()) {
    for (size_t i = 0; i < k; i++) { delete reinterpret_cast<Regex *>(out[i]); out[i] = nullptr; }
    set_error("out of host memory");
    return FMX_ERR_NOMEM;
  }
  const size_t bad = first_bad.load();
  if (bad < k) {        // the first failure's message, as the one-regex entry point would have left it
    try { (void)compile_regex(res[bad], line_only != 0); } catch (const RegexError &e) { set_error(e.msg + " (regex " + std::to_string(bad) + " of the batch)"); } catch (...) {}
  }
  return FMX_OK;
}

int fmx_regex_free_batch(fmx_regex *const *res, size_t k) {
  if (k && !res) { set_error("null argument"); return FMX_ERR_ARG; }
  auto free_range = [&](size_t a, size_t b) {
    for (size_t i = a; i < b; i++) delete reinterpret_cast<Regex *>(res[i]);
  };
  parallel_for(k, 4096, free_range);
  return FMX_OK;
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

int fmx_regex_batch_info(const fmx_regex_batch *b, uint64_t *n_regexes, uint64_t *n_states, uint64_t *n_follows,
                         uint64_t *n_firsts) {
  if (!b) { set_error("null argument"); return FMX_ERR_ARG; }
  const RegexBatch *rb = reinterpret_cast<const RegexBatch *>(b);
  if (n_regexes) *n_regexes = rb->k;
  if (n_states) *n_states = rb->n_states;
  if (n_follows) *n_follows = rb->n_fol;
  if (n_firsts) *n_firsts = rb->n_first;
  return FMX_OK;
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
    const RefTables rt{rb->nfa.st, rb->nfa.fol, rb->d_st_num, rb->d_first_off, rb->d_first_state, rb->d_fol_rec, rb->d_first_rec, rb->max_num};
    return regex_match_reference(h, rt, rb->k, rb->max_fanout, lim->max_branching, lim->max_iterations, out, cap, n_out,
                                 per_regex_count, nullptr);
  }
  if (lim && lim->mode != FMX_MATCH_FRONTIER) { set_error("unknown fmx_limits.mode"); return FMX_ERR_ARG; }
  return regex_batch_match(h, rb, lim, out, cap, n_out, per_regex_count);
}

int fmx_regex_batch_match_dev(const fmx_index *idx, fmx_regex_batch *b, const fmx_limits *lim, void *d_out, size_t cap,
                              size_t *n_out, void *d_per_regex_count) {
  if (!idx || !b || !n_out || (cap && !d_out)) { set_error("null argument"); return FMX_ERR_ARG; }
  if (lim && lim->mode != FMX_MATCH_FRONTIER) { set_error("the device-resident form runs the frontier mode"); return FMX_ERR_UNSUPPORTED; }
  return regex_batch_match(reinterpret_cast<const Index *>(idx), reinterpret_cast<RegexBatch *>(b), lim,
                           static_cast<fmx_result *>(d_out), cap, n_out, static_cast<uint32_t *>(d_per_regex_count), true);
}

// ---- one process, several GPUs: the batch is cut into contiguous slices of about equal ESTIMATED
// frontier work, slice r is made resident on idxs[r]'s device, slices are matched from one host thread each and
// their result lists -- each already in canonical order, regex ids ascending across slices -- are concatenated.
struct RegexBatchMulti {
  size_t k = 0;
  std::vector<const Index *> idx;
  std::vector<RegexBatch *> part;
  std::vector<size_t> cut;       // n_idx + 1 slice bounds
  // per-slice result buffers, kept between calls and never value-initialised (a fresh zeroed 100 MB vector per
  // slice and call cost 25 ms)
  struct PinnedFree { void operator()(fmx_result *p) const { if (p) (void)hipHostFree(p); } };
  std::vector<std::unique_ptr<fmx_result[], PinnedFree>> buf;      // page-locked: the device writes a slice's results itself
  std::vector<size_t> buf_cap;
  ~RegexBatchMulti() {
    for (size_t r = 0; r < part.size(); r++)
      if (part[r]) { (void)hipSetDevice(part[r]->device); delete part[r]; }
  }
};

// What a regex is expected to cost: its start elements, its states (each is stepped at least once per path through
// it) and its follow entries (every one is a push); a starred class shows up as many follows.
static double regex_work_estimate(const Regex &re, double n, double sigma) {
  std::vector<double> a, b;
  return frontier_work_estimate(re, n, sigma, a, b) + 4.0 * (double)re.firsts.size();
}

int fmx_regex_batch_create_multi(fmx_index *const *idxs, size_t n_idx, fmx_regex *const *res, size_t k,
                                 fmx_regex_batch_multi **out) {
  if (!idxs || !n_idx || !out || (k && !res)) { set_error("null argument"); return FMX_ERR_ARG; }
  *out = nullptr;
  for (size_t r = 0; r < n_idx; r++) {
    if (!idxs[r]) { set_error("null index handle"); return FMX_ERR_ARG; }
    const Index *a = reinterpret_cast<const Index *>(idxs[r]), *b0 = reinterpret_cast<const Index *>(idxs[0]);
    if (a->n != b0->n || a->eof != b0->eof) { set_error("the handles are not replicas of one index"); return FMX_ERR_ARG; }
  }
  for (size_t r = 0; r < k; r++)
    if (!res[r]) { set_error("null regex handle"); return FMX_ERR_ARG; }
  std::unique_ptr<RegexBatchMulti> m(new RegexBatchMulti());
  m->k = k;
  std::vector<double> cum(k + 1, 0.0);
  {
    const Index *h0 = reinterpret_cast<const Index *>(idxs[0]);
    std::vector<double> w(k, 0.0);
    parallel_for(k, 1024, [&](size_t a, size_t b) {
      for (size_t r = a; r < b; r++)
        w[r] = regex_work_estimate(*reinterpret_cast<const Regex *>(res[r]), (double)h0->n, (double)std::max<uint32_t>(h0->nslots, 2u));
    });
    for (size_t r = 0; r < k; r++) cum[r + 1] = cum[r] + w[r];
  }
  m->cut.assign(n_idx + 1, k);
  m->cut[0] = 0;
  for (size_t r = 1; r < n_idx; r++) {
    const double want = cum[k] * (double)r / (double)n_idx;
    size_t c = (size_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
    if (c > k) c = k;
    m->cut[r] = std::max(c, m->cut[r - 1]);
  }
  for (size_t r = 0; r < n_idx; r++) {
    m->idx.push_back(reinterpret_cast<const Index *>(idxs[r]));
    m->part.push_back(nullptr);
    const size_t a = m->cut[r], b = m->cut[r + 1];
    int rc = regex_batch_create(m->idx[r], reinterpret_cast<const Regex *const *>(res) + a, b - a, &m->part[r]);
    if (rc != FMX_OK) return rc;
  }
  *out = reinterpret_cast<fmx_regex_batch_multi *>(m.release());
  return FMX_OK;
}

int fmx_regex_batch_free_multi(fmx_regex_batch_multi *mb) {
  delete reinterpret_cast<RegexBatchMulti *>(mb);
  return FMX_OK;
}

int fmx_regex_batch_match_multi(fmx_regex_batch_multi *mb, const fmx_limits *lim, fmx_result *out, size_t cap,
                                size_t *n_out, uint32_t *per_regex_count) {
  if (!mb || !n_out || (cap && !out)) { set_error("null argument"); return FMX_ERR_ARG; }
  RegexBatchMulti *m = reinterpret_cast<RegexBatchMulti *>(mb);
  if (lim && lim->mode != FMX_MATCH_FRONTIER) { set_error("the multi-device form runs the frontier mode"); return FMX_ERR_UNSUPPORTED; }
  const size_t np = m->part.size();
  {   // one slice holds the whole batch (one handle, or every regex in one slice): no thread, no merge
    size_t only = np, busy = 0;
    for (size_t r = 0; r < np; r++)
      if (m->cut[r] != m->cut[r + 1]) { only = r; busy++; }
    if (busy == 1 && m->cut[only] == 0) return regex_batch_match(m->idx[only], m->part[only], lim, out, cap, n_out, per_regex_count);
  }
  m->buf.resize(np);
  m->buf_cap.resize(np, 0);
  for (size_t r = 0; r < np; r++)
    if (m->cut[r] != m->cut[r + 1] && m->buf_cap[r] < (cap ? cap : 1)) {
      void *p = nullptr;
      m->buf[r].reset();
      if (hipSetDevice(m->idx[r]->device) != hipSuccess || hipHostMalloc(&p, (cap ? cap : 1) * sizeof(fmx_result), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error("out of page-locked host memory");
        return FMX_ERR_NOMEM;
      }
      m->buf[r].reset(static_cast<fmx_result *>(p));
      m->buf_cap[r] = cap ? cap : 1;
    }
  std::vector<size_t> cnt(np, 0);
  std::vector<int> rc(np, FMX_OK);
  std::vector<std::string> msg(np);
  std::vector<Worker *> busy;          // slice r runs on handle r's own host thread (kept with the handle)
  for (size_t r = 0; r < np; r++) {
    if (m->cut[r] == m->cut[r + 1]) continue;
    Worker *w = worker_of(m->idx[r]);
    w->submit([&, r]() {
      // every slice may fill the caller's whole capacity
      uint32_t *per = per_regex_count ? per_regex_count + m->cut[r] : nullptr;
      rc[r] = regex_batch_match(m->idx[r], m->part[r], lim, m->buf[r].get(), cap, &cnt[r], per);
      if (rc[r] != FMX_OK) msg[r] = fmx_last_error();
    });
    busy.push_back(w);
  }
  for (Worker *w : busy) w->wait();
  size_t total = 0;
  bool truncated = false;
  for (size_t r = 0; r < np; r++) {
    if (rc[r] == FMX_TRUNCATED) { truncated = true; msg[np - 1] = msg[r]; }
    else if (rc[r] != FMX_OK) { set_error(msg[r]); *n_out = cnt[r]; return rc[r]; }
    total += cnt[r];
  }
  *n_out = total;
  if (total > cap) { set_error("result buffer too small"); return FMX_ERR_OVERFLOW; }
  size_t at = 0;
  for (size_t r = 0; r < np; r++) {
    for (size_t j = 0; j < cnt[r]; j++) {
      out[at] = m->buf[r][j];
      out[at].regex += (uint32_t)m->cut[r];
      at++;
    }
  }
  if (truncated) { set_error("some matches run past max_steps: results hold every match of length <= max_steps"); return FMX_TRUNCATED; }
  return FMX_OK;
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

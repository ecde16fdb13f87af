// fmx_host.h -- host-side handle and the launch entry points shared by the .hip/.cpp units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fmx_device.h"
#include "fmx_hostpar.h"

namespace fmx {

// What one host-pointer call needs on the device side: a stream, two events and grow-only scratch buffers.
// A handle keeps a few of them between calls (fmx_api.cpp, class Call), so that a call costs copies and
// launches only -- creating them afresh took longer than a small batch's kernel.
struct CallCtx {
  static constexpr int kBufs = 8;
  hipStream_t stream = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  void *buf[kBufs] = {nullptr};
  size_t cap[kBufs] = {0};
  std::vector<hipEvent_t> chunk_ev;   // per-chunk events of the pipelined batch search (created on first use)
  void *pin = nullptr;            // pinned host staging for small calls (operands packed into one copy each way)
  size_t pin_cap = 0;
  size_t total() const { size_t t = 0; for (size_t c : cap) t += c; return t; }
};

// The k-mer jump table of an index (fmx_ktab.hip): level[j] holds the intervals of all (j+1)-mers, tab = the last.
struct KTab {
  const uint4 *tab = nullptr;
  const uint4 *level[16] = {nullptr};
  const uint4 *const *level_dev = nullptr;   // the same 16 pointers in device memory (kernels that pick a level at run time
                                             // stage them in LDS: indexing the by-value array would go through scratch)
  const uint8_t *dense = nullptr;   // [256] byte -> dense symbol id, 0xFF for bytes without a bit-vector
  uint32_t k = 0, sigma = 0;
  // the regex frontier's row table (fmx_jump.hip, row1_get): per row BWT'[r] << 40 | LF r, or nullptr
  const unsigned long long *row1 = nullptr;
};

// Host-resident rank dictionary of a (block-sized) index, for ONE dependent chain of rank queries (fmx_hostrank.cpp):
// per 256 positions one record = the running count of every symbol that occurs (nslots x u32), then the 256 BWT' bytes --
// a query touches its symbol's count and the bytes behind it, neighbours in memory.
struct HostRank {
  std::vector<uint8_t> blob;        // (n / 256 + 1) records of `stride` bytes
  size_t stride = 0;                // nslots * 4 + 256
};

// What a handle may build beside its rank dictionary, and when (round 5: per handle -- until round 4 these were process-global
// atomics in fmx_jump.hip, so two handles in one JVM could not differ).  fmx_config_set sets the DEFAULTS a handle takes at
// open; fmx_index_config_set changes one handle's own copy afterwards (a table that exists stays until fmx_drop_tables).
struct TablePolicy {
  std::atomic<int> ktab{1};                    // the k-mer jump table: 1 auto, 0 off
  std::atomic<int> jump_mode{7};               // bit 0: row table (R1), bit 1: row jump table (J), bit 2: three-step row table (R3)
  std::atomic<int> jump_pairs{-1};             // -1 auto (indexes of 2^30 rows and more, when 32 n bytes fit), 0 never, 1 whenever they fit
  std::atomic<int> jump_chars{9};
  std::atomic<long long> tables_after{-1};     // patterns before a search builds tables; -1 auto
  // the BUDGET: device bytes all derived tables of this handle (k-mer, row, row jump, select) may hold together
  std::atomic<uint64_t> budget_bytes{~0ull};   // ~0: none (what fits beside the margins)
  std::atomic<uint32_t> budget_ppb{0};         // != 0: a fraction (billionths) of the HBM that is free when a table is decided, the handle's own tables counted as free
  TablePolicy() = default;
  TablePolicy(const TablePolicy &o) { *this = o; }
  TablePolicy &operator=(const TablePolicy &o) {
    ktab.store(o.ktab.load()); jump_mode.store(o.jump_mode.load()); jump_pairs.store(o.jump_pairs.load());
    jump_chars.store(o.jump_chars.load()); tables_after.store(o.tables_after.load());
    budget_bytes.store(o.budget_bytes.load()); budget_ppb.store(o.budget_ppb.load());
    return *this;
  }
};
TablePolicy &default_policy();          // fmx_jump.hip: what fmx_config_set writes and fmx_open* copies
// One policy key = value into `p`: 0 ok, 1 unknown key (not a table key), 2 bad value (*why says which values there are)
int policy_set(TablePolicy &p, const char *key, const char *value, const char **why);

struct Index {
  uint64_t serial = 0;          // unique per open in this process: what a resident regex batch remembers of its index
  int device = 0;
  int cu_count = 256;
  uint64_t n = 0, eof = 0, nblocks = 0;
  uint32_t nslots = 0;
  int64_t counts[256] = {0};
  uint64_t cf[256] = {0};
  uint16_t slot[256] = {0};
  // device memory owned by the handle
  // NaiveBWTSearcher form (fmx_open_block, findex.scala:459-506): bucket starts come from the caller, symbol 0 is an
  // ordinary (absent) symbol, and the block's first byte decides one quirk (build_index)
  bool block_mode = false;
  int64_t block_bs[256] = {0};
  int block_first = -1;         // BWT byte at position 0 (-1: position 0 is the skipped row)
  int block_skipped = -1;       // BWT byte at the skipped row rk0
  uint32_t layout = 0;          // kLayoutOneHot | kLayoutBytes
  void *d_bv = nullptr;
  void *d_chk = nullptr;        // bytes layout: checkpoints
  void *d_sup = nullptr;        // bytes layout: superblock counts
  void *d_bwt = nullptr;
  void *d_cf = nullptr;
  void *d_slot = nullptr;
  unsigned long long *d_counters = nullptr;   // kCounterSlots slots of kCounterStride counters (fmx_device.h)
  DevIndex dev{};
  uint64_t index_bytes = 0;
  double build_ms = 0.0;
  // host-call bookkeeping
  mutable std::mutex mu;
  mutable std::vector<CallCtx *> ctx_pool;      // idle call contexts (guarded by mu)
  // k-mer jump table (fmx_ktab.hip), built on first use
  mutable std::mutex kt_mu;
  mutable bool kt_ready = false;
  mutable KTab kt;
  mutable void *d_ktab = nullptr, *d_kt_dense = nullptr, *d_kt_levels = nullptr;
  mutable uint64_t kt_bytes = 0;
  // row jump table (fmx_jump.hip), built at the first literal search
  mutable std::mutex jt_mu;
  mutable bool jt_ready = false;
  mutable void *d_jump = nullptr;
  mutable uint64_t jump_bytes = 0;
  mutable bool jump_pairs = false;              // the table holds pairs J[r] | J[LF^jc r] (32 bytes per row): up to 2 jc steps per request
  mutable uint32_t jump_chars = 0;              // characters (backward steps) one entry of the row jump table stands for: 8 .. 11
  // row table of the regex frontier (fmx_jump.hip): (BWT'[r], LF r) per row, 8 bytes; built at the first regex match
  mutable std::mutex r1_mu;
  mutable bool r1_ready = false;
  mutable void *d_row1 = nullptr;
  mutable uint64_t row1_bytes = 0;
  // three-step row table (fmx_jump.hip): (BWT'[r], BWT'[LF r], BWT'[LF^2 r]; LF^3 r) per row, 8 bytes; built at the first
  // literal search of a handle whose row jump table does not fit
  mutable std::mutex r3_mu;
  mutable bool r3_ready = false;
  mutable void *d_row3 = nullptr;
  mutable uint64_t row3_bytes = 0;
  mutable double tables_ms = 0.0;             // host time spent building the k-mer table and the select directory (under their mutexes)
  // when the derived tables are built (fmx_jump.hip, tables_due): patterns searched so far, fmx_prepare seen
  // ticket areas of the literal search kernel (fmx_device.h, kTixAreas): which stream owns which.  A launch draws its last
  // batches from the area of the stream it is enqueued on; launches on ONE stream follow each other, so an area serves one
  // launch at a time.  Streams beyond the areas (and streams being captured) get none: their launches stride statically.
  mutable std::mutex tix_mu;
  mutable void *tix_owner[16] = {};
  mutable bool tix_used[16] = {};
  mutable std::atomic<uint64_t> patterns_seen{0};
  mutable std::atomic<uint32_t> search_residency{0};     // fmx_stats.search_residency
  // fmx_prepare has been asked for the k-mer table / the row tables: searches use them (and build what a drop took away)
  // whatever the pattern count says.  One flag per table (one shared flag made prepare(KTAB) build 40 n bytes
  // of row tables inside the next search).
  mutable std::atomic<bool> prepared_ktab{false}, prepared_rows{false};
  mutable TablePolicy policy;                                   // this handle's own (copied from the defaults at open)
  mutable std::atomic<uint64_t> tables_held{0};                 // device bytes of all derived tables right now (what the budget counts)
  mutable std::atomic<uint64_t> tables_alloc_us{0};             // of tables_ms: microseconds spent inside hipMalloc for the tables (the driver wiping memory somebody released: profiles/r05_alloc.md)
  mutable std::atomic<uint64_t> hbm_free_after_tables{0};       // hipMemGetInfo's free bytes right after the last table build (0: none built)
  mutable std::atomic<uint64_t> peak_table_build_bytes{0};      // most device memory a table build held at once (table + its scratch)
  // select directory for Psi (fmx_select.hip), built on first use
  mutable std::mutex sel_mu;
  mutable bool sel_ready = false;
  mutable void *d_sel_dir = nullptr, *d_sel_off = nullptr, *d_sel_shift = nullptr;
  mutable uint64_t sel_bytes = 0;
  mutable uint64_t launches = 0;
  mutable double last_kernel_ms = 0.0;
  // the handle's host thread for the one-process-several-GPUs entry points (made at the first such call; fmx_hostpar.h)
  mutable std::unique_ptr<Worker> worker;
  // host-side rank dictionary (fmx_occ_host / fmx_calc_gaps_chain), built at first use
  mutable std::mutex hr_mu;
  mutable std::unique_ptr<HostRank> hr;
};

Worker *worker_of(const Index *h);     // fmx_api.cpp

void set_error(const std::string &msg);
// Borrow / return one of the handle's call contexts (stream + events created on first use).  ctx_acquire
// returns nullptr and records the error when the stream or events cannot be created.
CallCtx *ctx_acquire(const Index *h);
void ctx_release(const Index *h, CallCtx *c);
hipError_t ctx_scratch(CallCtx *c, int i, size_t bytes, void **out);   // grow-only device scratch buffer i
struct CtxLease {            // scope guard around ctx_acquire / ctx_release
  const Index *h;
  CallCtx *c;
  explicit CtxLease(const Index *hh) : h(hh), c(ctx_acquire(hh)) {}
  ~CtxLease() { if (c) ctx_release(h, c); }
  CtxLease(const CtxLease &) = delete;
  CtxLease &operator=(const CtxLease &) = delete;
};
// The derived tables are built when they have a chance to pay: by fmx_prepare, or at the search that brings the patterns a
// handle has been asked for to `threshold` (fmx_config_set("tables_after", ..)); `build` = false only looks.
bool tables_due(const Index *h, uint64_t k, bool small_table);      // fmx_jump.hip: counts k, then decides
void note_table_build(const Index *h, uint64_t bytes_held);
// Room for one more derived table of this handle: min(free HBM - margin, what the handle's budget leaves); 0 when hipMemGetInfo fails.
uint64_t table_room(const Index *h, uint64_t margin);
void tables_account(const Index *h, int64_t delta);
hipError_t table_malloc(const Index *h, void **p, size_t bytes);    // hipMalloc for a derived table, its time added to tables_alloc_us                 // a table of |delta| bytes was built (+) or freed (-)
hipError_t ktab_get(const Index *h, hipStream_t st, KTab *out, bool build = true);     // fmx_ktab.hip
hipError_t select_prepare(const Index *h, hipStream_t st);          // fmx_select.hip: builds the select directory now
hipError_t jump_get(const Index *h, hipStream_t st, const uint4 **out, bool build = true);   // fmx_jump.hip (nullptr: the handle has none)
hipError_t row1_get(const Index *h, hipStream_t st, const unsigned long long **out, bool build = true);   // fmx_jump.hip (nullptr: none)
hipError_t row3_get(const Index *h, hipStream_t st, const unsigned long long **out, bool build = true);   // fmx_jump.hip (nullptr: none)
int drop_tables(Index *h, unsigned what);       // fmx_jump.hip: fmx_drop_tables
// fmx_search.hip: the residency census of the k_search4 instantiation this handle's full-size searches use now, taken with
// calibration launches on `st` (synchronises it): fmx_prepare's last step, never a _dev call's.
hipError_t search_calibrate(const Index *h, hipStream_t st);
bool force_superblocks();                       // fmx_config_set("checkpoints", "superblock"): the bytes layout's >= 2^32-count form
int layout_preference();                        // -1 auto, else kLayoutOneHot / kLayoutBytes (fmx_config_set)
int hip_fail(hipError_t e, const char *what);   // records the message, returns FMX_ERR_HIP

// fmx_build.hip
int build_index(Index *h, hipStream_t st, const int64_t *given_counts);   // returns an FMX_* status

// fmx_kernels.hip (launch_psi, launch_next_substr: fmx_select.hip)
hipError_t launch_occ(const Index *h, const void *d_c, const void *d_i, void *d_out, uint64_t k, hipStream_t st);
hipError_t launch_prev_range(const Index *h, const void *d_sp, const void *d_ep, const void *d_c, void *d_sp1,
                             void *d_ep1, uint64_t k, hipStream_t st);
hipError_t launch_search(const Index *h, const void *d_pat, const void *d_off, void *d_sp, void *d_ep, uint64_t k,
                         hipStream_t st, uint32_t fixed_len = 0,       // d_off == nullptr: k patterns of fixed_len bytes each
                         uint64_t pack_cap = ~0ull,                    // != ~0: the 8-byte form into d_sp (escape list of pack_cap), d_ep scratch
                         uint32_t flags = 0);                          // kSearchMissNone: fmx.h, FMX_SEARCH_MISS_NONE
constexpr uint32_t kSearchMissNone = 1u;
// the 8-byte form of a batch's intervals (fmx.h: fmx_pack_intervals_dev); in place when d_packed == d_sp
hipError_t launch_pack_intervals(const Index *h, const void *d_sp, const void *d_ep, uint64_t k, uint64_t escape_cap,
                                 void *d_packed, hipStream_t st);
hipError_t launch_unpack_intervals(const Index *h, const void *d_packed, uint64_t k, uint64_t escape_cap, void *d_sp,
                                   void *d_ep, hipStream_t st);
// fmx_config_set("validate", "1"): is d_off[0..k] non-decreasing?  Synchronises the stream.
hipError_t check_offsets(const Index *h, const void *d_off, uint64_t k, hipStream_t st, bool *ok);
hipError_t launch_lf_walk(const Index *h, const void *d_rows, uint64_t k, uint32_t len, void *d_out, void *d_end,
                          hipStream_t st);
hipError_t launch_fm_fill(const Index *h, void *d_fm, hipStream_t st);
hipError_t launch_psi(const Index *h, const void *d_rows, void *d_out, uint64_t k, hipStream_t st);
hipError_t launch_next_substr(const Index *h, const void *d_sps, uint64_t k, uint32_t len, void *d_out,
                              void *d_out_len, hipStream_t st);

// fmx_refmatch.hip
struct RefTables;
}  // namespace fmx
struct fmx_result;
namespace fmx {
int regex_match_reference(const Index *h, const RefTables &rt, size_t k, uint32_t max_fanout, uint32_t max_branching,
                          uint32_t max_iterations, fmx_result *out, size_t cap, size_t *n_out,
                          uint32_t *per_regex_count, uint32_t *front_left);

}  // namespace fmx

/*
 * fmx.h -- C ABI of libfmx.so, the MI355X-native FM-index backward-search engine.
 *
 * This is the drop-in boundary for findex's search hot path.  The reference
 * (martende/findex, Scala) has no FFI layer; its seam is the Scala trait
 * SuffixAlgo / SuffixWalkingAlgo (src/main/scala/org/fmindex/findex.scala:9-57)
 * that every search engine is written against.  Each entry point below names the
 * reference member it replaces; INTEGRATION.md shows the JNI stub and the Scala
 * adapter class (`HipFMSearcher extends SuffixWalkingAlgo`) a maintainer would
 * add on the reference side.
 *
 * Conventions
 *  - Every function returns an int status (FMX_OK == 0).  No exception crosses the
 *    ABI; fmx_last_error() gives a thread-local message for the last failure.
 *  - "No match" is a value, not an error: a miss is reported as sp == ep
 *    (the Scala adapter maps that to None, findex.scala:30,35).
 *  - Positions are uint64 (the reference's Int widened; identical for n < 2^31).
 *  - Symbols are unsigned bytes.  The reference indexes with a signed Byte and
 *    throws on bytes >= 0x80 (findex.scala:21,26); accepting them is a documented
 *    superset.
 *  - The caller owns every buffer it passes; the library keeps no caller pointer
 *    after a call returns (fmx_open_dev copies the BWT).  Handles own their device
 *    memory.  An index handle is immutable after open: concurrent calls on one
 *    handle are allowed from different host threads.
 *  - Plain entry points take HOST pointers and move data themselves.  The `_dev`
 *    twins take DEVICE pointers plus a hipStream_t (as void*) and only enqueue
 *    work: inputs already resident in HBM, outputs left in HBM.  (One exception, stated at
 *    fmx_prepare: the single search that builds a handle's derived tables at its threshold when the
 *    caller never called fmx_prepare; fmx_regex_batch_match_dev is synchronous by design.)
 *  - There is no CPU fallback anywhere: without a usable HIP device every compute
 *    entry point fails with FMX_ERR_HIP.
 */
#ifndef FMX_H
#define FMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMX_ABI_VERSION 5

enum {
  FMX_OK = 0,
  FMX_ERR_IO = 1,          /* cannot open/read a file (reference: java.io exceptions) */
  FMX_ERR_FORMAT = 2,      /* bad size/header ("File %s bad size", bwtmerger.scala:153,261-262) */
  FMX_ERR_ARG = 3,         /* null handle / out-of-range argument (reference: ArrayIndexOutOfBounds) */
  FMX_ERR_NOMEM = 4,       /* host or device allocation failed */
  FMX_ERR_HIP = 5,         /* HIP runtime error / no device */
  FMX_ERR_UNSUPPORTED = 6, /* valid input this build cannot serve */
  FMX_ERR_SYNTAX = 7,      /* "re2post syntax" (re2/re2.scala:84,87,109,133,141,159,174) */
  FMX_ERR_MATCH = 8,       /* scala.MatchError from ReTree.apply (re2/retree.scala:235-238,291-294) */
  FMX_ERR_OVERFLOW = 9,    /* a caller-sized output or a device work queue was too small */
  FMX_TRUNCATED = 10       /* not an error: outputs are valid but a search was cut at a limit (regex max_steps) */
};

typedef struct fmx_index fmx_index;   /* replaces class NaiveFMSearcher, bwtmerger.scala:335-421 */
typedef struct fmx_regex fmx_regex;   /* replaces class ReTree, re2/retree.scala:485 */
typedef struct fmx_regex_batch fmx_regex_batch;   /* a set of compiled regexes made resident on a device */

const char *fmx_last_error(void);
int fmx_abi_version(void);
/* Process-wide options.  The table keys ("ktab", "jump", "jump_pairs", "jump_chars", "tables_after", "table_budget") are the
 * DEFAULTS a handle copies when it is opened; fmx_index_config_set changes one handle's own copy afterwards, so two handles
 * in one process (one JVM) can differ.  key "layout": "auto" (default: one-hot bit-vectors, one 64-byte block per
 * rank query, when sigma*n/7 bytes fit in free HBM and n < 2^37; else BWT bytes + checkpoints, two
 * lines per rank query), "onehot", "bytes".  key "ktab": "auto" (default) / "off": the k-mer jump table that
 * answers a search's first K backward steps with one lookup (built on a handle's first search; K is chosen from n
 * and the alphabet: fmx_stats_t.ktab_k).  key "validate": "0" (default) / "1": the device-pointer search entry
 * point then checks on the device that d_off is non-decreasing and fails with FMX_ERR_ARG otherwise (one more small
 * kernel and a synchronisation per call: a debugging aid; the host-pointer form always checks).  key "checkpoints": "auto" (default: the bytes layout keeps absolute
 * 32-bit checkpoints whenever every symbol occurs fewer than 2^32 times) or "superblock" (always the relative
 * checkpoints + 64-bit superblock counts that larger counts need; for tests).  Affects indexes opened afterwards.
 * key "jump": "auto" (default) / "jumps" / "rows3" / "rows" / "off": the derived tables that serve searches whose
 * interval has narrowed to ONE ROW, where a backward step is a comparison with the text in front of that row's suffix:
 *   - the row jump table: per row the nine BWT characters an LF walk from it reads and the row it ends on, 16 n
 *     bytes -- nine steps of a literal search with one 16-byte lookup when the pattern's next nine characters match
 *     (key "jump_chars": "8" .. "11" characters per entry for tables built afterwards; 9 is the default)
 *     (built when 16 n bytes + 8 GiB of HBM are free -- one allocation: key "tables_after" says when).
 *     Key "jump_pairs": "auto" (default) / "on" / "off": the table holds PAIRS of entries, a row's own and that of the
 *     row it lands on, side by side in 32 bytes -- one memory request (a 64-byte sector either way) then serves up to
 *     2 x jump_chars steps.  32 n bytes; "auto" builds pairs for the one-hot layout when the index has 2^30 rows or more
 *     and 32 n bytes + 8 GiB are free ("on": whenever they are free);
 *   - the three-step row table: the same with three characters, 8 n bytes -- built instead where the jump table does
 *     not fit; the one-row part of every pattern is then walked by one lane per pattern;
 *   - the row table: BWT'[r] and LF r in 8 bytes per row -- the regex frontier steps its one-row elements with it
 *     (built at a handle's first regex match when 8 n bytes + 4 GiB are free); a literal search on a handle that has
 *     it, and no jump table, uses it like the three-step table.
 * "auto" builds what fits, at first use or in fmx_prepare; "jumps" / "rows3" / "rows" allow only the one; "off" none.
 * fmx_stats_t.jump_bytes, .row_bytes.  Results and executed-step counts are the same with and without them.
 * key "tables_after": "auto" (default) / N: WHEN a handle's derived tables are built.  A table pays only for a caller that
 * searches much: at C3 (n = 2^32) the row tables take ~1.5 s and 96 GiB and save 0.4 ms per million patterns.  So they are
 * built by fmx_prepare, or by the search that brings the patterns the handle has been asked for to the threshold -- auto:
 * n / 64 patterns (at least 65536) for the row tables, 1024 patterns for the k-mer table; N: that many for both (0: at the
 * first search).  A per-call adapter's single queries never build one.  (The regex frontier builds the k-mer table and its
 * row table at a handle's first match: a frontier is thousands of steps.)  fmx_stats_t.patterns_seen,
 * .peak_table_build_bytes; fmx_drop_tables frees them again.
 * key "table_budget": "auto" (default) / N / F: the most device memory ALL derived tables of a handle (k-mer table, row
 * tables, row jump table, select directory) may hold together -- N bytes, or a fraction F written with a point ("0.5": that
 * share of the HBM that is free when a table is decided, the handle's own tables counted as free); "auto": whatever fits
 * beside the built-in margins (8 GiB for the row jump table, 4-8 GiB for the others).  Under a budget the k-mer table takes
 * at most a quarter of it, then the three-step row table (8 n), then the row jump table as pairs (32 n) if that still fits,
 * else as single entries (16 n), else not at all -- searches give the same answers with any subset.  At C3 (n = 2^32) "auto"
 * ends up holding 4 + 32 + 128 GiB beside the 81 GiB index; "100000000000" keeps it to 4 + 32 + 64 GiB.
 * fmx_stats_t.tables_held_bytes, .table_budget_bytes, .hbm_free_after_tables; fmx_prepare_ex sets it with the build.
 * key "pipeline": "off" (default) / "on": fmx_search_batch with 128 k patterns or more in page-locked buffers cut into
 * chunks whose uploads, searches and downloads overlap on three streams, instead of whole arrays up, one search, whole
 * arrays down.  Same results; which is faster depends on how the platform's asynchronous copies compare with its
 * synchronous ones (here: the synchronous form, hence the default).
 * key "threads": host threads the library's own parallel parts use (fmx_regex_compile_batch, making a regex batch
 * resident); "0" = detect (default). */
int fmx_config_set(const char *key, const char *value);
/* Page-locked host memory for batch buffers (hipHostMalloc): the host-pointer entry points move such buffers
 * by DMA at link speed; pageable memory works everywhere too, through the runtime's staging copies.  A JNI adapter
 * wraps these in direct ByteBuffers. */
int fmx_host_alloc(size_t bytes, void **out);
int fmx_host_free(void *p);
/* Number of HIP devices visible (0 and FMX_OK when there is none). */
int fmx_device_count(int *count);

/* ---- open / close ------------------------------------------------------------
 * fmx_open      : NaiveFMSearcher(filename, bigEndian) constructor, bwtmerger.scala:335-353,
 *                 reading X.bwt (BWTLoader :144-174) and X.aux (AUXLoader :130-142).  The
 *                 reference's third file, X.fm (FMLoader :252-290), is not needed: its content
 *                 is a function of the other two (FMCreator :424-533) and the device rank
 *                 dictionary is built from them directly.  An X.fm that IS there (the .bwt's name
 *                 with the extension swapped, :33-36) is held to FMLoader's checks -- element size
 *                 4, size * 4 + 9 == file length (:259-262) -- and to n = fm.size (:339) being the
 *                 .bwt's n: FMX_ERR_FORMAT otherwise, as the reference throws.
 * fmx_open_mem  : the same from host memory (bwt[n] raw bytes incl. the filler at slot eof,
 *                 counts[256] = the .aux array).
 * fmx_open_dev  : the same with the BWT bytes already in device memory; counts may be NULL
 *                 (then they are computed on the device).
 * counts[0] must be 0 (the reference's readers escape byte 0, bwtreader.scala:136-155, and its
 * FMCreator gives symbol 0 exactly the one EOF slot, bwtmerger.scala:440-450).  Unlike the
 * reference, open verifies counts against the BWT and fails with FMX_ERR_FORMAT on mismatch. */
int fmx_open(const char *bwt_path, const char *aux_path, int big_endian, int device, fmx_index **out);
int fmx_open_mem(const uint8_t *bwt, uint64_t n, uint64_t eof, const int64_t counts[256], int device,
                 fmx_index **out);
int fmx_open_dev(const void *d_bwt, uint64_t n, uint64_t eof, const int64_t *counts_or_null, int device,
                 void *stream, fmx_index **out);
/* fmx_open_block : class NaiveBWTSearcher(bwt, bucketStarts, rk0), findex.scala:459-506 -- the searcher
 *                  BWTMerger2.calcGaps (bwtmerger.scala:981-1023) uses over one block's raw BWT.  cf(c) is the
 *                  caller's bucket_starts[c]; occ(c, key) = occurrences of byte c in bwt[0..key] without row rk0
 *                  (c is taken & 0xff by the byte-typed batch entry points, like :480), including the reference's
 *                  last-slot rule (:500-502): a symbol whose only occurrence is position 0 answers 0.  The block
 *                  must not contain byte 0 (FMX_ERR_UNSUPPORTED; the merger's input is 0-free).  The handle
 *                  serves every entry point (search, getPrevRange ... are inherited from SuffixAlgo there too). */
int fmx_open_block(const uint8_t *bwt, uint64_t n, const int64_t bucket_starts[256], uint64_t rk0, int device,
                   fmx_index **out);
int fmx_close(fmx_index *idx);          /* NULL: nothing to close, FMX_OK */
/* One handle's own table policy (the keys fmx_config_set lists as table keys; FMX_ERR_ARG for any other key or a bad value).
 * Tables that exist are not touched: fmx_drop_tables + fmx_prepare rebuild under the new policy. */
int fmx_index_config_set(fmx_index *idx, const char *key, const char *value);
/* Builds now what a handle otherwise builds when its searches reach the "tables_after" threshold (literal search) or
 * at first use (Psi, regex match) -- FMX_PREPARE_KTAB: the k-mer jump table (up to min(16 GiB, a quarter of the free
 * HBM, a quarter of the budget)); FMX_PREPARE_SELECT: the select directory (first Psi / nextSubstr; at most ~n bytes);
 * FMX_PREPARE_JUMP: the tables of the literal search's one-row part -- the three-step row table and the row jump table
 * (8 n + 16 n or 32 n bytes, one allocation each; each skipped when that much HBM is not free or not in the handle's
 * budget); FMX_PREPARE_FRONTIER: the regex frontier's row table (8 n bytes, otherwise built at the first regex match);
 * FMX_PREPARE_SEARCH (implied by KTAB and JUMP): calibrates the literal search kernel the handle's tables select -- a few
 * 60-us launches on the library's own stream that count how many of its workgroups a CU really holds
 * (fmx_stats_t.search_residency).  Each flag stands for its own table only: FMX_PREPARE_KTAB alone leaves the row tables to
 * their threshold.
 * THE CONTRACT: after fmx_prepare has built a table, no later call builds, allocates for, or synchronises a stream for
 * THAT table; with KTAB | JUMP (| FRONTIER | SELECT for regex / Psi callers) prepared, every _dev entry point only
 * enqueues work -- safe inside a stream capture, and a caller that times its first search times a search.  Without
 * fmx_prepare, the ONE search that brings a handle to its "tables_after" threshold builds the tables inside the call
 * (it allocates and synchronises `stream`; FMX_ERR_HIP under a stream capture), and calibrates; every other _dev
 * call only enqueues.  A table that cannot be built (no memory, no budget) is left out: searches then walk every step on
 * the rank dictionary, with the same results.  The time spent is reported as fmx_stats_t.tables_build_ms.
 * fmx_prepare_ex: the same under a budget -- budget_bytes != 0 becomes the handle's "table_budget" first. */
enum { FMX_PREPARE_KTAB = 1, FMX_PREPARE_SELECT = 2, FMX_PREPARE_JUMP = 4, FMX_PREPARE_FRONTIER = 8, FMX_PREPARE_SEARCH = 16 };
int fmx_prepare(const fmx_index *idx, unsigned what);
int fmx_prepare_ex(fmx_index *idx, unsigned what, uint64_t budget_bytes);
/* Frees derived tables again (what = FMX_PREPARE_KTAB | FMX_PREPARE_JUMP | FMX_PREPARE_FRONTIER in any combination: the
 * k-mer table / the row jump table and the three-step row table, 16-32 n + 8 n bytes / the frontier's row table, 8 n
 * bytes) and forgets the handle's pattern count, so that they come back only by fmx_prepare or when the threshold is met
 * anew (under the handle's policy as it is then).  No other call may be using the handle. */
int fmx_drop_tables(fmx_index *idx, unsigned what);

/* ---- scalars: SuffixAlgo.n / cf, findex.scala:10-12; NaiveFMSearcher.cf bwtmerger.scala:346-352 */
int fmx_n(const fmx_index *idx, uint64_t *n);
int fmx_eof(const fmx_index *idx, uint64_t *eof);
int fmx_cf(const fmx_index *idx, int c, uint64_t *out);
int fmx_counts(const fmx_index *idx, int64_t out[256]);
int fmx_device(const fmx_index *idx, int *device);

/* ---- batched rank: SuffixAlgo.occ(c,i), findex.scala:13; NaiveFMSearcher.occ bwtmerger.scala:354-375.
 * out[q] = #{p <= i[q] : BWT'[p] == c[q]}, BWT' = BWT with slot eof read as symbol 0; i = -1 gives 0;
 * i >= n is clamped to n-1 (what the binary search returns for any key past the last entry). */
int fmx_occ_batch(const fmx_index *idx, const uint8_t *c, const int64_t *i, uint64_t *out, size_t k);
int fmx_occ_batch_dev(const fmx_index *idx, const void *d_c, const void *d_i, void *d_out, size_t k, void *stream);

/* ---- batched literal backward search: SuffixAlgo.search, findex.scala:15-31.
 * Pattern q is pat[off[q] .. off[q+1]) (off has k+1 entries), matched last byte first from (0, n);
 * sp[q], ep[q] receive the loop's final values: a hit iff sp < ep, a miss has sp == ep.
 * An empty pattern yields (0, n).  The host form validates the offsets (non-decreasing); the device form
 * cannot look at them: d_off must hold k+1 non-decreasing offsets into the d_pat buffer, or the kernel reads
 * outside it.  A host-pointer batch of 128k patterns or more whose four buffers are page-locked (fmx_host_alloc) is
 * pipelined in chunks over two streams (copies of one chunk beside the kernel of another);
 * fmx_stats_t.last_kernel_ms is then the whole device side of the call. */
int fmx_search_batch(const fmx_index *idx, const uint8_t *pat, const uint64_t *off, uint64_t *sp, uint64_t *ep,
                     size_t k);
int fmx_search_batch_dev(const fmx_index *idx, const void *d_pat, const void *d_off, void *d_sp, void *d_ep,
                         size_t k, void *stream);
/* The lean forms of the same search (round 4; what the host link and the multi-GPU exchange carry):
 *   fixed_len > 0 : every pattern has this many bytes and pattern q is pat[q * fixed_len ..): `off` is not read and may be
 *                   NULL -- 8 bytes per pattern less up the link, and the kernels compute the offsets instead of loading them;
 *   packed & FMX_SEARCH_PACKED : the intervals come back in the 8-BYTE FORM, into `sp` (fmx_packed_words(k, escape_cap) words; `ep` is
 *                   not written and may be NULL in the host form; in the device form d_ep is k words of scratch and d_sp's
 *                   first k words are overwritten in place).
 * The 8-byte form: word q = sp[q] | w << 40 with w = min(ep[q] - sp[q], 0xFFFFFF) -- rows are < 2^38, and a miss (sp == ep:
 * None in the reference, findex.scala:30) keeps the loop's sp at its failing step with w = 0.  An interval of 2^24 - 1 rows
 * or more (patterns of a character or two) has w = 0xFFFFFF and its ep in the ESCAPE LIST behind the k words: word k = the
 * number of such intervals, then pairs (q, ep[q]) in no particular order, room for escape_cap of them.  When more than
 * escape_cap intervals are that wide, word k still counts them all, fmx_unpack_intervals returns FMX_ERR_OVERFLOW and the
 * caller asks for the 16-byte form instead (device side: compare word k with escape_cap).
 *   packed & FMX_SEARCH_MISS_NONE : a pattern that does not occur MAY come back as (0, 0) instead of the loop's values at its
 *                   failing step.  SuffixAlgo.search returns None for it either way (findex.scala:30: `if (sp < ep) Some((sp, ep))
 *                   else None`) -- the values a miss ends with are not observable through the reference's API, and finding them
 *                   costs a walk of up to five dependent memory round trips per miss where a row-table lookup has already shown
 *                   that the pattern's text differs from its row's.  Hits are unchanged; fmx_stats' rank_queries counts the
 *                   reference loop's steps on every pattern as before (they are known from where the texts differ).  Which
 *                   misses are canonicalised is the kernels' business (those found by a table lookup; a miss found by a rank
 *                   query keeps its values): test sp < ep, nothing else.  The JVM adapter, which returns Option, always asks.
 * opts == NULL is fmx_search_batch[_dev] exactly. */
#define FMX_SEARCH_PACKED 1u
#define FMX_SEARCH_MISS_NONE 2u
typedef struct fmx_search_opts {
  uint32_t fixed_len;
  uint32_t packed;       /* bit set: FMX_SEARCH_PACKED | FMX_SEARCH_MISS_NONE (1 = the packed form, as until ABI 5) */
  uint64_t escape_cap;
} fmx_search_opts;
int fmx_search_batch_ex(const fmx_index *idx, const uint8_t *pat, const uint64_t *off, uint64_t *sp, uint64_t *ep,
                        size_t k, const fmx_search_opts *opts);
int fmx_search_batch_ex_dev(const fmx_index *idx, const void *d_pat, const void *d_off, void *d_sp, void *d_ep,
                            size_t k, const fmx_search_opts *opts, void *stream);
/* The 8-byte form on its own: pack device-resident (sp, ep) arrays (d_packed may be d_sp: in place), unpack them on the
 * device (the first min(word k, escape_cap) escape entries are applied) or on the host (a decode of the caller's own
 * buffer: FMX_ERR_OVERFLOW as above, FMX_ERR_FORMAT when an escape entry names a pattern >= k). */
size_t fmx_packed_words(size_t k, size_t escape_cap);
int fmx_pack_intervals_dev(const fmx_index *idx, const void *d_sp, const void *d_ep, size_t k, size_t escape_cap,
                           void *d_packed, void *stream);
int fmx_unpack_intervals_dev(const fmx_index *idx, const void *d_packed, size_t k, size_t escape_cap, void *d_sp,
                             void *d_ep, void *stream);
int fmx_unpack_intervals(const uint64_t *packed, size_t k, size_t escape_cap, uint64_t *sp, uint64_t *ep);

/* One process, several GPUs: the batch is cut into contiguous slices balanced by pattern bytes, slice r is
 * searched on idxs[r] (one handle per device, every handle a replica of the same index) from its own host
 * thread, and each slice's intervals land in their range of sp / ep -- the single-process form of SURVEY.md 8e
 * (no collective; the one-process-per-GPU form with an RCCL all-gather is findex_amd/distributed.py). */
int fmx_search_batch_multi(fmx_index *const *idxs, size_t n_idx, const uint8_t *pat, const uint64_t *off,
                           uint64_t *sp, uint64_t *ep, size_t k);

/* ---- batched single step: SuffixAlgo.getPrevRange(sp,ep,c), findex.scala:32-36.
 * sp1 = cf(c)+occ(c,sp-1), ep1 = cf(c)+occ(c,ep-1); empty iff sp1 >= ep1.  Needs sp <= ep <= n. */
int fmx_prev_range_batch(const fmx_index *idx, const uint64_t *sp, const uint64_t *ep, const uint8_t *c,
                         uint64_t *sp1, uint64_t *ep1, size_t k);
int fmx_prev_range_batch_dev(const fmx_index *idx, const void *d_sp, const void *d_ep, const void *d_c,
                             void *d_sp1, void *d_ep1, size_t k, void *stream);

/* ---- character-class step: SuffixAlgo.getIntervalPrevRange(sp,ep,cstart,cend), findex.scala:37-51.
 * All c in [cstart, cend] (inclusive, 0 <= cstart, cend <= 255); only non-empty ranges are
 * returned, in DESCENDING c like the reference's prepended list.  out arrays need cend-cstart+1
 * slots; *n_out = number written. */
int fmx_interval_prev_range(const fmx_index *idx, uint64_t sp, uint64_t ep, int cstart, int cend,
                            uint64_t *out_sp, uint64_t *out_ep, uint8_t *out_c, size_t *n_out);

/* ---- LF walks: NaiveFMSearcher.getPrevI / prevSubstr, bwtmerger.scala:386-389,409-419.
 * For each start row: `len` times emit BWT'[row] (0 at the EOF row) and step row = LF(row).
 * out_bytes is k*len bytes (walk q at q*len, in emission order = prevSubstr's string);
 * end_rows (optional) receives the row after the last step. */
int fmx_lf_walk_batch(const fmx_index *idx, const uint64_t *rows, size_t k, uint32_t len, uint8_t *out_bytes,
                      uint64_t *end_rows);
int fmx_lf_walk_batch_dev(const fmx_index *idx, const void *d_rows, size_t k, uint32_t len, void *d_out_bytes,
                          void *d_end_rows, void *stream);

/* ---- Psi walks: NaiveFMSearcher.getNextI / nextSubstr, bwtmerger.scala:390-405.
 * fmx_psi_batch: out[q] = fm[rows[q]] (the inverted-list entry = select on the rank dictionary).
 * fmx_next_substr: the reference's nextSubstr(sp,len): walk Psi, stop after a 0 byte, reversed;
 * out needs len bytes, *out_len = bytes written. */
int fmx_psi_batch(const fmx_index *idx, const uint64_t *rows, uint64_t *out, size_t k);
int fmx_psi_batch_dev(const fmx_index *idx, const void *d_rows, void *d_out, size_t k, void *stream);
/* Device form of the batched nextSubstr: d_out is k*len bytes in WALK order (the reference's `ret` before its
 * final .reverse, bwtmerger.scala:404), d_out_len[q] (uint32) = bytes of walk q written. */
int fmx_next_substr_batch_dev(const fmx_index *idx, const void *d_rows, size_t k, uint32_t len, void *d_out,
                              void *d_out_len, void *stream);
/* The first Psi / nextSubstr call on a handle builds a select directory on the device (at most ~n bytes; it is
 * counted in fmx_stats_t.index_bytes from then on). */
int fmx_next_substr(const fmx_index *idx, uint64_t sp, uint32_t len, uint8_t *out, uint32_t *out_len);
/* the same for k rows at once (rendering a result list): out is k*len bytes, row q's string at q*len, out_len[q]
 * bytes of it written. */
int fmx_next_substr_batch(const fmx_index *idx, const uint64_t *rows, size_t k, uint32_t len, uint8_t *out,
                          uint32_t *out_len);
int fmx_prev_substr(const fmx_index *idx, uint64_t sp, uint32_t len, uint8_t *out);
/* Both directions behind one entry point (the name SURVEY.md 8b lists): direction > 0 = nextSubstr (the text
 * that starts at row's suffix, what SAResult.toString prints, re2.scala:11-15), direction < 0 = prevSubstr
 * (len bytes, *out_len = len).  out needs len bytes. */
int fmx_extract(const fmx_index *idx, uint64_t row, uint32_t len, int direction, uint8_t *out, uint32_t *out_len);

/* ---- BWTMerger2.calcGaps' rank loop, bwtmerger.scala:981-1023.  calcGaps keeps ONE running rank over
 * the bytes of the older text -- curRank = bucketStarts(c) + searcher.occ(c, curRank - 1): each rank query needs the
 * one before, so the loop cannot be batched, and one such chain runs slower on the GPU than on a host core
 * (profiles/r02_calcgaps_chain.txt).  These two entry points therefore answer on the HOST, from a host copy of the
 * handle's BWT' with symbol counts every 256 positions (built at first use; n <= 2^31: a merge block) -- one
 * count + a scan of < 256 bytes per query where NaiveBWTSearcher.occ (findex.scala:479-505) binary-searches an
 * inverted list.  Same answers as fmx_occ_batch on the handle (fmx_open_block quirks included).  Not a fallback:
 * every batch entry point stays on the device.
 * fmx_occ_host        : one occ(c, i) -- what a `searcher: SuffixAlgo` adapter hands calcGaps unchanged.
 * fmx_calc_gaps_chain : the loop's rank chain itself over k bytes from curRank = rank0: ranks[j] = the rank after byte
 *                       j, with the loop's own correction for c == last_char (last_char < 0: none): a rank above
 *                       rklst is bumped by one (:1011-1012); a rank EQUAL to rklst needs the caller's KMP buffer /
 *                       longSuffixCmp (:1004-1010), so the chain stops there: *done = j < k, ranks[j] = the
 *                       uncorrected rank; the caller fixes ranks[j] and calls again from byte j + 1 with rank0 =
 *                       the fixed value.  *done == k: all bytes processed. */
int fmx_occ_host(const fmx_index *idx, int c, int64_t i, uint64_t *out);
int fmx_calc_gaps_chain(const fmx_index *idx, const uint8_t *c, size_t k, uint64_t rank0, int last_char, uint64_t rklst,
                        uint64_t *ranks, size_t *done);

/* ---- FMCreator.create, bwtmerger.scala:424-533: writes the reference's .fm file (inverted position
 * lists, 4-byte big-endian entries) from the device structure, so that findex's own NaiveFMSearcher
 * and tests can consume an index this engine prepared.  n must be < 0xffffffff (the reference has
 * no 8-byte entry format, :465-469). */
int fmx_write_fm(const fmx_index *idx, const char *path);

/* ---- regex: REParser.re2post (re2/re2.scala:50-185) + ReTree.apply (re2/retree.scala:156-370).
 * Bytes of `re` are Latin-1 characters.  FMX_ERR_SYNTAX / FMX_ERR_MATCH mirror the reference's
 * "re2post syntax" exception and scala.MatchError. */
int fmx_regex_compile(const char *re, int line_only, fmx_regex **out);
int fmx_regex_free(fmx_regex *re);
/* The same for k regexes at once, compiled on all the host cores the process may use (its affinity mask capped by the
 * cgroup CPU quota; fmx_config_set("threads", "N") overrides): out[i] = the handle of res[i] or NULL, status[i]
 * (optional) = FMX_OK / FMX_ERR_SYNTAX / FMX_ERR_MATCH exactly as fmx_regex_compile(res[i]) returns.  The call itself
 * returns FMX_OK when it ran (whatever the regexes' own statuses; fmx_last_error() then describes the first one that
 * failed), FMX_ERR_ARG / FMX_ERR_NOMEM otherwise.  The reference compiles one regex per REParser.re2post + ReTree
 * call (re2/re2.scala:50-185, re2/retree.scala:156-423); a batch of 100 k is what config C4 hands over at once.
 * fmx_regex_free_batch frees k handles (NULL entries allowed). */
int fmx_regex_compile_batch(const char *const *res, size_t k, int line_only, fmx_regex **out, int *status);
int fmx_regex_free_batch(fmx_regex *const *res, size_t k);
/* Flat Glushkov tables (what ReTree._matchSA touches): per CharNode its byte, `num`
 * (retree.scala:393-423), isLast (:40-50), and `follows` (:14-38) as a CSR list that keeps the
 * reference's order and multiplicity; `firsts` = root.firsts.  Any output pointer may be NULL;
 * sizes come back through the n_* pointers. */
int fmx_regex_tables(const fmx_regex *re, uint32_t *n_states, uint8_t *st_c, int32_t *st_num, uint8_t *st_last,
                     int32_t *fol_off /* n_states+1 */, uint32_t *n_follows, int32_t *fol, uint32_t *n_firsts,
                     int32_t *firsts);
/* re2poststr, re2/re2.scala:187 (UTF-8). */
int fmx_regex_post_string(const char *re, int line_only, char *out, size_t cap);

enum {
  FMX_MATCH_FRONTIER = 0,   /* breadth-first over the whole batch: every match, the throughput path */
  FMX_MATCH_REFERENCE = 1   /* the reference's own pop order and limits, one regex per lane group */
};

/* ---- the reference's two other SA-interval engines, served by the same frontier kernel:
 * fmx_nfa_compile : REParser.createNFA (re2/re2.scala:264-334) for REParser.matchSA (:568-693): Thompson
 *                   NFA, epsilon closures folded into the tables.  `src` is a regex for re2post or, with
 *                   src_is_postfix, a postfix string for post2re (:188-205) as the reference's tests use.
 *                   FMX_ERR_MATCH where the reference throws scala.MatchError ([..] sets; a regex that
 *                   matches the empty string).  matchSA's maxLength is fmx_limits.max_steps.
 * fmx_dfa_compile : DFA.compileBuckets + DFA.matchSA (dfa.scala:190-213,231-289) over a caller-built
 *                   transition table moves[nstates][nchars] (-1 = none), finish[nstates]: as in the
 *                   reference only single-character actions expand (runs of equal targets are "buckets",
 *                   which StatePoint.expand ignores, :247-251).
 * Both return an fmx_regex handle for fmx_regex_match_batch / fmx_regex_batch_* in frontier mode: every match.
 * The reference's cuts for these two engines -- REParser.matchSA's maxIterations (re2.scala:612) and DFA.matchSA's
 * 500 iterations (dfa.scala:268) -- stop the search after a number of POPS, and the pop order is not defined by the
 * reference's source: the NFA engine seeds its queue from an immutable Set of state objects hashed by identity and
 * takes equal-length elements in heap-layout order, the DFA engine takes `head` of an immutable HashSet.  Two
 * orders the source allows return different results once a cut binds (and the same multiset while it does not):
 * tests/test_oracle_engines.py::test_order_dependent_cuts_are_not_reproducible_without_the_jvm.  So there is no
 * reference-order mode for them; FMX_MATCH_REFERENCE exists for ReTree, whose order IS defined by its source
 * (CharNode.num and Scala's binary heap). */
int fmx_nfa_compile(const char *src, int line_only, int src_is_postfix, fmx_regex **out);
int fmx_dfa_compile(const int32_t *moves, uint32_t nstates, uint32_t nchars, const uint8_t *finish,
                    fmx_regex **out);

typedef struct fmx_limits {
  /* mode = FMX_MATCH_FRONTIER.  The frontier kernel steps every element of every regex's frontier (in the order
   * that suits the device: elements are independent), so its results equal the reference's (as a multiset) whenever
   * the reference's limits do not bind:
   * max_steps   = longest match explored (levels); 0 = default 4096.  When the frontier is still alive
   *               there, the call returns FMX_TRUNCATED with every match of length <= max_steps.  (On
   *               the BWT of a real text a frontier always dies -- no match is longer than the text --
   *               but on a synthetic "BWT" that is just a random string, LF has short cycles and x* can
   *               run forever.)
   * max_frontier= capacity of the device work queue in elements, 0 = default (1<<22): what the waves cannot
   *               hold in their own pools is queued in HBM (64 slices of two buffers each, about
   *               max_frontier / 50 entries per buffer); FMX_ERR_OVERFLOW when a buffer fills up.
   * Results and per-regex counts are written by the device itself when `out` / `per_regex_count` are
   * page-locked memory (fmx_host_alloc); pageable buffers are filled by a copy after the search.
   * mode = FMX_MATCH_REFERENCE.  ReTree._matchSA exactly (re2/retree.scala:618-653): the priority queue
   * of Scala 2.10 replayed per regex, loop while queue non-empty && queue.length < max_branching &&
   * (max_iterations == 0 || i < max_iterations), i from 1.  Results come back per regex in the
   * reference's list order (newest first).  ReTree.matchSA's defaults are 1024 / 1000 (:570). */
  uint32_t max_steps;
  uint32_t mode;
  uint64_t max_frontier;
  uint32_t max_branching;
  uint32_t max_iterations;
} fmx_limits;

typedef struct fmx_result {   /* SAResult(sa,len,sp,ep), re2/re2.scala:9-19, + which regex */
  uint32_t regex;
  uint32_t len;
  uint64_t sp;
  uint64_t ep;
} fmx_result;

/* ReTree.matchSA over a batch of compiled regexes (re2/retree.scala:570-653): frontier items
 * (regex, CharNode, len, sp, ep) start at root.firsts x (0, 0, n); each is stepped with
 * getPrevRange; isLast states emit a result, the others push their follows.  In frontier mode results
 * are written sorted by (regex, len, sp, ep); in reference mode per regex in the reference's own
 * list order.  per_regex_count (optional, k entries) = results per
 * regex.  FMX_ERR_OVERFLOW if out (cap entries) or the work queue was too small: *n_out then
 * holds the number of results found so far / needed.  FMX_TRUNCATED: see fmx_limits.max_steps. */
int fmx_regex_match_batch(const fmx_index *idx, fmx_regex *const *res, size_t k, const fmx_limits *lim,
                          fmx_result *out, size_t cap, size_t *n_out, uint32_t *per_regex_count);

/* The same in two stages for serving: make a batch of compiled regexes resident on idx's device
 * once (concatenated tables + the level-0 frontier), then match it any number of times, in either mode
 * (FMX_MATCH_REFERENCE needs a batch of fmx_regex_compile handles only).  One match at a time per batch. */
int fmx_regex_batch_create(const fmx_index *idx, fmx_regex *const *res, size_t k, fmx_regex_batch **out);
int fmx_regex_batch_free(fmx_regex_batch *batch);
/* Sizes of a resident batch: regexes, CharNode states, follow entries, start elements (any pointer may be NULL). */
int fmx_regex_batch_info(const fmx_regex_batch *batch, uint64_t *n_regexes, uint64_t *n_states, uint64_t *n_follows,
                         uint64_t *n_firsts);
int fmx_regex_batch_match(const fmx_index *idx, fmx_regex_batch *batch, const fmx_limits *lim, fmx_result *out,
                          size_t cap, size_t *n_out, uint32_t *per_regex_count);
/* The same with the results left in HBM (frontier mode): d_out = device memory for cap fmx_result records,
 * d_per_regex_count = device memory for k counts or NULL; *n_out = the number of results.  The device-pointer
 * form of the regex path, like fmx_search_batch_dev for literals: what a caller that post-processes on the device
 * (or gathers over RCCL, findex_amd/distributed.py) uses -- nothing crosses the host link but the count.  The call
 * is synchronous (the search needs the host between launch chains). */
int fmx_regex_batch_match_dev(const fmx_index *idx, fmx_regex_batch *batch, const fmx_limits *lim, void *d_out,
                              size_t cap, size_t *n_out, void *d_per_regex_count);

/* One process, several GPUs (the single-process form of SURVEY.md 8e for regexes; one process per GPU uses
 * findex_amd/distributed.py and an RCCL gather instead): the batch is cut into contiguous slices of about equal
 * estimated frontier work (start elements, states and follow entries of each regex), slice r is made resident on
 * idxs[r]'s device (every handle a replica of one index), the slices are matched concurrently from one host thread
 * each with no device-to-device traffic, and the result lists are concatenated: same output as
 * fmx_regex_batch_match on one handle (frontier mode only). */
typedef struct fmx_regex_batch_multi fmx_regex_batch_multi;
int fmx_regex_batch_create_multi(fmx_index *const *idxs, size_t n_idx, fmx_regex *const *res, size_t k,
                                 fmx_regex_batch_multi **out);
int fmx_regex_batch_free_multi(fmx_regex_batch_multi *mb);
int fmx_regex_batch_match_multi(fmx_regex_batch_multi *mb, const fmx_limits *lim, fmx_result *out, size_t cap,
                                size_t *n_out, uint32_t *per_regex_count);

/* Gathers per-device result slices into one host array (the host-side counterpart of the RCCL all-gather): slice r
 * is cnt[r] elements of `elem` bytes at DEVICE pointer d_src[r] on idxs[r]'s device and lands at
 * dst + (cnt[0] + .. + cnt[r-1]) * elem.  The copies of all devices run concurrently and the call returns when all
 * have landed.  For callers that keep their batches on the devices (fmx_*_dev entry points) and search slices on
 * several handles themselves. */
int fmx_gather(fmx_index *const *idxs, size_t n_idx, const void *const *d_src, const size_t *cnt, size_t elem,
               void *dst);

/* ---- the exchange itself: an RCCL all-gather of the ranks' device-resident result slices over xGMI (SURVEY.md 8e:
 * "ncclAllGather of uint64 sp[], ep[] slices after the kernels finish"), for callers below the Python mirror -- a JVM.
 * RCCL is loaded at first use (dlopen); FMX_ERR_UNSUPPORTED when the process has none.
 *   one process per GPU : rank 0 calls fmx_comm_unique_id, ships the 128 bytes to the other ranks by its own
 *                         means, every rank calls fmx_comm_create_rank(its handle, n_ranks, rank, id).
 *   one process, N GPUs : fmx_comm_create_all(one handle per device) -- ncclCommInitAll.
 * fmx_allgather_dev(comm, d_send, d_recv, bytes, producer_streams): every rank contributes `bytes` bytes at d_send[i] and
 * receives all ranks' contributions, in rank order, at d_recv[i] (n_ranks * bytes); the arrays have one entry per LOCAL
 * rank of the communicator (1, or N for fmx_comm_create_all, in the order of `idxs`).
 * fmx_gather_dev(.., root, ..): the same exchange delivered to ONE rank -- every rank sends its slice to `root`, only the
 * root's d_recv entry is written (the others may be NULL): the "final gather of hit intervals" when one rank consumes
 * the answer; with the 8-byte interval form (fmx_search_opts.packed) it moves half of what the all-gather of (sp, ep) did
 * per link and nothing at all into the other ranks.
 * Ordering: the collective runs on the communicator's own streams.  producer_streams[i] (a hipStream_t; one per local
 * rank) is the stream on which the caller enqueued the work that writes d_send[i] -- fmx_search_batch_dev is asynchronous --
 * and that last touched d_recv[i]: the collective waits for everything enqueued there so far (an event, no host
 * synchronisation).  producer_streams == NULL: the caller vouches that d_send and d_recv are idle (it has synchronised).
 * Both calls return when the exchange has completed on every local rank, so anything enqueued afterwards may read d_recv.
 * Slices of different lengths (regex result lists): gather the counts first, then the payload padded to the longest, as
 * findex_amd/distributed.py all_gather_varlen does. */
#define FMX_COMM_ID_BYTES 128
typedef struct fmx_comm fmx_comm;
int fmx_comm_unique_id(void *id /* FMX_COMM_ID_BYTES */);
int fmx_comm_create_rank(const fmx_index *idx, int n_ranks, int rank, const void *id, fmx_comm **out);
int fmx_comm_create_all(fmx_index *const *idxs, size_t n_idx, fmx_comm **out);
int fmx_comm_info(const fmx_comm *comm, int *n_ranks, int *n_local);
int fmx_comm_free(fmx_comm *comm);
int fmx_allgather_dev(fmx_comm *comm, const void *const *d_send, void *const *d_recv, size_t bytes,
                      void *const *producer_streams);
int fmx_gather_dev(fmx_comm *comm, const void *const *d_send, void *const *d_recv, size_t bytes, int root,
                   void *const *producer_streams);

/* ---- statistics (since open or the last reset; device counters are read with a sync).
 * rank_queries counts occ(c,i) evaluations in the REFERENCE's terms: two per backward step (findex.scala:26-27,
 * 32-36), one per occ_batch operand or LF step -- the number the Scala path would execute on the same inputs
 * (early exits included), which is what the CPU oracle counts too.  The kernels do less memory work than that:
 * a search's first step needs no block, narrow intervals share one, a single-row step needs one; what they
 * really asked of memory is in the *_requests fields (64-byte one-hot blocks, or in the bytes layout 128-byte
 * BWT blocks and their 4-byte checkpoints, one request each). */
typedef struct fmx_stats_t {
  uint64_t rank_queries;     /* occ evaluations the executed steps stand for (see above) */
  uint64_t backward_steps;   /* getPrevRange-equivalents executed (2 rank queries each) */
  uint64_t launches;         /* kernels launched by this handle */
  double last_kernel_ms;     /* device time of the last host-pointer call's kernel(s), HIP events (calls whose operands
                              * and results fit in 2 KB are not timed: they leave it as it was) */
  uint64_t index_bytes;      /* device bytes held: rank dictionary + BWT + tables */
  uint64_t n_blocks;         /* rank-dictionary blocks per symbol */
  uint32_t n_symbols;        /* symbols that own a bit-vector */
  uint32_t block_bytes;      /* bytes fetched per rank query: 64 (one-hot block) or 132 (BWT block + checkpoint) */
  double build_ms;           /* device time of the kernels that built the rank dictionary at open */
  uint32_t layout;           /* 0 = one-hot bit-vectors, 1 = BWT bytes + checkpoints */
  uint32_t search_residency; /* workgroups per CU the last fmx_search_batch[_dev] launch was sized for (0: none yet); | 0x100 once
                                the kernel's residency census (fmx_prepare) has confirmed the number (the occupancy query can
                                answer one too many: DESIGN.md 3, "residency") */
  uint64_t search_requests;  /* memory requests for rank-dictionary lines issued by fmx_search_batch[_dev]'s kernel */
  /* the regex frontier kernels (fmx_regex_*match*), for their roofline: */
  uint64_t frontier_requests;   /* memory requests for rank-dictionary lines */
  uint64_t frontier_elements;   /* frontier elements stepped (= their backward steps) */
  uint64_t frontier_queue_reads;   /* elements read from the HBM work queues */
  uint64_t frontier_queue_writes;  /* elements appended to the HBM work queues */
  uint64_t frontier_results;    /* results written */
  uint64_t frontier_records;    /* 32-byte state records loaded (none inside a literal stretch) */
  uint64_t ktab_lookups;        /* 16-byte k-mer table entries fetched (each stands for up to ktab_k backward steps) */
  uint32_t ktab_k;              /* K of the k-mer jump table (0: none, or not built yet) */
  uint32_t jump_chars;          /* characters (backward steps) one row-jump-table entry stands for (0: no such table) */
  double tables_build_ms;       /* host time spent building the k-mer table, the row jump table and the select directory
                                 * (at first use or in fmx_prepare): what a handle's first search / first Psi pays on top
                                 * of build_ms */
  uint64_t jump_lookups;        /* row-jump-table lookups (one memory request each) by fmx_search_batch[_dev]'s kernel: an entry
                                 * stands for jump_chars backward steps when the pattern's next jump_chars characters match
                                 * it, a pair of entries ("jump_pairs") for up to twice as many */
  uint64_t jump_bytes;          /* device bytes of the row jump table (0: the handle has none; 16 n, or 32 n with pairs);
                                 * part of index_bytes */
  uint64_t row_lookups;         /* 8-byte row-table words fetched (one or three backward steps of a one-row interval each):
                                 * by the one-row part of fmx_search_batch[_dev] and by the regex frontier */
  uint64_t row_bytes;           /* device bytes of the row table and the three-step row table (0: none); part of index_bytes */
  uint64_t peak_table_build_bytes;   /* most device memory one derived table's build held at once: since round 4 the table
                                      * itself (the row jump table was built through a second buffer of its size before) */
  uint64_t patterns_seen;       /* patterns this handle's literal searches have been asked for (what "tables_after" counts) */
  uint64_t tables_held_bytes;   /* device bytes of all derived tables of this handle now (what "table_budget" counts) */
  uint64_t table_budget_bytes;  /* the handle's budget in bytes (a fraction resolved against the HBM free now); ~0: none */
  uint64_t hbm_free_after_tables;   /* free device memory right after the handle's last table build (hipMemGetInfo; 0: none built) */
  double tables_alloc_ms;       /* of tables_build_ms: the time spent inside hipMalloc for the tables -- ~0 on memory nobody has held
                                 * since the box came up, 45-60 ms per GiB on memory a process released shortly before (the driver
                                 * wipes it first: profiles/r05_alloc.md); the rest of tables_build_ms is the build kernels */
} fmx_stats_t;
int fmx_stats(const fmx_index *idx, fmx_stats_t *out);
/* fmx_stats_t.last_kernel_ms alone, without the device synchronisation and counter read-back of fmx_stats. */
int fmx_last_kernel_ms(const fmx_index *idx, double *ms);
int fmx_stats_reset(fmx_index *idx);

#ifdef __cplusplus
}
#endif
#endif /* FMX_H */

// fmx_device.h -- device-side rank dictionary layouts and the rank primitive (gfx950).
//
// Layout kLayoutOneHot (DESIGN.md "Rank dictionary"):
//   For every symbol c that occurs in the BWT there is one bit-vector of n one-hot bits
//   (bit p set <=> BWT'[p] == c), cut into 64-byte blocks:
//       bytes 0..7    uint64  number of set bits in all earlier blocks of this vector
//       bytes 8..63   448 payload bits, position p of the block at dword 2 + p/32, bit p%32
//   Block b of symbol-slot s lives at bv + (s * nblocks + b) * 64.  nblocks = n/448 + 1, so
//   the query position x == n has a block too (its header is the symbol's total count).
//   One rank query = ONE aligned 64-byte request: header + in-register popcount.
//   Symbol 0 (the EOF row, BWT' only) needs no vector: rank0(x) = (x > eof).
//
// A query is served by 4 adjacent lanes (a "quad"): lane t loads bytes 16t..16t+15 of the block
// with one global_load_dwordx4, so every wave-level load instruction fetches 16 whole blocks and
// each block is requested exactly once.  Why 64 bytes and a quad: the memory system's limit for
// this access pattern is distinct requests per second, whatever the granule up to 128 B
// (tools/ubench/gather.hip: ~46-48 G independent requests/s; tools/ubench/chain.hip: dependent
// chains reach ~53 G requests/s with 64-byte granules at 16 chains per wave against ~46 G/s with
// 128-byte granules at 8 per wave), and a second load instruction to the same line is a second
// request.  Against the 128-byte / 8-lane form this layout replaced (round 1, profiles/) the half
// block costs nothing in request rate, halves the vector instructions per query and doubles the
// queries a wave keeps in flight.
//
// The popcount side is written for instruction count: the first search kernels were bound by
// vector-instruction issue, not by HBM (profiles/).  Per dword: saturating
// subtract, bit-field extract, compare, select, popcount-accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmx {

constexpr uint32_t kBlockBits = 448;     // payload positions per one-hot block
constexpr uint32_t kBlockBytes = 64;
constexpr uint32_t kBlockPayloadDwords = kBlockBits / 32;   // 14
constexpr uint16_t kSlotNone = 0xFFFF;   // symbol absent from the BWT: rank is 0
constexpr uint16_t kSlotEof = 0xFFFE;    // symbol 0: rank0(x) = (x > eof)
constexpr uint64_t kOneHotMaxN = 1ull << 37;   // split448 is exact for positions below this

// Second layout, for indexes whose one-hot vectors (sigma * n / 7 bytes) do not fit in HBM
// (BASELINE config C5: n = 2^34, sigma = 128 -> 314 GB): the BWT bytes themselves in 128-byte
// blocks (slot eof holds 0, which no real symbol has) + per-block checkpoints
//     chk[blk][slot]  uint32  occurrences of the symbol before this block -- absolute when every symbol occurs
//                             fewer than 2^32 times (C5: n = 2^34, sigma = 128 -> ~2^27 each), else since the
//                             start of the block's superblock, with
//     sup[sb][slot]   uint64  occurrences before superblock sb (2^15 blocks = 2^22 positions)
// rank = [sup +] chk + #{bytes of the block below the boundary that equal c}: two HBM lines per rank
// query (the block and the checkpoint), 132 B algorithmic as SURVEY 8d; the superblock form adds a third
// (cache-resident) request.
// A query is served by 8 lanes (an "octet"): 16 of the block's 128 bytes per lane.
constexpr uint32_t kLayoutOneHot = 0;
constexpr uint32_t kLayoutBytes = 1;
constexpr uint32_t kByteBlock = 128;     // BWT positions per byte-layout block
constexpr uint32_t kSuperShift = 15;     // blocks per superblock = 2^15

// Lanes per query of a layout.
template <uint32_t LAYOUT> struct Lay { static constexpr int G = LAYOUT == kLayoutBytes ? 8 : 4; };

struct DevIndex {
  const uint4 *bv;        // one-hot layout: rank dictionary
  const uint8_t *bwt;     // BWT bytes, slot eof holds 0, zero-padded to whole 128-byte blocks
  const uint64_t *cf;     // [256] C[] = first row of each symbol, NaiveFMSearcher.cf
  const uint16_t *slot;   // [256] symbol -> slot | kSlotNone | kSlotEof
  uint64_t n;
  uint64_t eof;
  uint64_t nblocks;       // one-hot: blocks per vector; bytes: number of 128-position blocks
  const uint32_t *chk;    // bytes layout
  const uint64_t *sup;    // bytes layout
  uint32_t layout;
  uint32_t nslots;
};

constexpr unsigned long long kPackWide = 0xFFFFFFull;      // width field of a packed interval that stands for "look in the escape list"

// Where pattern q of a batch lies in the pattern buffer: pat[b, e) -- from the caller's k + 1 offsets, or, for a batch of
// equal-length patterns handed over WITHOUT offsets (fmx_search_opts.fixed_len), q * fixed.  Branch-free on purpose: the
// two offsets are loaded in both forms (an implicit batch reads the same 16 readable bytes every time: `base` is then the
// handle's C[] array) and a select picks -- a branch here would put a wait for the loads right behind them, and the
// search kernel requests a batch's offsets two batches before it needs them.
struct PatOff {
  const uint64_t *base;
  uint64_t fixed;          // 0: base[q], base[q + 1]; else pattern q = [q * fixed, (q + 1) * fixed)
  __device__ __forceinline__ const uint64_t *at(uint64_t q) const { return base + (fixed ? 0ull : q); }
  __device__ __forceinline__ void get(uint64_t q, uint64_t &b, uint64_t &e) const {
    const uint64_t *p = at(q);
    const uint64_t v0 = p[0], v1 = p[1];
    b = fixed ? q * fixed : v0;
    e = fixed ? (q + 1) * fixed : v1;
  }
};

// ---- DPP helpers: reductions inside a group of 4 or 8 lanes stay in the VALU (no LDS crossbar).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <-> 7-i inside each 8

template <int G>
__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
  v += dpp<kDppXor1>(v);
  v += dpp<kDppXor2>(v);
  if (G == 8) v += dpp<kDppHalfMirror>(v);
  return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
  v |= dpp<kDppXor1>(v);
  v |= dpp<kDppXor2>(v);
  if (G == 8) v |= dpp<kDppHalfMirror>(v);
  return v;
}

// Value of lane U of the group, to every lane of the group (U < 4).
template <int G, int U>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v) {
  if (G == 4) return dpp<U * 0x55>(v);                  // quad_perm [U,U,U,U]
  return (uint32_t)__shfl((int)v, (int)((__lane_id() & ~7u) | U), 64);
}
template <int G, int U>
__device__ __forceinline__ uint64_t group_bcast64(uint64_t v) {
  return ((uint64_t)group_bcast<G, U>((uint32_t)(v >> 32)) << 32) | group_bcast<G, U>((uint32_t)v);
}

// popcount(x) + acc in one instruction (the compiler otherwise splits it into bcnt + add3)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t d;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
  return d;
}

// Per-lane constants of a group of G lanes.
struct LaneConst {
  uint32_t k[4];      // one-hot: first payload position of dword j of this lane (0xFFFFFFFF: header dword)
  uint32_t t;         // lane index inside the group
};

template <int G>
__device__ __forceinline__ LaneConst lane_const() {
  LaneConst lc;
  lc.t = threadIdx.x & (G - 1);
#pragma unroll
  for (int j = 0; j < 4; j++) lc.k[j] = (lc.t == 0 && j < 2) ? 0xFFFFFFFFu : (128u * lc.t - 64u + 32u * j);
  return lc;
}

// x / 448 and x % 448 for x < 2^37: one multiply (mul_hi) and shifts; 448 = 64 * 7, and the
// multiply-shift division by 7 is exact for operands below 2^31.
__device__ __forceinline__ void split448(uint64_t x, uint32_t &blk, uint32_t &rem) {
  const uint32_t y = (uint32_t)(x >> 6);
  blk = __umulhi(y, 0x92492493u) >> 2;                 // y / 7
  rem = (uint32_t)x - (blk << 9) + (blk << 6);         // x - 448 * blk (mod 2^32, exact: rem < 448)
}

// The in-register half of a one-hot rank query: given this lane's 16 bytes `w` of the block and
// the in-block boundary `rem`, returns (to every lane of the quad) header + #{set payload bits
// below rem}.  WIDE = false when every count fits 32 bits (n <= 2^32): the header then rides in
// the same 2-step DPP sum; WIDE = true carries the 64-bit header apart.
template <bool WIDE>
__device__ __forceinline__ uint64_t rank_finish(uint4 w, uint32_t rem, const LaneConst &lc) {
  const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
  const uint32_t hlo = lc.t == 0 ? w.x : 0u;
  uint32_t cnt = WIDE ? 0u : hlo;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    // bits of dword j below the boundary: nb = max(rem - first position of the dword, 0)
    const uint32_t nb = __builtin_elementwise_sub_sat(rem, lc.k[j]);
    uint32_t x = __builtin_amdgcn_ubfe(ww[j], 0u, nb);  // low nb bits (width taken mod 32)
    x = nb > 31u ? ww[j] : x;                          // whole dword below the boundary
    cnt = bcnt_acc(x, cnt);
  }
  if (!WIDE) return group_sum<4>(cnt);
  const uint32_t hhi = group_or<4>(lc.t == 0 ? w.y : 0u);
  return (((uint64_t)hhi << 32) | group_or<4>(hlo)) + group_sum<4>(cnt);
}

// Bit `rem` of the block's payload, to every lane of the quad (BWT'[x] == c for the block's symbol).
__device__ __forceinline__ uint32_t payload_bit(uint4 w, uint32_t rem, const LaneConst &lc) {
  const uint32_t d = (rem >> 5) + 2;                     // dword of the block that holds the bit
  const uint32_t comp = d & 3u;
  const uint32_t word = comp < 2u ? (comp == 0u ? w.x : w.y) : (comp == 2u ? w.z : w.w);
  uint32_t bit = __builtin_amdgcn_ubfe(word, rem, 1u);   // offset taken mod 32
  bit = (d >> 2) == lc.t ? bit : 0u;
  return group_or<4>(bit);
}

// ---- the same block served by a PAIR of lanes (round 5, k_search4<.., G2>): lane u of a pair (u = lane & 1) loads bytes
// 16u .. 16u+15 with one instruction and bytes 32+16u .. 32+16u+15 with a second, so a wave carries 32 patterns instead of 16
// -- twice the dependent chains per wave for the same registers per lane, at the price of a second request for the block's
// upper half.  Why: tools/c3_halfbatch.py -- a launch's time is its waves' round trips, not its requests.
constexpr int kDppPairLane0 = 0xA0;    // quad_perm [0,0,2,2]: lane 0 of each pair to both
constexpr int kDppPairLane1 = 0xF5;    // quad_perm [1,1,3,3]
__device__ __forceinline__ uint32_t pair_sum(uint32_t v) { return v + dpp<kDppXor1>(v); }
__device__ __forceinline__ uint32_t pair_or(uint32_t v) { return v | dpp<kDppXor1>(v); }
template <int U>
__device__ __forceinline__ uint32_t pair_bcast(uint32_t v) { return dpp<U ? kDppPairLane1 : kDppPairLane0>(v); }
struct Blk2 { uint4 a, b; };           // a: dwords 4u .. 4u+3 of the block, b: dwords 8+4u .. 8+4u+3
// header + #{set payload bits below rem}, to both lanes of the pair (rank_finish's counterpart).  u = lane & 1.
template <bool WIDE>
__device__ __forceinline__ uint64_t rank_finish_g2(const Blk2 &w, uint32_t rem, uint32_t u) {
  const uint32_t wa[4] = {w.a.x, w.a.y, w.a.z, w.a.w}, wb[4] = {w.b.x, w.b.y, w.b.z, w.b.w};
  const uint32_t hlo = u == 0 ? w.a.x : 0u;
  uint32_t cnt = WIDE ? 0u : hlo;
  // payload position of dword d of the block = 32 (d - 2); a.j is dword 4u + j (the header where u = 0, j < 2), b.j dword 8 + 4u + j
  const uint32_t base_a = 128u * u - 64u, base_b = 128u * u + 192u;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t nb = __builtin_elementwise_sub_sat(rem, base_a + 32u * j);
    if (j < 2) nb = u == 0 ? 0u : nb;                     // lane 0's first two dwords are the count, not payload
    uint32_t x = __builtin_amdgcn_ubfe(wa[j], 0u, nb);
    x = nb > 31u ? wa[j] : x;
    cnt = bcnt_acc(x, cnt);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t nb = __builtin_elementwise_sub_sat(rem, base_b + 32u * j);
    uint32_t x = __builtin_amdgcn_ubfe(wb[j], 0u, nb);
    x = nb > 31u ? wb[j] : x;
    cnt = bcnt_acc(x, cnt);
  }
  if (!WIDE) return pair_sum(cnt);
  const uint32_t hhi = pair_or(u == 0 ? w.a.y : 0u);
  return (((uint64_t)hhi << 32) | pair_or(hlo)) + pair_sum(cnt);
}
// bit `rem` of the block's payload, to both lanes of the pair
__device__ __forceinline__ uint32_t payload_bit_g2(const Blk2 &w, uint32_t rem, uint32_t u) {
  const uint32_t d = (rem >> 5) + 2;                     // dword of the block that holds the bit: 2 .. 15
  const bool in_b = d >= 8;
  const uint32_t owner = (d >> 2) & 1u;                  // dwords 0-3, 8-11: lane 0; 4-7, 12-15: lane 1
  const uint4 h = in_b ? w.b : w.a;
  const uint32_t comp = d & 3u;
  const uint32_t word = comp < 2u ? (comp == 0u ? h.x : h.y) : (comp == 2u ? h.z : h.w);
  uint32_t bit = __builtin_amdgcn_ubfe(word, rem, 1u);   // offset taken mod 32
  bit = owner == u ? bit : 0u;
  return pair_or(bit);
}

// Loads in the global (not flat) address space from an integer address.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_line16(uint64_t addr) {
  const u32x4 v = *(const u32x4 __attribute__((address_space(1))) *)addr;
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Byte address of this lane's 16 bytes of block `blk` of slot `s`.
__device__ __forceinline__ uint64_t block_addr(const DevIndex &ix, uint32_t s, uint32_t blk, const LaneConst &lc) {
  return (uint64_t)(uintptr_t)ix.bv + ((uint64_t)s * ix.nblocks + blk) * kBlockBytes + lc.t * 16u;
}

// ---- bytes layout: occurrences of byte c among the first `nbytes` (0..16, may exceed) of this
// lane's 16 block bytes.  Per dword: exact SWAR zero-byte detect on w ^ cccc, keep the wanted bytes.
__device__ __forceinline__ uint32_t match_count16(uint4 w, uint32_t c, uint32_t nbytes, uint32_t acc) {
  const uint32_t pat = c * 0x01010101u;
  const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t z = ww[j] ^ pat;
    const uint32_t eq = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);   // 0x80 per equal byte
    const uint32_t kb = __builtin_elementwise_sub_sat(nbytes, 4u * j);            // bytes wanted of this dword
    uint32_t m = __builtin_amdgcn_ubfe(eq, 0u, 8u * kb);
    m = kb > 3u ? eq : m;
    acc = bcnt_acc(m, acc);
  }
  return acc;
}

struct ByteRankReq {      // the two lines of a bytes-layout rank query, requested before use
  uint4 w;
  uint32_t chk;
  uint64_t sup;
  uint32_t rem;
};

__device__ __forceinline__ ByteRankReq byte_rank_issue(const DevIndex &ix, uint16_t slot, uint64_t x, const LaneConst &lc) {
  ByteRankReq q;
  const uint64_t blk = x >> 7;
  q.rem = (uint32_t)x & 127u;
  q.w = load_line16((uint64_t)(uintptr_t)ix.bwt + blk * kByteBlock + lc.t * 16u);
  q.chk = lc.t == 0 ? ix.chk[blk * ix.nslots + slot] : 0u;
  // When every symbol occurs fewer than 2^32 times the checkpoints are absolute counts and there are no
  // superblocks (ix.sup == nullptr): two requests per rank query -- the block and its checkpoint -- not three.
  q.sup = ix.sup ? ix.sup[(blk >> kSuperShift) * ix.nslots + slot] : 0ull;
  return q;
}

__device__ __forceinline__ uint64_t byte_rank_finish(const ByteRankReq &q, uint32_t c, const LaneConst &lc) {
  const uint32_t nbytes = __builtin_elementwise_sub_sat(q.rem, 16u * lc.t);
  return q.sup + group_sum<8>(match_count16(q.w, c, nbytes, q.chk));
}

// BWT'[x] == c from the block line of a bytes-layout request, to every lane of the octet.
__device__ __forceinline__ uint32_t byte_match_bit(const ByteRankReq &q, uint32_t c, const LaneConst &lc) {
  const uint32_t bidx = q.rem & 15u;                     // byte of this lane that holds row x
  const uint32_t comp = bidx >> 2;
  // the dword that holds the byte, by 64-bit shifts (a chain of selects here is turned into an indexed load from
  // scratch memory by the compiler)
  const uint32_t sh = 32u * (comp & 1u);
  const uint32_t lo = (uint32_t)((((uint64_t)q.w.y << 32) | q.w.x) >> sh);
  const uint32_t hi = (uint32_t)((((uint64_t)q.w.w << 32) | q.w.z) >> sh);
  const uint32_t word = (uint32_t)((((uint64_t)hi << 32) | lo) >> (16u * (comp & 2u)));
  const uint32_t byte = __builtin_amdgcn_ubfe(word, 8u * (bidx & 3u), 8u);
  return group_or<8>(((q.rem >> 4) == lc.t && byte == c) ? 1u : 0u);
}

// ---- either layout.  A rank query in two halves, so that a kernel can have several independent
// queries in flight per lane group: rank_issue requests the line(s), rank_complete consumes them.
// rank_excl(c, x) = #{p < x : BWT'[p] == c}, 0 <= x <= n; occ(c, i) of the reference is
// rank_excl(c, i + 1).
struct RankReq {
  uint4 w;            // one-hot: the block; bytes: the BWT block line
  uint32_t rem;
  uint32_t chk;       // bytes layout
  uint64_t sup;       // bytes layout; immediate value when kind == 0
  uint32_t kind;      // 0 immediate, 1 from memory
};

template <uint32_t LAYOUT>
__device__ __forceinline__ RankReq rank_issue(const DevIndex &ix, uint16_t slot, uint64_t x, const LaneConst &lc) {
  RankReq q;
  q.w = make_uint4(0, 0, 0, 0);
  q.rem = 0; q.chk = 0; q.sup = 0; q.kind = 0;
  if (slot == kSlotNone) return q;
  if (slot == kSlotEof) { q.sup = x > ix.eof ? 1 : 0; return q; }
  q.kind = 1;
  if (LAYOUT == kLayoutBytes) {
    const ByteRankReq b = byte_rank_issue(ix, slot, x, lc);
    q.w = b.w; q.rem = b.rem; q.chk = b.chk; q.sup = b.sup;
    return q;
  }
  uint32_t blk;
  split448(x, blk, q.rem);
  q.w = load_line16(block_addr(ix, slot, blk, lc));
  return q;
}

template <bool WIDE, uint32_t LAYOUT>
__device__ __forceinline__ uint64_t rank_complete(const RankReq &q, uint32_t c, const LaneConst &lc) {
  if (q.kind == 0) return q.sup;
  if (LAYOUT == kLayoutBytes) {
    ByteRankReq b;
    b.w = q.w; b.rem = q.rem; b.chk = q.chk; b.sup = q.sup;
    return byte_rank_finish(b, c, lc);
  }
  return rank_finish<WIDE>(q.w, q.rem, lc);
}

template <bool WIDE, uint32_t LAYOUT>
__device__ __forceinline__ uint64_t rank_excl(const DevIndex &ix, uint32_t c, uint16_t slot, uint64_t x,
                                              const LaneConst &lc) {
  return rank_complete<WIDE, LAYOUT>(rank_issue<LAYOUT>(ix, slot, x, lc), c, lc);
}

// One backward step for the whole lane group: (sp, ep) -> (C[c]+rank(c,sp), C[c]+rank(c,ep)), the
// body of SuffixAlgo.getPrevRange (findex.scala:32-36).  All lines are requested before any is consumed;
// when sp and ep fall into the same block (narrow intervals: most steps of a search or a regex frontier)
// the block is requested once.  Returns the number of memory requests for rank-dictionary lines it made.
template <bool WIDE, uint32_t LAYOUT>
__device__ __forceinline__ uint32_t backward_step(const DevIndex &ix, uint32_t c, uint16_t slot, uint64_t cfc,
                                                  const LaneConst &lc, uint64_t &sp, uint64_t &ep) {
  if (slot >= kSlotEof) {                       // absent symbol, or the EOF symbol 0
    const uint64_t r1 = (slot == kSlotEof && sp > ix.eof) ? 1 : 0;
    const uint64_t r2 = (slot == kSlotEof && ep > ix.eof) ? 1 : 0;
    sp = cfc + r1;
    ep = cfc + r2;
    return 0;
  }
  if (LAYOUT == kLayoutBytes) {
    const ByteRankReq q1 = byte_rank_issue(ix, slot, sp, lc);
    ByteRankReq q2 = q1;
    const bool two = (ep >> 7) != (sp >> 7);
    if (two) q2 = byte_rank_issue(ix, slot, ep, lc);
    else q2.rem = (uint32_t)ep & 127u;
    sp = cfc + byte_rank_finish(q1, c, lc);
    ep = cfc + byte_rank_finish(q2, c, lc);
    return two ? 4u : 2u;                        // block line + checkpoint line per distinct block
  }
  uint32_t b1, b2, m1, m2;
  split448(sp, b1, m1);
  split448(ep, b2, m2);
  const uint4 w1 = load_line16(block_addr(ix, slot, b1, lc));
  uint4 w2 = w1;
  if (b2 != b1) w2 = load_line16(block_addr(ix, slot, b2, lc));
  sp = cfc + rank_finish<WIDE>(w1, m1, lc);
  ep = cfc + rank_finish<WIDE>(w2, m2, lc);
  return b2 != b1 ? 2u : 1u;
}

// The same step for an interval of exactly one row, [sp, sp + 1): it maps to [C[c] + rank(c, sp), + [BWT'[sp] == c]),
// and BWT'[sp] == c is bit sp of c's own vector (a byte compare in the bytes layout), i.e. part of the block already
// fetched: one request and half the popcount work.  Same result as backward_step.
template <bool WIDE, uint32_t LAYOUT>
__device__ __forceinline__ uint32_t single_row_step(const DevIndex &ix, uint32_t c, uint16_t slot, uint64_t cfc,
                                                    const LaneConst &lc, uint64_t &sp, uint64_t &ep) {
  if (slot >= kSlotEof) {
    const uint64_t r1 = (slot == kSlotEof && sp > ix.eof) ? 1 : 0;
    const uint64_t r2 = (slot == kSlotEof && ep > ix.eof) ? 1 : 0;
    sp = cfc + r1;
    ep = cfc + r2;
    return 0;
  }
  if (LAYOUT == kLayoutBytes) {
    const ByteRankReq q1 = byte_rank_issue(ix, slot, sp, lc);
    sp = cfc + byte_rank_finish(q1, c, lc);
    ep = sp + byte_match_bit(q1, c, lc);
    return 2u;
  }
  uint32_t b1, m1;
  split448(sp, b1, m1);
  const uint4 w1 = load_line16(block_addr(ix, slot, b1, lc));
  sp = cfc + rank_finish<WIDE>(w1, m1, lc);
  ep = sp + payload_bit(w1, m1, lc);
  return 1u;
}

// ---- statistics counters without a hot spot.  Same-address device atomics complete at roughly 100
// per microsecond on MI355X; a kernel whose 8192 waves all end together and each add to one shared
// counter spends its last ~100 us per counter draining them (measured: round 1).  So the
// counters are kCounterSlots 128-byte slots, a workgroup adds to slot blockIdx.x % kCounterSlots
// (one atomic per wave and counter after a wave-level reduction), and the host sums the slots.
constexpr uint32_t kCounterSlots = 2048;
constexpr uint32_t kCounterStride = 16;        // uint64 per slot: [0] rank queries, [1] backward steps, [2] search requests,
                                               // frontier kernels: [3] rank-line requests, [4] queue appends, [5] results,
                                               // [6] elements stepped, [7] queue entries read, [8] state records loaded; [9] k-mer table lookups; [10] row jump table lookups (16 B); [11] row table lookups (8 B)
constexpr size_t kCounterBytes = (size_t)kCounterSlots * kCounterStride * 8;
// Behind the counters, in the same buffer: the RESIDENCY CENSUS of the search kernel -- begin and end (100 MHz clock) of
// each workgroup of the last k_search4 launch.  The occupancy query answers one workgroup per CU too many for kernels
// with 81..96 scalar registers (MI355X_MICROARCH.md, "Residency"): the surplus workgroups start when the first ones
// end, a second generation that owns an eighth of the batches.  The host counts how many workgroups began before
// the first one ended and sizes later grids by that (fmx_search.hip, Residency).
constexpr uint32_t kCensusBlocks = 4096;
constexpr size_t kCensusBytes = (size_t)kCensusBlocks * 16 + 16;      // + one word behind the entries: the grid of the launch that wrote them
// ... and behind that the TICKET areas of the literal search kernel (fmx_search.hip, "the last rounds are drawn"): kTixShards
// counters per area, each on a 128-byte line of its own (all waves of a launch drawing from ONE counter are ~12 000 returning
// atomics on one address, 8 ns each: 0.1 ms, measured); area a belongs to one stream at a time (host bookkeeping)
constexpr uint32_t kTixAreas = 16;
constexpr uint32_t kTixShards = 64;
constexpr size_t kTixStride = 16;                                      // words between two counters
constexpr size_t kTixBytes = (size_t)kTixAreas * kTixShards * kTixStride * 8;
constexpr size_t kCalibScratchBytes = 64;                             // behind the census: the calibration launch's empty pattern (two zero offsets) and its output words

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Called by EVERY lane of the workgroup (convergent), each with its private partial sums.
__device__ __forceinline__ void counters_add(unsigned long long *__restrict__ counters, unsigned long long ranks,
                                             unsigned long long steps, unsigned long long reqs) {
  ranks = wave_sum(ranks);
  steps = wave_sum(steps);
  reqs = wave_sum(reqs);
  if ((threadIdx.x & 63u) == 0) {
    unsigned long long *slot = counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride;
    if (ranks) atomicAdd(slot + 0, ranks);
    if (steps) atomicAdd(slot + 1, steps);
    if (reqs) atomicAdd(slot + 2, reqs);
  }
}

// The frontier kernels' own counters (slots 3..7), same calling rule.
__device__ __forceinline__ void counters_add_frontier(unsigned long long *__restrict__ counters, unsigned long long reqs,
                                                      unsigned long long pushes, unsigned long long results,
                                                      unsigned long long elems, unsigned long long reads,
                                                      unsigned long long recs) {
  reqs = wave_sum(reqs);
  recs = wave_sum(recs);
  pushes = wave_sum(pushes);
  results = wave_sum(results);
  elems = wave_sum(elems);
  reads = wave_sum(reads);
  if ((threadIdx.x & 63u) == 0) {
    unsigned long long *slot = counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride;
    if (reqs) atomicAdd(slot + 3, reqs);
    if (pushes) atomicAdd(slot + 4, pushes);
    if (results) atomicAdd(slot + 5, results);
    if (elems) atomicAdd(slot + 6, elems);
    if (reads) atomicAdd(slot + 7, reads);
    if (recs) atomicAdd(slot + 8, recs);
  }
}

// Launch-side dispatch over the three kernel instantiations an index can need: the bytes layout
// (counts are 64-bit sums there), and the one-hot layout with 32-bit or 64-bit counts.
#define FMX_LAYOUT_DISPATCH(h, CALL)                                                        \
  do {                                                                                      \
    if ((h)->layout == kLayoutBytes) { CALL(true, kLayoutBytes); }                          \
    else if ((h)->n > (1ull << 32)) { CALL(true, kLayoutOneHot); }                          \
    else { CALL(false, kLayoutOneHot); }                                                    \
  } while (0)

}  // namespace fmx

// fmx_device.h -- device-side rank dictionary layout and the rank primitive (gfx950).
//
// Layout in HBM (DESIGN.md "Rank dictionary"):
//   For every symbol c that occurs in the BWT there is one bit-vector of n one-hot bits
//   (bit p set <=> BWT'[p] == c), cut into 128-byte blocks:
//       bytes 0..7    uint64  number of set bits in all earlier blocks of this vector
//       bytes 8..127  960 payload bits, position p of the block at dword 2 + p/32, bit p%32
//   Block b of symbol-slot s lives at bv + (s * nblocks + b) * 128.  nblocks = n/960 + 1, so
//   the query position x == n has a block too (its header is the symbol's total count).
//   One rank query = ONE aligned 128-byte line: header + in-register popcount.
//   Symbol 0 (the EOF row, BWT' only) needs no vector: rank0(x) = (x > eof).
//
// A query is served by 8 adjacent lanes ("octet"): lane t loads bytes 16t..16t+15 of the line
// with one global_load_dwordx4, so every wave-level load instruction fetches 8 whole lines and
// each line is requested exactly once.  (Measured on MI355X: the memory system delivers about
// 46-48 G distinct-line requests/s whatever the granule size up to 128 B, and a second load
// instruction to the same line is a second request -- tools/ubench/gather.hip -- so wider
// per-lane loads or fewer lanes per query do not pay.)
//
// The popcount side is written for instruction count: the search kernels turned out to be bound
// by vector-instruction issue, not by HBM (profiles/).  Per dword: saturating
// subtract, bit-field extract, compare, select, popcount-accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmx {

constexpr uint32_t kBlockBits = 960;     // payload positions per block
constexpr uint32_t kBlockBytes = 128;
constexpr uint32_t kOctet = 8;           // lanes per query
constexpr uint16_t kSlotNone = 0xFFFF;   // symbol absent from the BWT: rank is 0
constexpr uint16_t kSlotEof = 0xFFFE;    // symbol 0: rank0(x) = (x > eof)

// Second layout, for indexes whose one-hot vectors (sigma * n / 8 bytes) do not fit in HBM
// (BASELINE config C5: n = 2^34, sigma = 128 -> 289 GB): the BWT bytes themselves in 128-byte
// blocks (slot eof holds 0, which no real symbol has) + per-block checkpoints
//     chk[blk][slot]  uint32  occurrences of the symbol in its superblock before this block
//     sup[sb][slot]   uint64  occurrences before superblock sb (2^15 blocks = 2^22 positions)
// rank = sup + chk + #{bytes of the block below the boundary that equal c}: two HBM lines per rank
// query (the block and the checkpoint; sup stays cache-resident), 132 B algorithmic as SURVEY 8d.
constexpr uint32_t kLayoutOneHot = 0;
constexpr uint32_t kLayoutBytes = 1;
constexpr uint32_t kByteBlock = 128;     // BWT positions per byte-layout block
constexpr uint32_t kSuperShift = 15;     // blocks per superblock = 2^15

struct DevIndex {
  const uint4 *bv;        // one-hot layout: rank dictionary
  const uint8_t *bwt;     // BWT bytes (one-hot: raw, slot eof holds a filler; bytes: slot eof holds 0,
                          // zero-padded to whole blocks)
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

// ---- DPP helpers: reductions inside an octet stay in the VALU (no LDS crossbar).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <-> 7-i inside each 8

__device__ __forceinline__ uint32_t octet_sum(uint32_t v) {
  v += dpp<kDppXor1>(v);
  v += dpp<kDppXor2>(v);
  v += dpp<kDppHalfMirror>(v);
  return v;
}
__device__ __forceinline__ uint32_t octet_or(uint32_t v) {
  v |= dpp<kDppXor1>(v);
  v |= dpp<kDppXor2>(v);
  v |= dpp<kDppHalfMirror>(v);
  return v;
}

// popcount(x) + acc in one instruction (the compiler otherwise splits it into bcnt + add3)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t d;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
  return d;
}

// Per-lane constants of the octet layout.
struct LaneConst {
  uint32_t k[4];      // first payload position of dword j of this lane (0xFFFFFFFF: header dword)
  uint32_t t;         // lane index inside the octet
};

__device__ __forceinline__ LaneConst lane_const() {
  LaneConst lc;
  lc.t = threadIdx.x & (kOctet - 1);
#pragma unroll
  for (int j = 0; j < 4; j++) lc.k[j] = (lc.t == 0 && j < 2) ? 0xFFFFFFFFu : (128u * lc.t - 64u + 32u * j);
  return lc;
}

// x / 960 and x % 960 for x < 2^38: one multiply (mul_hi) and shifts; 960 = 64 * 15.
__device__ __forceinline__ void split960(uint64_t x, uint32_t &blk, uint32_t &rem) {
  const uint32_t y = (uint32_t)(x >> 6);
  blk = __umulhi(y, 0x88888889u) >> 3;                 // y / 15
  rem = (uint32_t)x - (blk << 10) + (blk << 6);        // x - 960 * blk (mod 2^32, exact: rem < 960)
}

// The in-register half of a rank query: given this lane's 16 bytes `w` of the block and the
// in-block boundary `rem`, returns (to every lane of the octet) header + #{set payload bits below
// rem}.  WIDE = false when every count fits 32 bits (n <= 2^32): the header then rides in the
// same 3-step DPP sum; WIDE = true carries the 64-bit header apart.
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
  if (!WIDE) return octet_sum(cnt);
  const uint32_t hhi = octet_or(lc.t == 0 ? w.y : 0u);
  return (((uint64_t)hhi << 32) | octet_or(hlo)) + octet_sum(cnt);
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
  q.sup = ix.sup[(blk >> kSuperShift) * ix.nslots + slot];
  return q;
}

__device__ __forceinline__ uint64_t byte_rank_finish(const ByteRankReq &q, uint32_t c, const LaneConst &lc) {
  const uint32_t nbytes = __builtin_elementwise_sub_sat(q.rem, 16u * lc.t);
  return q.sup + octet_sum(match_count16(q.w, c, nbytes, q.chk));
}

// rank_excl(c, x) = #{p < x : BWT'[p] == c}, 0 <= x <= n, evaluated by the whole octet, for
// either layout.  occ(c, i) of the reference is rank_excl(c, i + 1).
template <bool WIDE>
__device__ __forceinline__ uint64_t rank_excl(const DevIndex &ix, uint32_t c, uint16_t slot, uint64_t x,
                                              const LaneConst &lc) {
  if (slot == kSlotNone) return 0;
  if (slot == kSlotEof) return x > ix.eof ? 1 : 0;
  if (ix.layout == kLayoutBytes) return byte_rank_finish(byte_rank_issue(ix, slot, x, lc), c, lc);
  uint32_t blk, rem;
  split960(x, blk, rem);
  return rank_finish<WIDE>(load_line16(block_addr(ix, slot, blk, lc)), rem, lc);
}

// The same in two halves, so that a kernel can have several independent rank queries in flight per
// octet: rank_issue requests the line(s), rank_complete consumes them.
struct RankReq {
  uint4 w;            // one-hot: the block line; bytes: the BWT block line
  uint32_t rem;
  uint32_t chk;       // bytes layout
  uint64_t sup;       // bytes layout; immediate value when kind == 0
  uint32_t kind;      // 0 immediate, 1 one-hot, 2 bytes
};

__device__ __forceinline__ RankReq rank_issue(const DevIndex &ix, uint16_t slot, uint64_t x, const LaneConst &lc) {
  RankReq q;
  q.w = make_uint4(0, 0, 0, 0);
  q.rem = 0; q.chk = 0; q.sup = 0; q.kind = 0;
  if (slot == kSlotNone) return q;
  if (slot == kSlotEof) { q.sup = x > ix.eof ? 1 : 0; return q; }
  if (ix.layout == kLayoutBytes) {
    const ByteRankReq b = byte_rank_issue(ix, slot, x, lc);
    q.w = b.w; q.rem = b.rem; q.chk = b.chk; q.sup = b.sup; q.kind = 2;
    return q;
  }
  uint32_t blk;
  split960(x, blk, q.rem);
  q.w = load_line16(block_addr(ix, slot, blk, lc));
  q.kind = 1;
  return q;
}

template <bool WIDE>
__device__ __forceinline__ uint64_t rank_complete(const RankReq &q, uint32_t c, const LaneConst &lc) {
  if (q.kind == 0) return q.sup;
  if (q.kind == 2) {
    ByteRankReq b;
    b.w = q.w; b.rem = q.rem; b.chk = q.chk; b.sup = q.sup;
    return byte_rank_finish(b, c, lc);
  }
  return rank_finish<WIDE>(q.w, q.rem, lc);
}

// One backward step for the whole octet: (sp, ep) -> (C[c]+rank(c,sp), C[c]+rank(c,ep)), the body
// of SuffixAlgo.getPrevRange (findex.scala:32-36).  All lines are requested before any is consumed.
template <bool WIDE>
__device__ __forceinline__ void backward_step(const DevIndex &ix, uint32_t c, uint16_t slot, uint64_t cfc,
                                              const LaneConst &lc, uint64_t &sp, uint64_t &ep) {
  uint64_t r1 = 0, r2 = 0;
  if (slot < kSlotEof) {
    if (ix.layout == kLayoutBytes) {
      const ByteRankReq q1 = byte_rank_issue(ix, slot, sp, lc);
      const ByteRankReq q2 = byte_rank_issue(ix, slot, ep, lc);
      r1 = byte_rank_finish(q1, c, lc);
      r2 = byte_rank_finish(q2, c, lc);
    } else {
      uint32_t b1, b2, m1, m2;
      split960(sp, b1, m1);
      split960(ep, b2, m2);
      const uint4 w1 = load_line16(block_addr(ix, slot, b1, lc));
      const uint4 w2 = load_line16(block_addr(ix, slot, b2, lc));
      r1 = rank_finish<WIDE>(w1, m1, lc);
      r2 = rank_finish<WIDE>(w2, m2, lc);
    }
  } else if (slot == kSlotEof) {
    r1 = sp > ix.eof ? 1 : 0;
    r2 = ep > ix.eof ? 1 : 0;
  }
  sp = cfc + r1;
  ep = cfc + r2;
}

}  // namespace fmx

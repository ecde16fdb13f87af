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
// with one global_load_dwordx4, so every wave-level load instruction fetches 8 whole lines.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmx {

constexpr uint32_t kBlockBits = 960;     // payload positions per block
constexpr uint32_t kBlockBytes = 128;
constexpr uint32_t kOctet = 8;           // lanes per query
constexpr uint16_t kSlotNone = 0xFFFF;   // symbol absent from the BWT: rank is 0
constexpr uint16_t kSlotEof = 0xFFFE;    // symbol 0: rank0(x) = (x > eof)

struct DevIndex {
  const uint4 *bv;        // rank dictionary
  const uint8_t *bwt;     // raw BWT bytes (slot eof holds a filler)
  const uint64_t *cf;     // [256] C[] = first row of each symbol, NaiveFMSearcher.cf
  const uint16_t *slot;   // [256] symbol -> bit-vector slot | kSlotNone | kSlotEof
  uint64_t n;
  uint64_t eof;
  uint64_t nblocks;
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

// x / 960 and x % 960 for x < 2^38 without 64-bit division: 960 = 64 * 15.
__device__ __forceinline__ void split960(uint64_t x, uint64_t &blk, uint32_t &rem) {
  uint32_t y = (uint32_t)(x >> 6);
  uint32_t q = y / 15u;
  blk = q;
  rem = ((y - q * 15u) << 6) | ((uint32_t)x & 63u);
}

// Address of this lane's 16 bytes of block `blk` of slot `s`.
__device__ __forceinline__ const uint4 *block_ptr(const DevIndex &ix, uint32_t s, uint64_t blk, uint32_t t) {
  return ix.bv + ((uint64_t)s * ix.nblocks + blk) * (kBlockBytes / 16) + t;
}

// Number of set payload bits below position `rem` in this lane's 16 bytes, plus (lane 0 only)
// the header: returns the octet-wide total = rank_excl for the block.
__device__ __forceinline__ uint64_t rank_finish(uint4 w, uint32_t rem, uint32_t t) {
  uint32_t hlo = 0, hhi = 0;
  if (t == 0) { hlo = w.x; hhi = w.y; w.x = 0; w.y = 0; }
  const int base = (int)rem - 32 * (4 * (int)t - 2);   // bits wanted from dword j: base - 32 j
  uint32_t cnt = 0;
  const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int nb = base - 32 * j;
    uint32_t m = nb >= 32 ? 0xFFFFFFFFu : ((1u << (nb > 0 ? nb : 0)) - 1u);
    cnt += __builtin_popcount(ww[j] & m);
  }
  cnt = octet_sum(cnt);
  hlo = octet_or(hlo);
  hhi = octet_or(hhi);
  return (((uint64_t)hhi << 32) | hlo) + cnt;
}

// rank_excl(c, x) = #{p < x : BWT'[p] == c}, 0 <= x <= n, evaluated by the whole octet.
// occ(c, i) of the reference is rank_excl(c, i + 1).
__device__ __forceinline__ uint64_t rank_excl(const DevIndex &ix, uint16_t slot, uint64_t x, uint32_t t) {
  if (slot == kSlotNone) return 0;
  if (slot == kSlotEof) return x > ix.eof ? 1 : 0;
  uint64_t blk;
  uint32_t rem;
  split960(x, blk, rem);
  uint4 w = *block_ptr(ix, slot, blk, t);
  return rank_finish(w, rem, t);
}

}  // namespace fmx

// fmx_nfa.h -- device-side regex tables shared by the frontier kernel (fmx_frontier.hip) and the
// reference-order kernel (fmx_refmatch.hip).  All regexes of a batch are concatenated; state ids are global.
#pragma once
#include <stdint.h>

namespace fmx {

constexpr uint32_t kInlineFollows = 4;
constexpr uint32_t kMaxFollows = 0xFFFFu;   // fol_cnt is 16 bits wide
constexpr uint32_t kMaxChain = 12;          // bytes of a literal stretch one record can carry
struct StateRec {        // 32 bytes: everything an element needs about its state
  uint32_t fol_off;      // first entry of its follows in `fol` / `fol_c`
  uint32_t cnt_c_emit;   // fol_cnt in bits 0..15 (0 for a state that emits and does not expand: ReTree isLast,
                         // retree.scala:636-641), the state's byte in bits 16..23, emit flag in bit 24, chain in 25..28
  uint32_t f[kInlineFollows];   // the first follows, so that short lists need no further load.
                         // LITERAL STRETCHES: a state is "single" when it does not emit and its one follow is the
                         // next state id (s + 1) -- every character of a literal but the last.  chain = r >= 1 says
                         // that s, s+1 .. s+r-1 are all single; then f[1..3] are not follows but 12 bytes rr[]: the
                         // bytes of the states s+1 .. s+r, stored backwards (rr[r-1-j] = byte of state s+1+j).  An
                         // element that has this record walks the r-1 states s+1 .. s+r-1 without loading theirs.
  uint32_t fc;           // the bytes of f[0..3] (byte j = byte of state f[j]): an element that moves on to f[j] can
                         // request its next rank blocks without waiting for f[j]'s own record
  uint32_t regex;
};
__host__ __device__ inline uint32_t rec_cnt(const StateRec &r) { return r.cnt_c_emit & 0xFFFFu; }
__host__ __device__ inline uint32_t rec_c(const StateRec &r) { return (r.cnt_c_emit >> 16) & 0xFFu; }
__host__ __device__ inline uint32_t rec_emit(const StateRec &r) { return (r.cnt_c_emit >> 24) & 1u; }
__host__ __device__ inline uint32_t rec_chain(const StateRec &r) { return (r.cnt_c_emit >> 25) & 0xFu; }

struct NfaTables {
  const StateRec *st;
  const uint32_t *fol;
  const uint8_t *fol_c;    // byte of state fol[i], parallel to fol
};

// What ReTree._matchSA's replay needs on top: CharNode.num (the heap key) and each regex's firsts.
// FolRec: everything a push needs about the state it pushes, one 16-byte load per follow entry (parallel to `fol`;
// `first_rec` the same for the firsts) -- the state's own follow list and byte ride along, so the element is stepped
// and later expanded without a load of its StateRec.
struct FolRec {
  uint32_t fc;           // the bytes of that state's first four follows (StateRec::fc): when the element is popped, its
                         // follows' rank blocks can be requested before their own records have arrived
  uint32_t fol_off;      // that state's follows (StateRec::fol_off)
  uint32_t cnt_c_emit;   // that state's StateRec::cnt_c_emit (count, byte, isLast)
  uint32_t num;          // CharNode.num
};
struct RefTables {
  const StateRec *st;
  const uint32_t *fol;
  const uint32_t *st_num;
  const uint32_t *first_off;   // k + 1: firsts of regex r are first[first_off[r] .. first_off[r+1])
  const uint32_t *first;
  const FolRec *fol_rec;       // parallel to fol
  const FolRec *first_rec;     // parallel to first
  uint32_t max_num;            // largest CharNode.num of the batch
};

}  // namespace fmx

// fmx_nfa.h -- device-side regex tables shared by the frontier kernels (fmx_frontier.hip) and the
// reference-order kernel (fmx_refmatch.hip).  All regexes of a batch are concatenated; state ids are global.
#pragma once
#include <stdint.h>

namespace fmx {

constexpr uint32_t kInlineFollows = 4;
struct StateRec {        // 32 bytes: everything an element needs about its state, two 16-byte loads
  uint32_t fol_off;      // first entry of its follows in `fol`
  uint32_t fol_cnt;      // 0 for a state that emits and does not expand (ReTree isLast, retree.scala:636-641)
  uint32_t regex;
  uint32_t c_emit;       // byte in bits 0..7, emit flag in bit 8
  uint32_t f[kInlineFollows];   // the first follows, so that short lists need no further load
};

struct NfaTables {
  const StateRec *st;
  const uint32_t *fol;
};

// What ReTree._matchSA's replay needs on top: CharNode.num (the heap key) and each regex's firsts.
struct RefTables {
  const StateRec *st;
  const uint32_t *fol;
  const uint32_t *st_num;
  const uint32_t *first_off;   // k + 1: firsts of regex r are first[first_off[r] .. first_off[r+1])
  const uint32_t *first;
};

}  // namespace fmx

// fmx_regex.h -- host-side regex front-end: REParser.re2post + ReTree (Glushkov position
// automaton) flattened to the tables the frontier kernel walks.
// Reference: src/main/scala/org/fmindex/re2/re2.scala:21-185, re2/retree.scala:9-484.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace fmx {

struct RegexError {
  int code;            // FMX_ERR_SYNTAX or FMX_ERR_MATCH
  std::string msg;
};

// One postfix token (REParser.PostPoint and subclasses, re2.scala:24-48).
struct PostPoint {
  enum Kind { Char, Interval, Alt, Concat, Star, Question, Plus, Or } kind;
  int c = 0;                  // Char
  int start = 0, end = 0;     // Interval (end exclusive once expanded, retree.scala:165-173)
  std::vector<int> alts;      // Alt: newest first, like the Scala list
};

std::vector<PostPoint> re2post(const std::string &re, bool line_only);   // throws RegexError
std::string re2poststr(const std::string &re, bool line_only);           // UTF-8, re2.scala:187

// What ReTree._matchSA (retree.scala:618-653) touches, per CharNode in tree order.
struct Regex {
  std::string source;
  int engine = 0;                  // 0 ReTree (Glushkov), 1 REParser.createNFA (Thompson), 2 DFA table
  bool last_stops = true;          // ReTree: an isLast state emits and does not expand (retree.scala:636-641)
  bool start_is_final = false;     // DFA whose state 0 is final: the reference reports (len 0, 0, n)
  std::vector<uint8_t> st_c;       // CharNode.c
  std::vector<int32_t> st_num;     // CharNode.num (retree.scala:393-423)
  std::vector<uint8_t> st_last;    // isLast (retree.scala:40-50)
  std::vector<int32_t> fol_off;    // CSR offsets, n_states + 1
  std::vector<int32_t> fol;        // follows (retree.scala:14-38): order and multiplicity kept
  std::vector<int32_t> firsts;     // root.firsts
};

Regex compile_regex(const std::string &re, bool line_only);               // throws RegexError

// REParser.post2re (re2.scala:188-205): a postfix string where '.' is the concat token.
std::vector<PostPoint> post2re(const std::string &s);
// REParser.createNFA (re2.scala:264-334) + the closure REParser.matchSA walks (outStates, :213-224):
// per TermState char one kernel state; st_last = MatchState is among next.outStates (emit), fol = the
// TermStates among them (push).  Throws RegexError(FMX_ERR_MATCH) where the reference throws
// scala.MatchError (AltPoint tokens; a nullable regex, whose MatchState start point cannot expand).
Regex compile_thompson(const std::vector<PostPoint> &post, const std::string &source);
// DFA.compileBuckets + DFA.matchSA's expansion rule (dfa.scala:190-213,242-259): one kernel state per
// single-character action (runs of two or more characters to one target are DFABuckets, which the
// reference's expand ignores); moves is nstates x nchars, -1 = no transition; finish[s] != 0 marks
// the final states.
Regex compile_dfa(const int32_t *moves, uint32_t nstates, uint32_t nchars, const uint8_t *finish);

}  // namespace fmx

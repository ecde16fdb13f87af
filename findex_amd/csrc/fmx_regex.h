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
  std::vector<uint8_t> st_c;       // CharNode.c
  std::vector<int32_t> st_num;     // CharNode.num (retree.scala:393-423)
  std::vector<uint8_t> st_last;    // isLast (retree.scala:40-50)
  std::vector<int32_t> fol_off;    // CSR offsets, n_states + 1
  std::vector<int32_t> fol;        // follows (retree.scala:14-38): order and multiplicity kept
  std::vector<int32_t> firsts;     // root.firsts
};

Regex compile_regex(const std::string &re, bool line_only);               // throws RegexError

}  // namespace fmx

// Fuzz harness for the host-side regex front-end (findex_amd/csrc/fmx_regex.cpp: re2post, ReTree tables,
// createNFA, post2re).  Built by tests/test_regex_fuzz_cpu.py with -fsanitize=address,undefined and fed random
// strings over the grammar's characters plus junk: every input must either compile into tables whose indexes
// are in range or raise RegexError -- never crash, overrun or hit undefined behaviour.
//   fuzz_regex <seed> <iterations>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include "fmx_regex.h"
using namespace fmx;
int main(int argc, char **argv) {
  unsigned seed = argc > 1 ? atoi(argv[1]) : 1;
  int iters = argc > 2 ? atoi(argv[2]) : 200000;
  std::mt19937 rng(seed);
  const char alpha[] = "ab.c*+?|()[]-\\wd^$z09 A{},\x01\xff\x80";
  size_t ok = 0, syn = 0, mat = 0, states = 0;
  for (int it = 0; it < iters; it++) {
    std::string s;
    int len = rng() % 14;
    for (int j = 0; j < len; j++) s.push_back(alpha[rng() % (sizeof(alpha) - 1)]);
    for (int lineOnly = 0; lineOnly < 2; lineOnly++) {
      try {
        Regex r = compile_regex(s, lineOnly != 0);
        ok++; states += r.st_c.size();
        // table invariants
        if (r.fol_off.size() != r.st_c.size() + 1) { printf("bad fol_off for %s\n", s.c_str()); return 1; }
        for (int32_t f : r.fol) if (f < 0 || (size_t)f >= r.st_c.size()) { printf("bad follow for %s\n", s.c_str()); return 1; }
        for (int32_t f : r.firsts) if (f < 0 || (size_t)f >= r.st_c.size()) { printf("bad first for %s\n", s.c_str()); return 1; }
      } catch (const RegexError &e) {
        if (e.code == 7) syn++; else mat++;
      }
      try {
        std::vector<PostPoint> post = re2post(s, lineOnly != 0);
        (void)re2poststr(s, lineOnly != 0);
        Regex t = compile_thompson(post, s);
        for (int32_t f : t.fol) if (f < 0 || (size_t)f >= t.st_c.size()) { printf("bad thompson follow for %s\n", s.c_str()); return 1; }
      } catch (const RegexError &) {
      }
      try { (void)post2re(s); } catch (const RegexError &) {}
    }
  }
  printf("seed %u: %zu compiled (%zu states), %zu syntax errors, %zu match errors\n", seed, ok, states, syn, mat);
  return 0;
}

// fmx_regex.cpp -- REParser.re2post and ReTree, host side (see fmx_regex.h).
//
// The reference keeps its trees in Scala immutable Lists that are built by prepending; here a
// list is a std::vector whose element 0 is the head, `xs ::= x` is insert-at-front and
// `xs :::= ys` is "ys then xs".  Node identity is the arena index.
#include "fmx_regex.h"

#include <fmx.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

namespace fmx {
namespace {

constexpr int MIN_CHAR = 2;     // re2.scala:22
constexpr int MAX_CHAR = 255;   // re2.scala:23

[[noreturn]] void syntax() { throw RegexError{FMX_ERR_SYNTAX, "re2post syntax"}; }

struct Paren { int nalt, natom; };

}  // namespace

// REParser.re2post, re2.scala:50-185
std::vector<PostPoint> re2post(const std::string &s, bool line_only) {
  const int l = (int)s.size();
  int natom = 0, nalt = 0;
  std::vector<PostPoint> dst;          // appended in emission order (the Scala code prepends, then reverses)
  dst.reserve(2 * (size_t)l + 2);
  std::vector<Paren> stack;
  auto emit = [&](PostPoint::Kind k) { PostPoint p; p.kind = k; dst.push_back(p); };
  auto concat_if = [&]() {
    if (natom > 1) { natom -= 1; emit(PostPoint::Concat); }
  };
  auto process_char = [&](int c, bool quoted) {          // :60-75
    concat_if();
    PostPoint p;
    p.kind = PostPoint::Char;
    p.c = c;
    if (quoted) {
      if (c == 'w') { p.kind = PostPoint::Interval; p.start = 'A'; p.end = 'z'; }
      else if (c == 'd') { p.kind = PostPoint::Interval; p.start = '0'; p.end = '9'; }
    } else if (c == '.') {
      p.kind = PostPoint::Interval;
      p.start = line_only ? 0x20 : MIN_CHAR;
      p.end = MAX_CHAR;
    }
    dst.push_back(p);
    natom += 1;
  };
  auto process_alt = [&](int i) -> int {                  // :76-119
    std::vector<int> alts;                               // newest first
    bool quoted = false, end = false, interval = false;
    auto pc = [&](int c) {
      if (interval) {
        if (alts.empty()) syntax();
        int cAlt = alts.front() + 1;
        const int eAlt = c;
        if (cAlt > eAlt) syntax();
        while (cAlt <= eAlt) { alts.insert(alts.begin(), cAlt); cAlt += 1; }
        interval = false;
      } else {
        alts.insert(alts.begin(), c);
      }
    };
    while (i < l && !end) {
      const int c = (unsigned char)s[i];
      if (quoted) { pc(c); quoted = false; }
      else if (c == '\\') quoted = true;
      else if (c == '-') interval = true;
      else if (c == ']') end = true;
      else pc(c);
      i += 1;
    }
    if (!end || interval) syntax();
    concat_if();
    PostPoint p;
    p.kind = PostPoint::Alt;
    p.alts = alts;
    dst.push_back(p);
    natom += 1;
    return i;
  };

  int i = 0;
  bool quoted = false;
  while (i < l) {
    const int c = (unsigned char)s[i];
    if (!quoted) {
      switch (c) {
        case '(':                                         // :124-131
          concat_if();
          stack.push_back({nalt, natom});
          nalt = 0;
          natom = 0;
          break;
        case '|':                                         // :132-139
          if (natom == 0) syntax();
          natom -= 1;
          while (natom > 0) { emit(PostPoint::Concat); natom -= 1; }
          nalt += 1;
          break;
        case ')': {                                       // :140-153
          if (natom == 0) syntax();
          natom -= 1;
          while (natom > 0) { emit(PostPoint::Concat); natom -= 1; }
          while (nalt > 0) { emit(PostPoint::Or); nalt -= 1; }
          if (stack.empty()) syntax();                    // Stack.pop on empty: NoSuchElementException there
          const Paren t = stack.back();
          stack.pop_back();
          nalt = t.nalt;
          natom = t.natom + 1;
          break;
        }
        case '[':                                         // :154-155
          i = process_alt(i + 1) - 1;
          break;
        case '\\':
          quoted = true;
          break;
        case '*': case '+': case '?':                     // :158-164
          if (natom == 0) syntax();
          emit(c == '*' ? PostPoint::Star : c == '+' ? PostPoint::Plus : PostPoint::Question);
          break;
        default:
          process_char(c, false);
      }
    } else {
      process_char(c, true);
      quoted = false;
    }
    i += 1;
  }
  if (!stack.empty()) syntax();
  natom -= 1;
  while (natom > 0) { emit(PostPoint::Concat); natom -= 1; }
  while (nalt > 0) { emit(PostPoint::Or); nalt -= 1; }
  return dst;
}

static void put_utf8(std::string &o, int c) {
  if (c < 0x80) o.push_back((char)c);
  else { o.push_back((char)(0xC0 | (c >> 6))); o.push_back((char)(0x80 | (c & 0x3F))); }
}

// PostPoint.toString, re2.scala:25-48
std::string re2poststr(const std::string &re, bool line_only) {
  std::string o;
  for (const PostPoint &p : re2post(re, line_only)) {
    switch (p.kind) {
      case PostPoint::Char: put_utf8(o, p.c); break;
      case PostPoint::Interval:
        if (p.start == MIN_CHAR && p.end == MAX_CHAR) o += ".";
        else { o += "["; put_utf8(o, p.start); o += "-"; put_utf8(o, p.end); o += "]"; }
        break;
      case PostPoint::Alt:
        o += "[";
        for (auto it = p.alts.rbegin(); it != p.alts.rend(); ++it) put_utf8(o, *it);
        o += "]";
        break;
      case PostPoint::Concat: o += "\xC2\xB7"; break;     // U+00B7
      case PostPoint::Star: o += "*"; break;
      case PostPoint::Question: o += "?"; break;
      case PostPoint::Plus: o += "+"; break;
      case PostPoint::Or: o += "|"; break;
    }
  }
  return o;
}

// ------------------------------------------------------------------------------ ReTree
namespace {

enum NodeKind { N_CHAR, N_STAR, N_QUESTION, N_PLUS, N_OR, N_FOLLOW };
constexpr int ROOT = -1;

// The lists of the tree code -- childs, firsts, follows -- hold a handful of node ids and are built by prepending;
// a regex compile made ~300 of them, each a heap allocation (12 us per regex, most of it malloc/free).  IVec keeps up
// to 12 ids in place and only longer lists (a '.' has 253 children) on the heap.
class IVec {
 public:
  IVec() = default;
  IVec(const int *b, const int *e) { assign(b, e); }
  IVec(const IVec &o) { assign(o.begin(), o.end()); }
  IVec(IVec &&o) noexcept { steal(o); }
  IVec &operator=(const IVec &o) { if (this != &o) assign(o.begin(), o.end()); return *this; }
  IVec &operator=(IVec &&o) noexcept { if (this != &o) { release(); steal(o); } return *this; }
  ~IVec() { release(); }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  int *begin() { return p_; }
  int *end() { return p_ + n_; }
  const int *begin() const { return p_; }
  const int *end() const { return p_ + n_; }
  int front() const { return p_[0]; }
  int operator[](size_t i) const { return p_[i]; }
  void push_back(int v) { grow(n_ + 1); p_[n_++] = v; }
  void push_front(int v) { grow(n_ + 1); std::memmove(p_ + 1, p_, n_ * sizeof(int)); p_[0] = v; n_++; }
  void append(const IVec &o) { const size_t m = o.n_; grow(n_ + m); std::memmove(p_ + n_, o.p_, m * sizeof(int)); n_ += m; }   // o may be *this
  void prepend(const IVec &o) {             // o ++ this
    const size_t m = o.n_;
    if (&o == this) { append(o); return; }
    grow(n_ + m);
    std::memmove(p_ + m, p_, n_ * sizeof(int));
    std::memcpy(p_, o.p_, m * sizeof(int));
    n_ += m;
  }
  void drop_front(size_t k) { std::memmove(p_, p_ + k, (n_ - k) * sizeof(int)); n_ -= k; }
  void swap(IVec &o) { IVec t(std::move(o)); o = std::move(*this); *this = std::move(t); }

 private:
  static constexpr size_t kInline = 12;
  void assign(const int *b, const int *e) { n_ = 0; grow((size_t)(e - b)); std::memmove(p_, b, (size_t)(e - b) * sizeof(int)); n_ = (size_t)(e - b); }
  void grow(size_t want) {
    if (want <= cap_) return;
    size_t c = cap_ * 2;
    if (c < want) c = want;
    int *q = static_cast<int *>(std::malloc(c * sizeof(int)));
    if (!q) throw std::bad_alloc();
    std::memcpy(q, p_, n_ * sizeof(int));
    if (p_ != in_) std::free(p_);
    p_ = q;
    cap_ = c;
  }
  void release() { if (p_ != in_) std::free(p_); p_ = in_; cap_ = kInline; n_ = 0; }
  void steal(IVec &o) {
    n_ = o.n_;
    if (o.p_ == o.in_) { p_ = in_; cap_ = kInline; std::memcpy(in_, o.in_, o.n_ * sizeof(int)); }
    else { p_ = o.p_; cap_ = o.cap_; o.p_ = o.in_; o.cap_ = kInline; }
    o.n_ = 0;
  }
  int in_[kInline];
  int *p_ = in_;
  size_t n_ = 0, cap_ = kInline;
};

struct Node {
  NodeKind kind;
  int c = 0;
  int num = 0;
  int parent = ROOT;
  IVec childs;    // head first
};

struct Tree {
  std::vector<Node> a;

  int make(NodeKind k, int c = 0) {
    Node n;
    n.kind = k;
    n.c = c;
    a.push_back(n);
    return (int)a.size() - 1;
  }
  static bool unar(NodeKind k) { return k == N_STAR || k == N_QUESTION || k == N_PLUS; }

  // Node.append of each class: UnarOp :73-77, Or :99-110, Follow :130-133
  void append(int self, int n) {
    Node &s = a[self];
    if (s.kind == N_OR && a[n].kind == N_OR) {
      for (int ch : a[n].childs) a[ch].parent = self;
      s.childs.prepend(a[n].childs);
    } else {
      a[n].parent = self;
      s.childs.push_front(n);
    }
  }

  bool is_null(int x) const {
    const Node &n = a[x];
    switch (n.kind) {
      case N_CHAR: return false;
      case N_STAR: case N_QUESTION: return true;
      case N_PLUS: case N_FOLLOW:
        for (int ch : n.childs) if (!is_null(ch)) return false;
        return true;
      case N_OR:
        for (int ch : n.childs) if (is_null(ch)) return true;
        return false;
    }
    return false;
  }

  IVec firsts(int x) const {
    const Node &n = a[x];
    IVec ret;
    if (n.kind == N_CHAR) { ret.push_back(x); return ret; }
    if (n.kind == N_FOLLOW) {                              // :117-127
      size_t p = 0;
      auto prepend = [&](int ch) { ret.prepend(firsts(ch)); };
      while (p < n.childs.size() && is_null(n.childs[p])) { prepend(n.childs[p]); p++; }
      if (p < n.childs.size()) prepend(n.childs[p]);
      return ret;
    }
    for (int ch : n.childs) ret.append(firsts(ch));        // flatMap, :78,97
    return ret;
  }

  IVec siblings_after(int parent, int me) const {   // childs.dropWhile(_ != this).tail
    const IVec &ch = a[parent].childs;
    size_t k = 0;
    while (k < ch.size() && ch[k] != me) k++;
    if (k >= ch.size()) return {};
    return IVec(ch.begin() + k + 1, ch.end());
  }

  IVec follows(int x) const {                  // :14-38
    const int p = a[x].parent;
    if (p == ROOT) return {};
    const Node &pn = a[p];
    switch (pn.kind) {
      case N_OR: return follows(p);
      case N_FOLLOW: {
        const IVec last = siblings_after(p, x);
        if (last.empty()) return follows(p);
        IVec ret = firsts(last[0]);
        auto prepend = [&](int ch) { ret.prepend(firsts(ch)); };
        if (is_null(last[0])) {
          size_t k = 1;
          while (k < last.size() && is_null(last[k])) { prepend(last[k]); k++; }
          if (k < last.size()) prepend(last[k]);
        }
        return ret;
      }
      case N_STAR: {
        IVec ret = firsts(x);
        ret.append(follows(p));
        return ret;
      }
      case N_QUESTION: return follows(p);
      default: return {};                                  // PlusNode parent: List()
    }
  }

  bool is_last(int x) const {                              // :40-50
    const int p = a[x].parent;
    if (p == ROOT) return true;
    const Node &pn = a[p];
    if (pn.kind == N_OR || unar(pn.kind)) return is_last(p);
    if (pn.kind == N_FOLLOW) {
      const IVec last = siblings_after(p, x);
      bool all_null = true;
      for (int s : last) if (!is_null(s)) { all_null = false; break; }
      return (last.empty() || all_null) ? is_last(p) : false;
    }
    return true;
  }
};

char kind_letter(NodeKind k) {
  switch (k) {
    case N_CHAR: return 'C';
    case N_OR: return 'O';
    case N_FOLLOW: return 'F';
    default: return 'U';
  }
}

[[noreturn]] void match_error(const char *what) { throw RegexError{FMX_ERR_MATCH, what}; }

int pop(std::vector<int> &args) {
  if (args.empty()) throw RegexError{FMX_ERR_MATCH, "pop of an empty operand stack (NoSuchElementException)"};
  int v = args.back();
  args.pop_back();
  return v;
}

// PlusPoint / StarPoint / QuestionPoint, retree.scala:296-337
int unary(Tree &t, int a1, PostPoint::Kind op) {
  const NodeKind k = t.a[a1].kind;
  auto star_of_child = [&]() {
    int el = t.make(N_STAR);
    t.append(el, t.a[a1].childs.front());
    return el;
  };
  if (op == PostPoint::Plus || op == PostPoint::Star) {
    if (k == N_STAR) return a1;
    if (k == N_QUESTION || k == N_PLUS) return star_of_child();
    int el = t.make(op == PostPoint::Plus ? N_PLUS : N_STAR);
    t.append(el, a1);
    return el;
  }
  if (k == N_QUESTION) {
    int el = t.make(N_QUESTION);
    t.append(el, t.a[a1].childs.front());
    return el;
  }
  if (k == N_STAR) return a1;
  if (k == N_PLUS) return star_of_child();
  int el = t.make(N_QUESTION);
  t.append(el, a1);
  return el;
}

// postProcess, retree.scala:439-482: rebuilds the tree (which undoes the prepend order of
// `childs`) and rewrites x+ as x x*.
int post_process(Tree &t, int r) {
  const NodeKind k = t.a[r].kind;
  if (k == N_CHAR) return t.make(N_CHAR, t.a[r].c);
  if (k == N_PLUS) match_error("postProcess: PlusNode has no case");
  const int nc = t.make(k);
  const IVec old = t.a[r].childs;
  for (int ch : old) {
    if (t.a[ch].kind == N_PLUS) {
      const int inner = t.a[ch].childs.front();
      const int a1 = post_process(t, inner);
      const int a2 = t.make(N_STAR);
      t.append(a2, post_process(t, inner));
      IVec &c = t.a[nc].childs;                          // a1 :: a2 :: newL
      c.push_front(a2);
      c.push_front(a1);
    } else {
      const int pc = post_process(t, ch);
      t.a[nc].childs.push_front(pc);
    }
  }
  return nc;
}

int remove_border_nulls(Tree &t, int a1) {                 // :371-385
  const int n = t.make(N_FOLLOW);
  IVec p = t.a[a1].childs;
  size_t b = 0;
  while (b < p.size() && t.is_null(p[b])) b++;
  p.drop_front(b);
  std::reverse(p.begin(), p.end());
  b = 0;
  while (b < p.size() && t.is_null(p[b])) b++;
  p.drop_front(b);
  for (int x : p) t.append(n, x);
  return n;
}

void set_parents(Tree &t, int r, int parent) {             // :386-391
  t.a[r].parent = parent;
  for (int ch : t.a[r].childs) set_parents(t, ch, r);
}

int set_nums_scope(Tree &t, int r, int idx0);

int set_nums_inner(Tree &t, int r, int &idx) {             // __setNums, :396-418
  if (t.a[r].kind == N_OR) {
    int nidx = idx;
    for (int ch : t.a[r].childs) {
      if (t.a[ch].kind == N_CHAR) {
        t.a[ch].num = idx;
        nidx = std::max(nidx, idx + 1);
      } else {
        nidx = std::max(nidx, set_nums_scope(t, ch, idx));
      }
    }
    idx = nidx;
  } else {
    for (int ch : t.a[r].childs) {
      if (t.a[ch].kind == N_CHAR) { t.a[ch].num = idx; idx += 1; }
      else set_nums_inner(t, ch, idx);
    }
  }
  return idx;
}

int set_nums_scope(Tree &t, int r, int idx0) {             // _setNums, :394-420
  int idx = idx0;
  return set_nums_inner(t, r, idx);
}

void collect_chars(const Tree &t, int r, std::vector<int> &out) {
  if (t.a[r].kind == N_CHAR) out.push_back(r);
  for (int ch : t.a[r].childs) collect_chars(t, ch, out);
}

}  // namespace

// ReTree.apply, retree.scala:156-370, then the flattening the kernels use.
Regex compile_regex(const std::string &re, bool line_only) {
  const std::vector<PostPoint> post = re2post(re, line_only);
  // the node arena and the operand stack keep their capacity from one compile of this thread to the next
  static thread_local Tree t;
  static thread_local std::vector<int> args;               // mutable.Stack, top = back
  t.a.clear();
  args.clear();
  for (const PostPoint &c : post) {
    switch (c.kind) {
      case PostPoint::Interval: {                          // :165-173, end exclusive
        int el = t.make(N_OR);
        for (int j = c.start; j < c.end; j++) t.append(el, t.make(N_CHAR, j));
        args.push_back(el);
        break;
      }
      case PostPoint::Alt: {                               // :174-179
        int el = t.make(N_OR);
        for (int ch : c.alts) t.append(el, t.make(N_CHAR, ch));
        args.push_back(el);
        break;
      }
      case PostPoint::Char:
        args.push_back(t.make(N_CHAR, c.c));
        break;
      case PostPoint::Or: {                                // :181-239
        const int a2 = pop(args), a1 = pop(args);
        const char k1 = kind_letter(t.a[a1].kind), k2 = kind_letter(t.a[a2].kind);
        if (k2 == 'O' && (k1 == 'C' || k1 == 'U' || k1 == 'F' || k1 == 'O')) {
          t.append(a2, a1);
          args.push_back(a2);
        } else if ((k1 == 'F' && k2 == 'F') || (k1 == 'C' && k2 == 'C') || (k1 == 'U' && k2 == 'F') ||
                   (k1 == 'C' && k2 == 'F') || (k1 == 'U' && k2 == 'C') || (k1 == 'F' && k2 == 'C') ||
                   (k1 == 'U' && k2 == 'U')) {
          int el = t.make(N_OR);
          t.append(el, a1);
          t.append(el, a2);
          args.push_back(el);
        } else {
          match_error("OrPoint have no match for these operands");
        }
        break;
      }
      case PostPoint::Concat: {                            // :240-295
        const int a2 = pop(args), a1 = pop(args);
        const char k1 = kind_letter(t.a[a1].kind), k2 = kind_letter(t.a[a2].kind);
        if ((k1 == 'O' && k2 == 'O') || (k1 == 'C' && k2 == 'O') || (k1 == 'C' && k2 == 'C') ||
            (k1 == 'U' && k2 == 'C') || (k1 == 'U' && k2 == 'O') || (k1 == 'C' && k2 == 'U') ||
            (k1 == 'U' && k2 == 'U')) {
          int el = t.make(N_FOLLOW);
          t.append(el, a1);
          t.append(el, a2);
          args.push_back(el);
        } else if (k1 == 'F' && (k2 == 'C' || k2 == 'O' || k2 == 'U')) {
          t.append(a1, a2);
          args.push_back(a1);
        } else {
          match_error("ConcatPoint have no match for these operands");
        }
        break;
      }
      case PostPoint::Plus: case PostPoint::Star: case PostPoint::Question:
        args.push_back(unary(t, pop(args), c.kind));
        break;
    }
  }
  const int a0 = pop(args);
  int a2 = a0;                                             // :345-360
  if (t.a[a0].kind != N_FOLLOW) {
    a2 = t.make(N_FOLLOW);
    t.append(a2, a0);
  }
  const int a1 = post_process(t, a2);
  const int a3 = remove_border_nulls(t, a1);
  set_parents(t, a3, ROOT);
  set_nums_scope(t, a3, 1);

  static thread_local std::vector<int> chars, index_of;
  chars.clear();
  collect_chars(t, a3, chars);
  index_of.assign(t.a.size(), -1);
  for (size_t k = 0; k < chars.size(); k++) index_of[chars[k]] = (int)k;
  Regex out;
  out.source = re;
  out.st_c.reserve(chars.size());
  out.st_num.reserve(chars.size());
  out.st_last.reserve(chars.size());
  out.fol_off.reserve(chars.size() + 1);
  out.fol.reserve(2 * chars.size());
  out.fol_off.push_back(0);
  for (int x : chars) {
    if (t.a[x].c < 0 || t.a[x].c > 255) throw RegexError{FMX_ERR_SYNTAX, "character outside 0..255"};
    out.st_c.push_back((uint8_t)t.a[x].c);
    out.st_num.push_back(t.a[x].num);
    out.st_last.push_back(t.is_last(x) ? 1 : 0);
    for (int f : t.follows(x)) out.fol.push_back(index_of[f]);
    out.fol_off.push_back((int32_t)out.fol.size());
  }
  for (int f : t.firsts(a3)) out.firsts.push_back(index_of[f]);
  return out;
}

}  // namespace fmx

// ------------------------------------------------------------------------------ Thompson NFA (REParser)
namespace fmx {

std::vector<PostPoint> post2re(const std::string &s) {        // re2.scala:188-205
  std::vector<PostPoint> out;
  for (unsigned char ch : s) {
    PostPoint p;
    switch (ch) {
      case '*': p.kind = PostPoint::Star; break;
      case '.': p.kind = PostPoint::Concat; break;
      case '|': p.kind = PostPoint::Or; break;
      case '?': p.kind = PostPoint::Question; break;
      case '+': p.kind = PostPoint::Plus; break;
      default: p.kind = PostPoint::Char; p.c = ch;
    }
    out.push_back(p);
  }
  return out;
}

namespace {

enum TKind { T_CONST, T_INTERVAL, T_SPLIT, T_MATCH };
struct TState { TKind kind; int c = 0, start = 0, end = 0; int out = -1, out1 = -1, out2 = -1; };   // link ids
struct Frag0 { int start; std::vector<int> out; };

struct TNfa {
  std::vector<TState> st;
  std::vector<int> link;      // LinkState.s (state id or -1)
  int new_link(int s = -1) { link.push_back(s); return (int)link.size() - 1; }
  int add(TState t) { st.push_back(t); return (int)st.size() - 1; }

  // BaseState.outStates, re2.scala:213-224 (a Set there: no duplicates; order not part of the results)
  void out_states(int s, std::vector<char> &seen, std::vector<int> &acc) const {
    if (s < 0 || seen[s]) return;
    const TState &t = st[s];
    if (t.kind == T_SPLIT) {
      seen[s] = 1;
      out_states(link[t.out1], seen, acc);
      if (t.out2 >= 0) out_states(link[t.out2], seen, acc);
    } else {
      seen[s] = 1;
      acc.push_back(s);
    }
  }
  std::vector<int> out_states(int s) const {
    std::vector<char> seen(st.size(), 0);
    std::vector<int> acc;
    out_states(s, seen, acc);
    return acc;
  }
};

Frag0 pop_frag(std::vector<Frag0> &s0) {
  if (s0.empty()) throw RegexError{FMX_ERR_MATCH, "createNFA: pop of an empty stack (NoSuchElementException)"};
  Frag0 f = s0.back();
  s0.pop_back();
  return f;
}

}  // namespace

Regex compile_thompson(const std::vector<PostPoint> &post, const std::string &source) {
  TNfa nfa;
  std::vector<Frag0> s0;
  const int match = nfa.add(TState{T_MATCH});
  auto patch = [&](const Frag0 &f, int s) { for (int l : f.out) nfa.link[l] = s; };
  for (const PostPoint &c : post) {                      // createNFA, re2.scala:286-326
    switch (c.kind) {
      case PostPoint::Question: {
        Frag0 e = pop_frag(s0);
        const int open = nfa.new_link();
        TState t{T_SPLIT}; t.out1 = nfa.new_link(e.start); t.out2 = open;
        const int ns = nfa.add(t);
        std::vector<int> out{open};
        out.insert(out.end(), e.out.begin(), e.out.end());
        s0.push_back(Frag0{ns, out});
        break;
      }
      case PostPoint::Star: {
        Frag0 e = pop_frag(s0);
        const int open = nfa.new_link();
        TState t{T_SPLIT}; t.out1 = nfa.new_link(e.start); t.out2 = open;
        const int ns = nfa.add(t);
        patch(e, ns);
        s0.push_back(Frag0{ns, {open}});
        break;
      }
      case PostPoint::Plus: {
        Frag0 e = pop_frag(s0);
        const int open = nfa.new_link();
        TState t{T_SPLIT}; t.out1 = nfa.new_link(e.start); t.out2 = open;
        const int ns = nfa.add(t);
        patch(e, ns);
        s0.push_back(Frag0{e.start, {open}});
        break;
      }
      case PostPoint::Concat: {
        Frag0 e2 = pop_frag(s0), e1 = pop_frag(s0);
        patch(e1, e2.start);
        s0.push_back(Frag0{e1.start, e2.out});
        break;
      }
      case PostPoint::Or: {
        Frag0 e2 = pop_frag(s0), e1 = pop_frag(s0);
        TState t{T_SPLIT}; t.out1 = nfa.new_link(e1.start); t.out2 = nfa.new_link(e2.start);
        const int ns = nfa.add(t);
        std::vector<int> out = e1.out;
        out.insert(out.end(), e2.out.begin(), e2.out.end());
        s0.push_back(Frag0{ns, out});
        break;
      }
      case PostPoint::Char: {
        TState t{T_CONST}; t.c = c.c; t.out = nfa.new_link();
        const int ns = nfa.add(t);
        s0.push_back(Frag0{ns, {nfa.st[ns].out}});
        break;
      }
      case PostPoint::Interval: {
        TState t{T_INTERVAL}; t.start = c.start; t.end = c.end; t.out = nfa.new_link();
        const int ns = nfa.add(t);
        s0.push_back(Frag0{ns, {nfa.st[ns].out}});
        break;
      }
      case PostPoint::Alt:
        throw RegexError{FMX_ERR_MATCH, "createNFA has no case for AltPoint ([..] sets)"};
    }
  }
  Frag0 e0 = pop_frag(s0);
  patch(e0, match);
  const int start = e0.start;

  // flatten: one kernel state per (TermState, char)
  std::vector<int> first_id(nfa.st.size(), -1), n_chars(nfa.st.size(), 0);
  Regex out;
  out.source = source;
  out.engine = 1;
  out.last_stops = false;
  std::vector<int> term_of;     // kernel state -> term state
  for (size_t s = 0; s < nfa.st.size(); s++) {
    const TState &t = nfa.st[s];
    int lo = 0, hi = 0;
    if (t.kind == T_CONST) { lo = t.c; hi = t.c + 1; }
    else if (t.kind == T_INTERVAL) { lo = t.start; hi = t.end; }       // `start until end`, re2.scala:472
    else continue;
    first_id[s] = (int)out.st_c.size();
    n_chars[s] = std::max(0, hi - lo);
    for (int ch = lo; ch < hi; ch++) {
      if (ch < 0 || ch > 255) throw RegexError{FMX_ERR_SYNTAX, "character outside 0..255"};
      out.st_c.push_back((uint8_t)ch);
      out.st_num.push_back(0);
      term_of.push_back((int)s);
    }
  }
  auto expand_targets = [&](const std::vector<int> &states, std::vector<int32_t> &dst, bool &has_match) {
    has_match = false;
    for (int s : states) {
      if (nfa.st[s].kind == T_MATCH) { has_match = true; continue; }
      for (int j = 0; j < n_chars[s]; j++) dst.push_back(first_id[s] + j);
    }
  };
  out.fol_off.push_back(0);
  for (size_t q = 0; q < out.st_c.size(); q++) {
    const TState &t = nfa.st[term_of[q]];
    bool m = false;
    expand_targets(nfa.out_states(nfa.link[t.out]), out.fol, m);       // nextStates, re2.scala:450-453
    out.st_last.push_back(m ? 1 : 0);
    out.fol_off.push_back((int32_t)out.fol.size());
  }
  bool start_match = false;
  expand_targets(nfa.out_states(start), out.firsts, start_match);
  if (start_match)
    throw RegexError{FMX_ERR_MATCH, "the regex matches the empty string: REParser.matchSA would expand a MatchState "
                                    "start point, which StatePoint.expand has no case for (re2.scala:457-482)"};
  return out;
}

Regex compile_dfa(const int32_t *moves, uint32_t nstates, uint32_t nchars, const uint8_t *finish) {
  if (!moves || !finish || nstates == 0 || nchars == 0 || nchars > 256)
    throw RegexError{FMX_ERR_ARG, "bad DFA table"};
  Regex out;
  out.source = "<dfa>";
  out.engine = 2;
  out.last_stops = false;
  out.start_is_final = finish[0] != 0;
  // compileBuckets, dfa.scala:190-213: maximal runs of equal targets; only runs of one character
  // (DFAChar) are expanded by StatePoint.expand (:242-259)
  std::vector<std::vector<std::pair<int, int>>> act(nstates);     // per state: (char, target)
  for (uint32_t i = 0; i < nstates; i++) {
    int last = -1, start_bucket = -1;
    for (uint32_t j = 0; j < nchars; j++) {
      const int v = moves[(size_t)i * nchars + j];
      if (v < -1 || v >= (int)nstates) throw RegexError{FMX_ERR_ARG, "DFA target out of range"};
      if (last != v) {
        if (last != -1 && start_bucket == (int)j - 1) act[i].push_back({start_bucket, last});
        start_bucket = (int)j;
        last = v;
      }
    }
    if (last != -1 && start_bucket == (int)nchars - 1) act[i].push_back({start_bucket, last});
  }
  std::vector<int> first_id(nstates, 0);
  for (uint32_t i = 0; i < nstates; i++) {
    first_id[i] = (int)out.st_c.size();
    for (auto &a : act[i]) {
      out.st_c.push_back((uint8_t)a.first);
      out.st_num.push_back((int32_t)i);
      out.st_last.push_back(finish[a.second] ? 1 : 0);          // the target is reported when popped, :270-273
    }
  }
  out.fol_off.push_back(0);
  for (uint32_t i = 0; i < nstates; i++)
    for (auto &a : act[i]) {
      for (size_t j = 0; j < act[a.second].size(); j++) out.fol.push_back(first_id[a.second] + (int)j);
      out.fol_off.push_back((int32_t)out.fol.size());
    }
  for (size_t j = 0; j < act[0].size(); j++) out.firsts.push_back(first_id[0] + (int)j);
  return out;
}

}  // namespace fmx

"""CPU restatement of findex's regex front-end and Glushkov tree (ReTree).

TEST INFRASTRUCTURE ONLY -- see oracle/README.md.  Only tests/, smoke() and
bench.py's cpu_baseline leg may import this; the product never does.

Follows (paths relative to /root/reference):
  re2post ........ src/main/scala/org/fmindex/re2/re2.scala:21-185
  ReTree ......... src/main/scala/org/fmindex/re2/retree.scala:9-484
  _matchSA ....... src/main/scala/org/fmindex/re2/retree.scala:562-653

Scala's immutable List is modelled with Python lists whose index 0 is the head:
`xs ::= x` is `xs = [x] + xs` and `xs :::= ys` is `xs = ys + xs` (an assignment
operator `l op= r` means `l = l.op(r)`).  Node identity is object identity, as
for Scala classes without `equals`.

Parity status: pinned by the reference's own known answers (re2post strings,
`follows`, `num`, removeBorderNulls, the 30 must-parse regexes, matchSA result
counts; tests/test_oracle_regex.py).  One reference vector is stale against the
reference's own code (T/REParser.scala:27-31 expects a stray ']' token that
re2.scala:96-118,154-155 cannot produce); this file follows the code.
"""

MIN_CHAR = 2      # re2.scala:22
MAX_CHAR = 255    # re2.scala:23


class Re2PostSyntax(Exception):
    """`throw new Exception("re2post syntax")`, re2.scala:84,87,109,133,141,159,174"""


class MatchError(Exception):
    """scala.MatchError thrown by ReTree.apply for operand shapes it has no case
    for, retree.scala:235-238,291-294 (and postProcess, :448-481)."""


# ------------------------------------------------------------------ postfix tokens
class PostPoint:
    pass


class CharPoint(PostPoint):           # re2.scala:25-27
    def __init__(self, c):
        self.c = c

    def __str__(self):
        return self.c


class IntervalPoint(PostPoint):       # re2.scala:28-30
    def __init__(self, start, end):
        self.start, self.end = start, end     # ints (Char codes)

    def __str__(self):
        if self.start == MIN_CHAR and self.end == MAX_CHAR:
            return "."
        return "[%c-%c]" % (self.start, self.end)


class AltPoint(PostPoint):            # re2.scala:31-33
    def __init__(self, alts):
        self.alts = alts              # list of 1-char strings, newest first

    def __str__(self):
        return "[" + "".join(reversed(self.alts)) + "]"


class ConcatPoint(PostPoint):
    def __str__(self):
        return "·"


class StarPoint(PostPoint):
    def __str__(self):
        return "*"


class QuestionPoint(PostPoint):
    def __str__(self):
        return "?"


class PlusPoint(PostPoint):
    def __str__(self):
        return "+"


class OrPoint(PostPoint):
    def __str__(self):
        return "|"


def re2post(s, lineOnly=False):
    """REParser.re2post, re2.scala:50-185."""
    l = len(s)
    st = {"natom": 0, "nalt": 0}
    dst = []                 # built newest-first (`dst ::= x`), reversed at the end
    stack = []               # Stack[Paren(nalt, natom)]

    def emit(p):
        dst.insert(0, p)

    def processChar(c, quoted):                       # :60-75
        if st["natom"] > 1:
            st["natom"] -= 1
            emit(ConcatPoint())
        if quoted:
            if c == "w":
                emit(IntervalPoint(ord("A"), ord("z")))
            elif c == "d":
                emit(IntervalPoint(ord("0"), ord("9")))
            else:
                emit(CharPoint(c))
        else:
            if c == ".":
                emit(IntervalPoint(0x20, MAX_CHAR) if lineOnly else IntervalPoint(MIN_CHAR, MAX_CHAR))
            else:
                emit(CharPoint(c))
        st["natom"] += 1

    def processAltChar(i):                            # :76-119
        alts = []
        quoted = False
        end = False
        interval = False

        def pc(c):
            nonlocal alts, interval
            if interval:
                if not alts:
                    raise Re2PostSyntax("re2post syntax")
                cAlt = ord(alts[0]) + 1
                eAlt = ord(c)
                if cAlt > eAlt:
                    raise Re2PostSyntax("re2post syntax")
                while cAlt <= eAlt:
                    alts = [chr(cAlt)] + alts
                    cAlt += 1
                interval = False
            else:
                alts = [c] + alts

        while i < l and not end:
            c = s[i]
            if quoted:
                pc(c)
                quoted = False
            elif c == "\\":
                quoted = True
            elif c == "-":
                interval = True
            elif c == "]":
                end = True
            else:
                pc(c)
            i += 1
        if not end or interval:
            raise Re2PostSyntax("re2post syntax")
        if st["natom"] > 1:
            st["natom"] -= 1
            emit(ConcatPoint())
        emit(AltPoint(alts))
        st["natom"] += 1
        return i

    i = 0
    quoted = False
    while i < l:
        c = s[i]
        if not quoted:
            if c == "(":                              # :124-131
                if st["natom"] > 1:
                    st["natom"] -= 1
                    emit(ConcatPoint())
                stack.append((st["nalt"], st["natom"]))
                st["nalt"] = 0
                st["natom"] = 0
            elif c == "|":                            # :132-139
                if st["natom"] == 0:
                    raise Re2PostSyntax("re2post syntax")
                st["natom"] -= 1
                while st["natom"] > 0:
                    emit(ConcatPoint())
                    st["natom"] -= 1
                st["nalt"] += 1
            elif c == ")":                            # :140-153
                if st["natom"] == 0:
                    raise Re2PostSyntax("re2post syntax")
                st["natom"] -= 1
                while st["natom"] > 0:
                    emit(ConcatPoint())
                    st["natom"] -= 1
                while st["nalt"] > 0:
                    emit(OrPoint())
                    st["nalt"] -= 1
                if not stack:
                    raise Re2PostSyntax("pop of empty stack")   # NoSuchElementException there
                nalt, natom = stack.pop()
                st["nalt"] = nalt
                st["natom"] = natom + 1
            elif c == "[":                            # :154-155
                i = processAltChar(i + 1) - 1
            elif c == "\\":
                quoted = True
            elif c in "*+?":                          # :158-164
                if st["natom"] == 0:
                    raise Re2PostSyntax("re2post syntax")
                emit({"*": StarPoint, "+": PlusPoint, "?": QuestionPoint}[c]())
            else:
                processChar(c, False)
        else:
            processChar(c, True)
            quoted = False
        i += 1
    if stack:
        raise Re2PostSyntax("re2post syntax")
    st["natom"] -= 1
    while st["natom"] > 0:
        emit(ConcatPoint())
        st["natom"] -= 1
    while st["nalt"] > 0:
        emit(OrPoint())
        st["nalt"] -= 1
    return list(reversed(dst))


def re2poststr(s):
    """re2.scala:187"""
    return "".join(str(p) for p in re2post(s))


def post2re(s):
    """REParser.post2re, re2.scala:188-205: '.' is the concat token here."""
    m = {"*": StarPoint, ".": ConcatPoint, "|": OrPoint, "?": QuestionPoint, "+": PlusPoint}
    return [m[c]() if c in m else CharPoint(c) for c in s]


# ----------------------------------------------------------------------- tree
class _Root:
    childs = []

    def __repr__(self):
        return "<<<ROOT>>>"


RootNode = _Root()


def _after(childs, me):
    """x.childs.dropWhile(_ != this).tail"""
    k = 0
    while k < len(childs) and childs[k] is not me:
        k += 1
    return childs[k + 1:]


class Node:
    def __init__(self):
        self.childs = []
        self.parent = RootNode

    # retree.scala:14-38
    @property
    def follows(self):
        p = self.parent
        if p is RootNode:
            return []
        if isinstance(p, OrNode):
            return p.follows
        if isinstance(p, FollowNode):
            last = _after(p.childs, self)
            if last:
                ret = last[0].firsts
                if last[0].isNull:
                    last = last[1:]
                    while last and last[0].isNull:
                        ret = last[0].firsts + ret
                        last = last[1:]
                    if last:
                        ret = last[0].firsts + ret
                return ret
            return p.follows
        if isinstance(p, StarNode):
            return self.firsts + p.follows
        if isinstance(p, QuestionNode):
            return p.follows
        return []

    # retree.scala:40-50
    @property
    def isLast(self):
        p = self.parent
        if p is RootNode:
            return True
        if isinstance(p, OrNode):
            return p.isLast
        if isinstance(p, UnarOpNode):
            return p.isLast
        if isinstance(p, FollowNode):
            last = _after(p.childs, self)
            if not last or all(x.isNull for x in last):
                return p.isLast
            return False
        return True


class CharNode(Node):                 # retree.scala:62-68
    def __init__(self, c):
        super().__init__()
        self.c = c                    # int char code
        self.num = 0

    isNull = False

    @property
    def firsts(self):
        return [self]

    def __repr__(self):
        return chr(self.c) if 0x20 <= self.c < 0x7F else "%02x" % self.c


class UnarOpNode(Node):               # retree.scala:72-79
    def append(self, n):
        assert not self.childs
        n.parent = self
        self.childs = [n] + self.childs

    @property
    def firsts(self):
        return [f for ch in self.childs for f in ch.firsts]


class StarNode(UnarOpNode):
    isNull = True

    def __repr__(self):
        return "*[" + ",".join(map(repr, self.childs)) + "]"


class QuestionNode(UnarOpNode):
    isNull = True

    def __repr__(self):
        return "?[" + ",".join(map(repr, self.childs)) + "]"


class PlusNode(UnarOpNode):
    @property
    def isNull(self):
        return all(ch.isNull for ch in self.childs)

    def __repr__(self):
        return "+[" + ",".join(map(repr, self.childs)) + "]"


class OrNode(Node):                   # retree.scala:96-112
    @property
    def firsts(self):
        return [f for ch in self.childs for f in ch.firsts]

    @property
    def isNull(self):
        return any(ch.isNull for ch in self.childs)

    def append(self, n):
        if isinstance(n, OrNode):
            for ch in n.childs:
                ch.parent = self
            self.childs = n.childs + self.childs
        else:
            n.parent = self
            self.childs = [n] + self.childs

    def __repr__(self):
        return "O[" + "|".join(map(repr, self.childs)) + "]"


class FollowNode(Node):               # retree.scala:114-133
    @property
    def firsts(self):
        p = self.childs
        ret = []
        while p and p[0].isNull:
            ret = p[0].firsts + ret
            p = p[1:]
        if p:
            ret = p[0].firsts + ret
        return ret

    @property
    def isNull(self):
        return all(ch.isNull for ch in self.childs)

    def append(self, n):
        n.parent = self
        self.childs = [n] + self.childs

    def __repr__(self):
        return "F[" + ",".join(map(repr, self.childs)) + "]"


def _kind(n):
    if isinstance(n, CharNode):
        return "C"
    if isinstance(n, UnarOpNode):
        return "U"
    if isinstance(n, OrNode):
        return "O"
    if isinstance(n, FollowNode):
        return "F"
    return "?"


# OrPoint operand shapes, retree.scala:184-239.  "into2": x2.append(a1); "new": fresh OrNode.
_OR_CASES = {
    ("C", "O"): "into2", ("U", "O"): "into2", ("F", "O"): "into2", ("O", "O"): "into2",
    ("F", "F"): "new", ("C", "C"): "new", ("U", "F"): "new", ("C", "F"): "new",
    ("U", "C"): "new", ("F", "C"): "new", ("U", "U"): "new",
}
# ConcatPoint operand shapes, retree.scala:243-295.  "new": fresh FollowNode; "into1": x1.append(x2).
_CONCAT_CASES = {
    ("O", "O"): "new", ("C", "O"): "new", ("C", "C"): "new", ("U", "C"): "new",
    ("U", "O"): "new", ("C", "U"): "new", ("U", "U"): "new",
    ("F", "C"): "into1", ("F", "O"): "into1", ("F", "U"): "into1",
}


def _unary(a1, kind):
    """PlusPoint / StarPoint / QuestionPoint, retree.scala:296-337."""
    if kind == "+":
        if isinstance(a1, StarNode):
            return a1
        if isinstance(a1, (QuestionNode, PlusNode)):
            el = StarNode()
            el.append(a1.childs[0])
            return el
        el = PlusNode()
        el.append(a1)
        return el
    if kind == "*":
        if isinstance(a1, StarNode):
            return a1
        if isinstance(a1, (QuestionNode, PlusNode)):
            el = StarNode()
            el.append(a1.childs[0])
            return el
        el = StarNode()
        el.append(a1)
        return el
    # '?'
    if isinstance(a1, QuestionNode):
        el = QuestionNode()
        el.append(a1.childs[0])
        return el
    if isinstance(a1, StarNode):
        return a1
    if isinstance(a1, PlusNode):
        el = StarNode()
        el.append(a1.childs[0])
        return el
    el = QuestionNode()
    el.append(a1)
    return el


def postProcess(r):
    """retree.scala:439-482: rebuilds the tree (undoing the prepend order of
    `childs`) and rewrites x+ as x x*."""
    def processChild(newL, oldC):
        if isinstance(oldC, PlusNode):
            a1 = postProcess(oldC.childs[0])
            a2 = StarNode()
            a2.append(postProcess(oldC.childs[0]))
            return [a1, a2] + newL
        return [postProcess(oldC)] + newL

    if isinstance(r, CharNode):
        return CharNode(r.c)
    for cls in (FollowNode, QuestionNode, OrNode, StarNode):
        if type(r) is cls:
            nc = cls()
            for ch in r.childs:
                nc.childs = processChild(nc.childs, ch)
            return nc
    raise MatchError("postProcess: %r" % (r,))


def removeBorderNulls(a1):
    """retree.scala:371-385"""
    n = FollowNode()
    p = a1.childs
    while p and p[0].isNull:
        p = p[1:]
    p = list(reversed(p))
    while p and p[0].isNull:
        p = p[1:]
    while p:
        n.append(p[0])
        p = p[1:]
    return n


def setParents(r, parent=RootNode):
    """retree.scala:386-391"""
    r.parent = parent
    for ch in r.childs:
        setParents(ch, r)


def setNums(r):
    """retree.scala:393-423"""
    def _setNums(r, _idx):
        idx = [_idx]

        def __setNums(r):
            if isinstance(r, OrNode):
                nidx = idx[0]
                for ch in r.childs:
                    if isinstance(ch, CharNode):
                        ch.num = idx[0]
                        nidx = max(nidx, idx[0] + 1)
                    else:
                        nidx = max(nidx, _setNums(ch, idx[0]))
                idx[0] = nidx
            else:
                for ch in r.childs:
                    if isinstance(ch, CharNode):
                        ch.num = idx[0]
                        idx[0] += 1
                    else:
                        __setNums(ch)
            return idx[0]

        return __setNums(r)

    return _setNums(r, 1)


class ReTree:
    """ReTree.apply + class ReTree, retree.scala:156-370,485-653."""

    def __init__(self, postfix, removeNulls=True):
        args = []                                   # mutable.Stack: top is the end
        for c in postfix:
            if isinstance(c, IntervalPoint):        # :165-173, END-EXCLUSIVE
                el = OrNode()
                for j in range(c.start, c.end):
                    el.append(CharNode(j))
                args.append(el)
            elif isinstance(c, AltPoint):           # :174-179
                el = OrNode()
                for ch in c.alts:
                    el.append(CharNode(ord(ch)))
                args.append(el)
            elif isinstance(c, CharPoint):
                args.append(CharNode(ord(c.c)))
            elif isinstance(c, OrPoint):            # :181-239
                a2 = self._pop(args)
                a1 = self._pop(args)
                act = _OR_CASES.get((_kind(a1), _kind(a2)))
                if act == "into2":
                    a2.append(a1)
                    args.append(a2)
                elif act == "new":
                    el = OrNode()
                    el.append(a1)
                    el.append(a2)
                    args.append(el)
                else:
                    raise MatchError("OrPoint have no match for a1=%r a2=%r" % (a1, a2))
            elif isinstance(c, ConcatPoint):        # :240-295
                a2 = self._pop(args)
                a1 = self._pop(args)
                act = _CONCAT_CASES.get((_kind(a1), _kind(a2)))
                if act == "new":
                    el = FollowNode()
                    el.append(a1)
                    el.append(a2)
                    args.append(el)
                elif act == "into1":
                    a1.append(a2)
                    args.append(a1)
                else:
                    raise MatchError("ConcatPoint have no match for a1=%r a2=%r" % (a1, a2))
            elif isinstance(c, PlusPoint):
                args.append(_unary(self._pop(args), "+"))
            elif isinstance(c, StarPoint):
                args.append(_unary(self._pop(args), "*"))
            elif isinstance(c, QuestionPoint):
                args.append(_unary(self._pop(args), "?"))
            else:
                raise MatchError(repr(c))
        a0 = self._pop(args)
        if isinstance(a0, FollowNode):              # :345-360
            a2 = a0
        elif isinstance(a0, (OrNode, UnarOpNode, CharNode)):
            a2 = FollowNode()
            a2.append(a0)
        else:
            raise Exception("Nonfollow Stack End%r" % (a0,))
        a1 = postProcess(a2)
        a3 = removeBorderNulls(a1) if removeNulls else a1
        setParents(a3, RootNode)
        setNums(a3)
        self.root = a3

    @staticmethod
    def _pop(args):
        if not args:
            raise Exception("NoSuchElementException: pop of empty stack")
        return args.pop()

    # ------------------------------------------------------------ flat tables
    def char_nodes(self):
        """All CharNodes in tree order (depth first, childs head first)."""
        out = []

        def walk(n):
            if isinstance(n, CharNode):
                out.append(n)
            for ch in n.childs:
                walk(ch)

        walk(self.root)
        return out

    def tables(self):
        """Flatten to what _matchSA touches: per CharNode (c, num, isLast,
        follows as index lists keeping order and multiplicity) + root.firsts."""
        nodes = self.char_nodes()
        idx = {id(n): k for k, n in enumerate(nodes)}
        return {
            "c": [n.c for n in nodes],
            "num": [n.num for n in nodes],
            "isLast": [bool(n.isLast) for n in nodes],
            "follows": [[idx[id(f)] for f in n.follows] for n in nodes],
            "firsts": [idx[id(f)] for f in self.root.firsts],
        }

    # ----------------------------------------------------------------- search
    def matchSA(self, sa, maxBranching=1024, maxIterations=1000):
        """ReTree.matchSA, retree.scala:570-617: returns the first pass's `ret`
        (list of (len, sp, ep), newest first); the exploratory restarts at
        :578-614 never change the returned value and are not restated."""
        ret, _front, _pops = self._matchSA(sa, maxBranching, maxIterations)
        return ret

    def _matchSA(self, sa, maxBranching=1024, maxIterations=1000):
        """Pure-Python _matchSA (retree.scala:618-653) for small cases; `sa`
        needs .n and .getPrevRange(sp, ep, c) -> (sp1, ep1) | None."""
        t = self.tables()
        num = t["num"]
        heap = [None]                       # scala PriorityQueue, 1-based (see fmx_oracle.c)

        def lt(x, y):                       # StatePoint.compare, :564
            return num[x[3]] > num[y[3]]

        def push(e):
            heap.append(e)
            k = len(heap) - 1
            while k > 1 and lt(heap[k // 2], heap[k]):
                heap[k], heap[k // 2] = heap[k // 2], heap[k]
                k //= 2

        def pop():
            last = len(heap) - 1
            heap[1], heap[last] = heap[last], heap[1]
            n = last - 1
            k = 1
            while n >= 2 * k:
                j = 2 * k
                if j < n and lt(heap[j], heap[j + 1]):
                    j += 1
                if not lt(heap[k], heap[j]):
                    break
                heap[k], heap[j] = heap[j], heap[k]
                k = j
            return heap.pop()

        for f in t["firsts"]:
            push((0, 0, sa.n, f))
        ret = []
        i = 1
        pops = 0
        while len(heap) > 1 and (len(heap) - 1) < maxBranching and (maxIterations == 0 or i < maxIterations):
            ln, sp, ep, s = pop()
            pops += 1
            r = sa.getPrevRange(sp, ep, t["c"][s])
            if r is not None:
                sp1, ep1 = r
                if t["isLast"][s]:
                    ret = [(ln + 1, sp1, ep1)] + ret
                else:
                    for f in t["follows"][s]:
                        push((ln + 1, sp1, ep1, f))
            i += 1
        return ret, heap[1:], pops

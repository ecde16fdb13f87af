"""CPU restatement of findex's two other SA-interval regex engines (TEST INFRASTRUCTURE ONLY).

  REParser.createNFA / matchSA ... src/main/scala/org/fmindex/re2/re2.scala:207-334,437-484,568-693
  DFA.compileBuckets / matchSA ... src/main/scala/org/fmindex/dfa.scala:114-289

Both reference engines keep their frontier in collections whose iteration order is not part of
any contract (immutable Set of case-class instances, a priority queue with ties); their results
are therefore compared as multisets.  Pinned by the reference's own vectors:
T/REParser.scala:219-234,292-307 and T/dfa.scala:110-122 (tests/test_oracle_engines.py).
"""
from . import retree as R


# ------------------------------------------------------------------ Thompson NFA (re2.scala)
class LinkState:                      # re2.scala:228-231
    def __init__(self, s=None):
        self.s = s


class BaseState:
    def outStates(self):              # re2.scala:213-224
        seen = []

        def add(s):
            if s is None or any(s is x for x in seen):
                return
            if isinstance(s, SplitState):
                seen.append(s)
                add(s.out1.s)
                if s.out2 is not None:
                    add(s.out2.s)
            else:
                seen.append(s)

        add(self)
        return [s for s in seen if not isinstance(s, SplitState)]


class ConstState(BaseState):          # re2.scala:237-241
    def __init__(self, c):
        self.c = c
        self.out = LinkState()

    @property
    def next(self):
        return self.out.s


class IntervalState(BaseState):       # re2.scala:242-250
    def __init__(self, start, end):
        self.start, self.end = start, end
        self.out = LinkState()

    @property
    def next(self):
        return self.out.s


class SplitState(BaseState):          # re2.scala:251-253
    def __init__(self, out1, out2):
        self.out1, self.out2 = out1, out2


class _Match(BaseState):
    pass


MatchState = _Match()                 # re2.scala:254-256


def createNFA(postfix):
    """REParser.createNFA, re2.scala:264-334."""
    s0 = []

    def pop():
        if not s0:
            raise R.MatchError("createNFA: pop of an empty stack")
        return s0.pop()

    def patch(frag, s):
        for l in frag[1]:
            l.s = s

    for c in postfix:
        if isinstance(c, R.QuestionPoint):
            e = pop()
            op = LinkState()
            ns = SplitState(LinkState(e[0]), op)
            s0.append((ns, [op] + e[1]))
        elif isinstance(c, R.StarPoint):
            e = pop()
            op = LinkState()
            ns = SplitState(LinkState(e[0]), op)
            patch(e, ns)
            s0.append((ns, [op]))
        elif isinstance(c, R.PlusPoint):
            e = pop()
            op = LinkState()
            ns = SplitState(LinkState(e[0]), op)
            patch(e, ns)
            s0.append((e[0], [op]))
        elif isinstance(c, R.ConcatPoint):
            e2 = pop()
            e1 = pop()
            patch(e1, e2[0])
            s0.append((e1[0], e2[1]))
        elif isinstance(c, R.OrPoint):
            e2 = pop()
            e1 = pop()
            ns = SplitState(LinkState(e1[0]), LinkState(e2[0]))
            s0.append((ns, e1[1] + e2[1]))
        elif isinstance(c, R.CharPoint):
            ns = ConstState(ord(c.c))
            s0.append((ns, [ns.out]))
        elif isinstance(c, R.IntervalPoint):
            ns = IntervalState(c.start, c.end)
            s0.append((ns, [ns.out]))
        else:
            raise R.MatchError("createNFA has no case for %r" % (c,))
    e0 = pop()
    patch(e0, MatchState)
    return e0[0]


def nfa_matchSA(nfa, sa, maxIterations=0, maxLength=0, tie="first"):
    """REParser.matchSA, re2.scala:568-693 -> list of (len, sp, ep) (order not meaningful).
    StatePoint = (len, state, [intervals]); the queue pops the largest len (:446).  Which of several elements
    of equal len comes out is not defined by the reference's source: the start states come from an immutable
    Set of objects hashed by identity (`liststates`, :570-578) and the queue breaks ties by its internal layout.
    `tie` picks one legal order ("first" / "last" element of maximal len): with maxIterations = 0 the result
    multiset does not depend on it, with a binding maxIterations it does (tests/test_oracle_engines.py)."""
    front = [(0, s, [(0, sa.n)]) for s in nfa.outStates()]
    results = []
    i = 0
    while front and (maxIterations == 0 or i < maxIterations):
        top = max(f[0] for f in front)
        cand = [j for j in range(len(front)) if front[j][0] == top]
        k = cand[0] if tie == "first" else cand[-1]
        ln, state, intervals = front.pop(k)
        if isinstance(state, ConstState):
            chars = [state.c]
        elif isinstance(state, IntervalState):
            chars = range(state.start, state.end)            # `start until end`, :472
        else:
            raise R.MatchError("StatePoint.expand has no case for %r" % (state,))    # :457-482
        ret = []
        for sp, ep in intervals:
            for ch in chars:
                r = sa.getPrevRange(sp, ep, ch)
                if r is not None:
                    ret.insert(0, r)
        new = [(ln + 1, ns, ret) for ns in state.next.outStates()] if ret else []
        for nl, ns, ivs in new:
            if ns is MatchState:
                for sp, ep in ivs:
                    results.insert(0, (nl, sp, ep))
            elif maxLength == 0 or nl < maxLength:
                front.append((nl, ns, ivs))
        i += 1
    return results


# ------------------------------------------------------------------ DFA (dfa.scala)
class DFA:
    """class DFA(nstates, nchars), dfa.scala:114-289, with a transition table filled by addLink."""

    def __init__(self, nstates, nchars=256):
        self.moves = [[-1] * nchars for _ in range(nstates)]
        self.nchars = nchars
        self.finishStates = set()
        self.buckets = None

    def addLink(self, frm, to, ch):
        self.moves[frm][ch] = to

    def compileBuckets(self):
        """dfa.scala:190-213 -> per state a list of ('char', target, c) / ('bucket', target, c1, c2)."""
        bkt = []
        for row in self.moves:
            acts, last, start = [], -1, -1
            for j, v in enumerate(row):
                if last != v:
                    if last != -1:
                        acts.append(self._action(last, start, j - 1))
                    start, last = j, v
            if last != -1:
                acts.append(self._action(last, start, self.nchars - 1))
            bkt.append(acts)
        self.buckets = bkt

    @staticmethod
    def _action(state, c1, c2):                       # DFAAction.create, :177-185
        return ("char", state, c1) if c1 == c2 else ("bucket", state, c1, c2)

    def matchSA(self, sa, cap=500, take="first"):
        """dfa.scala:261-289 -> list of (len, sp, ep).  The reference's frontier is an immutable Set whose
        head/tail order is the hash trie's (no contract); `take` picks a legal order ("first" = oldest, "last" =
        newest).  Results do not depend on it unless the 500-iteration cap (:268) binds."""
        front = [(0, 0, 0, sa.n)]                     # StatePoint(state, len, sp, ep); a Set there
        visited = set()
        results = []
        i = 0
        while front and i < cap:
            st = front.pop(0 if take == "first" else -1)
            visited.add(st)
            state, ln, sp, ep = st
            new = []
            for act in self.buckets[state]:           # StatePoint.expand, :242-259
                if act[0] == "char":
                    r = sa.getPrevRange(sp, ep, act[2])
                    if r is not None:
                        new.append((act[1], ln + 1, r[0], r[1]))
            if state in self.finishStates:
                results.insert(0, (ln, sp, ep))
            for s in new:
                if s not in visited and s not in front:
                    front.append(s)
            i += 1
        return results

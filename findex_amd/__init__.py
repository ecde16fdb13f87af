"""findex_amd -- MI355X-native FM-index backward search behind findex's SuffixAlgo API.

The compute lives in libfmx.so (hand-written HIP for gfx950, C ABI in include/fmx.h); this
package is the thin host-side mirror of the reference's Scala interface for that path:

    reference (Scala)                              here
    NaiveFMSearcher(filename, bigEndian)           HipFMSearcher(filename, bigEndian=True)
      .n .cf .occ .search .getPrevRange ...          same names, same meaning
    REParser.re2post(str, lineOnly)                REParser.re2post(str, lineOnly=False)
    ReTree(postfix).matchSA(sa)                    ReTree(postfix).matchSA(sa)
    SAResult(sa,len,sp,ep)                         SAResult

There is no CPU fallback: importing works anywhere, but every compute call needs the built
library and a HIP device and fails loudly otherwise.
"""
from ._lib import FmxError, MatchError, Re2PostSyntax, LIB_PATH, load  # noqa: F401
from .searcher import HipFMSearcher  # noqa: F401
from .regex import DFA, NFA, REParser, ReTree, SAResult, CompiledRegexes  # noqa: F401



def set_layout(name):
    """fmx_config_set("layout", ...): "auto" | "onehot" | "bytes" for indexes opened afterwards."""
    from . import _lib
    _lib.check(_lib.load().fmx_config_set(b"layout", name.encode()))


def set_checkpoints(name):
    """fmx_config_set("checkpoints", ...): "auto" | "superblock" (bytes layout, indexes opened afterwards)."""
    from . import _lib
    _lib.check(_lib.load().fmx_config_set(b"checkpoints", name.encode()))


def set_ktab(name):
    """fmx_config_set("ktab", ...): "auto" | "off" (handles that have not searched yet)."""
    from . import _lib
    _lib.check(_lib.load().fmx_config_set(b"ktab", name.encode()))


def set_jump(name):
    """fmx_config_set("jump", ...): "auto" | "jumps" | "rows3" | "rows" | "off" -- the row jump table, the three-step row table and
    the row table (handles that have not searched yet)."""
    from . import _lib
    _lib.check(_lib.load().fmx_config_set(b"jump", name.encode()))


def set_pipeline(name):
    """fmx_config_set("pipeline", ...): "off" | "on" -- large host batches in page-locked memory as overlapping chunks."""
    from . import _lib
    _lib.check(_lib.load().fmx_config_set(b"pipeline", name.encode()))


def config_set(key, value):
    """fmx_config_set(key, value): any of the library's process-wide options (include/fmx.h)."""
    from . import _lib
    _lib.check(_lib.load().fmx_config_set(key.encode(), str(value).encode()))


__all__ = ["config_set", "set_layout", "set_checkpoints", "set_ktab", "set_jump", "set_pipeline", "HipFMSearcher", "REParser", "ReTree", "SAResult", "NFA", "DFA", "CompiledRegexes", "FmxError", "MatchError", "Re2PostSyntax"]

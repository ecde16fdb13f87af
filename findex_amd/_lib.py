"""ctypes binding of libfmx.so (include/fmx.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback in this package."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMX_LIB") or os.path.join(_HERE, "lib", "libfmx.so")     # FMX_LIB: A/B runs of two builds

FMX_OK = 0
ERR_NAMES = {1: "FMX_ERR_IO", 2: "FMX_ERR_FORMAT", 3: "FMX_ERR_ARG", 4: "FMX_ERR_NOMEM", 5: "FMX_ERR_HIP",
             6: "FMX_ERR_UNSUPPORTED", 7: "FMX_ERR_SYNTAX", 8: "FMX_ERR_MATCH", 9: "FMX_ERR_OVERFLOW"}


class FmxError(Exception):
    """A non-zero status from libfmx (the Scala adapter would rethrow it as Exception)."""

    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


class Re2PostSyntax(FmxError):
    """`throw new Exception("re2post syntax")`, re2/re2.scala"""


class MatchError(FmxError):
    """scala.MatchError from ReTree.apply, re2/retree.scala:235-238,291-294"""


FMX_MATCH_FRONTIER, FMX_MATCH_REFERENCE = 0, 1


class fmx_limits(ctypes.Structure):
    _fields_ = [("max_steps", ctypes.c_uint32), ("mode", ctypes.c_uint32), ("max_frontier", ctypes.c_uint64),
                ("max_branching", ctypes.c_uint32), ("max_iterations", ctypes.c_uint32)]


class fmx_result(ctypes.Structure):
    _fields_ = [("regex", ctypes.c_uint32), ("len", ctypes.c_uint32), ("sp", ctypes.c_uint64),
                ("ep", ctypes.c_uint64)]


class fmx_search_opts(ctypes.Structure):
    _fields_ = [("fixed_len", ctypes.c_uint32), ("packed", ctypes.c_uint32), ("escape_cap", ctypes.c_uint64)]


class fmx_stats_t(ctypes.Structure):
    _fields_ = [("rank_queries", ctypes.c_uint64), ("backward_steps", ctypes.c_uint64),
                ("launches", ctypes.c_uint64), ("last_kernel_ms", ctypes.c_double),
                ("index_bytes", ctypes.c_uint64), ("n_blocks", ctypes.c_uint64), ("n_symbols", ctypes.c_uint32),
                ("block_bytes", ctypes.c_uint32), ("build_ms", ctypes.c_double), ("layout", ctypes.c_uint32),
                ("search_residency", ctypes.c_uint32), ("search_requests", ctypes.c_uint64),
                ("frontier_requests", ctypes.c_uint64), ("frontier_elements", ctypes.c_uint64),
                ("frontier_queue_reads", ctypes.c_uint64), ("frontier_queue_writes", ctypes.c_uint64),
                ("frontier_results", ctypes.c_uint64), ("frontier_records", ctypes.c_uint64),
                ("ktab_lookups", ctypes.c_uint64), ("ktab_k", ctypes.c_uint32), ("jump_chars", ctypes.c_uint32),
                ("tables_build_ms", ctypes.c_double), ("jump_lookups", ctypes.c_uint64), ("jump_bytes", ctypes.c_uint64),
                ("row_lookups", ctypes.c_uint64), ("row_bytes", ctypes.c_uint64),
                ("peak_table_build_bytes", ctypes.c_uint64), ("patterns_seen", ctypes.c_uint64),
                ("tables_held_bytes", ctypes.c_uint64), ("table_budget_bytes", ctypes.c_uint64),
                ("hbm_free_after_tables", ctypes.c_uint64), ("tables_alloc_ms", ctypes.c_double)]


# name -> (restype, argtypes); every symbol include/fmx.h declares
_vp, _u64, _i32, _u32, _sz, _cp = (ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t,
                                   ctypes.c_char_p)
_P = ctypes.POINTER
SYMBOLS = {
    "fmx_last_error": (_cp, []),
    "fmx_abi_version": (_i32, []),
    "fmx_config_set": (_i32, [_cp, _cp]),
    "fmx_device_count": (_i32, [_P(_i32)]),
    "fmx_host_alloc": (_i32, [_sz, _P(_vp)]),
    "fmx_host_free": (_i32, [_vp]),
    "fmx_open": (_i32, [_cp, _cp, _i32, _i32, _P(_vp)]),
    "fmx_open_mem": (_i32, [_vp, _u64, _u64, _vp, _i32, _P(_vp)]),
    "fmx_open_dev": (_i32, [_vp, _u64, _u64, _vp, _i32, _vp, _P(_vp)]),
    "fmx_open_block": (_i32, [_vp, _u64, _vp, _u64, _i32, _P(_vp)]),
    "fmx_close": (_i32, [_vp]),
    "fmx_index_config_set": (_i32, [_vp, _cp, _cp]),
    "fmx_prepare": (_i32, [_vp, _u32]),
    "fmx_prepare_ex": (_i32, [_vp, _u32, _u64]),
    "fmx_drop_tables": (_i32, [_vp, _u32]),
    "fmx_n": (_i32, [_vp, _P(_u64)]),
    "fmx_eof": (_i32, [_vp, _P(_u64)]),
    "fmx_cf": (_i32, [_vp, _i32, _P(_u64)]),
    "fmx_counts": (_i32, [_vp, _vp]),
    "fmx_device": (_i32, [_vp, _P(_i32)]),
    "fmx_occ_batch": (_i32, [_vp, _vp, _vp, _vp, _sz]),
    "fmx_occ_batch_dev": (_i32, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "fmx_search_batch": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "fmx_search_batch_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fmx_search_batch_ex": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _P(fmx_search_opts)]),
    "fmx_search_batch_ex_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _P(fmx_search_opts), _vp]),
    "fmx_packed_words": (_sz, [_sz, _sz]),
    "fmx_pack_intervals_dev": (_i32, [_vp, _vp, _vp, _sz, _sz, _vp, _vp]),
    "fmx_unpack_intervals_dev": (_i32, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "fmx_unpack_intervals": (_i32, [_vp, _sz, _sz, _vp, _vp]),
    "fmx_prev_range_batch": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "fmx_prev_range_batch_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fmx_interval_prev_range": (_i32, [_vp, _u64, _u64, _i32, _i32, _vp, _vp, _vp, _P(_sz)]),
    "fmx_lf_walk_batch": (_i32, [_vp, _vp, _sz, _u32, _vp, _vp]),
    "fmx_lf_walk_batch_dev": (_i32, [_vp, _vp, _sz, _u32, _vp, _vp, _vp]),
    "fmx_psi_batch": (_i32, [_vp, _vp, _vp, _sz]),
    "fmx_psi_batch_dev": (_i32, [_vp, _vp, _vp, _sz, _vp]),
    "fmx_next_substr_batch_dev": (_i32, [_vp, _vp, _sz, _u32, _vp, _vp, _vp]),
    "fmx_next_substr": (_i32, [_vp, _u64, _u32, _vp, _P(_u32)]),
    "fmx_prev_substr": (_i32, [_vp, _u64, _u32, _vp]),
    "fmx_next_substr_batch": (_i32, [_vp, _vp, _sz, _u32, _vp, _vp]),
    "fmx_search_batch_multi": (_i32, [_vp, _sz, _vp, _vp, _vp, _vp, _sz]),
    "fmx_extract": (_i32, [_vp, _u64, _u32, _i32, _vp, _P(_u32)]),
    "fmx_write_fm": (_i32, [_vp, _cp]),
    "fmx_occ_host": (_i32, [_vp, _i32, ctypes.c_int64, _P(_u64)]),
    "fmx_calc_gaps_chain": (_i32, [_vp, _vp, _sz, _u64, _i32, _u64, _vp, _P(_sz)]),
    "fmx_regex_compile": (_i32, [_cp, _i32, _P(_vp)]),
    "fmx_regex_compile_batch": (_i32, [_vp, _sz, _i32, _vp, _vp]),
    "fmx_regex_free_batch": (_i32, [_vp, _sz]),
    "fmx_nfa_compile": (_i32, [_cp, _i32, _i32, _P(_vp)]),
    "fmx_dfa_compile": (_i32, [_vp, _u32, _u32, _vp, _P(_vp)]),
    "fmx_regex_free": (_i32, [_vp]),
    "fmx_regex_tables": (_i32, [_vp, _P(_u32), _vp, _vp, _vp, _vp, _P(_u32), _vp, _P(_u32), _vp]),
    "fmx_regex_post_string": (_i32, [_cp, _i32, _vp, _sz]),
    "fmx_regex_match_batch": (_i32, [_vp, _vp, _sz, _vp, _vp, _sz, _P(_sz), _vp]),
    "fmx_regex_batch_create": (_i32, [_vp, _vp, _sz, _P(_vp)]),
    "fmx_regex_batch_free": (_i32, [_vp]),
    "fmx_regex_batch_info": (_i32, [_vp, _P(_u64), _P(_u64), _P(_u64), _P(_u64)]),
    "fmx_regex_batch_match": (_i32, [_vp, _vp, _vp, _vp, _sz, _P(_sz), _vp]),
    "fmx_regex_batch_match_dev": (_i32, [_vp, _vp, _vp, _vp, _sz, _P(_sz), _vp]),
    "fmx_regex_batch_create_multi": (_i32, [_vp, _sz, _vp, _sz, _P(_vp)]),
    "fmx_regex_batch_free_multi": (_i32, [_vp]),
    "fmx_regex_batch_match_multi": (_i32, [_vp, _vp, _vp, _sz, _P(_sz), _vp]),
    "fmx_gather": (_i32, [_vp, _sz, _vp, _vp, _sz, _vp]),
    "fmx_comm_unique_id": (_i32, [_vp]),
    "fmx_comm_create_rank": (_i32, [_vp, _i32, _i32, _vp, _P(_vp)]),
    "fmx_comm_create_all": (_i32, [_vp, _sz, _P(_vp)]),
    "fmx_comm_info": (_i32, [_vp, _P(_i32), _P(_i32)]),
    "fmx_comm_free": (_i32, [_vp]),
    "fmx_allgather_dev": (_i32, [_vp, _vp, _vp, _sz, _vp]),
    "fmx_gather_dev": (_i32, [_vp, _vp, _vp, _sz, _i32, _vp]),
    "fmx_stats": (_i32, [_vp, _P(fmx_stats_t)]),
    "fmx_stats_reset": (_i32, [_vp]),
    "fmx_last_kernel_ms": (_i32, [_vp, _P(ctypes.c_double)]),
}

_lib = None


def load():
    """Loads libfmx.so once.  torch (if installed) is imported first so that both share the one
    HIP runtime torch ships; raises ImportError when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("findex_amd: %s is missing -- run `python -m findex_amd.build` "
                          "(there is no CPU fallback)" % LIB_PATH)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


FMX_TRUNCATED = 10


def check(rc):
    """Raises for error statuses; returns the status otherwise (FMX_OK or FMX_TRUNCATED)."""
    if rc == FMX_TRUNCATED:
        return rc
    if rc != FMX_OK:
        msg = (load().fmx_last_error() or b"").decode("utf-8", "replace")
        if rc == 7:
            raise Re2PostSyntax(rc, msg)
        if rc == 8:
            raise MatchError(rc, msg)
        raise FmxError(rc, msg)
    return rc

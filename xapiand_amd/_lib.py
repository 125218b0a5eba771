"""ctypes binding of libxgm.so — the C ABI declared in include/xgm.h.

The library is built in-tree by xapiand_amd/build.py (hipcc, gfx950).  Loading it does not need a
GPU; the search entry points do, and fail loudly (XGM_E_NO_DEVICE) without one — there is no CPU
fallback anywhere in this package.
"""
import ctypes as C
import os

XGM_MAX_TERMS = 16
XGM_MAX_TREE = 40
XGM_OP_TREE = 8
XGM_MAX_K = 1024
XGM_OK, XGM_UNSUPPORTED = 0, 1
XGM_E_INVALID, XGM_E_IO, XGM_E_NO_DEVICE, XGM_E_DEVICE, XGM_E_REVISION, XGM_E_NOMEM = -1, -2, -3, -4, -5, -6
XGM_OP_AND, XGM_OP_OR, XGM_OP_PHRASE = 1, 2, 3
XGM_OP_AND_NOT, XGM_OP_AND_MAYBE, XGM_OP_FILTER, XGM_OP_NEAR = 4, 5, 6, 7
XGM_REPLAY_BATCH_COUNT = 2
XGM_REPLAY_BATCH_FROZEN = 1         # xgm_query.replay / xgm_query_desc.replay (include/xgm.h)
XGM_KNOWN_LOWER_BOUND = 1 << 63
XGM_MATCHES_LOWER_BOUND = 1 << 63    # xgm_result_hdr.matches_exact: the count is a lower bound (include/xgm.h)
XGM_DEVICE_NONE = -1          # xgm_index_open: dictionary and statistics only (host memory), no searches
UINT64_MAX = (1 << 64) - 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XGM_LIB_PATH") or os.path.join(_HERE, "csrc", "libxgm.so")   # override: A/B builds


class RawPostings(C.Structure):
    _fields_ = [("n_terms", C.c_uint32), ("lastdocid", C.c_uint32), ("doccount", C.c_uint32),
                ("has_positions", C.c_uint32), ("total_length", C.c_uint64), ("n_postings", C.c_uint64),
                ("n_positions", C.c_uint64), ("revision", C.c_uint64),
                ("doclen", C.POINTER(C.c_uint32)), ("terms", C.POINTER(C.c_char_p)),
                ("term_len", C.POINTER(C.c_uint32)), ("df", C.POINTER(C.c_uint32)),
                ("did", C.POINTER(C.c_uint32)), ("wdf", C.POINTER(C.c_uint32)),
                ("pos_off", C.POINTER(C.c_uint64)), ("pos", C.POINTER(C.c_uint32)),
                ("doclen_lower_bound", C.c_uint32), ("wdf_upper_bound", C.c_uint32),
                ("doclen_upper_bound", C.c_uint32), ("reserved", C.c_uint32)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("vocab", C.c_uint32), ("len_lo", C.c_uint32), ("len_hi", C.c_uint32),
                ("n_docs_global", C.c_uint64), ("n_shards", C.c_uint32), ("shard", C.c_uint32),
                ("stripe_bits", C.c_uint32), ("with_positions", C.c_uint32)]


class IndexInfo(C.Structure):
    _fields_ = [("n_terms", C.c_uint32), ("lastdocid", C.c_uint32), ("doccount", C.c_uint32),
                ("has_positions", C.c_uint32), ("total_length", C.c_uint64), ("revision", C.c_uint64),
                ("n_postings", C.c_uint64), ("n_positions", C.c_uint64), ("n_blocks", C.c_uint64),
                ("device_bytes", C.c_uint64), ("payload_bytes", C.c_uint64), ("stripe_bits", C.c_uint32),
                ("block_size", C.c_uint32), ("doclen_lower_bound", C.c_uint32), ("wdf_upper_bound", C.c_uint32)]


class TreeOp(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("arity", C.c_uint8), ("term", C.c_uint16)]


class QueryDesc(C.Structure):
    _fields_ = [("op", C.c_uint32), ("n_terms", C.c_uint32), ("terms", C.c_char_p * XGM_MAX_TERMS),
                ("term_len", C.c_uint32 * XGM_MAX_TERMS), ("window", C.c_uint32), ("first", C.c_uint32),
                ("maxitems", C.c_uint32), ("check_at_least", C.c_uint32),
                ("k1", C.c_double), ("k2", C.c_double), ("k3", C.c_double), ("b", C.c_double),
                ("min_normlen", C.c_double), ("n_required", C.c_uint32), ("n_tree", C.c_uint32),
                ("wqf", C.c_uint32 * XGM_MAX_TERMS), ("tree", TreeOp * XGM_MAX_TREE), ("tree_scale", C.c_double * XGM_MAX_TREE),
                ("replay", C.c_uint32), ("reserved", C.c_uint32)]


class GlobalStats(C.Structure):
    _fields_ = [("total_length", C.c_uint64), ("collection_size", C.c_uint32),
                ("full_db_has_positions", C.c_uint32), ("termfreq", C.c_uint32 * XGM_MAX_TERMS)]


class Term(C.Structure):
    _fields_ = [("term_id", C.c_uint32), ("phrase_index", C.c_uint32), ("termweight", C.c_double)]


class Query(C.Structure):
    _fields_ = [("op", C.c_uint32), ("n_terms", C.c_uint32), ("terms", Term * XGM_MAX_TERMS),
                ("sum_prog", C.c_int8 * (2 * XGM_MAX_TERMS)), ("sum_len", C.c_uint32), ("window", C.c_uint32),
                ("phrase_active", C.c_uint32), ("len_factor", C.c_double), ("k1", C.c_double), ("b", C.c_double),
                ("min_normlen", C.c_double), ("first", C.c_uint32), ("maxitems", C.c_uint32),
                ("check_at_least", C.c_uint32), ("max_possible", C.c_double), ("req_mask", C.c_uint32), ("neg_mask", C.c_uint32),
                ("tree_len", C.c_uint32), ("n_groups", C.c_uint32), ("tree_root", C.c_uint32), ("total_subqs", C.c_uint32),
                ("group_scored", C.c_uint32), ("group_of", C.c_uint8 * XGM_MAX_TERMS), ("group_weight", C.c_double * XGM_MAX_TERMS),
                ("tree_op", C.c_uint8 * XGM_MAX_TREE), ("tree_a", C.c_uint8 * XGM_MAX_TREE), ("tree_b", C.c_uint8 * XGM_MAX_TREE),
                ("est_min", C.c_uint32), ("est_est", C.c_uint32), ("est_max", C.c_uint32), ("replay", C.c_uint32)]


class SortSpec(C.Structure):
    _fields_ = [("sort_by", C.c_uint32), ("slot", C.c_uint32), ("reverse", C.c_uint32), ("reserved", C.c_uint32)]


XGM_SORT_VALUE, XGM_SORT_VALUE_RELEVANCE, XGM_SORT_RELEVANCE_VALUE = 1, 2, 3


class Hit(C.Structure):
    _fields_ = [("docid", C.c_uint32), ("subqs_matched", C.c_uint32), ("weight", C.c_double)]


class ResultHdr(C.Structure):
    _fields_ = [("n_hits", C.c_uint32), ("max_weight_subqs_matched", C.c_uint32), ("matches_exact", C.c_uint64),
                ("max_attained", C.c_double), ("max_possible", C.c_double)]


assert C.sizeof(Hit) == 16 and C.sizeof(ResultHdr) == 32

# every symbol include/xgm.h declares: (name, restype, argtypes)
_P = C.POINTER
_API = [
    ("xgm_segment_build", C.c_int, [_P(RawPostings), C.c_uint32, C.c_char_p]),
    ("xgm_segment_build_from_file", C.c_int, [C.c_char_p, C.c_uint32, C.c_char_p]),
    ("xgm_segment_build_from_glass", C.c_int, [C.c_char_p, C.c_uint32, C.c_char_p]),
    ("xgm_glass_export_column", C.c_int, [C.c_char_p, C.c_uint32, C.c_char_p]),
    ("xgm_index_attach_column", C.c_int, [C.c_void_p, C.c_char_p]),
    ("xgm_search_sorted", C.c_int, [C.c_void_p, _P(Query), _P(SortSpec), _P(Hit), _P(C.c_uint32), _P(ResultHdr)]),
    ("xgm_search_sorted_batch", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, _P(SortSpec), C.c_uint32, _P(Hit), _P(C.c_uint32), _P(ResultHdr)]),
    ("xgm_search_sorted_spy_batch", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, _P(SortSpec), C.c_uint32, _P(Hit), _P(C.c_uint32), _P(ResultHdr),
                                     C.c_uint32, _P(C.c_uint32), C.c_uint32]),
    ("xgm_search_collapsed", C.c_int, [C.c_void_p, _P(Query), _P(SortSpec), C.c_uint32, C.c_uint32, _P(Hit), _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32),
                                       _P(ResultHdr), _P(C.c_uint64)]),
    ("xgm_search_sorted_spy", C.c_int, [C.c_void_p, _P(Query), _P(SortSpec), _P(Hit), _P(C.c_uint32), _P(ResultHdr), C.c_uint32, _P(C.c_uint32), C.c_uint32]),
    ("xgm_search_collapsed_batch", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, _P(SortSpec), C.c_uint32, C.c_uint32, C.c_uint32, _P(Hit), _P(C.c_uint32),
                                    _P(C.c_uint32), _P(C.c_uint32), _P(ResultHdr), _P(C.c_uint64)]),
    ("xgm_segment_refresh_from_glass", C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p]),
    ("xgm_glass_export_raw", C.c_int, [C.c_char_p, C.c_char_p]),
    ("xgm_glass_info", C.c_int, [C.c_char_p, _P(C.c_uint64), _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint64)]),
    ("xgm_segment_decode_term", C.c_int64, [C.c_char_p, C.c_char_p, C.c_size_t, _P(C.c_uint32), _P(C.c_uint32), C.c_uint64]),
    ("xgm_index_open", C.c_int, [C.c_char_p, C.c_int, C.c_uint64, _P(C.c_void_p)]),
    ("xgm_index_build_synthetic", C.c_int, [_P(SynthParams), C.c_int, _P(C.c_void_p)]),
    ("xgm_index_close", None, [C.c_void_p]),
    ("xgm_index_save", C.c_int, [C.c_void_p, C.c_char_p]),
    ("xgm_index_get_info", C.c_int, [C.c_void_p, _P(IndexInfo)]),
    ("xgm_index_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("xgm_lookup_term", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32)]),
    ("xgm_index_termfreqs", C.c_int, [C.c_void_p, _P(C.c_uint32), C.c_uint32]),
    ("xgm_plan_query", C.c_int, [C.c_void_p, _P(QueryDesc), _P(GlobalStats), _P(Query)]),
    ("xgm_mset_bounds", None, [_P(Query), _P(ResultHdr), _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32)]),
    ("xgm_search", C.c_int, [C.c_void_p, _P(Query), _P(Hit), _P(ResultHdr)]),
    ("xgm_search_all", C.c_int, [C.c_void_p, _P(Query), _P(Hit), C.c_uint64, _P(C.c_uint64), _P(ResultHdr)]),
    ("xgm_search_replay", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, _P(Hit), _P(ResultHdr), _P(C.c_uint64)]),
    ("xgm_debug_replay_info", C.c_int, [_P(C.c_uint64)]),
    ("xgm_debug_batch_replay_info", C.c_int, [_P(C.c_uint64)]),
    ("xgm_search_batch", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, C.c_uint32, _P(Hit), _P(ResultHdr)]),
    ("xgm_search_batch_known", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, C.c_uint32, _P(Hit), _P(ResultHdr), _P(C.c_uint64)]),
    ("xgm_search_batch_begin", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, C.c_uint32, _P(C.c_void_p)]),
    ("xgm_get_mset_batch_begin", C.c_int, [C.c_void_p, _P(QueryDesc), _P(GlobalStats), C.c_uint32, C.c_uint32, _P(C.c_void_p)]),
    ("xgm_batch_end", C.c_int, [C.c_void_p, _P(_P(Hit)), _P(_P(ResultHdr))]),
    ("xgm_batch_known", C.c_int, [C.c_void_p, _P(_P(C.c_uint64))]),
    ("xgm_batch_poll", C.c_int, [C.c_void_p]),
    ("xgm_batch_release", None, [C.c_void_p]),
    ("xgm_search_batch_device", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("xgm_get_mset_batch", C.c_int, [C.c_void_p, _P(QueryDesc), _P(GlobalStats), C.c_uint32, C.c_uint32, _P(Hit), _P(ResultHdr)]),
    ("xgm_get_mset_batch_device", C.c_int, [C.c_void_p, _P(QueryDesc), _P(GlobalStats), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("xgm_search_sharded", C.c_int, [_P(C.c_void_p), C.c_uint32, _P(QueryDesc), C.c_uint32, C.c_uint32, _P(Hit), _P(ResultHdr)]),
    ("xgm_debug_sharded_info", C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    ("xgm_merge_shards_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_uint32), C.c_void_p, C.c_void_p]),
    ("xgm_merge_shards_packed_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_uint32), C.c_void_p, C.c_void_p]),
    ("xgm_shard_record_bytes", C.c_size_t, [C.c_uint32, C.c_uint32]),
    ("xgm_index_set_profiling", C.c_int, [C.c_void_p, C.c_int]),
    ("xgm_index_set_near_colocated", C.c_int, [C.c_void_p, C.c_int]),
    ("xgm_last_kernel_ms", C.c_double, [C.c_void_p]),
    ("xgm_last_kernel_name", C.c_char_p, [C.c_void_p]),
    ("xgm_last_batch_traffic", C.c_int, [C.c_void_p, _P(C.c_uint64), C.c_uint32]),
    ("xgm_query_postings_bytes", C.c_uint64, [C.c_void_p, _P(Query)]),
    ("xgm_debug_decode_term_device", C.c_int64, [C.c_void_p, C.c_uint32, _P(C.c_uint32), _P(C.c_uint32), C.c_uint64]),
    ("xgm_debug_read_doclen", C.c_int64, [C.c_void_p, _P(C.c_uint32), C.c_uint64]),
    ("xgm_debug_read_positions", C.c_int64, [C.c_void_p, C.c_uint32, _P(C.c_uint32), C.c_uint64]),
    ("xgm_debug_plan_us", C.c_double, [C.c_void_p, _P(QueryDesc), _P(GlobalStats), C.c_uint32, C.c_uint32]),
    ("xgm_debug_batch_launches", C.c_int, [C.c_void_p, _P(Query), C.c_uint32, C.c_char_p, C.c_uint32]),
    ("xgm_index_set_batching", C.c_int, [C.c_void_p, C.c_uint32]),
    ("xgm_debug_batching_info", C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    ("xgm_debug_host_ns", C.c_int, [_P(C.c_uint64)]),
    ("xgm_debug_concurrent_searches", C.c_double, [C.c_void_p, _P(QueryDesc), _P(GlobalStats), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_double)]),
    ("xgm_debug_phase_cycles", C.c_int, [_P(C.c_ulonglong)]),
    ("xgm_debug_merge_cycles", C.c_int, [_P(C.c_ulonglong)]),
    ("xgm_debug_orw_phase_cycles", C.c_int, [_P(C.c_ulonglong)]),
    ("xgm_debug_plan_batch", C.c_int64, [C.c_void_p, _P(Query), C.c_uint32, C.c_char_p, _P(C.c_uint32), C.c_uint64]),
    ("xgm_debug_or_bounds", C.c_int, [C.c_void_p, _P(Query), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    ("xgm_debug_last_units", C.c_int64, [C.c_void_p, _P(C.c_ulonglong), C.c_uint64]),
    ("xgm_debug_last_units2", C.c_int64, [C.c_void_p, _P(C.c_ulonglong), _P(C.c_ulonglong), C.c_uint64]),
    ("xgm_mset_bounds_known", None, [_P(Query), _P(ResultHdr), C.c_uint64, _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32)]),
    ("xgm_index_attach_column_ordinals", C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_uint32), C.c_uint32, C.c_uint32]),
    ("xgm_expand_prefix", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, _P(C.c_uint32), _P(C.c_uint32)]),
    ("xgm_term_info", C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_char_p), _P(C.c_size_t), _P(C.c_uint32), _P(C.c_uint32)]),
    ("xgm_known_matching_docs", C.c_uint64, [_P(C.c_double), C.c_uint64, C.c_uint32, C.c_uint32]),
    ("xgm_round_estimate", C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    ("xgm_last_error", C.c_char_p, []),
    ("xgm_version", C.c_char_p, []),
]

_lib = None


class XgmError(RuntimeError):
    """Hard failure (< 0) reported by libxgm; `.code` holds the XGM_E_* value."""

    def __init__(self, code, msg):
        super().__init__("xgm error %d: %s" % (code, msg))
        self.code = code


class XgmUnsupported(Exception):
    """The device path declines this query shape (> 0): the caller must use the CPU matcher."""


def lib():
    """Load libxgm.so (once).  Raises if it has not been built — never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libxgm.so is not built: run `python -m xapiand_amd.build` (or __graft_entry__.build())")
        # A process must hold ONE HIP runtime.  PyTorch bundles its own libamdhip64.so (soname
        # libamdhip64.so.7, the same as /opt/rocm's): when torch is going to be used in this process
        # (device tensors, RCCL), it has to be loaded first so libxgm.so binds to that copy instead of
        # pulling in a second runtime that cannot see the GPU.  A C++ host without torch just gets
        # /opt/rocm's runtime through libxgm.so's RUNPATH.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, res, args in _API:
            fn = getattr(l, name)       # AttributeError if the ABI symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc < 0:
        raise XgmError(rc, lib().xgm_last_error().decode("utf-8", "replace"))
    if rc > 0:
        raise XgmUnsupported()
    return rc


def api_symbols():
    return [name for name, _, _ in _API]

"""ctypes binding of csrc/libmrk.so (the C-ABI declared in include/mrk.h).

The HIP extension is the product path: if it is missing this module raises -- there is
no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MRK_LIB_PATH") or os.path.join(CSRC, "libmrk.so")  # override: kernel experiments only

MRK_OK, MRK_E_INVAL, MRK_E_UNSUPPORTED, MRK_E_HIP, MRK_E_NOMEM, MRK_E_FORMAT = 0, -1, -2, -3, -4, -5
MRK_MAX_K = 1024
MRK_MAX_AND_TERMS = 8


class MrkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mrk error {code}: {msg}")
        self.code = code


class DictEntry(C.Structure):
    _fields_ = [("wordid", C.c_uint64), ("doclist_off", C.c_uint64), ("doclist_len", C.c_uint64),
                ("skiplist_off", C.c_uint64), ("docs", C.c_uint32), ("hits", C.c_uint32)]


class SegmentDesc(C.Structure):
    _fields_ = [("spd", C.c_void_p), ("spd_len", C.c_uint64), ("spp", C.c_void_p), ("spp_len", C.c_uint64),
                ("spe", C.c_void_p), ("spe_len", C.c_uint64), ("dict", C.c_void_p), ("n_terms", C.c_uint32),
                ("total_docs", C.c_uint64), ("skiplist_block_size", C.c_uint32), ("hit_format", C.c_uint32),
                ("n_fields", C.c_uint32), ("rowid_base", C.c_uint32)]


class Node(C.Structure):
    _fields_ = [("op", C.c_int32), ("n_children", C.c_int32), ("first_child", C.c_int32), ("term_id", C.c_int32),
                ("atom_pos", C.c_int32), ("field_mask", C.c_uint32), ("boost", C.c_float), ("opt", C.c_int32),
                ("not_weighted", C.c_int32), ("term_pos", C.c_int32), ("field_max_pos", C.c_int32)]


class Filter(C.Structure):
    """mrk_filter (include/mrk.h)"""
    _fields_ = [("kind", C.c_int32), ("bit_offset", C.c_int32), ("bit_count", C.c_int32), ("exclude", C.c_int32),
                ("has_equal_min", C.c_int32), ("has_equal_max", C.c_int32), ("open_left", C.c_int32), ("open_right", C.c_int32),
                ("min_value", C.c_int64), ("max_value", C.c_int64), ("values", C.POINTER(C.c_int64)), ("n_values", C.c_int32),
                ("fmin", C.c_float), ("fmax", C.c_float), ("mva_bits", C.c_int32), ("mva_all", C.c_int32), ("blob_attr_id", C.c_int32),
                ("n_blob_attrs", C.c_int32)]


class Query(C.Structure):
    _fields_ = [("nodes", C.POINTER(Node)), ("n_nodes", C.c_int32), ("children", C.POINTER(C.c_int32)),
                ("root", C.c_int32), ("ranker", C.c_int32), ("max_matches", C.c_int32),
                ("field_weights", C.POINTER(C.c_int32)), ("n_weights", C.c_int32), ("index_weight", C.c_int32),
                ("plain_idf", C.c_int32), ("normalized_tfidf", C.c_int32), ("total_docs_override", C.c_int64),
                ("local_docs", C.POINTER(C.c_int64)), ("cutoff", C.c_int32), ("filters", C.POINTER(Filter)),
                ("n_filters", C.c_int32), ("weight_filters", C.POINTER(Filter)), ("n_weight_filters", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("n", C.c_int32), ("total_found", C.c_int64), ("rowid", C.POINTER(C.c_uint32)),
                ("weight", C.POINTER(C.c_int32)), ("status", C.c_int32)]


class BatchStats(C.Structure):
    _fields_ = [("scan_ms", C.c_float), ("merge_ms", C.c_float), ("algo_bytes", C.c_uint64), ("n_items", C.c_uint64),
                ("dev_bytes", C.c_uint64), ("packed", C.c_uint32), ("n_cands", C.c_uint64),
                ("n_items_bm", C.c_uint64), ("plan_ms", C.c_float), ("submit_ms", C.c_float)]


class RtSegmentDesc(C.Structure):
    """mrk_rt_segment_desc (include/mrk.h)"""
    _fields_ = [("words", C.c_void_p), ("words_len", C.c_uint64), ("docs", C.c_void_p), ("docs_len", C.c_uint64), ("hits", C.c_void_p), ("hits_len", C.c_uint64),
                ("rows", C.c_uint32), ("word_dict", C.c_uint32), ("words_checkpoint", C.c_uint32), ("skiplist_block_size", C.c_uint32),
                ("hit_format", C.c_uint32), ("n_fields", C.c_uint32)]


class BatcherStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("queries", C.c_uint64), ("max_batch", C.c_uint32), ("submit_ms", C.c_double), ("collect_ms", C.c_double),
                ("flight_ms", C.c_double)]


class IndexInfo(C.Structure):
    """mrk_index_info (include/mrk.h)"""
    _fields_ = [("version", C.c_uint32), ("n_fields", C.c_uint32), ("n_attrs", C.c_uint32),
                ("skiplist_block_size", C.c_uint32), ("hit_format", C.c_uint32), ("hitless", C.c_uint32),
                ("word_dict", C.c_uint32), ("min_prefix_len", C.c_uint32), ("min_infix_len", C.c_uint32),
                ("index_sp", C.c_uint32), ("index_field_lens", C.c_uint32), ("n_checkpoints", C.c_uint32),
                ("total_docs", C.c_uint64), ("total_bytes", C.c_uint64), ("n_dead", C.c_uint64)]


class AttrInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_uint32), ("bit_offset", C.c_int32), ("bit_count", C.c_int32)]


class PairStats(C.Structure):
    _fields_ = [("matches", C.c_uint64), ("docs_a", C.c_uint64), ("docs_b", C.c_uint64), ("lines128_a", C.c_uint64), ("lines128_b", C.c_uint64),
                ("blocks_a", C.c_uint64), ("blocks_b", C.c_uint64), ("lines128s_a", C.c_uint64), ("lines128s_b", C.c_uint64)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_docs", C.c_uint64), ("rowid_base", C.c_uint64), ("term_prob", C.POINTER(C.c_double)),
                ("n_terms", C.c_uint32), ("n_fields", C.c_uint32), ("title_frac", C.c_double), ("max_pos", C.c_uint32),
                ("skiplist_block_size", C.c_uint32), ("hit_format", C.c_uint32), ("end_markers", C.c_uint32),
                ("n_threads", C.c_uint32)]


# every symbol include/mrk.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mrk_last_error", C.c_char_p, []),
    ("mrk_ctx_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("mrk_ctx_destroy", C.c_int, [C.c_void_p]),
    ("mrk_batcher_create", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mrk_batcher_destroy", None, [C.c_void_p]),
    ("mrk_batcher_search", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Query), C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(Result)]),
    ("mrk_batcher_stats_get", C.c_int, [C.c_void_p, C.POINTER(BatcherStats)]),
    ("mrk_ctx_set", C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("mrk_segment_create", C.c_int, [C.c_void_p, C.POINTER(SegmentDesc), C.POINTER(C.c_void_p)]),
    ("mrk_segment_validate", C.c_int, [C.POINTER(SegmentDesc)]),
    ("mrk_segment_destroy", None, [C.c_void_p]),
    ("mrk_segment_device_bytes", C.c_uint64, [C.c_void_p]),
    ("mrk_batch_export_rows", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mrk_batch_set_rows_dst", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mrk_topk_merge_rows_async", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_uint32]),
    ("mrk_merge_wait", C.c_int, [C.c_void_p, C.c_uint32]),
    ("mrk_batch_record_event", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mrk_topk_merge_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("mrk_segment_set_dead_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    ("mrk_segment_set_attrs", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64]),
    ("mrk_batch_create", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mrk_batch_destroy", None, [C.c_void_p]),
    ("mrk_batch_submit", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Query), C.c_uint32]),
    ("mrk_batch_wait", C.c_int, [C.c_void_p]),
    ("mrk_batch_test", C.c_int, [C.c_void_p]),
    ("mrk_batch_result", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Result)]),
    ("mrk_batch_stats_get", C.c_int, [C.c_void_p, C.POINTER(BatchStats)]),
    ("mrk_batch_device_results", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("mrk_batch_export_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mrk_topk_merge", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("mrk_comm_unique_id", C.c_int, [C.c_void_p]),
    ("mrk_comm_init", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("mrk_comm_destroy", None, [C.c_void_p]),
    ("mrk_comm_allreduce_i64", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    ("mrk_shard_exchange", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]),
    ("mrk_shard_slice", C.c_int, [C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("mrk_shard_flags", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("mrk_shard_partitioned", C.c_int, [C.c_void_p]),
    ("mrk_topk_merge_rows_part", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("mrk_idf", C.c_float, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float]),
    ("mrk_index_from_hits", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mrk_synth_generate", C.c_int, [C.POINTER(SynthParams), C.POINTER(C.c_void_p)]),
    ("mrk_host_index_free", None, [C.c_void_p]),
    ("mrk_host_index_spd", C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("mrk_host_index_spp", C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("mrk_host_index_spe", C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("mrk_host_index_dict", C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint32)]),
    ("mrk_host_index_pair_stats", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(PairStats)]),
    ("mrk_index_open", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("mrk_host_index_info", C.c_int, [C.c_void_p, C.POINTER(IndexInfo)]),
    ("mrk_host_index_field_name", C.c_char_p, [C.c_void_p, C.c_uint32]),
    ("mrk_host_index_find_word", C.c_int32, [C.c_void_p, C.c_char_p, C.c_int32]),
    ("mrk_host_index_word", C.c_void_p, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("mrk_host_index_find_wordid", C.c_int32, [C.c_void_p, C.c_uint64]),
    ("mrk_host_index_dead_rows", C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("mrk_host_index_attr", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(AttrInfo)]),
    ("mrk_host_index_attr_rows", C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("mrk_segment_set_blobs", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint64]),
    ("mrk_host_index_blobs", C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("mrk_rt_ram_open", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("mrk_rt_ram_segments", C.c_uint32, [C.c_void_p]),
    ("mrk_rt_ram_take", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mrk_rt_ram_free", None, [C.c_void_p]),
    ("mrk_rt_segment_open", C.c_int, [C.POINTER(RtSegmentDesc), C.POINTER(C.c_void_p)]),
    ("mrk_query_parse", C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mrk_parsed_transform", C.c_int, [C.c_void_p]),
    ("mrk_parsed_free", None, [C.c_void_p]),
    ("mrk_parsed_n_nodes", C.c_int32, [C.c_void_p]),
    ("mrk_parsed_root", C.c_int32, [C.c_void_p]),
    ("mrk_parsed_nodes", C.POINTER(Node), [C.c_void_p]),
    ("mrk_parsed_children", C.POINTER(C.c_int32), [C.c_void_p, C.POINTER(C.c_int32)]),
    ("mrk_parsed_keyword", C.c_char_p, [C.c_void_p, C.c_int32]),
    ("mrk_parsed_resolve", C.c_int, [C.c_void_p, C.c_void_p]),
]


def build(force: bool = False) -> str:
    """Compile libmrk.so for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "mrk.h"))
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", CSRC, "libmrk.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP extension is the only execution path; there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != MRK_OK:
        raise MrkError(rc, lib().mrk_last_error().decode(errors="replace"))

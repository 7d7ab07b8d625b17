"""ctypes wrapper over oracle/libcpu_ref.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcpu_ref.so")

RANK_PROXIMITY_BM25, RANK_BM25, RANK_NONE, RANK_WORDCOUNT, RANK_PROXIMITY = 0, 1, 2, 3, 4
RANK_MATCHANY, RANK_FIELDMASK, RANK_SPH04 = 5, 6, 7
OP_TERM, OP_AND, OP_OR, OP_MAYBE, OP_ANDNOT, OP_PHRASE, OP_PROXIMITY, OP_QUORUM, OP_BEFORE = 0, 1, 2, 3, 4, 5, 6, 7, 8
OP_NEAR, OP_NOTNEAR = 9, 10  # 'a NEAR/N b', 'a NOTNEAR/N b': opt = N
OP_SENTENCE, OP_PARAGRAPH = 11, 12  # 'a SENTENCE b', 'a PARAGRAPH b': the node's term_id = the boundary keyword (index_sp), < 0 = none
ALL_FIELDS = 0xFFFFFFFF


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cpu_ref.c")
    hdr = os.path.join(_HERE, "cpu_ref.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcpu_ref.so"])
    return _LIB_PATH


class DictEntry(C.Structure):
    _fields_ = [("wordid", C.c_uint64), ("doclist_off", C.c_uint64), ("doclist_len", C.c_uint64),
                ("skiplist_off", C.c_uint64), ("docs", C.c_uint32), ("hits", C.c_uint32)]


DICT_DTYPE = np.dtype([("wordid", "<u8"), ("doclist_off", "<u8"), ("doclist_len", "<u8"),
                       ("skiplist_off", "<u8"), ("docs", "<u4"), ("hits", "<u4")])
assert DICT_DTYPE.itemsize == C.sizeof(DictEntry)


class _Index(C.Structure):
    _fields_ = [("spd", C.c_void_p), ("spd_len", C.c_size_t), ("spp", C.c_void_p), ("spp_len", C.c_size_t),
                ("spe", C.c_void_p), ("spe_len", C.c_size_t), ("dict", C.c_void_p), ("n_terms", C.c_uint32),
                ("total_docs", C.c_int64), ("skiplist_block_size", C.c_int), ("inline_hits", C.c_int),
                ("n_fields", C.c_int), ("dead_rows", C.c_void_p), ("attrs", C.c_void_p), ("attr_stride", C.c_int), ("blobs", C.c_void_p)]


class _Node(C.Structure):
    _fields_ = [("op", C.c_int), ("n_children", C.c_int), ("first_child", C.c_int), ("term_id", C.c_int32),
                ("atom_pos", C.c_int), ("field_mask", C.c_uint32), ("boost", C.c_float), ("opt", C.c_int),
                ("not_weighted", C.c_int), ("term_pos", C.c_int), ("field_max_pos", C.c_int), ("field_mask_hi", C.c_uint32 * 7)]


class _Filter(C.Structure):
    _fields_ = [("kind", C.c_int), ("bit_offset", C.c_int), ("bit_count", C.c_int), ("exclude", C.c_int),
                ("has_equal_min", C.c_int), ("has_equal_max", C.c_int), ("open_left", C.c_int), ("open_right", C.c_int),
                ("min_value", C.c_int64), ("max_value", C.c_int64), ("values", C.POINTER(C.c_int64)), ("n_values", C.c_int),
                ("fmin", C.c_float), ("fmax", C.c_float), ("mva_bits", C.c_int), ("mva_all", C.c_int), ("blob_attr_id", C.c_int),
                ("n_blob_attrs", C.c_int)]


class _Query(C.Structure):
    _fields_ = [("nodes", C.POINTER(_Node)), ("n_nodes", C.c_int), ("children", C.POINTER(C.c_int)),
                ("root", C.c_int), ("ranker", C.c_int), ("max_matches", C.c_int),
                ("field_weights", C.POINTER(C.c_int32)), ("n_weights", C.c_int), ("index_weight", C.c_int),
                ("plain_idf", C.c_int), ("normalized_tfidf", C.c_int), ("total_docs_override", C.c_int64),
                ("local_docs", C.POINTER(C.c_int64)), ("cutoff", C.c_int), ("filters", C.POINTER(_Filter)), ("n_filters", C.c_int),
                ("weight_filters", C.POINTER(_Filter)), ("n_weight_filters", C.c_int)]


class _Result(C.Structure):
    _fields_ = [("n", C.c_int), ("total_found", C.c_int64), ("rowid", C.POINTER(C.c_uint32)),
                ("weight", C.POINTER(C.c_int32)), ("fetched_docs", C.c_int64), ("fetched_hits", C.c_int64),
                ("skips", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_zip_u64.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_zip_u64.restype = C.c_int
        L.orc_writer_new.argtypes = [C.c_int, C.c_int]
        L.orc_writer_new.restype = C.c_void_p
        L.orc_writer_free.argtypes = [C.c_void_p]
        L.orc_writer_hit.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_writer_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_writer_finish.argtypes = [C.c_void_p]
        for nm in ("spd", "spp", "spe", "dict"):
            f = getattr(L, "orc_writer_" + nm)
            f.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
            f.restype = C.c_void_p
        L.orc_search.argtypes = [C.POINTER(_Index), C.POINTER(_Query), C.POINTER(_Result)]
        L.orc_search.restype = C.c_int
        L.orc_last_error.restype = C.c_char_p
        L.orc_search_many.argtypes = [C.POINTER(_Index), C.POINTER(C.POINTER(_Query)), C.c_int, C.c_int, C.c_int]
        L.orc_search_many.restype = C.c_double
        L.orc_idf.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float]
        L.orc_idf.restype = C.c_float
        L.orc_decode_doclist.argtypes = [C.POINTER(_Index), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_decode_doclist.restype = C.c_int
        L.orc_decode_hits.argtypes = [C.POINTER(_Index), C.c_uint64, C.c_void_p, C.c_int]
        L.orc_decode_hits.restype = C.c_int
        L.orc_decode_skiplist.argtypes = [C.POINTER(_Index), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_decode_skiplist.restype = C.c_int
        _lib = L
    return _lib


def zip_bytes(v: int) -> bytes:
    buf = (C.c_uint8 * 16)()
    n = lib().orc_zip_u64(buf, v)
    return bytes(buf[:n])


def hitpos(fld: int, pos: int, end: bool = False) -> int:
    return (fld << 24) | ((1 if end else 0) << 23) | (pos & 0x7FFFFF)


@dataclass
class Index:
    """A segment in the reference's on-disk layout (byte arrays + flat term table)."""
    spd: np.ndarray
    spp: np.ndarray
    spe: np.ndarray
    dict: np.ndarray  # DICT_DTYPE, indexed by term id
    total_docs: int
    skiplist_block_size: int = 128
    inline_hits: int = 1
    n_fields: int = 2
    dead_rows: Optional[np.ndarray] = None  # uint32 bitmap, DeadRowMap_c layout
    attrs: Optional[np.ndarray] = None      # uint32 [total_docs, stride]: the .spa rows
    blobs: Optional[np.ndarray] = None      # uint8: the blob pool (MVA filters)

    def c_struct(self) -> _Index:
        s = _Index()
        s.spd, s.spd_len = self.spd.ctypes.data, self.spd.size
        s.spp, s.spp_len = self.spp.ctypes.data, self.spp.size
        s.spe, s.spe_len = self.spe.ctypes.data, self.spe.size
        s.dict, s.n_terms = self.dict.ctypes.data, len(self.dict)
        s.total_docs = self.total_docs
        s.skiplist_block_size = self.skiplist_block_size
        s.inline_hits = self.inline_hits
        s.n_fields = self.n_fields
        s.dead_rows = self.dead_rows.ctypes.data if self.dead_rows is not None else None
        s.attrs = self.attrs.ctypes.data if self.attrs is not None else None
        s.attr_stride = int(self.attrs.shape[1]) if self.attrs is not None else 0
        s.blobs = self.blobs.ctypes.data if self.blobs is not None else None
        return s

    def decode_doclist(self, term_id: int):
        n = int(self.dict[term_id]["docs"])
        rowid = np.zeros(n, np.uint32)
        fields = np.zeros(n, np.uint32)
        hits = np.zeros(n, np.uint32)
        hp = np.zeros(n, np.uint64)
        s = self.c_struct()
        got = lib().orc_decode_doclist(C.byref(s), term_id, rowid.ctypes.data, fields.ctypes.data,
                                       hits.ctypes.data, hp.ctypes.data)
        assert got == n, (got, n)
        return rowid, fields, hits, hp

    def decode_hits(self, hitpos64: int) -> List[int]:
        s = self.c_struct()
        out = np.zeros(4096, np.uint32)
        n = lib().orc_decode_hits(C.byref(s), int(hitpos64), out.ctypes.data, out.size)
        return [int(x) for x in out[:n]]

    def decode_skiplist(self, term_id: int):
        s = self.c_struct()
        cap = int(self.dict[term_id]["docs"]) // self.skiplist_block_size + 2
        b = np.zeros(cap, np.uint32)
        o = np.zeros(cap, np.uint64)
        h = np.zeros(cap, np.uint64)
        n = lib().orc_decode_skiplist(C.byref(s), term_id, b.ctypes.data, o.ctypes.data, h.ctypes.data, cap)
        return b[:n], o[:n], h[:n]


def build_index(wordid: np.ndarray, rowid: np.ndarray, hitp: np.ndarray, total_docs: int,
                skiplist_block_size: int = 128, inline_hits: int = 1, n_fields: int = 2,
                n_terms: Optional[int] = None) -> Index:
    """Run the CSphHitBuilder restatement over hits sorted by (wordid, rowid, hitpos).

    wordid values are 1-based term ids + 1 (0 is the flush marker); the returned term
    table is indexed by (wordid - 1) and has n_terms rows (absent words: docs = 0).
    """
    L = lib()
    wordid = np.ascontiguousarray(wordid, np.uint64)
    rowid = np.ascontiguousarray(rowid, np.uint32)
    hitp = np.ascontiguousarray(hitp, np.uint32)
    assert wordid.size == rowid.size == hitp.size
    assert (wordid > 0).all()
    w = L.orc_writer_new(skiplist_block_size, inline_hits)
    try:
        L.orc_writer_hits(w, wordid.ctypes.data, rowid.ctypes.data, hitp.ctypes.data, wordid.size)
        L.orc_writer_finish(w)
        n = C.c_size_t()
        bufs = {}
        for nm in ("spd", "spp", "spe"):
            p = getattr(L, "orc_writer_" + nm)(w, C.byref(n))
            # 64 bytes of zero padding: device loaders may over-read a little past the end
            a = np.zeros(n.value + 64, np.uint8)
            C.memmove(a.ctypes.data, p, n.value)
            bufs[nm] = a[: n.value]
        p = L.orc_writer_dict(w, C.byref(n))
        d = np.zeros(n.value, DICT_DTYPE)
        if n.value:
            C.memmove(d.ctypes.data, p, n.value * DICT_DTYPE.itemsize)
    finally:
        L.orc_writer_free(w)
    nt = int(n_terms if n_terms is not None else (int(wordid.max()) if wordid.size else 0))
    table = np.zeros(nt, DICT_DTYPE)
    for e in d:
        table[int(e["wordid"]) - 1] = e
    return Index(bufs["spd"], bufs["spp"], bufs["spe"], table, total_docs, skiplist_block_size, inline_hits, n_fields)


# ---------------------------------------------------------------- query trees
@dataclass
class QNode:
    op: int
    children: List["QNode"] = field(default_factory=list)
    term_id: int = -1
    atom_pos: int = 0
    field_mask: int = ALL_FIELDS
    boost: float = 1.0
    opt: int = 0  # XQNode_t::m_iOpArg (proximity distance, quorum threshold)
    term_pos: int = 0  # TERMPOS_*: '^word', 'word$', '@field[N] word'
    field_max_pos: int = 0


FILTER_VALUES, FILTER_RANGE = 0, 1
TERMPOS_NONE, TERMPOS_START, TERMPOS_END, TERMPOS_STARTEND, TERMPOS_LIMIT = 0, 1, 2, 3, 4


def term(term_id: int, atom_pos: int, field_mask: int = ALL_FIELDS, boost: float = 1.0, term_pos: int = 0,
         field_max_pos: int = 0) -> QNode:
    return QNode(OP_TERM, [], term_id, atom_pos, field_mask, boost, term_pos=term_pos, field_max_pos=field_max_pos)


def op(kind: int, *children: QNode, field_mask: int = ALL_FIELDS, opt: int = 0, unit_term: int = -1) -> QNode:
    return QNode(kind, list(children), term_id=unit_term, field_mask=field_mask, opt=opt)


@dataclass
class Result:
    rowid: np.ndarray
    weight: np.ndarray
    total_found: int
    fetched_docs: int = 0
    fetched_hits: int = 0
    skips: int = 0


class FlatQuery:
    """Flattened query kept alive for repeated orc_search calls (cpu_baseline timing)."""

    def __init__(self, root: QNode, ranker: int = RANK_BM25, max_matches: int = 1000,
                 field_weights: Optional[Sequence[int]] = None, index_weight: int = 1,
                 plain_idf: bool = False, normalized_tfidf: bool = True, total_docs_override: int = 0,
                 local_docs: Optional[dict] = None, cutoff: int = 0, filters: Optional[Sequence[dict]] = None,
                 weight_filters: Optional[Sequence[dict]] = None):
        nodes: List[QNode] = []

        def walk(n: QNode) -> int:
            i = len(nodes)
            nodes.append(n)
            n._kids = [walk(c) for c in n.children]  # type: ignore[attr-defined]
            return i

        walk(root)
        self.nodes = (_Node * len(nodes))()
        kids: List[int] = []
        for i, n in enumerate(nodes):
            cn = self.nodes[i]
            cn.op, cn.n_children, cn.first_child = n.op, len(n._kids), len(kids)  # type: ignore[attr-defined]
            kids.extend(n._kids)  # type: ignore[attr-defined]
            cn.term_id, cn.atom_pos, cn.field_mask, cn.boost = n.term_id, n.atom_pos, n.field_mask & 0xFFFFFFFF, n.boost
            # fields 32..255 (indexes with more than 32 fields): ALL_FIELDS = any field; else the mask's bits above 32
            for d in range(7):
                cn.field_mask_hi[d] = 0xFFFFFFFF if n.field_mask == ALL_FIELDS else (n.field_mask >> (32 * (d + 1))) & 0xFFFFFFFF
            cn.opt = n.opt
            cn.term_pos, cn.field_max_pos = n.term_pos, n.field_max_pos
        self.children = (C.c_int * max(1, len(kids)))(*kids)
        q = _Query()
        q.nodes, q.n_nodes = self.nodes, len(nodes)
        q.children = self.children
        q.root, q.ranker, q.max_matches = 0, ranker, max_matches
        if field_weights is not None:
            self.fw = (C.c_int32 * len(field_weights))(*field_weights)
            q.field_weights, q.n_weights = self.fw, len(field_weights)
        q.index_weight = index_weight
        q.plain_idf, q.normalized_tfidf = int(plain_idf), int(normalized_tfidf)
        q.total_docs_override = total_docs_override
        if local_docs:
            arr = [-1] * len(nodes)
            for i, n in enumerate(nodes):
                if n.op == OP_TERM and n.term_id in local_docs:
                    arr[i] = int(local_docs[n.term_id])
            self.ld = (C.c_int64 * len(nodes))(*arr)
            q.local_docs = self.ld
        q.cutoff = cutoff
        self.fv = []

        def fill(fs):  # dicts with the fields of orc_filter; "values" = ascending ints, "fmin" / "fmax" = a float range
            arr = (_Filter * len(fs))()
            for i, f in enumerate(fs):
                c = arr[i]
                c.kind = FILTER_VALUES if "values" in f else 2 if "fmin" in f else FILTER_RANGE  # 2 = ORC_FILTER_FLOATRANGE
                c.fmin, c.fmax = float(f.get("fmin", 0.0)), float(f.get("fmax", 0.0))
                c.mva_bits, c.mva_all = int(f.get("mva_bits", 0)), int(f.get("mva_all", False))
                c.blob_attr_id, c.n_blob_attrs = int(f.get("blob_attr_id", 0)), int(f.get("n_blob_attrs", 0))
                c.bit_offset, c.bit_count, c.exclude = f["bit_offset"], f["bit_count"], int(f.get("exclude", False))
                c.has_equal_min, c.has_equal_max = int(f.get("has_equal_min", True)), int(f.get("has_equal_max", True))
                c.open_left, c.open_right = int(f.get("open_left", False)), int(f.get("open_right", False))
                c.min_value, c.max_value = int(f.get("min", 0)), int(f.get("max", 0))
                if "values" in f:
                    vals = (C.c_int64 * len(f["values"]))(*sorted(int(v) for v in f["values"]))
                    self.fv.append(vals)
                    c.values, c.n_values = vals, len(f["values"])
            return arr

        if filters:
            self.fl = fill(filters)
            q.filters, q.n_filters = self.fl, len(filters)
        if weight_filters:
            self.wfl = fill(weight_filters)
            q.weight_filters, q.n_weight_filters = self.wfl, len(weight_filters)
        self.q = q
        self.K = max_matches

    def run(self, index: Index, cidx: Optional[_Index] = None) -> Result:
        s = cidx if cidx is not None else index.c_struct()
        rowid = np.zeros(self.K, np.uint32)
        weight = np.zeros(self.K, np.int32)
        r = _Result()
        r.rowid = rowid.ctypes.data_as(C.POINTER(C.c_uint32))
        r.weight = weight.ctypes.data_as(C.POINTER(C.c_int32))
        rc = lib().orc_search(C.byref(s), C.byref(self.q), C.byref(r))
        if rc != 0:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        return Result(rowid[: r.n].copy(), weight[: r.n].copy(), int(r.total_found), int(r.fetched_docs),
                      int(r.fetched_hits), int(r.skips))


def search(index: Index, root: QNode, **kw) -> Result:
    return FlatQuery(root, **kw).run(index)


def idf(term_docs: int, total_docs: int, plain: bool = False, normalized: bool = True, n_qwords: int = 1,
        boost: float = 1.0) -> float:
    return float(lib().orc_idf(term_docs, total_docs, int(plain), int(normalized), n_qwords, boost))


def search_many(index: Index, flat_queries: Sequence["FlatQuery"], repeat: int, n_threads: int) -> float:
    """Wall seconds for running every query `repeat` times on n_threads threads (C worker pool)."""
    arr = (C.POINTER(_Query) * len(flat_queries))(*[C.pointer(q.q) for q in flat_queries])
    s = index.c_struct()
    t = lib().orc_search_many(C.byref(s), arr, len(flat_queries), repeat, n_threads)
    if t < 0:
        raise RuntimeError("oracle: a query failed in search_many")
    return float(t)


def usable_cpus() -> int:
    """CPUs this process may really use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n

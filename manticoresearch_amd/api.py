"""Host-side Python mirror of the reference's interface for this path.

Names follow the reference (Manticore 3.6.1): an index *segment* (plain index / RT disk
chunk) holds doclists (.spd), hitlists (.spp) and skiplists (.spe); a query is an
``XQNode`` tree of keywords (``XQKeyword``) plus the ``CSphQuery`` knobs that reach the
ranker (ranker mode, max_matches, field_weights, idf flags, local_df); results are what an
``ISphMatchSorter`` hands back: matches best-first + total_found.

Everything here is thin plumbing over the C-ABI (include/mrk.h); the work happens in the
HIP kernels of csrc/mrk_kernels.hip.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import MrkError, check, lib

# ESphRankMode / XQOperator_e subsets (include/mrk.h)
SPH_RANK_PROXIMITY_BM25, SPH_RANK_BM25, SPH_RANK_NONE, SPH_RANK_WORDCOUNT, SPH_RANK_PROXIMITY = 0, 1, 2, 3, 4
SPH_RANK_MATCHANY, SPH_RANK_FIELDMASK, SPH_RANK_SPH04 = 5, 6, 7
SPH_QUERY_TERM, SPH_QUERY_AND, SPH_QUERY_OR, SPH_QUERY_MAYBE, SPH_QUERY_ANDNOT, SPH_QUERY_PHRASE = 0, 1, 2, 3, 4, 5
SPH_QUERY_PROXIMITY = 6  # '"a b c"~N': XQNode.opt = N
SPH_QUERY_QUORUM = 7     # '"a b c"/N': XQNode.opt = N
SPH_QUERY_BEFORE = 8     # 'a << b << c' (ExtOrder_c)
SPH_QUERY_NEAR = 9       # 'a NEAR/N b' (ExtNWay_T<FSMmultinear_c>): XQNode.opt = N
SPH_QUERY_NOTNEAR = 10   # 'a NOTNEAR/N b' (ExtNotNear_c): XQNode.opt = N
SPH_QUERY_SENTENCE, SPH_QUERY_PARAGRAPH = 11, 12  # 'a SENTENCE b' / 'a PARAGRAPH b' (ExtUnit_c): XQNode.unit_term = the index_sp boundary keyword
SPH_HIT_FORMAT_PLAIN, SPH_HIT_FORMAT_INLINE = 0, 1
ALL_FIELDS = 0xFFFFFFFF

DICT_DTYPE = np.dtype([("wordid", "<u8"), ("doclist_off", "<u8"), ("doclist_len", "<u8"),
                       ("skiplist_off", "<u8"), ("docs", "<u4"), ("hits", "<u4")])
assert DICT_DTYPE.itemsize == C.sizeof(_lib.DictEntry)


# --------------------------------------------------------------------------- host index
@dataclass
class HostIndex:
    """One segment's bytes in host memory, in the reference's on-disk layout."""
    spd: np.ndarray
    spp: np.ndarray
    spe: np.ndarray
    dict: np.ndarray  # DICT_DTYPE, indexed by term id
    total_docs: int
    skiplist_block_size: int = 128
    hit_format: int = SPH_HIT_FORMAT_INLINE
    n_fields: int = 2

    @property
    def doclist_bytes(self) -> int:
        return int(self.dict["doclist_len"].sum())


class _HostIndexOwner:
    """Keeps the C++ mrk_host_index alive while numpy views its buffers (no copy of GBs)."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                lib().mrk_host_index_free(self.h)
                self.h = None
        except Exception:
            pass


def _take_host_index(h, total_docs: int, skiplist_block_size: int, hit_format: int, n_fields: int) -> HostIndex:
    L = lib()
    owner = _HostIndexOwner(h)
    n = C.c_uint64()
    out = {}
    for nm in ("spd", "spp", "spe"):
        p = getattr(L, "mrk_host_index_" + nm)(h, C.byref(n))
        buf = (C.c_uint8 * (n.value + 64)).from_address(p)  # the writer leaves 64 zero bytes of slack
        a = np.frombuffer(buf, dtype=np.uint8)
        out[nm] = a[: n.value]
    nt = C.c_uint32()
    p = L.mrk_host_index_dict(h, C.byref(nt))
    d = np.zeros(nt.value, DICT_DTYPE)
    if nt.value:
        C.memmove(d.ctypes.data, p, nt.value * DICT_DTYPE.itemsize)
    hi = HostIndex(out["spd"], out["spp"], out["spe"], d, total_docs, skiplist_block_size, hit_format, n_fields)
    hi._owner = owner
    return hi


def open_index(path_prefix: str) -> HostIndex:
    """Read the files of a real Manticore 3.x index / RT disk chunk (<prefix>.sph .spi .spd .spp .spe [.spm]).

    The result carries what the header says (`info`, `fields`), the keywords (`words`, dict=keywords), `find_word()`
    -- the dictionary lookup DiskIndexQwordSetup_c::Setup does (sphinx.cpp:12953-13060) -- and the dead-row bitmap
    of the .spm file (`dead_bitmap` for Segment.set_dead_rows(), `dead_rows` as rowids; None when no row is dead)."""
    L = lib()
    h = C.c_void_p()
    check(L.mrk_index_open(os.fsencode(path_prefix), C.byref(h)))
    return _wrap_file_index(h)


def open_rt_ram(path_prefix: str) -> List[HostIndex]:
    """Read an RT index's RAM chunk (<prefix>.meta + <prefix>.ram, RtIndex_c::SaveMeta / SaveRamChunk): one HostIndex per
    RAM segment, each with the fields open_index() gives (info, fields, words, attrs, attr_rows, dead rows, find_word):
    the segment's postings decoded from the RT codecs and re-emitted in the disk format (include/mrk.h, mrk_rt_ram_open)."""
    L = lib()
    rt = C.c_void_p()
    check(L.mrk_rt_ram_open(os.fsencode(path_prefix), C.byref(rt)))
    try:
        out = []
        for i in range(L.mrk_rt_ram_segments(rt)):
            h = C.c_void_p()
            check(L.mrk_rt_ram_take(rt, i, C.byref(h)))
            out.append(_wrap_file_index(h))
        return out
    finally:
        L.mrk_rt_ram_free(rt)


def open_rt_segment(words: bytes, docs: bytes, hits: bytes, rows: int, word_dict: bool = True, words_checkpoint: int = 64,
                    skiplist_block_size: int = 128, hit_format: int = 1, n_fields: int = 1) -> HostIndex:
    """A LIVE RAM segment handed over in memory (mrk_rt_segment_open): the three byte vectors of an RtSegment_t."""
    wb, db, hb = (np.frombuffer(bytes(x), np.uint8) if len(x) else np.zeros(0, np.uint8) for x in (words, docs, hits))
    d = _lib.RtSegmentDesc(wb.ctypes.data if wb.size else None, wb.size, db.ctypes.data if db.size else None, db.size, hb.ctypes.data if hb.size else None, hb.size,
                           rows, int(word_dict), words_checkpoint, skiplist_block_size, hit_format, n_fields)
    h = C.c_void_p()
    check(lib().mrk_rt_segment_open(C.byref(d), C.byref(h)))
    return _wrap_file_index(h)


def _wrap_file_index(h) -> HostIndex:
    L = lib()
    info = _lib.IndexInfo()
    check(L.mrk_host_index_info(h, C.byref(info)))
    hi = _take_host_index(h, int(info.total_docs), int(info.skiplist_block_size), int(info.hit_format), int(info.n_fields))
    hi.info = {k: int(getattr(info, k)) for k, _ in _lib.IndexInfo._fields_}
    hi.fields = [L.mrk_host_index_field_name(h, i).decode("utf-8", "surrogateescape") for i in range(info.n_fields)]
    hi.words = []
    if info.word_dict:
        n = C.c_uint32()
        for t in range(len(hi.dict)):
            p = L.mrk_host_index_word(h, t, C.byref(n))
            hi.words.append(C.string_at(p, n.value).decode("utf-8", "surrogateescape"))
    hi.attrs = {}  # name -> (ESphAttr type, bit_offset, bit_count): Filter(bit_offset, bit_count, ...)
    ai = _lib.AttrInfo()
    for i in range(info.n_attrs):
        check(L.mrk_host_index_attr(h, i, C.byref(ai)))
        hi.attrs[ai.name.decode("utf-8", "surrogateescape")] = (int(ai.type), int(ai.bit_offset), int(ai.bit_count))
    stride, arows = C.c_uint32(), C.c_uint64()
    p = L.mrk_host_index_attr_rows(h, C.byref(stride), C.byref(arows))
    hi.attr_rows = None  # the .spa rows for Segment.set_attrs()
    if p:
        hi.attr_rows = np.frombuffer((C.c_uint32 * (arows.value * stride.value)).from_address(p), dtype=np.uint32).reshape(arows.value, stride.value).copy()
    blen, nblob = C.c_uint64(), C.c_uint32()
    p = L.mrk_host_index_blobs(h, C.byref(blen), C.byref(nblob))
    hi.blobs = np.frombuffer((C.c_uint8 * blen.value).from_address(p), dtype=np.uint8).copy() if p else None  # blob pool (strings, MVAs)
    hi.n_blob_attrs = int(nblob.value)
    nrows = C.c_uint64()
    p = L.mrk_host_index_dead_rows(h, C.byref(nrows))
    hi.dead_bitmap = hi.dead_rows = None
    if p and info.n_dead:
        hi.dead_bitmap = np.frombuffer((C.c_uint32 * ((nrows.value + 31) // 32)).from_address(p), dtype=np.uint32).copy()
        bits = np.unpackbits(hi.dead_bitmap.view(np.uint8), bitorder="little")[: nrows.value]
        hi.dead_rows = np.flatnonzero(bits).astype(np.uint32)

    def find_word(word, _h=h, _owner=hi._owner):
        b = word.encode() if isinstance(word, str) else bytes(word)
        return int(L.mrk_host_index_find_word(_h, b, len(b)))

    hi.find_word = find_word
    hi.find_wordid = lambda wid, _h=h, _owner=hi._owner: int(L.mrk_host_index_find_wordid(_h, wid))
    return hi


def pair_stats(hi: HostIndex, pairs: Sequence[Sequence[int]], n_threads: int = 0):
    """[(matches, docs_a, docs_b, lines128_a, lines128_b, blocks_a, blocks_b, lines128s_a, lines128s_b)] per keyword pair: common
    docs, the distinct 128-byte lines of each keyword's packed tf / field words they touch, the distinct 128-doc blocks, and the
    lines of the slot-ordered two-byte plane (mrk_host_index_pair_stats; roofline accounting only)."""
    arr = np.ascontiguousarray(np.asarray(pairs, dtype=np.uint32).reshape(-1, 2))
    out = (_lib.PairStats * len(arr))()
    check(lib().mrk_host_index_pair_stats(hi._owner.h, hi.hit_format, arr.ctypes.data, len(arr), n_threads, out))
    return [(int(o.matches), int(o.docs_a), int(o.docs_b), int(o.lines128_a), int(o.lines128_b), int(o.blocks_a), int(o.blocks_b), int(o.lines128s_a), int(o.lines128s_b)) for o in out]


def index_from_hits(wordid: np.ndarray, rowid: np.ndarray, hitpos: np.ndarray, n_terms: int, total_docs: int,
                    skiplist_block_size: int = 128, hit_format: int = SPH_HIT_FORMAT_INLINE, n_fields: int = 2) -> HostIndex:
    """Encode explicit hits (sorted by wordid, rowid, hitpos; wordid = term id + 1)."""
    wordid = np.ascontiguousarray(wordid, np.uint64)
    rowid = np.ascontiguousarray(rowid, np.uint32)
    hitpos = np.ascontiguousarray(hitpos, np.uint32)
    h = C.c_void_p()
    check(lib().mrk_index_from_hits(wordid.ctypes.data, rowid.ctypes.data, hitpos.ctypes.data, wordid.size, n_terms,
                                    skiplist_block_size, hit_format, C.byref(h)))
    return _take_host_index(h, total_docs, skiplist_block_size, hit_format, n_fields)


def synth_index(n_docs: int, term_prob: Sequence[float], seed: int = 0x5EED0001, shard: int = 0, n_fields: int = 2,
                title_frac: float = 0.1, max_pos: int = 1024, skiplist_block_size: int = 128,
                hit_format: int = SPH_HIT_FORMAT_INLINE, end_markers: int = False, n_threads: int = 0,
                rowid_base: Optional[int] = None) -> HostIndex:
    """Deterministic synthetic postings for the given per-term document probabilities.

    Postings are a function of (seed, term, GLOBAL rowid): the segment holds the rows [rowid_base, rowid_base + n_docs)
    of the one corpus the seed defines (`shard` is shorthand for rowid_base = shard * n_docs), so equal-sized shards
    laid end to end ARE the unsharded corpus."""
    probs = (C.c_double * len(term_prob))(*[float(x) for x in term_prob])
    if rowid_base is None:
        rowid_base = shard * n_docs
    p = _lib.SynthParams(seed, n_docs, rowid_base, probs, len(term_prob), n_fields, title_frac, max_pos,
                         skiplist_block_size, hit_format, int(end_markers), n_threads)
    h = C.c_void_p()
    check(lib().mrk_synth_generate(C.byref(p), C.byref(h)))
    return _take_host_index(h, n_docs, skiplist_block_size, hit_format, n_fields)


# --------------------------------------------------------------------------- query tree
@dataclass
class XQKeyword:
    term_id: int              # dictionary slot of m_sWord (< 0: not in the dictionary)
    atom_pos: int             # m_iAtomPos
    boost: float = 1.0        # m_fBoost
    field_start: bool = False  # m_bFieldStart: '^word'
    field_end: bool = False    # m_bFieldEnd: 'word$'
    text: str = ""             # m_sWord, when the tree came from parse_query


@dataclass
class XQNode:
    op: int = SPH_QUERY_AND
    children: List["XQNode"] = field(default_factory=list)
    word: Optional[XQKeyword] = None
    field_mask: int = ALL_FIELDS   # m_dSpec.m_dFieldMask (low dword)
    opt: int = 0                   # m_iOpArg
    field_max_pos: int = 0         # m_dSpec.m_iFieldMaxPos: '@field[N] word'
    unit_term: int = -1            # SENTENCE / PARAGRAPH: dictionary slot of MAGIC_WORD_SENTENCE / _PARAGRAPH (< 0: the index holds none)

    @staticmethod
    def keyword(term_id: int, atom_pos: int, field_mask: int = ALL_FIELDS, boost: float = 1.0, field_start: bool = False,
                field_end: bool = False, field_max_pos: int = 0) -> "XQNode":
        return XQNode(SPH_QUERY_TERM, [], XQKeyword(term_id, atom_pos, boost, field_start, field_end), field_mask,
                      field_max_pos=field_max_pos)

    def term_pos(self) -> int:
        """TermPosFilter_e as ExtNode_i::Create derives it (searchnode.cpp:875-878, 1145-1146)."""
        if self.word is None:
            return 0
        if self.field_max_pos:
            return 4
        return (1 if self.word.field_start else 0) | (2 if self.word.field_end else 0)

    @staticmethod
    def AND(*kids: "XQNode") -> "XQNode":
        return XQNode(SPH_QUERY_AND, list(kids))


def parse_query(text: str, field_names: Sequence[str] = (), min_word_len: int = 1,
                lookup: "Union[None, HostIndex, Callable[[str], int]]" = None, transform: bool = False) -> Optional[XQNode]:
    """The extended query syntax -> XQNode tree (mrk_query_parse: the sphinxquery.y grammar + XQParser_t lexer restated in
    csrc/mrk_query.cpp).  lookup resolves a keyword's text to its dictionary slot: a HostIndex opened from files
    (dict=keywords), or a callable; without it every term_id is -1 and XQKeyword.text carries the word.  None = a query
    without keywords.  A syntax error raises MrkError with the reference's wording.  transform=True also applies what every query
    goes through between the parser and the ranker (mrk_parsed_transform: sphTransformExtendedQuery's quorum / NEAR rewrites)."""
    L = lib()
    names = (C.c_char_p * max(1, len(field_names)))(*[f.encode() for f in field_names])
    pq = C.c_void_p()
    check(L.mrk_query_parse(text.encode("utf-8"), names, len(field_names), min_word_len, C.byref(pq)))
    try:
        if transform:
            check(L.mrk_parsed_transform(pq))
        if isinstance(lookup, HostIndex):
            check(L.mrk_parsed_resolve(pq, lookup._owner.h))
        n, root = L.mrk_parsed_n_nodes(pq), L.mrk_parsed_root(pq)
        if root < 0:
            return None
        nodes, kids = L.mrk_parsed_nodes(pq), L.mrk_parsed_children(pq, None)
        built: List[XQNode] = []
        for i in range(n):  # post-order: children come before their parent
            m = nodes[i]
            if m.op == SPH_QUERY_TERM:
                w = L.mrk_parsed_keyword(pq, i).decode("utf-8")
                tid = m.term_id if not callable(lookup) else int(lookup(w))
                x = XQNode.keyword(tid, m.atom_pos, m.field_mask, m.boost, bool(m.term_pos & 1) and m.term_pos != 4,
                                   bool(m.term_pos & 2) and m.term_pos != 4, m.field_max_pos)
                x.word.text = w
            else:
                x = XQNode(m.op, [built[kids[m.first_child + j]] for j in range(m.n_children)], None, m.field_mask, m.opt)
                if m.op in (SPH_QUERY_SENTENCE, SPH_QUERY_PARAGRAPH):
                    w = L.mrk_parsed_keyword(pq, i).decode("utf-8")  # the boundary keyword's text
                    x.unit_term = m.term_id if not callable(lookup) else int(lookup(w))
            built.append(x)
        return built[root]
    finally:
        L.mrk_parsed_free(pq)


@dataclass
class Filter:
    """CSphFilterSettings over an integer attribute of the row-wise storage (sphinx.h:2461-2496), resolved to the
    attribute's locator.  values (ascending) => SPH_FILTER_VALUES, else SPH_FILTER_RANGE over [min, max]."""
    bit_offset: int
    bit_count: int
    values: Optional[Sequence[int]] = None
    min: int = 0
    max: int = 0
    exclude: bool = False
    has_equal_min: bool = True
    has_equal_max: bool = True
    open_left: bool = False
    open_right: bool = False
    fmin: Optional[float] = None  # both set => SPH_FILTER_FLOATRANGE over a 32-bit float attribute
    fmax: Optional[float] = None
    mva_bits: int = 0             # 32 / 64: a multi-value attribute in the blob pool (Segment.set_blobs); bit_offset / bit_count unused
    mva_all: bool = False         # every value of the doc must pass (ALL()) instead of any (ANY())
    blob_attr_id: int = 0         # CSphAttrLocator::m_iBlobAttrId / m_nBlobAttrs
    n_blob_attrs: int = 0

    def as_dict(self) -> dict:  # the oracle's spelling
        d = dict(bit_offset=self.bit_offset, bit_count=self.bit_count, exclude=self.exclude, has_equal_min=self.has_equal_min,
                 has_equal_max=self.has_equal_max, open_left=self.open_left, open_right=self.open_right, min=self.min, max=self.max)
        if self.values is not None:
            d["values"] = list(self.values)
        if self.fmin is not None:
            d["fmin"], d["fmax"] = float(self.fmin), float(self.fmax)
        if self.mva_bits:
            d.update(mva_bits=self.mva_bits, mva_all=self.mva_all, blob_attr_id=self.blob_attr_id, n_blob_attrs=self.n_blob_attrs)
        return d


@dataclass
class Query:
    """CSphQuery fields that reach the ranker, plus the parsed tree."""
    root: XQNode
    ranker: int = SPH_RANK_PROXIMITY_BM25
    max_matches: int = 1000
    field_weights: Optional[Sequence[int]] = None
    index_weight: int = 1
    plain_idf: bool = False
    normalized_tfidf: bool = True
    total_docs: int = 0                       # local_df: m_iTotalDocs override
    local_docs: Optional[Dict[int, int]] = None  # local_df: term id -> global docs
    cutoff: int = 0
    filters: Optional[Sequence["Filter"]] = None  # CSphQuery::m_dFilters, resolved to attribute locators
    weight_filters: Optional[Sequence["Filter"]] = None  # filters on the match weight (m_pWeightFilter); locator fields unused


class _CQueries:
    """Flattened C structs for a list of queries (kept alive while a batch runs)."""

    def __init__(self, queries: Sequence[Query]):
        self.keep = []
        self.arr = (_lib.Query * max(1, len(queries)))()
        for qi, q in enumerate(queries):
            nodes: List[XQNode] = []
            kids_of: List[List[int]] = []

            def walk(n: XQNode) -> int:
                i = len(nodes)
                nodes.append(n)
                kids_of.append([])
                kids_of[i] = [walk(c) for c in n.children]
                return i

            walk(q.root)
            cn = (_lib.Node * len(nodes))()
            flat: List[int] = []
            for i, n in enumerate(nodes):
                cn[i].op, cn[i].n_children, cn[i].first_child = n.op, len(kids_of[i]), len(flat)
                flat.extend(kids_of[i])
                cn[i].field_mask, cn[i].opt = n.field_mask, n.opt
                if n.word is not None:
                    cn[i].term_id, cn[i].atom_pos, cn[i].boost = n.word.term_id, n.word.atom_pos, n.word.boost
                    cn[i].term_pos, cn[i].field_max_pos = n.term_pos(), n.field_max_pos
                else:
                    cn[i].term_id, cn[i].boost = n.unit_term, 1.0
            ch = (C.c_int32 * max(1, len(flat)))(*flat)
            c = self.arr[qi]
            c.nodes, c.n_nodes, c.children, c.root = cn, len(nodes), ch, 0
            c.ranker, c.max_matches = q.ranker, q.max_matches
            if q.field_weights is not None:
                fw = (C.c_int32 * len(q.field_weights))(*q.field_weights)
                c.field_weights, c.n_weights = fw, len(q.field_weights)
                self.keep.append(fw)
            c.index_weight = q.index_weight
            c.plain_idf, c.normalized_tfidf = int(q.plain_idf), int(q.normalized_tfidf)
            c.total_docs_override = q.total_docs
            if q.local_docs:
                ld = (C.c_int64 * len(nodes))(*[
                    int(q.local_docs.get(n.word.term_id, -1)) if n.word is not None else -1 for n in nodes])
                c.local_docs = ld
                self.keep.append(ld)
            c.cutoff = q.cutoff
            def fill(fs):
                fl = (_lib.Filter * len(fs))()
                for i, f in enumerate(fs):
                    fl[i].kind = 0 if f.values is not None else 2 if f.fmin is not None else 1
                    if f.fmin is not None:
                        fl[i].fmin, fl[i].fmax = float(f.fmin), float(f.fmax)
                    fl[i].bit_offset, fl[i].bit_count, fl[i].exclude = f.bit_offset, f.bit_count, int(f.exclude)
                    fl[i].mva_bits, fl[i].mva_all, fl[i].blob_attr_id, fl[i].n_blob_attrs = f.mva_bits, int(f.mva_all), f.blob_attr_id, f.n_blob_attrs
                    fl[i].has_equal_min, fl[i].has_equal_max = int(f.has_equal_min), int(f.has_equal_max)
                    fl[i].open_left, fl[i].open_right = int(f.open_left), int(f.open_right)
                    fl[i].min_value, fl[i].max_value = int(f.min), int(f.max)
                    if f.values is not None:
                        vals = (C.c_int64 * len(f.values))(*sorted(int(v) for v in f.values))
                        self.keep.append(vals)
                        fl[i].values, fl[i].n_values = vals, len(f.values)
                self.keep.append(fl)
                return fl

            if q.filters:
                c.filters, c.n_filters = fill(q.filters), len(q.filters)
            if q.weight_filters:
                c.weight_filters, c.n_weight_filters = fill(q.weight_filters), len(q.weight_filters)
            self.keep += [cn, ch]


@dataclass
class Matches:
    """What the sorter hands back for one query: Flatten() order + GetTotalCount()."""
    rowid: np.ndarray
    weight: np.ndarray
    total_found: int
    status: int = 0


# --------------------------------------------------------------------------- device objects
class Context:
    def __init__(self, device: int = 0):
        import weakref

        self._h = C.c_void_p()
        check(lib().mrk_ctx_create(device, C.byref(self._h)))
        self.device = device
        # segments and batches belong to the context (their destructors run on its submission thread): close() takes the
        # ones still alive down first, so that no late __del__ ever reaches into a destroyed context
        self._children = weakref.WeakSet()

    def set(self, key: str, value: int) -> None:
        check(lib().mrk_ctx_set(self._h, key.encode(), value))

    # --- the shard exchange inside the library (RCCL loaded at run time; include/mrk.h "The shard exchange") ---
    has_comm = False

    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the communicator id to ship to every other rank (MRK_COMM_ID_BYTES)."""
        buf = (C.c_uint8 * 128)()
        check(lib().mrk_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, comm_id: bytes, n_ranks: int, rank: int) -> None:
        """Collective over all ranks (ncclCommInitRank on this context's device)."""
        assert len(comm_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(comm_id)
        check(lib().mrk_comm_init(self._h, buf, n_ranks, rank))
        self.has_comm = True

    def comm_allreduce(self, values: np.ndarray) -> np.ndarray:
        """Sum of an int64 array over the ranks (document frequencies + N: local_df)."""
        a = np.ascontiguousarray(values, dtype=np.int64).copy()
        check(lib().mrk_comm_allreduce_i64(self._h, a.ctypes.data, a.size))
        return a

    def close(self) -> None:
        if self._h:
            for child in list(self._children):
                child.close()
            # (the library refuses while a segment, batch or batcher of the context is alive: MRK_E_INVAL, nothing destroyed)
            check(lib().mrk_ctx_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _segment_desc(hi: HostIndex, rowid_base: int = 0) -> "_lib.SegmentDesc":
    d = _lib.SegmentDesc()
    d.spd, d.spd_len = hi.spd.ctypes.data, hi.spd.size
    d.spp, d.spp_len = hi.spp.ctypes.data, hi.spp.size
    d.spe, d.spe_len = hi.spe.ctypes.data, hi.spe.size
    d.dict, d.n_terms = hi.dict.ctypes.data, len(hi.dict)
    d.total_docs = hi.total_docs
    d.skiplist_block_size, d.hit_format, d.n_fields = hi.skiplist_block_size, hi.hit_format, hi.n_fields
    d.rowid_base = rowid_base
    return d


def validate_index(hi: HostIndex) -> None:
    """The host-only half of Segment(): limits, skiplists and a walk of every doclist (mrk_segment_validate); raises
    MrkError on bytes that must not reach a kernel.  Needs no GPU."""
    check(lib().mrk_segment_validate(C.byref(_segment_desc(hi))))


class Segment:
    """A HostIndex made resident in HBM (+ the device block index built from its skiplists)."""

    def __init__(self, ctx: Context, hi: HostIndex, rowid_base: int = 0):
        self.ctx = ctx
        self.host = hi
        d = _segment_desc(hi, rowid_base)
        self._h = C.c_void_p()
        ctx._children.add(self)
        check(lib().mrk_segment_create(ctx._h, C.byref(d), C.byref(self._h)))

    @property
    def device_bytes(self) -> int:
        return int(lib().mrk_segment_device_bytes(self._h))

    def set_attrs(self, rows: Optional[np.ndarray]) -> None:
        """Upload the row-wise attribute storage (.spa rows: uint32 [n_rows, stride]) for Query.filters; None drops it."""
        if rows is None:
            check(lib().mrk_segment_set_attrs(self._h, None, 0, 0))
            return
        a = np.ascontiguousarray(rows, dtype=np.uint32)
        assert a.ndim == 2
        check(lib().mrk_segment_set_attrs(self._h, a.ctypes.data, a.shape[1], a.shape[0]))

    def set_blobs(self, pool: Optional[np.ndarray], n_blob_attrs: int = 0, rows: Optional[np.ndarray] = None) -> None:
        """Upload the blob pool (.spb bytes / an RT segment's m_dBlobs) for MVA filters; rows = the attribute rows given to
        set_attrs (every row's blob row is bounds-checked against the pool here).  None drops it."""
        if pool is None:
            check(lib().mrk_segment_set_blobs(self._h, None, 0, 0, None, 0, 0))
            return
        b = np.ascontiguousarray(pool, dtype=np.uint8)
        a = np.ascontiguousarray(rows, dtype=np.uint32)
        assert a.ndim == 2
        check(lib().mrk_segment_set_blobs(self._h, b.ctypes.data, b.size, n_blob_attrs, a.ctypes.data, a.shape[1], a.shape[0]))

    def set_dead_rows(self, bitmap: Optional[np.ndarray]) -> None:
        """Install the segment's dead-row map (uint32 words, DeadRowMap_c layout); None clears it."""
        if bitmap is None:
            check(lib().mrk_segment_set_dead_rows(self._h, None, 0))
            return
        bm = np.ascontiguousarray(bitmap, dtype=np.uint32)
        check(lib().mrk_segment_set_dead_rows(self._h, bm.ctypes.data, bm.size * 32))

    def close(self) -> None:
        if self._h:
            lib().mrk_segment_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """Submits batches of queries against a segment; one in flight at a time."""

    def __init__(self, ctx: Context, max_queries: int):
        self.ctx = ctx
        self.max_queries = max_queries
        self._h = C.c_void_p()
        check(lib().mrk_batch_create(ctx._h, max_queries, C.byref(self._h)))
        ctx._children.add(self)
        self._cq = None
        self._n = 0

    def submit(self, seg: Segment, queries: Sequence[Query]) -> None:
        self._cq = _CQueries(queries)
        self._n = len(queries)
        check(lib().mrk_batch_submit(self._h, seg._h, self._cq.arr, self._n))

    def submit_prepared(self, seg: Segment, cq: "_CQueries", n: int) -> None:
        """Re-submit already flattened queries (benchmarks: no Python flattening in the loop)."""
        self._cq, self._n = cq, n
        check(lib().mrk_batch_submit(self._h, seg._h, cq.arr, n))

    def wait(self) -> None:
        check(lib().mrk_batch_wait(self._h))

    def results(self) -> List[Matches]:
        out = []
        r = _lib.Result()
        for i in range(self._n):
            check(lib().mrk_batch_result(self._h, i, C.byref(r)))
            n = r.n
            rowid = np.ctypeslib.as_array(r.rowid, (max(n, 1),))[:n].copy()
            weight = np.ctypeslib.as_array(r.weight, (max(n, 1),))[:n].copy()
            out.append(Matches(rowid, weight, int(r.total_found), int(r.status)))
        return out

    def stats(self) -> dict:
        s = _lib.BatchStats()
        check(lib().mrk_batch_stats_get(self._h, C.byref(s)))
        return {"scan_ms": s.scan_ms, "merge_ms": s.merge_ms, "algo_bytes": int(s.algo_bytes), "n_items": int(s.n_items),
                "dev_bytes": int(s.dev_bytes), "packed": int(s.packed), "n_cands": int(s.n_cands),
                "n_items_bm": int(s.n_items_bm), "plan_ms": float(s.plan_ms), "submit_ms": float(s.submit_ms)}

    def device_results(self):
        """(keys_ptr, counts_ptr, totals_ptr) of the last finished submit, HBM addresses."""
        k, c, t = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().mrk_batch_device_results(self._h, C.byref(k), C.byref(c), C.byref(t)))
        return k.value, c.value, t.value

    def search(self, seg: Segment, queries: Sequence[Query]) -> List[Matches]:
        self.submit(seg, queries)
        self.wait()
        return self.results()

    def close(self) -> None:
        if self._h:
            lib().mrk_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batcher:
    """mrk_batcher: many threads, one query each, common launches (include/mrk.h "the batching front")."""

    def __init__(self, ctx: Context, max_batch: int = 256, max_wait_us: int = 0):
        self.ctx = ctx
        self._h = C.c_void_p()
        check(lib().mrk_batcher_create(ctx._h, max_batch, max_wait_us, C.byref(self._h)))
        ctx._children.add(self)

    def search(self, seg: Segment, query: Query) -> Matches:
        """Blocks until the query's rows are there; callable from any number of threads (the GIL is released in the call)."""
        cq = _CQueries([query])
        cap = max(1, query.max_matches)
        rowid = np.empty(cap, np.uint32)
        weight = np.empty(cap, np.int32)
        r = _lib.Result()
        check(lib().mrk_batcher_search(self._h, seg._h, cq.arr, rowid.ctypes.data, weight.ctypes.data, cap, C.byref(r)))
        return Matches(rowid[: r.n].copy(), weight[: r.n].copy(), int(r.total_found), int(r.status))

    def stats(self) -> dict:
        s = _lib.BatcherStats()
        check(lib().mrk_batcher_stats_get(self._h, C.byref(s)))
        return {"launches": int(s.launches), "queries": int(s.queries), "max_batch": int(s.max_batch)}

    def close(self) -> None:
        if self._h:
            lib().mrk_batcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prepare(queries: Sequence[Query]) -> _CQueries:
    return _CQueries(queries)


def idf(term_docs: int, total_docs: int, plain: bool = False, normalized: bool = True, n_qwords: int = 1,
        boost: float = 1.0) -> float:
    return float(lib().mrk_idf(term_docs, total_docs, int(plain), int(normalized), n_qwords, boost))


__all__ = ["open_rt_ram", "open_rt_segment", "SPH_RANK_PROXIMITY_BM25", "SPH_RANK_BM25", "SPH_RANK_NONE", "SPH_RANK_WORDCOUNT", "SPH_RANK_PROXIMITY",
           "SPH_RANK_MATCHANY", "SPH_RANK_FIELDMASK", "SPH_RANK_SPH04",
           "parse_query", "SPH_QUERY_TERM", "SPH_QUERY_AND", "SPH_QUERY_OR", "SPH_QUERY_MAYBE", "SPH_QUERY_ANDNOT", "SPH_QUERY_PHRASE", "SPH_QUERY_PROXIMITY", "SPH_QUERY_QUORUM", "SPH_QUERY_BEFORE", "SPH_QUERY_NEAR", "SPH_QUERY_NOTNEAR", "SPH_QUERY_SENTENCE", "SPH_QUERY_PARAGRAPH",
           "SPH_HIT_FORMAT_PLAIN", "SPH_HIT_FORMAT_INLINE", "ALL_FIELDS", "DICT_DTYPE", "HostIndex", "open_index", "index_from_hits", "synth_index", "XQKeyword", "XQNode", "Query", "Filter", "Matches", "Context",
           "Segment", "Batch", "Batcher", "prepare", "idf", "MrkError", "validate_index", "pair_stats"]

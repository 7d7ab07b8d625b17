"""Pins the oracle (oracle/cpu_ref.c) to the reference's own golden vectors.

Sources (all under /root/reference, read as data while writing these tests; nothing is
read at run time):
  * doc/internals-index-format.txt:44-60      VLB byte examples
  * src/gtests/gtests_rtstuff.cpp:257-335     RTN.WeightBoundary  (weight == 1500)
  * test/test_037/test.xml + model.bin        rankers bm25 / none / proximity_bm25
  * test/test_019/test.xml + model.bin        extended queries (default ranker)
  * test/test_322/test.xml + model.bin        field weights (incl. 0 and negative)
Corpora are the rows of those tests' <db_insert> blocks; expected docid:weight pairs are
the values decoded from model.bin.
"""
import json
import os

import numpy as np
import pytest

from helpers import mini_index

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json"),
                        encoding="utf-8"))
def _field_text(f):  # a field the reference builds in a loop is stored as {"runs": [[text, count], ...]}
    return f if isinstance(f, str) else "".join(t * n for t, n in f["runs"])


for _c in GOLDEN["corpora"].values():  # corpora the reference builds in a loop are stored as (count, row) runs
    if "docs_spec" in _c:
        _c["docs"] = [[_field_text(f) for f in run["fields"]] for run in _c["docs_spec"] for _ in range(run["count"])]


# ------------------------------------------------------------------ VLB bytes
def test_vlb_golden_bytes(orc):
    assert orc.zip_bytes(0x12345) == bytes([0x84, 0xC6, 0x45])
    assert orc.zip_bytes(0) == b"\x00"
    assert orc.zip_bytes(127) == b"\x7f"
    assert orc.zip_bytes(128) == b"\x81\x00"
    assert orc.zip_bytes(0xFFFFFFFF) == bytes([0x8F, 0xFF, 0xFF, 0xFF, 0x7F])


def test_hitlist_golden_bytes(orc):
    # hits 2, 16777224, 16777229 -> 02 88 80 80 06 05 00 (one word, one doc, three hits)
    hits = [2, 16777224, 16777229]
    idx = orc.build_index(np.array([1, 1, 1], np.uint64), np.array([0, 0, 0], np.uint32),
                          np.array(hits, np.uint32), total_docs=1, n_terms=1)
    # .spp = dummy byte + the hitlist
    assert bytes(idx.spp) == bytes([0x01, 0x02, 0x88, 0x80, 0x80, 0x06, 0x05, 0x00])
    rowid, fields, nhits, hp = idx.decode_doclist(0)
    assert list(rowid) == [0] and list(nhits) == [3] and list(fields) == [0b11]
    assert idx.decode_hits(hp[0]) == hits


# ------------------------------------------------------------------ helpers
def run(orc, idx, root, ranker, ids, **kw):
    r = orc.search(idx, root, ranker=ranker, max_matches=1000, **kw)
    return [(ids[int(rid)], int(w)) for rid, w in zip(r.rowid, r.weight)], r


# ------------------------------------------------------------------ gtest WeightBoundary
def test_rt_weight_boundary(orc):
    docs = [["If I were a cat...", "We are the greatest cat"]]
    idx, v = mini_index(orc, docs)
    got, _ = run(orc, idx, orc.term(v["cat"], 1, field_mask=0b01), orc.RANK_PROXIMITY_BM25, [1])
    assert got == [(1, 1500)]


# ------------------------------------------------------------------ test_037
T037 = [["зимние шины диски чего то тут зимние шины", ""],
        ["test doc two", "second stupid test document with random content"]] + [["filler", "filler"]] * 8


@pytest.mark.parametrize("ranker,expect", [("RANK_PROXIMITY_BM25", 2800), ("RANK_BM25", 1800), ("RANK_NONE", 1)])
def test_037_phrase_rankers(orc, ranker, expect):
    idx, v = mini_index(orc, T037)
    ids = list(range(1, 11))
    root = orc.op(orc.OP_PHRASE, orc.term(v["зимние"], 1), orc.term(v["шины"], 2))
    got, r = run(orc, idx, root, getattr(orc, ranker), ids)
    assert got == [(1, expect)]
    assert r.total_found == 1


def test_037_title_test_bm25(orc):
    idx, v = mini_index(orc, T037)
    got, _ = run(orc, idx, orc.term(v["test"], 1, field_mask=0b01), orc.RANK_BM25, list(range(1, 11)))
    assert got == [(2, 1800)]


# ------------------------------------------------------------------ test_019
T019_IDS = [111, 222, 333, 444, 555, 666, 777, 888, 999, 901, 902, 903, 910]
T019 = [["", "basic query"],
        ["", "phrase query on steroids"],
        ["sample program", 'this is a test program that prints out "hello world" to the console'],
        ["", "china 吐我"],
        ["sample program two", "something written in basic | canon ef 16-35 lens"],
        ["sample program three", "something written in perl"],
        ["", "77 lies multiplied by 77"],
        ["", "agent 0077"],
        ["", "1234567812345678"],
        ["aaa", "aaa"],
        ["aaa", ""],
        ["", "aaa"],
        ["", "wordbefore\0\0wordafter"]]
TITLE, BODY = 0b01, 0b10


def t019(orc):
    return mini_index(orc, T019, min_word_len=2)


def test_019_basic_query(orc):
    idx, v = t019(orc)
    root = orc.op(orc.OP_AND, orc.term(v["basic"], 1), orc.term(v["query"], 2))
    got, _ = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(111, 2654)]


def test_019_phrase_query(orc):
    idx, v = t019(orc)
    root = orc.op(orc.OP_PHRASE, orc.term(v["phrase"], 1), orc.term(v["query"], 2))
    got, _ = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(222, 2687)]


def test_019_field_limits(orc):
    idx, v = t019(orc)
    root = orc.op(orc.OP_AND, orc.term(v["sample"], 1, field_mask=TITLE), orc.term(v["world"], 2, field_mask=BODY))
    got, _ = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(333, 2666)]


def test_019_or(orc):
    idx, v = t019(orc)
    root = orc.op(orc.OP_OR, orc.term(v["basic"], 1), orc.term(v["china"], 2))
    got, _ = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(444, 1610), (111, 1577), (555, 1577)]


def test_019_phrase_or_term(orc):
    idx, v = t019(orc)
    root = orc.op(orc.OP_OR, orc.op(orc.OP_PHRASE, orc.term(v["test"], 1), orc.term(v["program"], 2)),
                  orc.term(v["basic"], 3))
    got, _ = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(333, 2624), (111, 1551), (555, 1551)]


def test_019_andnot(orc):
    idx, v = t019(orc)
    root = orc.op(orc.OP_ANDNOT, orc.term(v["sample"], 1, field_mask=TITLE), orc.term(v["basic"], 2, field_mask=BODY))
    got, _ = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(333, 1555), (666, 1555)]


def test_019_single_word_77(orc):
    idx, v = t019(orc)
    got, _ = run(orc, idx, orc.term(v["77"], 1), orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(777, 1803)]


# ------------------------------------------------------------------ test_322
T322_IDS = [1, 2, 3, 100]
T322 = [["|sample program", "|program flow direct", "|sample program flow"],
        ["|one sample program", "|program rev flow", "|one rev flow"],
        ["|sample two program", "|sub program flow", "|two sub program"],
        ["unsigned", "", ""]]


@pytest.mark.parametrize("spam,expect", [
    (1, [(1, 7415), (3, 6426), (2, 4421)]),
    (10, [(1, 25415), (3, 15426), (2, 13421)]),
    (0, [(3, 5426), (1, 5415), (2, 3421)]),
    (-2, [(3, 3426), (2, 1421), (1, 1415)]),
    (-10, [(3, -4574), (2, -6579), (1, -14585)]),
])
def test_322_field_weights(orc, spam, expect):
    idx, v = mini_index(orc, T322)
    root = orc.op(orc.OP_AND, orc.term(v["program"], 1), orc.term(v["flow"], 2))
    got, _ = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T322_IDS, field_weights=[1, 2, spam])
    assert got == expect


def test_dead_rows_skip_the_sorter(orc):
    """MatchExtended (sphinx.cpp:12213-12217): a dead row is neither ranked into the queue nor counted."""
    idx, v = t019(orc)
    root = orc.op(orc.OP_OR, orc.term(v["basic"], 1), orc.term(v["china"], 2))
    got, r = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(444, 1610), (111, 1577), (555, 1577)] and r.total_found == 3
    dead = np.zeros(1, np.uint32)
    dead[0] |= 1 << T019_IDS.index(111)
    idx.dead_rows = dead
    got, r = run(orc, idx, root, orc.RANK_PROXIMITY_BM25, T019_IDS)
    assert got == [(444, 1610), (555, 1577)] and r.total_found == 2


# ------------------------------------------------------------------ tests/golden/reference_vectors.json, every case
def _golden_tree(orc, v, q):
    if "word" in q:
        tp = {None: 0, "start": orc.TERMPOS_START, "end": orc.TERMPOS_END, "startend": orc.TERMPOS_STARTEND, "limit": orc.TERMPOS_LIMIT}[q.get("tp")]
        return orc.term(v.get(q["word"], -1), q["pos"], field_mask=q["mask"], term_pos=tp, field_max_pos=q.get("max_pos", 0))  # -1: keyword not in the dictionary
    return orc.op(getattr(orc, "OP_" + q["op"].upper()), *[_golden_tree(orc, v, k) for k in q["kids"]], field_mask=q["mask"],
                  opt=q.get("opt", 0), unit_term=v.get(q["unit"], -1) if "unit" in q else -1)


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=[c["name"] for c in GOLDEN["cases"]])
def test_reference_vectors(orc, case):
    corpus = GOLDEN["corpora"][case["corpus"]]
    idx, v = mini_index(orc, corpus["docs"], min_word_len=corpus["min_word_len"], index_sp=bool(corpus.get("index_sp")))
    kw = {"field_weights": case["field_weights"]} if "field_weights" in case else {}
    if case.get("plain_idf"):
        kw["plain_idf"] = True
    if "total_docs" in case:  # local_df: the statistics of all the indexes searched together
        kw["total_docs_override"] = case["total_docs"]
        kw["local_docs"] = {v[w]: n for w, n in case["local_docs"].items() if w in v}
    got, r = run(orc, idx, _golden_tree(orc, v, case["query"]), getattr(orc, "RANK_" + case["ranker"].upper()), corpus["ids"], **kw)
    if "expect_row" in case:  # the reference's statement filters by id: one row must (not) be among the matches
        assert (case["expect_row"][0] in [i for i, _ in got]) == case["expect_row"][1]
    elif "expect_ids" in case:  # the reference's test lists the matching rows only
        assert sorted(i for i, _ in got) == sorted(case["expect_ids"])
        if "expect_weights" in case:  # rows listed in id order with their weights
            assert {str(i): w for i, w in got} == case["expect_weights"]
    else:
        assert got[:case.get("limit", len(got))] == [tuple(x) for x in case["expect"]]
    if "total_found" in case:
        assert r.total_found == case["total_found"]


def test_reference_byte_vectors(orc):
    for value, enc in GOLDEN["vlb_bytes"]["cases"]:
        assert orc.zip_bytes(value) == bytes(enc)
    hits = GOLDEN["hitlist_bytes"]["hits"]
    idx = orc.build_index(np.ones(len(hits), np.uint64), np.zeros(len(hits), np.uint32), np.array(hits, np.uint32),
                          total_docs=1, n_terms=1)
    assert bytes(idx.spp) == bytes(GOLDEN["hitlist_bytes"]["spp"])


# ------------------------------------------------------------------ ExtQuorum_c proper: no golden in the reference tree
def test_quorum_node_against_set_arithmetic(orc):
    """'"a b c d"/N' with 1 < N < words is a real ExtQuorum_c (searchnode.cpp:4342-4617).  The reference holds no golden
    for it (its test_019 quorum queries take the OR / AND rewrites), so the restatement is checked against what the
    operator means: a doc matches iff it holds at least N of the keywords; and N = words - 1 over two-word subsets."""
    from helpers import synth_postings

    rng = np.random.default_rng(11)
    n_docs, probs = 4000, [0.5, 0.3, 0.2, 0.1, 0.05]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=2, max_pos=30)
    idx = orc.build_index(W, R, H, total_docs=n_docs, n_fields=2, n_terms=len(probs))
    have = np.zeros((len(probs), n_docs), bool)
    for t in range(len(probs)):
        have[t, idx.decode_doclist(t)[0]] = True
    for words in ([0, 1, 2], [4, 2, 0, 1], [0, 1, 2, 3, 4]):
        for thr in range(2, len(words)):
            root = orc.op(orc.OP_QUORUM, *[orc.term(t, i + 1) for i, t in enumerate(words)], opt=thr)
            for ranker in (orc.RANK_BM25, orc.RANK_PROXIMITY_BM25, orc.RANK_WORDCOUNT):
                r = orc.search(idx, root, ranker=ranker, max_matches=n_docs)
                want = np.nonzero(have[words].sum(0) >= thr)[0]
                assert r.total_found == len(want)
                assert sorted(int(x) for x in r.rowid) == [int(x) for x in want]

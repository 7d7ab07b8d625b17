"""Field masks beyond 32 fields in the oracle (ISphQword::CollectHitMask, sphinxsearch.cpp:50-58; ExtTerm_T / ExtMultiAnd_T
FitsFields, searchnode.cpp:1925-1939, 2727-2747), pinned by the reference's own test for it: test/test_183 (256 fields; the rows its
model.bin lists, tests/golden/wide_fields_vectors.json).  The device path covers <= 32 fields and says so: tests/test_gpu_lifecycle.py."""
import json
import os

import numpy as np
import pytest

from helpers import make_hits

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wide_fields_vectors.json"), encoding="utf-8"))


def build(orc, corpus):
    c = GOLDEN["corpora"][corpus]
    docs = [[d.get(str(f), "") for f in range(c["n_fields"])] for d in c["docs"]]
    wordid, rowid, hitpos, vocab = make_hits(docs, 1)
    idx = orc.build_index(wordid, rowid, hitpos, total_docs=len(docs), n_fields=c["n_fields"], n_terms=len(vocab))
    return idx, vocab, c["ids"]


def mask_of(fields):
    if fields is None:
        return orc_all()
    m = 0
    for f in fields:
        m |= 1 << f
    return m


def orc_all():
    from oracle import oracle

    return oracle.ALL_FIELDS


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=[c["name"] for c in GOLDEN["cases"]])
def test_reference_rows_for_wide_field_limits(orc, case):
    idx, vocab, ids = build(orc, case["corpus"])
    terms = [orc.term(vocab.get(t["word"], -1), t["pos"], mask_of(t["fields"])) for t in case["query"]]
    root = terms[0] if len(terms) == 1 else orc.op(orc.OP_AND, *terms)
    for ranker in (orc.RANK_PROXIMITY_BM25, orc.RANK_BM25, orc.RANK_NONE):
        res = orc.search(idx, root, ranker=ranker, max_matches=100)
        got = sorted(ids[r] for r in res.rowid)
        assert got == sorted(case["expect_ids"]), (case["name"], ranker, got)
        assert res.total_found == len(case["expect_ids"])


def test_mask_comes_from_the_hits_beyond_32_fields(orc):
    """A doc whose only occurrence lies in field 37 has an empty low dword in its doclist entry: only the hits say where it is."""
    idx, vocab, ids = build(orc, "test_183_rt40")
    rowid, fields32, nhits, hp = idx.decode_doclist(vocab["kw37"])
    assert list(fields32) == [0, 0, 0]  # GetMask32: field 36 is not in it (inline single hits carry the field in the entry itself)
    # a word in a low field and in a high field of one doc: limited to the high field it still matches, to another high field it does not
    docs = [[""] * 40]
    docs[0][3], docs[0][35] = "alpha beta", "beta alpha alpha"
    wordid, rowid, hitpos, v = make_hits(docs, 1)
    ix = orc.build_index(wordid, rowid, hitpos, total_docs=1, n_fields=40, n_terms=len(v))
    assert orc.search(ix, orc.term(v["alpha"], 1, 1 << 35), max_matches=10).total_found == 1
    assert orc.search(ix, orc.term(v["alpha"], 1, 1 << 34), max_matches=10).total_found == 0
    assert orc.search(ix, orc.term(v["alpha"], 1, 1 << 3), max_matches=10).total_found == 1
    # the hits that travel on are the queried field's only: a phrase limited to field 35 sees 'beta alpha', not field 3's 'alpha beta'
    ph = orc.op(orc.OP_PHRASE, orc.term(v["alpha"], 1), orc.term(v["beta"], 2), field_mask=1 << 35)
    assert orc.search(ix, ph, max_matches=10).total_found == 0
    ph2 = orc.op(orc.OP_PHRASE, orc.term(v["beta"], 1), orc.term(v["alpha"], 2), field_mask=1 << 35)
    assert orc.search(ix, ph2, max_matches=10).total_found == 1

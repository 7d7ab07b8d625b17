"""CPU tests: the product's format writer / synthetic generator (libmrk.so host code)
against the oracle's CSphHitBuilder restatement and reader. No GPU calls."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_hits, synth_postings
from test_oracle_golden import T019, T322


def test_lib_exports_every_declared_symbol():
    import re, os
    from manticoresearch_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(os.path.dirname(_lib._HERE), "include", "mrk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mrk_[a-z_0-9]+)\s*\(", hdr))
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for n in declared:
        assert hasattr(L, n)


@pytest.mark.parametrize("block", [32, 128])
@pytest.mark.parametrize("inline", [1, 0])
def test_writer_bytes_match_oracle_writer(orc, block, inline):
    import manticoresearch_amd as m
    rng = np.random.default_rng(7)
    W, R, H = synth_postings(rng, 3000, [0.5, 0.2, 0.05, 0.01, 0.9], end_markers=True)
    a = orc.build_index(W, R, H, total_docs=3000, skiplist_block_size=block, inline_hits=inline, n_terms=5)
    b = m.index_from_hits(W, R, H, n_terms=5, total_docs=3000, skiplist_block_size=block, hit_format=inline)
    assert bytes(a.spd) == bytes(b.spd)
    assert bytes(a.spp) == bytes(b.spp)
    assert bytes(a.spe) == bytes(b.spe)
    for f in ("doclist_off", "doclist_len", "docs", "hits"):
        assert (a.dict[f] == b.dict[f]).all(), f
    big = a.dict["docs"] > block
    assert (a.dict["skiplist_off"][big] == b.dict["skiplist_off"][big]).all()


@pytest.mark.parametrize("docs", [T019, T322])
def test_writer_bytes_match_on_reference_corpora(orc, docs):
    import manticoresearch_amd as m
    W, R, H, vocab = make_hits(docs, min_word_len=2)
    a = orc.build_index(W, R, H, total_docs=len(docs), n_terms=len(vocab))
    b = m.index_from_hits(W, R, H, n_terms=len(vocab), total_docs=len(docs))
    assert bytes(a.spd) == bytes(b.spd) and bytes(a.spp) == bytes(b.spp) and bytes(a.spe) == bytes(b.spe)


def test_synth_index_decodes_with_oracle_reader(orc):
    import manticoresearch_amd as m
    probs = [0.3, 0.05, 0.002, 0.6]
    hi = m.synth_index(20000, probs, seed=123, skiplist_block_size=32, end_markers=True, n_threads=2)
    oi = orc.Index(hi.spd, hi.spp, hi.spe, hi.dict.view(orc.DICT_DTYPE), hi.total_docs, 32, 1, 2)
    for t, p in enumerate(probs):
        rowid, fields, hits, hp = oi.decode_doclist(t)
        n = len(rowid)
        assert n == hi.dict[t]["docs"]
        assert abs(n - p * 20000) < 6 * np.sqrt(p * 20000) + 5
        assert (np.diff(rowid.astype(np.int64)) > 0).all() and rowid[-1] < 20000
        assert hits.sum() == hi.dict[t]["hits"]
        for i in range(0, n, max(1, n // 50)):
            hl = oi.decode_hits(hp[i])
            assert len(hl) == hits[i] and hl == sorted(hl)
            mask = 0
            for h in hl:
                mask |= 1 << (h >> 24)
            assert mask == fields[i]
        b, o, hb = oi.decode_skiplist(t)
        if n > 32:
            assert len(b) == n // 32 and b[0] == 0
            assert (rowid[32::32][: len(b) - 1] >= b[1:]).all()
    # determinism
    hi2 = m.synth_index(20000, probs, seed=123, skiplist_block_size=32, end_markers=True, n_threads=1)
    assert bytes(hi.spd) == bytes(hi2.spd) and bytes(hi.spp) == bytes(hi2.spp)


def test_synth_end_markers_owned_by_positions(orc):
    """mrk_synth_params::end_markers = 2: the field-end flag sits on the hit at the field's LAST POSITION in that doc, whatever the word --
    every word at that position carries it, no hit lies beyond it (what the reference's indexer writes, sphinx.cpp:22424-22430; the
    tighter weight bound of the device path rests on it: csrc/mrk_kprune.h).  end_markers = 1 flags each word's own last hit instead."""
    import manticoresearch_amd as m
    probs = [0.8, 0.7, 0.6, 0.5]
    n_docs = 4000

    def collect(mode):
        hi = m.synth_index(n_docs, probs, seed=77, n_fields=3, max_pos=6, hit_format=0, end_markers=mode, n_threads=2)
        oi = orc.Index(hi.spd, hi.spp, hi.spe, hi.dict.view(orc.DICT_DTYPE), hi.total_docs, hi.skiplist_block_size, 0, 3)
        seen = {}  # (rowid, field) -> {pos: set of flags}
        for t in range(len(probs)):
            rowid, fields, hits, hp = oi.decode_doclist(t)
            for r, h in zip(rowid, hp):
                for x in oi.decode_hits(h):
                    seen.setdefault((int(r), x >> 24), {}).setdefault(x & 0x7FFFFF, set()).add((x >> 23) & 1)
        return seen

    seen = collect(2)
    n_flagged = n_shared = 0
    for poss in seen.values():
        top = max(poss)
        for pos, flags in poss.items():
            assert len(flags) == 1, "two words at one position differ in the end flag"
            if 1 in flags:
                assert pos == top, "a hit lies beyond the flagged position"
                n_flagged += 1
            n_shared += 1
    assert n_flagged > 1000 and n_shared > 10000
    # (mode 1 does flag per word: at crowded positions some position must carry both values)
    assert any(len(flags) == 2 for poss in collect(1).values() for flags in poss.values())


def test_index_from_hits_checks_its_arguments():
    """Unsorted hits, word ids outside 1..n_terms, a zero skiplist block size or an unknown hit format come back as MRK_E_INVAL."""
    import manticoresearch_amd as m
    from manticoresearch_amd._lib import MrkError
    W = np.array([1, 1, 2], np.uint64)
    R = np.array([0, 1, 0], np.uint32)
    H = np.array([1, 1, 1], np.uint32)
    assert m.index_from_hits(W, R, H, n_terms=2, total_docs=2).dict["docs"].tolist() == [2, 1]
    for kw, why in ((dict(W=np.array([2, 1, 1], np.uint64)), "sorted"), (dict(W=np.array([1, 1, 3], np.uint64)), "outside"),
                    (dict(W=np.array([0, 1, 2], np.uint64)), "outside"), (dict(block=0), "skiplist_block_size"), (dict(fmt=2), "hit_format")):
        with pytest.raises(MrkError, match=why):
            m.index_from_hits(kw.get("W", W), R, H, n_terms=2, total_docs=2, skiplist_block_size=kw.get("block", 32), hit_format=kw.get("fmt", 1))

"""Attribute filters in the oracle (EarlyReject): checked against plain numpy arithmetic over the attribute rows.
The reference holds no weight-bearing golden that isolates this step; its semantics are the comparison operators of
sphinxfilter.h:130-143 and the value search of sphinxfilter.cpp:69-91, restated in oracle/cpu_ref.c::filters_pass."""
import numpy as np

from helpers import synth_postings


def test_filters_against_numpy(orc):
    rng = np.random.default_rng(99)
    n_docs = 5000
    W, R, H = synth_postings(rng, n_docs, [0.6, 0.3], n_fields=2, max_pos=10)
    idx = orc.build_index(W, R, H, total_docs=n_docs, n_fields=2, n_terms=2)
    rows = np.zeros((n_docs, 4), np.uint32)
    rows[:, 0] = rng.integers(0, 100, n_docs)
    rows[:, 1] = rng.integers(0, 1 << 12, n_docs)  # bit field: 5 bits at offset 3
    big = rng.integers(-500, 500, n_docs).astype(np.int64)
    rows[:, 2], rows[:, 3] = (big.view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32), (big.view(np.uint64) >> np.uint64(32)).astype(np.uint32)
    idx.attrs = rows
    have = np.zeros(n_docs, bool)
    have[idx.decode_doclist(0)[0]] = True
    gid, bits = rows[:, 0].astype(np.int64), ((rows[:, 1] >> 3) & 31).astype(np.int64)
    cases = [
        (dict(bit_offset=0, bit_count=32, values=[3, 50, 99]), np.isin(gid, [3, 50, 99])),
        (dict(bit_offset=0, bit_count=32, values=[3, 50, 99], exclude=True), ~np.isin(gid, [3, 50, 99])),
        (dict(bit_offset=0, bit_count=32, min=10, max=20), (gid >= 10) & (gid <= 20)),
        (dict(bit_offset=0, bit_count=32, min=10, max=20, has_equal_min=False), (gid > 10) & (gid <= 20)),
        (dict(bit_offset=0, bit_count=32, min=10, max=20, has_equal_max=False), (gid >= 10) & (gid < 20)),
        (dict(bit_offset=0, bit_count=32, min=10, max=20, open_left=True), gid <= 20),
        (dict(bit_offset=0, bit_count=32, min=10, max=20, open_right=True, has_equal_min=False), gid > 10),
        (dict(bit_offset=35, bit_count=5, values=[0, 7, 31]), np.isin(bits, [0, 7, 31])),
        (dict(bit_offset=64, bit_count=64, min=-100, max=-1), (big >= -100) & (big <= -1)),
        (dict(bit_offset=64, bit_count=64, min=-100, max=100, exclude=True), ~((big >= -100) & (big <= 100))),
    ]
    for f, mask in cases:
        r = orc.search(idx, orc.term(0, 1), ranker=orc.RANK_BM25, max_matches=n_docs, filters=[f])
        want = np.nonzero(have & mask)[0]
        assert r.total_found == len(want), f
        assert sorted(int(x) for x in r.rowid) == [int(x) for x in want], f
    # two filters: both must pass
    r = orc.search(idx, orc.term(0, 1), ranker=orc.RANK_NONE, max_matches=n_docs, filters=[cases[2][0], cases[8][0]])
    want = np.nonzero(have & cases[2][1] & cases[8][1])[0]
    assert r.total_found == len(want) and sorted(int(x) for x in r.rowid) == [int(x) for x in want]
    # weights of the rows that pass are those of the unfiltered search
    full = orc.search(idx, orc.term(0, 1), ranker=orc.RANK_BM25, max_matches=n_docs)
    wmap = dict(zip(full.rowid.tolist(), full.weight.tolist()))
    r = orc.search(idx, orc.term(0, 1), ranker=orc.RANK_BM25, max_matches=n_docs, filters=[cases[0][0]])
    assert all(wmap[int(d)] == int(w) for d, w in zip(r.rowid, r.weight))

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


def mva_rows(rng, n_docs, widths=(32, 64)):
    """attribute rows [id lo, id hi, blob locator lo, hi] + a blob pool holding one MVA32 and one MVA64 per row (sorted, unique,
    often empty); -> rows, pool, python lists of the values"""
    from helpers import build_blob_pool

    v32 = [sorted(set(int(x) for x in rng.integers(0, 40, int(rng.integers(0, 6))))) if rng.random() < 0.85 else [] for _ in range(n_docs)]
    v64 = [sorted(set(int(x) for x in rng.integers(-20, 20, int(rng.integers(0, 5))) * (1 << 33))) if rng.random() < 0.85 else [] for _ in range(n_docs)]
    blobs = [[b"".join(int(x).to_bytes(4, "little") for x in a), b"".join(int(x).to_bytes(8, "little", signed=True) for x in b)] for a, b in zip(v32, v64)]
    pool, offs = build_blob_pool(blobs)
    rows = np.zeros((n_docs, 4), np.uint32)
    rows[:, 0] = np.arange(n_docs) + 1
    rows[:, 2] = np.array(offs, np.uint64) & 0xFFFFFFFF
    rows[:, 3] = np.array(offs, np.uint64) >> 32
    return rows, pool, v32, v64


def mva_expect(kind, vals, f):
    """what the reference's MvaEval_* say for one doc (sphinxfilter.h:160-253), written out plainly"""
    if not vals:
        ok = False
    elif kind == "values_any":
        ok = any(v in f["values"] for v in vals)
    elif kind == "values_all":
        ok = all(v in f["values"] for v in vals)
    elif kind == "range_any":
        lo, hi = f["min"], f["max"]
        if lo in vals:  # found the minimum itself: passes when the bound is inclusive -- or when ANY larger value exists (not tested against max)
            ok = f.get("has_equal_min", True) or vals.index(lo) + 1 < len(vals)
        else:
            nxt = [v for v in vals if v > lo]
            ok = bool(nxt) and (nxt[0] <= hi if f.get("has_equal_max", True) else nxt[0] < hi)
    else:
        lo, hi = f["min"], f["max"]
        ok = (vals[0] >= lo if f.get("has_equal_min", True) else vals[0] > lo) and (vals[-1] <= hi if f.get("has_equal_max", True) else vals[-1] < hi)
    return (not ok) if f.get("exclude") else ok


def test_mva_filters_against_the_definitions(orc):
    rng = np.random.default_rng(7)
    n_docs = 3000
    W, R, H = synth_postings(rng, n_docs, [0.7], n_fields=2, max_pos=10)
    idx = orc.build_index(W, R, H, total_docs=n_docs, n_fields=2, n_terms=1)
    rows, pool, v32, v64 = mva_rows(rng, n_docs)
    idx.attrs, idx.blobs = rows, pool
    have = np.zeros(n_docs, bool)
    have[idx.decode_doclist(0)[0]] = True
    B = 1 << 33
    cases = [("values_any", 0, dict(values=[3, 17, 39])), ("values_all", 0, dict(values=[1, 2, 3, 4, 5, 6, 7, 8], mva_all=True)),
             ("values_any", 0, dict(values=[5], exclude=True)), ("range_any", 0, dict(min=10, max=15)),
             ("range_any", 0, dict(min=10, max=15, has_equal_min=False)), ("range_any", 0, dict(min=10, max=15, has_equal_max=False)),
             ("range_all", 0, dict(min=5, max=30, mva_all=True)), ("range_all", 0, dict(min=5, max=30, mva_all=True, has_equal_min=False, exclude=True)),
             ("values_any", 1, dict(values=[-3 * B, 0, 7 * B])), ("values_all", 1, dict(values=[-2 * B, -B, 0, B, 2 * B], mva_all=True)),
             ("range_any", 1, dict(min=-5 * B, max=-B)), ("range_all", 1, dict(min=-10 * B, max=3 * B, mva_all=True))]
    for kind, attr, f in cases:
        flt = dict(bit_offset=0, bit_count=0, mva_bits=(32, 64)[attr], blob_attr_id=attr, n_blob_attrs=2, **f)
        r = orc.search(idx, orc.term(0, 1), ranker=orc.RANK_NONE, max_matches=n_docs, filters=[flt])
        vals = (v32, v64)[attr]
        want = [d for d in range(n_docs) if have[d] and mva_expect(kind, vals[d], f)]
        assert r.total_found == len(want) and sorted(int(x) for x in r.rowid) == want, (kind, attr, f)

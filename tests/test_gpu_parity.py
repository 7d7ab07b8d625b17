"""GPU parity: the HIP path (through the C-ABI) against the oracle on the same index bytes.

Bit-exact bar: identical rowid lists (order included), identical int weights, identical
total_found.  Run on the GPU box: python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from helpers import make_hits, mini_index, synth_postings
from test_oracle_golden import T019, T019_IDS, T037, T322

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["packed", "vlb"])
def dev(request):
    """Both device paths: packed doclists (load-time transcode, the default) and VLB-direct."""
    import manticoresearch_amd as m

    ctx = m.Context(0)
    ctx.set("path", 0 if request.param == "packed" else 1)
    ctx._path = 0 if request.param == "packed" else 1
    batch = m.Batch(ctx, 256)
    yield m, ctx, batch
    batch.close()
    ctx.close()


def orc_index_of(orc, hi):
    return orc.Index(hi.spd, hi.spp, hi.spe, hi.dict.view(orc.DICT_DTYPE), hi.total_docs, hi.skiplist_block_size,
                     hi.hit_format, hi.n_fields)


def to_orc(orc, q):
    def conv(n):
        if n.word is not None:
            return orc.term(n.word.term_id, n.word.atom_pos, n.field_mask, n.word.boost, term_pos=n.term_pos(), field_max_pos=n.field_max_pos)
        return orc.op(n.op, *[conv(c) for c in n.children], field_mask=n.field_mask, opt=n.opt, unit_term=n.unit_term)

    return orc.FlatQuery(conv(q.root), ranker=q.ranker, max_matches=q.max_matches, field_weights=q.field_weights,
                         index_weight=q.index_weight, plain_idf=q.plain_idf, normalized_tfidf=q.normalized_tfidf,
                         total_docs_override=q.total_docs, local_docs=q.local_docs, cutoff=q.cutoff,
                         filters=[f.as_dict() for f in q.filters] if q.filters else None,
                         weight_filters=[f.as_dict() for f in q.weight_filters] if q.weight_filters else None)


def check_batch(orc, dev, hi, queries, rowid_base=0):
    m, ctx, batch = dev
    seg = m.Segment(ctx, hi, rowid_base=rowid_base)
    oi = orc_index_of(orc, hi)
    try:
        for i in range(0, len(queries), batch.max_queries):
            chunk = queries[i:i + batch.max_queries]
            got = batch.search(seg, chunk)
            if hi.n_fields <= 8 and ctx_path(ctx) == 0:
                assert batch.stats()["packed"] == 1
            for q, g in zip(chunk, got):
                want = to_orc(orc, q).run(oi)
                assert g.status == 0
                assert g.total_found == want.total_found, (g.total_found, want.total_found)
                assert len(g.rowid) == len(want.rowid)
                assert (g.rowid == want.rowid).all(), (g.rowid[:10], want.rowid[:10])
                assert (g.weight == want.weight).all(), (g.weight[:10], want.weight[:10])
    finally:
        seg.close()


def ctx_path(ctx):
    return getattr(ctx, "_path", 0)


def kw(m, t, pos, mask=0xFFFFFFFF, boost=1.0):
    return m.XQNode.keyword(t, pos, mask, boost)


# ------------------------------------------------------------------ reference corpora
def test_reference_corpora_single_and_and(orc, dev):
    m = dev[0]
    for docs, mwl in ((T019, 2), (T037, 1), (T322, 1)):
        W, R, H, v = make_hits(docs, mwl)
        nf = max(len(d) for d in docs)
        hi = m.index_from_hits(W, R, H, n_terms=len(v), total_docs=len(docs), n_fields=nf)
        qs = []
        for t in range(len(v)):
            for rk in (m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_PROXIMITY_BM25):
                qs.append(m.Query(kw(m, t, 1), ranker=rk))
                qs.append(m.Query(kw(m, t, 1, mask=0b01), ranker=rk))
        ids = list(range(len(v)))
        for a in ids[:12]:
            for b in ids[:12]:
                if a != b:
                    qs.append(m.Query(m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), ranker=m.SPH_RANK_BM25))
        check_batch(orc, dev, hi, qs)


def test_golden_weights_on_device(dev):
    """Reference goldens straight from the device path (no oracle in the loop)."""
    m, ctx, batch = dev
    W, R, H, v = make_hits(T037)
    seg = m.Segment(ctx, m.index_from_hits(W, R, H, n_terms=len(v), total_docs=len(T037)))
    r = batch.search(seg, [m.Query(kw(m, v["test"], 1, mask=0b01), ranker=m.SPH_RANK_BM25)])[0]
    assert list(r.rowid) == [1] and list(r.weight) == [1800]  # test_037: "@title test" bm25 -> 2:1800
    seg.close()
    W, R, H, v = make_hits(T019, 2)
    seg = m.Segment(ctx, m.index_from_hits(W, R, H, n_terms=len(v), total_docs=len(T019)))
    r = batch.search(seg, [m.Query(kw(m, v["77"], 1))])[0]
    assert [T019_IDS[i] for i in r.rowid] == [777] and list(r.weight) == [1803]  # test_019: "77" -> 777:1803
    seg.close()
    docs = [["If I were a cat...", "We are the greatest cat"]]
    W, R, H, v = make_hits(docs)
    seg = m.Segment(ctx, m.index_from_hits(W, R, H, n_terms=len(v), total_docs=1))
    r = batch.search(seg, [m.Query(kw(m, v["cat"], 1, mask=0b01))])[0]
    assert list(r.weight) == [1500]  # gtests_rtstuff.cpp RTN.WeightBoundary
    seg.close()


# ------------------------------------------------------------------ random corpora
PROBS = [0.5, 0.3, 0.12, 0.05, 0.02, 0.006, 0.002, 0.0007, 0.9, 0.3, 0.0001]


@pytest.mark.parametrize("block,fmt", [(128, 1), (32, 1), (128, 0), (64, 1)])
def test_random_corpus_and_queries(orc, dev, block, fmt):
    m = dev[0]
    rng = np.random.default_rng(1234 + block + fmt)
    n_docs = 60000
    W, R, H = synth_postings(rng, n_docs, PROBS, n_fields=3, end_markers=True)
    nt = len(PROBS) + 1  # last term has no postings
    hi = m.index_from_hits(W, R, H, n_terms=nt, total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
    qs = []
    for t in range(nt):
        qs.append(m.Query(kw(m, t, 1), ranker=m.SPH_RANK_BM25))
        qs.append(m.Query(kw(m, t, 1, mask=0b010), ranker=m.SPH_RANK_BM25, max_matches=10))
    for _ in range(150):
        k = int(rng.integers(2, 5))
        ts = rng.choice(nt, size=k, replace=False)
        masks = [0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8)) for _ in ts]
        root = m.XQNode.AND(*[kw(m, int(t), i + 1, mk) for i, (t, mk) in enumerate(zip(ts, masks))])
        qs.append(m.Query(root, ranker=int(rng.choice([m.SPH_RANK_BM25, m.SPH_RANK_NONE])),
                          max_matches=int(rng.choice([1, 7, 100, 1000, 1024])),
                          field_weights=[int(x) for x in rng.integers(-3, 12, 3)] if rng.random() < 0.5 else None,
                          index_weight=int(rng.choice([1, 1, 3])), plain_idf=bool(rng.random() < 0.2),
                          normalized_tfidf=bool(rng.random() < 0.8)))
    # 8-way AND, a missing keyword, equal doc counts (terms 1 and 9 have the same probability)
    qs.append(m.Query(m.XQNode.AND(*[kw(m, t, t + 1) for t in (8, 0, 1, 9, 2, 3, 4, 5)]), ranker=m.SPH_RANK_BM25))
    qs.append(m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, nt - 1, 2)), ranker=m.SPH_RANK_BM25))
    qs.append(m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, -1, 2)), ranker=m.SPH_RANK_BM25))
    qs.append(m.Query(m.XQNode.AND(kw(m, 1, 1), kw(m, 9, 2), kw(m, 0, 3)), ranker=m.SPH_RANK_BM25))
    # local_df overrides (global IDF inputs for sharded search)
    qs.append(m.Query(m.XQNode.AND(kw(m, 2, 1), kw(m, 3, 2)), ranker=m.SPH_RANK_BM25, total_docs=10 * n_docs,
                      local_docs={2: 70000, 3: 31000}))
    check_batch(orc, dev, hi, qs)


def test_tiny_and_ragged_lists(orc, dev):
    """docs counts around the block edges: 1, block-1, block, block+1, 2*block, 2*block+1 ..."""
    m = dev[0]
    sizes = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 385, 1000]
    rng = np.random.default_rng(5)
    n_docs = 5000
    W, R, H = [], [], []
    for t, n in enumerate(sizes):
        rows = np.sort(rng.choice(n_docs, size=n, replace=False)).astype(np.uint32)
        for r in rows:
            tf = int(rng.integers(1, 4))
            hp = np.unique(((rng.integers(0, 2, tf).astype(np.uint32)) << 24) | rng.integers(1, 50, tf).astype(np.uint32))
            W += [t + 1] * len(hp)
            R += [r] * len(hp)
            H += list(hp)
    for block in (32, 128):
        hi = m.index_from_hits(np.array(W, np.uint64), np.array(R, np.uint32), np.array(H, np.uint32),
                               n_terms=len(sizes), total_docs=n_docs, skiplist_block_size=block)
        qs = [m.Query(kw(m, t, 1), ranker=m.SPH_RANK_BM25) for t in range(len(sizes))]
        for a in range(len(sizes)):
            for b in (len(sizes) - 1, len(sizes) - 2, 9):
                if a != b:
                    qs.append(m.Query(m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), ranker=m.SPH_RANK_BM25))
        check_batch(orc, dev, hi, qs)


def test_wide_rowid_gaps_and_first_rowid_zero(orc, dev):
    """5-byte rowid deltas (gaps >= 2^28), rowid 0 present, rowids near 2^32."""
    m = dev[0]
    rows_a = np.array([0, 1, 5, 1 << 28, (1 << 28) + 3, (1 << 31) + 7, 0xFFFFFFF0, 0xFFFFFFFE], np.uint32)
    rows_b = np.array([0, 5, 9, (1 << 28) + 3, (1 << 31) + 7, 0xFFFFFFFE], np.uint32)
    W = np.concatenate([np.full(len(rows_a), 1), np.full(len(rows_b), 2)]).astype(np.uint64)
    R = np.concatenate([rows_a, rows_b])
    H = np.full(len(R), (1 << 24) | 3, np.uint32)
    hi = m.index_from_hits(W, R, H, n_terms=2, total_docs=0xFFFFFFFF)
    qs = [m.Query(kw(m, 0, 1), ranker=m.SPH_RANK_BM25), m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2)), ranker=m.SPH_RANK_BM25)]
    check_batch(orc, dev, hi, qs)


def test_long_entries_slow_path(orc, dev):
    """Entries longer than the 8-byte fast window: huge tf, full 32-bit field masks, big hitlist offsets."""
    m = dev[0]
    rng = np.random.default_rng(11)
    W, R, H = [], [], []
    for r in range(0, 2000, 3):
        fields = rng.choice(32, size=int(rng.integers(1, 20)), replace=False)
        hp = np.unique(np.concatenate([(np.uint32(f) << 24) | rng.integers(1, 300, int(rng.integers(1, 60))).astype(np.uint32)
                                       for f in fields]))
        W += [1] * len(hp)
        R += [r] * len(hp)
        H += list(hp)
    for r in range(0, 2000, 2):
        W.append(2), R.append(r), H.append((31 << 24) | 1000)
    hi = m.index_from_hits(np.array(W, np.uint64), np.array(R, np.uint32), np.array(H, np.uint32), n_terms=2,
                           total_docs=2000, n_fields=32)
    fw = [int(x) for x in rng.integers(-5, 50, 32)]
    qs = [m.Query(kw(m, 0, 1), ranker=m.SPH_RANK_BM25, field_weights=fw),
          m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2)), ranker=m.SPH_RANK_BM25, field_weights=fw),
          m.Query(m.XQNode.AND(kw(m, 0, 1, 1 << 31), kw(m, 1, 2)), ranker=m.SPH_RANK_BM25)]
    check_batch(orc, dev, hi, qs)


def test_unsupported_shapes_fail_loudly(dev):
    m, ctx, batch = dev
    hi = m.synth_index(1000, [0.5, 0.5], seed=1)
    seg = m.Segment(ctx, hi)
    # an operator the device path does not know (SENTENCE / PARAGRAPH ...: op codes beyond NOTNEAR)
    q_or = m.Query(m.XQNode(13, [kw(m, 0, 1), kw(m, 1, 2)]), ranker=m.SPH_RANK_BM25)
    q_big = m.Query(kw(m, 0, 1), ranker=m.SPH_RANK_BM25, max_matches=5000)
    q_ok = m.Query(kw(m, 0, 1), ranker=m.SPH_RANK_BM25)
    r = batch.search(seg, [q_or, q_ok, q_big])
    assert r[0].status == -2 and r[2].status == -2 and r[1].status == 0 and r[1].total_found > 0
    seg.close()


def test_synth_medium_and_merge_across_shards(orc, dev):
    """Bigger lists (many work items per query) and the shard merge entry point."""
    import ctypes as C
    m, ctx, batch = dev
    from manticoresearch_amd import _lib
    hip = C.CDLL("libamdhip64.so")

    def dmalloc(n):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
        return p

    probs = [0.3, 0.2, 0.05, 0.01, 0.001]
    n_docs = 400000
    ctx.set("item_bytes", 16 << 10)  # many items per query
    his = [m.synth_index(n_docs, probs, seed=42, shard=s) for s in range(2)]
    qs = [m.Query(m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), ranker=m.SPH_RANK_BM25, total_docs=2 * n_docs,
                  local_docs={t: int(his[0].dict[t]["docs"]) + int(his[1].dict[t]["docs"]) for t in (a, b)})
          for a in range(5) for b in range(5) if a != b]
    nq = len(qs)
    for s in range(2):
        check_batch(orc, dev, his[s], qs, rowid_base=s * n_docs)
    # merged result == oracle top-K of the union with global rowids
    K = 1000
    in_keys, in_cnt = dmalloc(2 * nq * 1024 * 8), dmalloc(2 * nq * 4)
    out_keys, out_cnt = dmalloc(nq * 1024 * 8), dmalloc(nq * 4)
    want = []
    for s in range(2):
        seg = m.Segment(ctx, his[s], rowid_base=s * n_docs)
        batch.search(seg, qs)
        kp, cp, _ = batch.device_results()
        assert hip.hipMemcpy(C.c_void_p(in_keys.value + s * nq * 1024 * 8), C.c_void_p(kp), C.c_size_t(nq * 1024 * 8), 3) == 0
        assert hip.hipMemcpy(C.c_void_p(in_cnt.value + s * nq * 4), C.c_void_p(cp), C.c_size_t(nq * 4), 3) == 0
        seg.close()
        oi = orc_index_of(orc, his[s])
        want.append([to_orc(orc, q).run(oi) for q in qs])
    _lib.check(_lib.lib().mrk_topk_merge(ctx._h, in_keys, in_cnt, 2, nq, K, out_keys, out_cnt))
    ok = np.zeros((nq, 1024), np.uint64)
    oc = np.zeros(nq, np.uint32)
    assert hip.hipMemcpy(C.c_void_p(ok.ctypes.data), out_keys, C.c_size_t(ok.nbytes), 2) == 0
    assert hip.hipMemcpy(C.c_void_p(oc.ctypes.data), out_cnt, C.c_size_t(oc.nbytes), 2) == 0
    for p in (in_keys, in_cnt, out_keys, out_cnt):
        hip.hipFree(p)
    for qi in range(nq):
        allm = []
        for s in range(2):
            w = want[s][qi]
            allm += [(-int(wt), int(r) + s * n_docs) for r, wt in zip(w.rowid, w.weight)]
        allm.sort()  # weight desc, global docid asc
        allm = allm[:K]
        ks = [int(k) for k in ok[qi][:oc[qi]]]
        gw = [((k >> 32) ^ 0x80000000) - (1 << 32) if ((k >> 32) ^ 0x80000000) >= (1 << 31) else ((k >> 32) ^ 0x80000000) for k in ks]
        gr = [(~k) & 0xFFFFFFFF for k in ks]
        assert [(-w, r) for w, r in zip(gw, gr)] == allm
    ctx.set("item_bytes", 128 << 10)


# ------------------------------------------------------------------ proximity rankers (hit path)
def test_golden_proximity_weights_on_device(dev):
    """Reference goldens that need hitlists: test_019 'basic query' -> 111:2654, '@title sample @body world' ->
    333:2666; test_322 'program flow' under five field-weight settings (incl. negative)."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("proximity rankers run on the packed path")
    W, R, H, v = make_hits(T019, 2)
    seg = m.Segment(ctx, m.index_from_hits(W, R, H, n_terms=len(v), total_docs=len(T019)))
    r = batch.search(seg, [m.Query(m.XQNode.AND(kw(m, v["basic"], 1), kw(m, v["query"], 2))),
                           m.Query(m.XQNode.AND(kw(m, v["sample"], 1, mask=0b01), kw(m, v["world"], 2, mask=0b10)))])
    assert [(T019_IDS[i], int(w)) for i, w in zip(r[0].rowid, r[0].weight)] == [(111, 2654)]
    assert [(T019_IDS[i], int(w)) for i, w in zip(r[1].rowid, r[1].weight)] == [(333, 2666)]
    seg.close()
    W, R, H, v = make_hits(T322)
    seg = m.Segment(ctx, m.index_from_hits(W, R, H, n_terms=len(v), total_docs=len(T322), n_fields=3))
    ids = [1, 2, 3, 100]
    root = m.XQNode.AND(kw(m, v["program"], 1), kw(m, v["flow"], 2))
    want = {1: [(1, 7415), (3, 6426), (2, 4421)], 10: [(1, 25415), (3, 15426), (2, 13421)],
            0: [(3, 5426), (1, 5415), (2, 3421)], -2: [(3, 3426), (2, 1421), (1, 1415)],
            -10: [(3, -4574), (2, -6579), (1, -14585)]}
    for spam, exp in want.items():
        r = batch.search(seg, [m.Query(root, ranker=m.SPH_RANK_PROXIMITY_BM25, field_weights=[1, 2, spam])])[0]
        assert [(ids[i], int(w)) for i, w in zip(r.rowid, r.weight)] == exp
    seg.close()


@pytest.mark.parametrize("block,fmt", [(128, 1), (32, 1), (128, 0)])
def test_random_corpus_proximity(orc, dev, block, fmt):
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("proximity rankers run on the packed path")
    rng = np.random.default_rng(4321 + block + fmt)
    n_docs = 40000
    probs = [0.5, 0.3, 0.12, 0.05, 0.02, 0.004, 0.9]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=40, end_markers=True)
    hi = m.index_from_hits(W, R, H, n_terms=len(probs), total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
    qs = []
    for _ in range(120):
        k = int(rng.integers(2, 5))
        ts = rng.choice(len(probs), size=k, replace=bool(rng.random() < 0.25))  # repeated keywords: the HANDLE_DUPES update
        masks = [0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8)) for _ in ts]
        # atom positions as the parser numbers them, sometimes with a gap (stop word); some queries sit at positions
        # around 32: the HANDLE_DUPES masks are DWORDs fed by 1UL << qpos (sphinxsearch.cpp:1396, 1403)
        pos, ap = [], int(rng.choice([0, 0, 0, 0, 29, 31, 45]))
        for _ in ts:
            ap += 1 if rng.random() < 0.85 else 2
            pos.append(ap)
        root = m.XQNode.AND(*[kw(m, int(t), p, mk) for t, p, mk in zip(ts, pos, masks)])
        qs.append(m.Query(root, ranker=int(rng.choice([m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_PROXIMITY, m.SPH_RANK_WORDCOUNT,
                                                       m.SPH_RANK_MATCHANY, m.SPH_RANK_FIELDMASK, m.SPH_RANK_SPH04])),
                          max_matches=int(rng.choice([5, 100, 1000])),
                          field_weights=[int(x) for x in rng.integers(-3, 12, 3)] if rng.random() < 0.5 else None,
                          index_weight=int(rng.choice([1, 1, 2]))))
    for t in range(len(probs)):  # single keyword: WeightSum with / without BM25; the other state rankers still read hits
        qs.append(m.Query(kw(m, t, 1), ranker=m.SPH_RANK_PROXIMITY))
        qs.append(m.Query(kw(m, t, 1), ranker=m.SPH_RANK_PROXIMITY_BM25))
        for rk in (m.SPH_RANK_WORDCOUNT, m.SPH_RANK_MATCHANY, m.SPH_RANK_FIELDMASK, m.SPH_RANK_SPH04):
            qs.append(m.Query(kw(m, t, 1, int(rng.choice([0xFFFFFFFF, 0b011]))), ranker=rk,
                              field_weights=[int(x) for x in rng.integers(-3, 12, 3)]))
    check_batch(orc, dev, hi, qs)


def test_proximity_long_hitlists(orc, dev):
    """Docs with hundreds of hits per keyword, 5-byte hit deltas (field jumps), phrases that do line up."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("proximity rankers run on the packed path")
    rng = np.random.default_rng(77)
    W, R, H = [], [], []
    n_docs = 3000
    for t in range(3):
        for r in range(n_docs):
            if rng.random() < 0.6:
                n = int(rng.integers(1, 300)) if rng.random() < 0.2 else int(rng.integers(1, 6))
                fields = rng.choice([0, 1, 7], size=n)
                # keyword t often sits at position base+t so that consecutive keywords form runs
                base = rng.integers(1, 200, size=n)
                hp = np.unique((fields.astype(np.uint32) << 24) | (base + t).astype(np.uint32))
                W += [t + 1] * len(hp)
                R += [r] * len(hp)
                H += list(hp)
    order = np.lexsort((np.array(H), np.array(R), np.array(W)))
    W, R, H = np.array(W, np.uint64)[order], np.array(R, np.uint32)[order], np.array(H, np.uint32)[order]
    hi = m.index_from_hits(W, R, H, n_terms=3, total_docs=n_docs, n_fields=8)
    qs = [m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2), kw(m, 2, 3)), ranker=m.SPH_RANK_PROXIMITY_BM25,
                  field_weights=[3, 2, 1, 1, 1, 1, 1, 5]),
          m.Query(m.XQNode.AND(kw(m, 2, 1), kw(m, 0, 2)), ranker=m.SPH_RANK_PROXIMITY_BM25),
          m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2)), ranker=m.SPH_RANK_PROXIMITY, max_matches=50)]
    check_batch(orc, dev, hi, qs)


# ------------------------------------------------------------------ boolean trees (ExtAnd / ExtOr / ExtMaybe / ExtAndNot)
def _tree_only(dev):
    if ctx_path(dev[1]) != 0:
        pytest.skip("boolean trees run on the packed path")


def OR(m, *k):
    return m.XQNode(m.SPH_QUERY_OR, list(k))


def MAYBE(m, *k):
    return m.XQNode(m.SPH_QUERY_MAYBE, list(k))


def BEFORE(m, *k):
    return m.XQNode(m.SPH_QUERY_BEFORE, list(k))


def ANDNOT(m, *k):
    return m.XQNode(m.SPH_QUERY_ANDNOT, list(k))


def test_golden_boolean_weights_on_device(dev):
    """test_019 goldens: 'basic | china' -> 444:1610 111:1577 555:1577; '@title sample @body -basic' -> 333:1555 666:1555."""
    _tree_only(dev)
    m, ctx, batch = dev
    W, R, H, v = make_hits(T019, 2)
    seg = m.Segment(ctx, m.index_from_hits(W, R, H, n_terms=len(v), total_docs=len(T019)))
    r = batch.search(seg, [m.Query(OR(m, kw(m, v["basic"], 1), kw(m, v["china"], 2))),
                           m.Query(ANDNOT(m, kw(m, v["sample"], 1, mask=0b01), kw(m, v["basic"], 2, mask=0b10)))])
    assert r[0].status == 0 and r[1].status == 0
    assert [(T019_IDS[i], int(w)) for i, w in zip(r[0].rowid, r[0].weight)] == [(444, 1610), (111, 1577), (555, 1577)]
    assert [(T019_IDS[i], int(w)) for i, w in zip(r[1].rowid, r[1].weight)] == [(333, 1555), (666, 1555)]
    seg.close()


def _random_tree(m, rng, nt, depth=0):
    """A random boolean tree over distinct keywords, atom positions in traversal order."""
    state = {"pos": 0, "used": set()}

    def leaf():
        while True:
            t = int(rng.integers(0, nt))
            if t not in state["used"]:
                break
        state["used"].add(t)
        state["pos"] += 1
        mk = 0xFFFFFFFF if rng.random() < 0.75 else int(rng.integers(1, 8))
        return kw(m, t, state["pos"], mk)

    def node(d, budget):
        if budget <= 1 or d >= 3 or rng.random() < 0.25:
            return leaf()
        op = rng.choice(["and", "or", "maybe", "andnot"], p=[0.35, 0.35, 0.15, 0.15])
        k = 2 if op in ("maybe", "andnot") else int(rng.integers(2, min(3, budget) + 1))
        kids = [node(d + 1, max(1, budget // k)) for _ in range(k)]
        return {"and": m.XQNode.AND, "or": lambda *c: OR(m, *c), "maybe": lambda *c: MAYBE(m, *c),
                "andnot": lambda *c: ANDNOT(m, *c)}[op](*kids)

    return node(0, int(rng.integers(2, 7)))


@pytest.mark.parametrize("block,fmt", [(128, 1), (32, 0)])
def test_random_boolean_trees(orc, dev, block, fmt):
    _tree_only(dev)
    m, ctx, batch = dev
    rng = np.random.default_rng(777 + block + fmt)
    n_docs = 50000
    probs = [0.5, 0.3, 0.12, 0.05, 0.02, 0.006, 0.002, 0.9, 0.3, 0.0001]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=30, end_markers=True)
    nt = len(probs) + 1  # last keyword has no postings
    hi = m.index_from_hits(W, R, H, n_terms=nt, total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
    a, b, c, d = (kw(m, t, i + 1) for i, t in enumerate((1, 3, 4, 0)))
    fixed = [OR(m, a, b), OR(m, b, a), m.XQNode.AND(OR(m, a, b), c), m.XQNode.AND(c, OR(m, a, b)), ANDNOT(m, a, b),
             m.XQNode.AND(a, ANDNOT(m, b, c)), MAYBE(m, b, a), MAYBE(m, c, OR(m, a, b)), OR(m, a, b, c, d),
             OR(m, m.XQNode.AND(a, b), m.XQNode.AND(c, d)), m.XQNode.AND(OR(m, a, b), OR(m, c, d)),
             OR(m, a, kw(m, nt - 1, 2)), m.XQNode.AND(OR(m, a, kw(m, nt - 1, 2)), c), ANDNOT(m, a, kw(m, nt - 1, 2)),
             ANDNOT(m, OR(m, a, b), c), OR(m, ANDNOT(m, a, b), c), m.XQNode.AND(m.XQNode.AND(a, b), c),
             # quorum operator: threshold 1 = OR chain, threshold >= words = AND chain (searchnode.cpp:1638-1686)
             m.XQNode(m.SPH_QUERY_QUORUM, [a, b, c], None, 0xFFFFFFFF, 1), m.XQNode(m.SPH_QUERY_QUORUM, [c, a, b], None, 0b011, 3),
             m.XQNode(m.SPH_QUERY_QUORUM, [b, d], None, 0xFFFFFFFF, 5), m.XQNode(m.SPH_QUERY_QUORUM, [a, kw(m, nt - 1, 2)], None, 0xFFFFFFFF, 1)]
    qs = []
    for root in fixed:
        for rk in (m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_PROXIMITY_BM25):
            qs.append(m.Query(root, ranker=rk, max_matches=int(rng.choice([10, 1000]))))
    for _ in range(160):
        root = _random_tree(m, rng, nt)
        qs.append(m.Query(root, ranker=int(rng.choice([m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_PROXIMITY_BM25,
                                                       m.SPH_RANK_PROXIMITY])),
                          max_matches=int(rng.choice([1, 20, 1000])),
                          field_weights=[int(x) for x in rng.integers(-3, 12, 3)] if rng.random() < 0.5 else None,
                          index_weight=int(rng.choice([1, 1, 3])), normalized_tfidf=bool(rng.random() < 0.8)))
    m_, ctx_, batch_ = dev
    seg = m.Segment(ctx, hi)
    oi = orc_index_of(orc, hi)
    n_ok = 0
    try:
        for i in range(0, len(qs), batch.max_queries):
            chunk = qs[i:i + batch.max_queries]
            got = batch.search(seg, chunk)
            for q, g in zip(chunk, got):
                if g.status == -2:
                    continue  # shapes the device path declines (too many drivers / keywords for a hit ranker)
                want = to_orc(orc, q).run(oi)
                assert g.status == 0
                assert g.total_found == want.total_found, (g.total_found, want.total_found)
                assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
                n_ok += 1
    finally:
        seg.close()
    assert n_ok >= len(fixed) * 3 - 6 + 100, n_ok


# ------------------------------------------------------------------ PHRASE (ExtNWay_T<FSMphrase_c>)
def PHRASE(m, *k, mask=0xFFFFFFFF):
    return m.XQNode(m.SPH_QUERY_PHRASE, list(k), None, mask)


def PROXIMITY(m, dist, *k, mask=0xFFFFFFFF):
    return m.XQNode(m.SPH_QUERY_PROXIMITY, list(k), None, mask, dist)


def test_golden_phrase_weight_on_device(dev):
    """test_019: '"phrase query"' under the default ranker -> 222:2687."""
    _tree_only(dev)
    m, ctx, batch = dev
    W, R, H, v = make_hits(T019, 2)
    seg = m.Segment(ctx, m.index_from_hits(W, R, H, n_terms=len(v), total_docs=len(T019)))
    r = batch.search(seg, [m.Query(PHRASE(m, kw(m, v["phrase"], 1), kw(m, v["query"], 2)))])[0]
    assert r.status == 0
    assert [(T019_IDS[i], int(w)) for i, w in zip(r.rowid, r.weight)] == [(222, 2687)]
    seg.close()


@pytest.mark.parametrize("block,fmt", [(128, 1), (32, 0)])
def test_random_corpus_phrases(orc, dev, block, fmt):
    """Dense little docs (positions 1..12 over 3 fields) so that 2-4 word phrases do occur, repeat and overlap."""
    _tree_only(dev)
    m, ctx, batch = dev
    rng = np.random.default_rng(999 + block + fmt)
    n_docs = 30000
    probs = [0.6, 0.5, 0.4, 0.3, 0.1, 0.02, 0.9]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=12, end_markers=True)
    hi = m.index_from_hits(W, R, H, n_terms=len(probs), total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
    qs = []
    rankers = [m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_PROXIMITY]
    more = [m.SPH_RANK_WORDCOUNT, m.SPH_RANK_SPH04, m.SPH_RANK_MATCHANY, m.SPH_RANK_FIELDMASK]
    for i in range(140):
        k = int(rng.integers(2, 5))
        dupes = rng.random() < 0.15
        ts = rng.choice(len(probs), size=k, replace=dupes)
        pos, ap = [], 0
        for _ in ts:
            ap += 1 if rng.random() < 0.85 else 2  # a stop word leaves a gap in the atom positions
            pos.append(ap)
        mask = 0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8))
        words = [kw(m, int(t), p) for t, p in zip(ts, pos)]
        root = PHRASE(m, *words, mask=mask) if i % 2 else PROXIMITY(m, int(rng.integers(1, 6)), *words, mask=mask)
        rk = rankers[i % 4] if i % 3 else more[(i // 3) % 4]
        # (repeated words under the proximity rankers take RankerState_Proximity_fn<.., true>)
        qs.append(m.Query(root, ranker=rk, max_matches=int(rng.choice([5, 100, 1000])),
                          field_weights=[int(x) for x in rng.integers(-3, 12, 3)] if rng.random() < 0.5 else None,
                          index_weight=int(rng.choice([1, 1, 2]))))
    # a PHRASE below other operators: '"a b" | c', '"a b" c', 'c "a b"', 'c -"a b"', '"a b" MAYBE c', '("a b" | c) d'
    for i in range(60):
        ts = [int(t) for t in rng.choice(len(probs), size=4, replace=False)]
        npw = int(rng.integers(2, 4))
        pos, ap = [], 0
        for _ in range(4):
            ap += 1 if rng.random() < 0.85 else 2
            pos.append(ap)
        ph_mask = 0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8))
        ph_words = [kw(m, ts[j], pos[j]) for j in range(npw)]
        ph = PHRASE(m, *ph_words, mask=ph_mask) if i % 3 else PROXIMITY(m, int(rng.integers(1, 5)), *ph_words, mask=ph_mask)
        c = kw(m, ts[npw], pos[npw], 0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8)))
        shape = i % 6
        if shape == 0:
            root = OR(m, ph, c)
        elif shape == 1:
            root = m.XQNode.AND(ph, c)
        elif shape == 2:
            root = m.XQNode.AND(c, ph)
        elif shape == 3:
            root = ANDNOT(m, c, ph)
        elif shape == 4:
            root = MAYBE(m, ph, c)
        else:
            root = m.XQNode.AND(OR(m, ph, c), kw(m, ts[3], pos[3])) if npw == 2 else OR(m, c, ph)
        qs.append(m.Query(root, ranker=rankers[(i // 6) % 4], max_matches=int(rng.choice([5, 1000])),
                          field_weights=[int(x) for x in rng.integers(-3, 12, 3)] if rng.random() < 0.5 else None))
    # "a a" style phrases: back-to-back occurrences share a position
    qs.append(m.Query(PHRASE(m, kw(m, 6, 1), kw(m, 6, 2)), ranker=m.SPH_RANK_BM25))
    qs.append(m.Query(PHRASE(m, kw(m, 6, 1), kw(m, 6, 2), kw(m, 6, 3)), ranker=m.SPH_RANK_NONE))
    check_batch(orc, dev, hi, qs)
    # phrases were actually found
    seg = m.Segment(ctx, hi)
    r = batch.search(seg, [m.Query(PHRASE(m, kw(m, 0, 1), kw(m, 1, 2)))])[0]
    assert r.total_found > 100
    seg.close()


# ------------------------------------------------------------------ dead-row map (MatchExtended, sphinx.cpp:12213-12217)
def test_dead_rows_are_dropped_before_the_sorter(orc, dev):
    m, ctx, batch = dev
    rng = np.random.default_rng(31337)
    n_docs = 50000
    probs = [0.5, 0.3, 0.1, 0.02]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=20)
    hi = m.index_from_hits(W, R, H, n_terms=len(probs), total_docs=n_docs, n_fields=3)
    dead = np.zeros((n_docs + 31) // 32, np.uint32)
    killed = rng.choice(n_docs, size=n_docs // 3, replace=False)
    np.bitwise_or.at(dead, killed >> 5, (np.uint32(1) << (killed & 31).astype(np.uint32)))
    a, b, c = kw(m, 0, 1), kw(m, 1, 2), kw(m, 2, 3)
    roots = [a, m.XQNode.AND(a, b), m.XQNode.AND(a, b, c)]
    rankers = [m.SPH_RANK_BM25, m.SPH_RANK_NONE]
    if ctx_path(ctx) == 0:
        roots += [OR(m, b, c), ANDNOT(m, a, c), PHRASE(m, a, b)]
        rankers.append(m.SPH_RANK_PROXIMITY_BM25)
    qs = [m.Query(r, ranker=rk, max_matches=k) for r in roots for rk in rankers for k in (10, 1000)]
    seg = m.Segment(ctx, hi)
    oi = orc_index_of(orc, hi)
    try:
        before = batch.search(seg, qs)
        seg.set_dead_rows(dead)
        oi.dead_rows = dead
        after = batch.search(seg, qs)
        for q, g0, g in zip(qs, before, after):
            want = to_orc(orc, q).run(oi)
            assert g.status == 0
            assert g.total_found == want.total_found and g.total_found < g0.total_found
            assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
            assert not ((dead[g.rowid >> 5] >> (g.rowid & 31)) & 1).any()
        seg.set_dead_rows(None)  # an emptied map brings everything back
        again = batch.search(seg, qs[:4])
        for g0, g in zip(before, again):
            assert g.total_found == g0.total_found and (g.rowid == g0.rowid).all()
    finally:
        seg.close()


# ------------------------------------------------------------------ two dense keywords: the bitmap AND kernel
def test_bitmap_and_kernel(orc, dev):
    """Dense x dense 2-keyword ANDs run on doc-set bitmaps; same results as the oracle, with field limits,
    saturated tf (>= 255 hits), a dead-row map, a row count that is no multiple of the 2048-rowid window."""
    _tree_only(dev)
    m, ctx, batch = dev
    rng = np.random.default_rng(2048)
    n_docs = 70001
    probs = [0.5, 0.31, 0.3, 0.12, 0.05, 0.02, 0.9, 0.004]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=40, end_markers=True)
    # a few docs of keyword 1 with 300 hits: tf saturates the packed byte -> exception list
    fat = rng.choice(n_docs, 40, replace=False).astype(np.uint32)
    keep = ~((W == 2) & np.isin(R, fat))
    W, R, H = W[keep], R[keep], H[keep]
    W = np.concatenate([W, np.full(fat.size * 300, 2, np.uint64)])
    R = np.concatenate([R, np.repeat(fat, 300)])
    H = np.concatenate([H, np.tile((np.uint32(1) << 24) | np.arange(1, 301, dtype=np.uint32), fat.size)])
    o = np.lexsort((H, R, W))
    W, R, H = W[o], R[o], H[o]
    hi = m.index_from_hits(W, R, H, n_terms=len(probs), total_docs=n_docs, n_fields=3)
    qs = []
    for a_ in range(7):
        for b_ in range(7):
            if a_ == b_:
                continue
            for rk in (m.SPH_RANK_BM25, m.SPH_RANK_NONE):
                mk = [0xFFFFFFFF if rng.random() < 0.6 else int(rng.integers(1, 8)) for _ in range(2)]
                qs.append(m.Query(m.XQNode.AND(kw(m, a_, 1, mk[0]), kw(m, b_, 2, mk[1])), ranker=rk,
                                  max_matches=int(rng.choice([3, 100, 1000, 1024])),
                                  field_weights=[int(x) for x in rng.integers(-3, 12, 3)] if rng.random() < 0.5 else None,
                                  index_weight=int(rng.choice([1, 2]))))
    qs.append(m.Query(m.XQNode.AND(kw(m, 6, 1), kw(m, 6, 2)), ranker=m.SPH_RANK_BM25))  # the same keyword twice
    qs.append(m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 7, 2)), ranker=m.SPH_RANK_BM25))  # dense x sparse: block path
    ctx.set("item_bytes", 16 << 10)  # several work items per query
    try:
        check_batch(orc, dev, hi, qs)
        ctx.set("attr_nibbles", 1)  # the one-byte tf/field plane (tf >= 15 escapes to the attr words)
        try:
            check_batch(orc, dev, hi, qs)
        finally:
            ctx.set("attr_nibbles", 0)
        ctx.set("attr_seq", 0)  # without the slot-ordered plane the gathers read the block decoder's interleaved words
        try:
            check_batch(orc, dev, hi, qs)
        finally:
            ctx.set("attr_seq", 1)
        seg = m.Segment(ctx, hi)
        batch.search(seg, qs[:8])
        assert batch.stats()["n_items_bm"] > 0  # the bitmap kernel did run
        # dead rows are masked out of the match words
        dead = np.zeros((n_docs + 31) // 32, np.uint32)
        killed = rng.choice(n_docs, size=n_docs // 4, replace=False)
        np.bitwise_or.at(dead, killed >> 5, (np.uint32(1) << (killed & 31).astype(np.uint32)))
        seg.set_dead_rows(dead)
        oi = orc_index_of(orc, hi)
        oi.dead_rows = dead
        for q, g in zip(qs[:40], batch.search(seg, qs[:40])):
            want = to_orc(orc, q).run(oi)
            assert g.status == 0 and g.total_found == want.total_found
            assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
        seg.close()
        # bitmaps off: the block kernel answers the same queries
        ctx.set("bitmap_inv", 0)
        seg = m.Segment(ctx, hi)
        batch.search(seg, qs[:8])
        assert batch.stats()["n_items_bm"] == 0
        seg.close()
    finally:
        ctx.set("bitmap_inv", 64)
        ctx.set("item_bytes", 128 << 10)


def test_row_export_and_merge(orc, dev):
    """mrk_batch_export_rows / mrk_topk_merge_rows (the one-collective shard exchange): rows of two shards merged
    on the device == oracle top-K of the union with global docids, totals added up."""
    import ctypes as C
    m, ctx, batch = dev
    from manticoresearch_amd import _lib
    hip = C.CDLL("libamdhip64.so")

    def dmalloc(n):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
        return p

    probs = [0.3, 0.2, 0.05, 0.01]
    n_docs, K, RW = 150000, 1000, 1026
    his = [m.synth_index(n_docs, probs, seed=4242, shard=s) for s in range(2)]
    qs = [m.Query(m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), ranker=m.SPH_RANK_BM25, total_docs=2 * n_docs, max_matches=K,
                  local_docs={t: int(his[0].dict[t]["docs"]) + int(his[1].dict[t]["docs"]) for t in (a, b)})
          for a in range(4) for b in range(4) if a != b]
    nq = len(qs)
    rows_all, out_rows = dmalloc(2 * nq * RW * 8), dmalloc(nq * RW * 8)
    want = []
    for s in range(2):
        seg = m.Segment(ctx, his[s], rowid_base=s * n_docs)
        batch.submit(seg, qs)
        batch.wait()
        _lib.check(_lib.lib().mrk_batch_export_rows(batch._h, C.c_void_p(rows_all.value + s * nq * RW * 8)))
        seg.close()
        oi = orc_index_of(orc, his[s])
        want.append([to_orc(orc, q).run(oi) for q in qs])
    _lib.check(_lib.lib().mrk_topk_merge_rows(ctx._h, rows_all, 2, nq, K, out_rows))
    host = np.zeros((nq, RW), np.uint64)
    assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), out_rows, C.c_size_t(host.nbytes), 2) == 0
    for qi in range(nq):
        exp = sorted([(-int(w), int(r) + s * n_docs) for s in range(2) for r, w in zip(want[s][qi].rowid, want[s][qi].weight)])[:K]
        cnt = int(host[qi, 1024])
        k = host[qi, :cnt]
        weight = ((k >> np.uint64(32)).astype(np.uint32) ^ np.uint32(0x80000000)).view(np.int32)
        docid = ~k.astype(np.uint32)
        assert [(-int(w), int(d)) for w, d in zip(weight, docid)] == exp
        assert int(host[qi, 1025]) == want[0][qi].total_found + want[1][qi].total_found
        assert not host[qi, cnt:1024].any()
    for p in (rows_all, out_rows):
        hip.hipFree(p)


def test_sharded_equals_unsharded(orc, dev):
    """Rowid-range shards are slices of the one corpus (the generator is keyed on the global rowid), document
    frequencies and N are the corpus-wide ones (local_df), and the cross-shard order is (weight desc, global docid asc)
    = the unsharded sorter's (weight desc, rowid asc): three uneven shards, exchanged as rows and merged on the device,
    must equal the unsharded device result bit for bit -- ties at rank K included -- for AND pairs under BM25 / NONE,
    an OR / ANDNOT tree and a proximity-ranked 3-keyword AND; the unsharded result itself is checked against the oracle
    (sphinxrt.cpp:6018-6108: chunks searched apart, sphinxrt.cpp:5945-5950: sorters merged)."""
    import ctypes as C
    _tree_only(dev)
    m, ctx, batch = dev
    from manticoresearch_amd import _lib
    hip = C.CDLL("libamdhip64.so")

    def dmalloc(n):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
        return p

    probs = [0.35, 0.2, 0.04, 0.006]
    n_docs, K, RW = 700_001, 1000, 1026
    cuts = [0, 131_072, 400_003, n_docs]
    whole = m.synth_index(n_docs, probs, seed=2026, max_pos=64)
    shards = [m.synth_index(cuts[i + 1] - cuts[i], probs, seed=2026, max_pos=64, rowid_base=cuts[i]) for i in range(3)]
    gdocs = {t: int(whole.dict[t]["docs"]) for t in range(4)}
    assert all(gdocs[t] == sum(int(sh.dict[t]["docs"]) for sh in shards) for t in range(4))

    def mk(root, ranker, k=K):
        return m.Query(root, ranker=ranker, max_matches=k, total_docs=n_docs, local_docs=dict(gdocs))

    qs = [mk(m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), rk) for a in range(4) for b in range(4) if a < b for rk in (m.SPH_RANK_BM25, m.SPH_RANK_NONE)]
    qs.append(mk(m.XQNode.AND(OR(m, kw(m, 2, 1), kw(m, 3, 2)), kw(m, 0, 3)), m.SPH_RANK_BM25))
    qs.append(mk(m.XQNode(m.SPH_QUERY_ANDNOT, [kw(m, 1, 1), kw(m, 0, 2)]), m.SPH_RANK_BM25, 37))
    qs.append(mk(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2), kw(m, 2, 3)), m.SPH_RANK_PROXIMITY_BM25))
    # shapes of the generic per-doc evaluator: they shard like everything else (a segment's evaluator sees segment-local rowids)
    qs.append(mk(m.XQNode(m.SPH_QUERY_NEAR, [m.XQNode(m.SPH_QUERY_PROXIMITY, [kw(m, 0, 1), kw(m, 1, 2)], opt=6), kw(m, 2, 3)], opt=12), m.SPH_RANK_PROXIMITY_BM25))
    qs.append(mk(m.XQNode(m.SPH_QUERY_BEFORE, [OR(m, kw(m, 2, 1), kw(m, 3, 2)), kw(m, 0, 3)]), m.SPH_RANK_SPH04, 200))
    qs.append(mk(m.XQNode(m.SPH_QUERY_NOTNEAR, [kw(m, 1, 1), m.XQNode(m.SPH_QUERY_PHRASE, [kw(m, 0, 2), kw(m, 2, 3)])], opt=4), m.SPH_RANK_BM25))
    nq = len(qs)
    seg = m.Segment(ctx, whole)
    want = batch.search(seg, qs)
    seg.close()
    oi = orc_index_of(orc, whole)
    for q, w in zip(qs, want):
        o = to_orc(orc, q).run(oi)
        assert w.status == 0 and w.total_found == o.total_found
        assert np.array_equal(w.rowid, o.rowid) and np.array_equal(w.weight, o.weight)
    rows_all, out_rows = dmalloc(3 * nq * RW * 8), dmalloc(nq * RW * 8)
    for s in range(3):
        seg = m.Segment(ctx, shards[s], rowid_base=cuts[s])
        batch.submit(seg, qs)
        batch.wait()
        _lib.check(_lib.lib().mrk_batch_export_rows(batch._h, C.c_void_p(rows_all.value + s * nq * RW * 8)))
        seg.close()
    _lib.check(_lib.lib().mrk_topk_merge_rows(ctx._h, rows_all, 3, nq, 1024, out_rows))
    host = np.zeros((nq, RW), np.uint64)
    assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), out_rows, C.c_size_t(host.nbytes), 2) == 0
    for qi, (q, w) in enumerate(zip(qs, want)):
        k = host[qi, : min(int(host[qi, 1024]), q.max_matches)]  # the merged list, cut to the query's own K
        weight = ((k >> np.uint64(32)).astype(np.uint32) ^ np.uint32(0x80000000)).view(np.int32)
        docid = ~k.astype(np.uint32)
        assert int(host[qi, 1025]) == w.total_found, qi
        assert np.array_equal(docid, w.rowid) and np.array_equal(weight, w.weight), qi
    # the exchange partitioned by query (round 3), emulated on one device: "rank" r of 3 receives every shard's rows of ITS
    # queries (mrk_shard_slice) -- here a device copy stands in for the all-to-all -- and merges that slice only; the slices
    # together must be the rows the all-gather form merged above, bit for bit (keys, counts, totals)
    part_rows = dmalloc(nq * RW * 8)
    assert hip.hipMemset(part_rows, 0xEE, C.c_size_t(nq * RW * 8)) == 0
    per = (nq + 2) // 3
    recv = dmalloc(3 * per * RW * 8)
    covered = 0
    for r in range(3):
        f, c = C.c_uint32(), C.c_uint32()
        _lib.check(_lib.lib().mrk_shard_slice(nq, 3, r, C.byref(f), C.byref(c)))
        assert f.value == covered and c.value <= per
        covered += c.value
        for s in range(3):
            assert hip.hipMemcpy(C.c_void_p(recv.value + s * per * RW * 8), C.c_void_p(rows_all.value + (s * nq + f.value) * RW * 8),
                                 C.c_size_t(c.value * RW * 8), 3) == 0
        _lib.check(_lib.lib().mrk_topk_merge_rows_part(ctx._h, recv, 3, per, f.value, c.value, 1024, part_rows))
    assert covered == nq
    host2 = np.zeros((nq, RW), np.uint64)
    assert hip.hipMemcpy(C.c_void_p(host2.ctypes.data), part_rows, C.c_size_t(host2.nbytes), 2) == 0
    assert np.array_equal(host2, host)
    for p in (rows_all, out_rows, part_rows, recv):
        hip.hipFree(p)


@pytest.mark.parametrize("block,fmt", [(128, 1), (32, 0)])
def test_near_and_notnear(orc, dev, block, fmt):
    """'a NEAR/N b' (FSMmultinear_c, two operands) and 'a NOTNEAR/N b' (ExtNotNear_c) over plain keywords: at the root and below
    AND / OR / ANDNOT / MAYBE, all rankers, field limits, repeated keywords ('x NEAR/2 x'), docs with long hitlists."""
    _tree_only(dev)
    m, ctx, batch = dev
    rng = np.random.default_rng(115 + block)
    n_docs = 30000
    probs = [0.5, 0.3, 0.12, 0.04, 0.9]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=24, end_markers=True)
    hi = m.index_from_hits(W, R, H, n_terms=len(probs), total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
    rankers = [m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_SPH04, m.SPH_RANK_WORDCOUNT, m.SPH_RANK_PROXIMITY,
               m.SPH_RANK_MATCHANY, m.SPH_RANK_FIELDMASK]
    qs = []
    for i in range(160):
        a, b, c = (int(x) for x in rng.choice(len(probs), size=3, replace=bool(rng.random() < 0.2)))
        mk = lambda: 0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8))
        dist = int(rng.integers(1, 7))
        node = m.XQNode(m.SPH_QUERY_NEAR if i % 2 == 0 else m.SPH_QUERY_NOTNEAR, [kw(m, a, 1, mk()), kw(m, b, 2, mk())], None, 0xFFFFFFFF, dist)
        shape = (i // 2) % 6
        if shape == 1:
            root = m.XQNode.AND(node, kw(m, c, 3, mk()))
        elif shape == 2:
            root = OR(m, node, kw(m, c, 3))
        elif shape == 3:
            root = ANDNOT(m, node, kw(m, c, 3))
        elif shape == 4:
            root = MAYBE(m, kw(m, c, 3), m.XQNode(node.op, [kw(m, a, 4), kw(m, b, 5)], None, 0xFFFFFFFF, dist)) if False else MAYBE(m, node, kw(m, c, 3))
        elif shape == 5:
            root = m.XQNode.AND(kw(m, c, 1), m.XQNode(node.op, [kw(m, a, 2, mk()), kw(m, b, 3, mk())], None, 0xFFFFFFFF, dist))
        else:
            root = node
        qs.append(m.Query(root, ranker=rankers[(i // 12) % len(rankers)], max_matches=int(rng.choice([7, 1000])),
                          field_weights=[int(x) for x in rng.integers(-2, 9, 3)] if rng.random() < 0.4 else None))
    check_batch(orc, dev, hi, qs)
    seg = m.Segment(ctx, hi)
    r = batch.search(seg, [m.Query(m.XQNode(m.SPH_QUERY_NEAR, [kw(m, 0, 1), kw(m, 1, 2)], None, 0xFFFFFFFF, 2)),
                           m.Query(m.XQNode(m.SPH_QUERY_NOTNEAR, [kw(m, 0, 1), kw(m, 4, 2)], None, 0xFFFFFFFF, 3))])
    assert r[0].status == 0 and r[0].total_found > 100 and r[1].status == 0 and 0 < r[1].total_found < int(hi.dict[0]["docs"])
    seg.close()


# ------------------------------------------------------------------ tests/golden/reference_vectors.json on the device
@pytest.mark.parametrize("from_text", [False, True], ids=["trees", "query-text"])
def test_golden_vectors_on_device(dev, from_text):
    """Every case of the committed golden fixture, straight from the device path (no oracle in the loop).
    A case whose shape a path declines must say so, not answer wrongly.  "query-text": the tree comes from the library's
    query parser over the query's text (the way a caller hands it over) instead of the hand-built one."""
    from test_oracle_golden import GOLDEN
    from test_query_parser import FIELDS, query_text
    m, ctx, batch = dev
    rankers = {"proximity_bm25": m.SPH_RANK_PROXIMITY_BM25, "bm25": m.SPH_RANK_BM25, "none": m.SPH_RANK_NONE,
               "wordcount": m.SPH_RANK_WORDCOUNT, "sph04": m.SPH_RANK_SPH04, "fieldmask": m.SPH_RANK_FIELDMASK}
    ops = {"and": m.SPH_QUERY_AND, "or": m.SPH_QUERY_OR, "andnot": m.SPH_QUERY_ANDNOT, "phrase": m.SPH_QUERY_PHRASE,
           "proximity": m.SPH_QUERY_PROXIMITY, "quorum": m.SPH_QUERY_QUORUM, "before": m.SPH_QUERY_BEFORE, "near": m.SPH_QUERY_NEAR,
           "notnear": m.SPH_QUERY_NOTNEAR, "sentence": m.SPH_QUERY_SENTENCE, "paragraph": m.SPH_QUERY_PARAGRAPH}

    def near_beyond_device(q):
        """the NEAR shape the device declines by design: more than two operands below another operator (none in the fixture)"""
        return False

    def tree(v, q):
        if "word" in q:
            tp = q.get("tp")
            return m.XQNode.keyword(v.get(q["word"], -1), q["pos"], q["mask"], field_start=tp in ("start", "startend"),  # -1: not in the dictionary
                                    field_end=tp in ("end", "startend"), field_max_pos=q.get("max_pos", 0) if tp == "limit" else 0)
        return m.XQNode(ops[q["op"]], [tree(v, k) for k in q["kids"]], None, q["mask"], q.get("opt", 0), unit_term=v.get(q["unit"], -1) if "unit" in q else -1)

    n_ok = 0
    n_near_declined = 0
    declined = []
    for name, corpus in GOLDEN["corpora"].items():
        W, R, H, v = make_hits(corpus["docs"], corpus["min_word_len"], bool(corpus.get("index_sp")))
        nf = max(len(d) for d in corpus["docs"])
        seg = m.Segment(ctx, m.index_from_hits(W, R, H, n_terms=len(v), total_docs=len(corpus["docs"]), n_fields=nf))
        cases = [c for c in GOLDEN["cases"] if c["corpus"] == name]
        def root(c):
            text = query_text(c["name"]) if from_text else None
            if text is None:
                return tree(v, c["query"])
            # (transform: what every query goes through between the parser and the ranker, sphTransformExtendedQuery's always-on part)
            return m.parse_query(text, FIELDS.get(name, []), corpus["min_word_len"], lookup=lambda w: v.get(w, -1), transform=True)

        qs = [m.Query(root(c), ranker=rankers[c["ranker"]], field_weights=c.get("field_weights"), plain_idf=bool(c.get("plain_idf")),
                      total_docs=c.get("total_docs", 0), local_docs={v[w]: n for w, n in c["local_docs"].items() if w in v} if "local_docs" in c else None)
              for c in cases]
        for c, r in zip(cases, batch.search(seg, qs)):
            if r.status == -2:
                if near_beyond_device(c["query"]):
                    n_near_declined += 1
                else:
                    declined.append(c["name"])
                continue
            assert r.status == 0 and not near_beyond_device(c["query"])
            got = [(corpus["ids"][i], int(w)) for i, w in zip(r.rowid, r.weight)]
            if "expect_row" in c:
                assert (c["expect_row"][0] in [i for i, _ in got]) == c["expect_row"][1], c["name"]
            elif "expect_ids" in c:
                assert sorted(i for i, _ in got) == sorted(c["expect_ids"]), c["name"]
                if "expect_weights" in c:
                    assert {str(i): w for i, w in got} == c["expect_weights"], c["name"]
            else:
                assert got[:c.get("limit", len(got))] == [tuple(x) for x in c["expect"]], c["name"]
            if "total_found" in c:
                assert r.total_found == c["total_found"]
            n_ok += 1
        seg.close()
    # packed path: every case answers -- the shapes the specialised hit pass declines (more than four keywords under a hit
    # ranker, phrases of five and more words, BEFORE / NEAR / NOTNEAR over phrases, groups and quorums) go through the generic
    # per-doc evaluator, NEAR over three and more operands (test_115) with its probe launch for the reference's never-reset
    # m_uFirstQpos.  VLB path: keyword / AND cases under BM25 / NONE
    if ctx_path(ctx) == 0:
        assert declined == [] and n_near_declined == 0 and n_ok == len(GOLDEN["cases"]), (declined, n_ok)
    else:
        assert n_ok >= 1


def test_generic_evaluator_vs_oracle(orc, dev):
    """Shapes only the generic per-doc evaluator (mrk_keval.h) takes, on random corpora, against the oracle: five to eight
    keywords under the hit rankers, long phrases, several phrase-like nodes in one query, BEFORE / NEAR / NOTNEAR over
    phrases, OR groups and quorums, quorums below other operators, position modifiers next to them."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("generic evaluator: packed path only")
    import os
    rng = np.random.default_rng(int(os.environ.get("MRK_FUZZ_SEED", 20260406)))
    rankers = [m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_SPH04, m.SPH_RANK_WORDCOUNT, m.SPH_RANK_MATCHANY, m.SPH_RANK_FIELDMASK,
               m.SPH_RANK_PROXIMITY, m.SPH_RANK_NONE]
    n_checked = n_generic = 0
    for trial in range(int(os.environ.get("MRK_FUZZ_TRIALS", 10))):
        n_docs = int(rng.choice([300, 3000, 20000]))
        nt = 8
        probs = [float(rng.choice([0.9, 0.6, 0.4, 0.2])) for _ in range(nt)]
        W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=int(rng.choice([6, 14])), end_markers=True)
        hi = m.index_from_hits(W, R, H, n_terms=nt + 1, total_docs=n_docs, skiplist_block_size=int(rng.choice([32, 128])),
                               hit_format=int(rng.integers(0, 2)), n_fields=3)
        qs = []
        for _ in range(24):
            pos = [0]

            def term(tp=False):
                pos[0] += 1
                t = int(rng.integers(0, nt + 1))
                mask = 0xFFFFFFFF if rng.random() < 0.8 else int(rng.integers(1, 8))
                if tp and rng.random() < 0.25:
                    k = int(rng.integers(0, 4))
                    return m.XQNode.keyword(t, pos[0], mask, field_start=k in (0, 2), field_end=k in (1, 2), field_max_pos=int(rng.integers(1, 5)) if k == 3 else 0)
                return kw(m, t, pos[0], mask)

            def words(n):  # a phrase-like node's words: distinct keywords, ascending positions (sometimes a gap)
                ts = [int(t) for t in rng.choice(nt, size=n, replace=False)]
                out = []
                for t in ts:
                    pos[0] += 1 if rng.random() < 0.85 else 2
                    out.append(kw(m, t, pos[0]))
                return out

            def phrase_like():
                r = rng.random()
                if r < 0.5:
                    return m.XQNode(m.SPH_QUERY_PHRASE, words(int(rng.integers(2, 4))))
                if r < 0.8:
                    return m.XQNode(m.SPH_QUERY_PROXIMITY, words(int(rng.integers(2, 4))), opt=int(rng.integers(1, 6)))
                return term()

            def operand():
                r = rng.random()
                if r < 0.4:
                    return term(True)
                if r < 0.75:
                    return phrase_like()
                if r < 0.9:
                    return m.XQNode(m.SPH_QUERY_OR, [term(), term()])
                return m.XQNode(m.SPH_QUERY_AND, [term(), term()])

            shape = rng.choice(["and_many", "long_phrase", "long_prox", "two_phrases", "before_ops", "near_phrases", "notnear_ops", "quorum_in_tree",
                                "big_quorum", "nested_near", "before_quorum", "mix", "near_many", "unit", "unit_in_tree"])
            if shape == "and_many":
                root = m.XQNode(m.SPH_QUERY_AND, [term() for _ in range(int(rng.integers(5, 8)))])
            elif shape == "long_phrase":
                root = m.XQNode(m.SPH_QUERY_PHRASE, words(int(rng.integers(5, 8))))
            elif shape == "long_prox":
                root = m.XQNode(m.SPH_QUERY_PROXIMITY, words(int(rng.integers(5, 8))), opt=int(rng.integers(2, 12)))
            elif shape == "two_phrases":
                root = m.XQNode(rng.choice([m.SPH_QUERY_AND, m.SPH_QUERY_OR, m.SPH_QUERY_MAYBE, m.SPH_QUERY_ANDNOT]), [phrase_like(), phrase_like()])
            elif shape == "before_ops":
                root = m.XQNode(m.SPH_QUERY_BEFORE, [operand() for _ in range(int(rng.integers(2, 4)))])
            elif shape == "near_phrases":
                root = m.XQNode(m.SPH_QUERY_NEAR, [phrase_like(), phrase_like()], opt=int(rng.integers(1, 8)))
            elif shape == "nested_near":
                inner = m.XQNode(m.SPH_QUERY_NEAR, [term(), term()], opt=int(rng.integers(1, 5)))
                root = m.XQNode(m.SPH_QUERY_NEAR, [inner, phrase_like()], opt=int(rng.integers(2, 9)))
            elif shape == "near_many":  # three and more operands at the root: the folded hits' query position depends on the docs before
                root = m.XQNode(m.SPH_QUERY_NEAR, [phrase_like() if rng.random() < 0.3 else term() for _ in range(int(rng.integers(3, 5)))], opt=int(rng.integers(1, 8)))
            elif shape in ("unit", "unit_in_tree"):  # SENTENCE / PARAGRAPH: any keyword can play the boundary word ("dot")
                def sp_item():
                    return m.XQNode(m.SPH_QUERY_PHRASE, words(2)) if rng.random() < 0.3 else term()
                root = m.XQNode(int(rng.choice([m.SPH_QUERY_SENTENCE, m.SPH_QUERY_PARAGRAPH])), [sp_item() for _ in range(int(rng.integers(2, 4)))],
                                field_mask=0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8)), unit_term=int(rng.integers(-1, nt + 1)))
                if shape == "unit_in_tree":
                    root = m.XQNode(rng.choice([m.SPH_QUERY_AND, m.SPH_QUERY_OR, m.SPH_QUERY_ANDNOT]), [root, term()] if rng.random() < 0.5 else [term(), root])
            elif shape == "notnear_ops":
                root = m.XQNode(m.SPH_QUERY_NOTNEAR, [operand(), operand()], opt=int(rng.integers(1, 8)))
            elif shape == "quorum_in_tree":
                qn = m.XQNode(m.SPH_QUERY_QUORUM, words(int(rng.integers(3, 5))), opt=2)
                root = m.XQNode(rng.choice([m.SPH_QUERY_AND, m.SPH_QUERY_OR, m.SPH_QUERY_MAYBE]), [qn, term()] if rng.random() < 0.5 else [term(), qn])
            elif shape == "big_quorum":
                n = int(rng.integers(5, 8))
                root = m.XQNode(m.SPH_QUERY_QUORUM, words(n), opt=int(rng.integers(2, n)))
            elif shape == "before_quorum":
                root = m.XQNode(m.SPH_QUERY_BEFORE, [m.XQNode(m.SPH_QUERY_QUORUM, words(3), opt=int(rng.choice([1, 2, 3]))), term(), phrase_like()])
            else:
                root = m.XQNode(m.SPH_QUERY_AND, [m.XQNode(m.SPH_QUERY_NOTNEAR, [phrase_like(), term()], opt=int(rng.integers(1, 6))),
                                                  m.XQNode(m.SPH_QUERY_OR, [phrase_like(), term(True)])])
            fw = None if rng.random() < 0.6 else [int(x) for x in rng.integers(-2, 5, size=3)]
            qs.append(m.Query(root, ranker=int(rng.choice(rankers)), max_matches=int(rng.choice([5, 100, 1000])), field_weights=fw))
        seg = m.Segment(ctx, hi)
        oi = orc_index_of(orc, hi)
        try:
            for q, g in zip(qs, batch.search(seg, qs)):
                try:
                    want = to_orc(orc, q).run(oi)
                except Exception:
                    continue  # a shape the oracle itself does not restate
                def n_kws(x):
                    return 1 if x.word is not None else sum(n_kws(k) for k in x.children)

                if g.status == -2 and n_kws(q.root) > 8:  # the one limit these shapes can reach: eight keywords per query
                    continue
                ok = g.status == 0 and g.total_found == want.total_found and len(g.rowid) == len(want.rowid) and (g.rowid == want.rowid).all() and \
                    (g.weight == want.weight).all()
                if not ok:  # say which docs differ (every match, by rowid) before failing
                    import dataclasses
                    qa = dataclasses.replace(q, ranker=m.SPH_RANK_NONE, max_matches=1000)
                    ga, wa = batch.search(seg, [qa])[0], to_orc(orc, qa).run(oi)
                    only_dev, only_orc = sorted(set(ga.rowid.tolist()) - set(wa.rowid.tolist()))[:5], sorted(set(wa.rowid.tolist()) - set(ga.rowid.tolist()))[:5]
                    hits = {}
                    for r in (only_dev + only_orc)[:3]:
                        sel = R == r
                        hits[r] = sorted((int(h) >> 24, int(h) & 0x7FFFFF, int(h >> 23) & 1, int(w)) for w, h in zip(W[sel], H[sel]))
                    raise AssertionError(f"status {g.status} total {g.total_found} vs {want.total_found}; ranker {q.ranker} K {q.max_matches} fw {q.field_weights}\n"
                                         f"dev {g.rowid[:6]} {g.weight[:6]}\norc {want.rowid[:6]} {want.weight[:6]}\nonly dev {only_dev} only orc {only_orc}\n"
                                         f"hits (field, pos, end, term) {hits}\n{q.root}")
                n_checked += 1
        finally:
            seg.close()
    assert n_checked >= 150


def test_proximity_bounds_with_shared_positions(orc, dev):
    """Two keywords may share a position (exact forms next to lemmas, blended parts): a proximity run then carries on through the
    shared position and LCS exceeds the number of keywords.  The weight bounds in front of the hit pass (prox_bounds) must allow
    for it -- a bound of "one per keyword" once dropped such a doc from the top K (MRK_FUZZ_SEED=777, trial 22 of the fuzz below).
    Corpus: four positions per field, so positions collide all the time; OR / AND-OR shapes over dense keywords, small and large
    K, field weights: device == oracle."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("pruning in front of the hit pass: packed path only")
    rng = np.random.default_rng(777022)
    n_docs = 30000
    probs = [0.7, 0.7, 0.5, 0.3]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=4, end_markers=True)
    hi = m.index_from_hits(W, R, H, n_terms=len(probs) + 1, total_docs=n_docs, skiplist_block_size=128, hit_format=1, n_fields=3)
    seg = m.Segment(ctx, hi)
    oi = orc_index_of(orc, hi)
    qs = []
    for k_ in (1, 10, 100, 1000):
        for fw in (None, [7, 8, 8], [1, 5, -2]):
            a, b, c = kw(m, 0, 1), kw(m, 1, 2), kw(m, 2, 3)
            for root in (QUORUM(m, 1, a, b), OR(m, a, b), m.XQNode.AND(OR(m, a, b), c), OR(m, a, OR(m, b, c)), MAYBE(m, a, b)):
                qs.append(m.Query(root, ranker=m.SPH_RANK_PROXIMITY_BM25 if k_ != 10 else m.SPH_RANK_PROXIMITY, max_matches=k_, field_weights=fw))
    try:
        n_ok = 0
        for i in range(0, len(qs), 24):
            part = qs[i:i + 24]
            for q, g in zip(part, batch.search(seg, part)):
                assert g.status == 0
                want = to_orc(orc, q).run(oi)
                assert g.total_found == want.total_found
                assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all(), (q.max_matches, q.field_weights, q.ranker)
                n_ok += 1
        assert n_ok == len(qs)
    finally:
        seg.close()


# ------------------------------------------------------------------ many tiny corpora: boundaries of blocks / windows
def test_fuzz_tiny_corpora_all_shapes(orc, dev):
    """Corpora of 1 .. 5000 docs (below / around one 128-doc block and one 2048-rowid window), every query shape the
    device accepts, dense and sparse keywords, dead rows, rowid bases: device == oracle."""
    m, ctx, batch = dev
    import os
    rng = np.random.default_rng(int(os.environ.get("MRK_FUZZ_SEED", 424242)))
    tuned = [kv.split("=") for kv in os.environ.get("MRK_FUZZ_CTX", "").split(",") if kv]  # e.g. prox_prune=0,bt_phrase=0: localise a mismatch
    for k_, v_ in tuned:
        ctx.set(k_, int(v_))
    packed = ctx_path(ctx) == 0
    n_checked = 0
    for trial in range(int(os.environ.get("MRK_FUZZ_TRIALS", 28))):  # MRK_FUZZ_TRIALS=500 for a long soak
        n_docs = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 257, 2047, 2048, 2049, 4097, 5000, 20000]))
        nt = int(rng.integers(3, 7))
        probs = [float(rng.choice([1.0, 0.7, 0.3, 0.1, 0.02, 0.6])) for _ in range(nt)]
        W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=int(rng.choice([4, 30])), end_markers=True)
        if len(W) == 0:
            continue
        block = int(rng.choice([32, 64, 128]))
        fmt = int(rng.integers(0, 2))
        hi = m.index_from_hits(W, R, H, n_terms=nt + 1, total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
        qs = []
        for _ in range(14):
            k = int(rng.integers(1, min(4, nt) + 1))
            ts = [int(t) for t in rng.choice(nt + 1, size=k, replace=bool(packed and rng.random() < 0.15))]
            masks = [0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8)) for _ in ts]
            kws = [kw(m, t, i + 1, mk) for i, (t, mk) in enumerate(zip(ts, masks))]
            shape = (rng.choice(["and", "or", "andnot", "maybe", "phrase", "mixed", "proximity", "quorum", "phrase_in_tree", "before",
                                 "before_in_tree", "real_quorum"])
                     if (packed and k > 1) else "and")
            if packed and shape not in ("phrase", "proximity", "quorum", "real_quorum") and rng.random() < 0.3:
                for i in range(2 if shape == "phrase_in_tree" else 0, k):  # position modifiers ('^word', 'word$', '^word$',
                    # '@field[N] word') -- not on the words of a phrase
                    tp = int(rng.integers(0, 8))
                    if tp < 4:
                        kws[i] = m.XQNode.keyword(ts[i], i + 1, masks[i], field_start=tp in (0, 2), field_end=tp in (1, 2),
                                                  field_max_pos=int(rng.integers(1, 4)) if tp == 3 else 0)
            if k == 1:
                root = kws[0]
            elif shape == "and":
                root = m.XQNode.AND(*kws)
            elif shape == "or":
                root = OR(m, *kws)
            elif shape == "andnot":
                root = ANDNOT(m, kws[0], kws[1])
            elif shape == "maybe":
                root = MAYBE(m, kws[0], kws[1])
            elif shape == "phrase":
                root = PHRASE(m, *kws)
            elif shape == "proximity":
                root = PROXIMITY(m, int(rng.integers(1, 6)), *kws)
            elif shape == "quorum":
                root = m.XQNode(m.SPH_QUERY_QUORUM, kws, None, 0xFFFFFFFF, 1 if rng.random() < 0.5 else k)
            elif shape == "before":
                root = BEFORE(m, *kws)
            elif shape == "before_in_tree" and k > 2:
                bf = BEFORE(m, *kws[:2])
                root = [OR(m, bf, *kws[2:]), m.XQNode.AND(bf, *kws[2:]), ANDNOT(m, kws[2], bf), MAYBE(m, bf, kws[2])][int(rng.integers(0, 4))]
            elif shape == "real_quorum" and k > 2 and len(set(ts)) == k:
                root = m.XQNode(m.SPH_QUERY_QUORUM, [kw(m, t, i + 1) for i, t in enumerate(ts)], None, 0xFFFFFFFF, int(rng.integers(2, k)))
            elif shape == "phrase_in_tree" and k > 2:
                ph = PHRASE(m, *kws[:2]) if rng.random() < 0.5 else PROXIMITY(m, 3, *kws[:2])
                root = [OR(m, ph, *kws[2:]), m.XQNode.AND(ph, *kws[2:]), ANDNOT(m, kws[2], ph), MAYBE(m, ph, kws[2])][int(rng.integers(0, 4))]
            else:
                root = m.XQNode.AND(OR(m, *kws[:2]), *kws[2:]) if k > 2 else OR(m, *kws)
            rk = [m.SPH_RANK_BM25, m.SPH_RANK_NONE] + ([m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_PROXIMITY, m.SPH_RANK_WORDCOUNT,
                                                       m.SPH_RANK_MATCHANY, m.SPH_RANK_FIELDMASK, m.SPH_RANK_SPH04] if packed else [])
            fl = None
            if packed and rng.random() < 0.2:  # attribute filters over the rows set below
                fl = [m.Filter(0, 32, values=sorted(set(int(v) for v in rng.integers(0, 6, 2))), exclude=bool(rng.random() < 0.3))
                      if rng.random() < 0.5 else m.Filter(32, 64, min=int(rng.integers(-3, 3)), max=int(rng.integers(3, 9)))]
            qs.append(m.Query(root, ranker=int(rng.choice(rk)), max_matches=int(rng.choice([1, 3, 1000, 1024])),
                              field_weights=[int(x) for x in rng.integers(-2, 9, 3)] if rng.random() < 0.4 else None, filters=fl))
        base = int(rng.choice([0, 7, 1 << 20]))
        seg = m.Segment(ctx, hi, rowid_base=base)
        oi = orc_index_of(orc, hi)
        if packed:
            arows = np.zeros((n_docs, 3), np.uint32)
            arows[:, 0] = rng.integers(0, 6, n_docs)
            big = rng.integers(-5, 12, n_docs).astype(np.int64).view(np.uint64)
            arows[:, 1], arows[:, 2] = (big & np.uint64(0xFFFFFFFF)).astype(np.uint32), (big >> np.uint64(32)).astype(np.uint32)
            seg.set_attrs(arows)
            oi.attrs = arows
        if rng.random() < 0.5:
            dead = np.zeros((n_docs + 31) // 32, np.uint32)
            killed = rng.choice(n_docs, size=max(1, n_docs // 5), replace=False)
            np.bitwise_or.at(dead, killed >> 5, (np.uint32(1) << (killed & 31).astype(np.uint32)))
            seg.set_dead_rows(dead)
            oi.dead_rows = dead
        try:
            for qi_, (q, g) in enumerate(zip(qs, batch.search(seg, qs))):
                if g.status == -2:
                    continue
                want = to_orc(orc, q).run(oi)
                what = (trial, n_docs, qi_, q.ranker, q.max_matches, q.field_weights, base, block, fmt)
                assert g.status == 0
                if os.environ.get("MRK_FUZZ_CTX") is not None and not ((g.rowid == want.rowid).all() and (g.weight == want.weight).all()):
                    def tree(nd):
                        return ("t%d@%d" % (nd.word.term_id, nd.word.atom_pos)) if nd.word is not None else (nd.op, nd.opt, [tree(c) for c in nd.children])
                    bad = [(i_, int(a), int(b), int(c), int(d)) for i_, (a, b, c, d) in enumerate(zip(g.rowid, g.weight, want.rowid, want.weight)) if a != c or b != d]
                    print("MISMATCH", what, tree(q.root), "n", len(g.rowid), len(want.rowid), "total", g.total_found, "docs", [int(x) for x in hi.dict["docs"]],
                          "first", bad[:6], "last", bad[-3:], "dead" if oi.dead_rows is not None else "", flush=True)
                assert g.total_found == want.total_found, what + (g.total_found, want.total_found)
                assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all(), what + (
                    [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(g.rowid, g.weight, want.rowid, want.weight) if a != c or b != d][:4],)
                n_checked += 1
        finally:
            seg.close()
    assert n_checked > 200


# ------------------------------------------------------------------ more ties than the candidate list holds
def test_candidate_overflow_is_rerun(orc, dev):
    """1.3 M docs that all match with the very same weight: nothing can be pruned, the 2^20-slot candidate list
    overflows, and the query is rerun with a full-size list instead of failing or truncating."""
    _tree_only(dev)
    m, ctx, batch = dev
    n_docs = 1_300_000
    rows = np.arange(n_docs, dtype=np.uint32)
    W = np.concatenate([np.full(n_docs, 1, np.uint64), np.full(n_docs, 2, np.uint64)])
    R = np.concatenate([rows, rows])
    H = np.concatenate([np.full(n_docs, (1 << 24) | 1, np.uint32), np.full(n_docs, (1 << 24) | 2, np.uint32)])
    hi = m.index_from_hits(W, R, H, n_terms=2, total_docs=n_docs, n_fields=2)
    qs = [m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2)), ranker=m.SPH_RANK_BM25, max_matches=1000),
          m.Query(OR(m, kw(m, 0, 1), kw(m, 1, 2)), ranker=m.SPH_RANK_BM25, max_matches=10),
          m.Query(kw(m, 0, 1), ranker=m.SPH_RANK_FIELDMASK, max_matches=100),
          # the generic per-doc evaluator's rerun: 'a NOTNEAR/3 "b a"' -- the phrase never occurs, every doc of a stays
          m.Query(m.XQNode(m.SPH_QUERY_NOTNEAR, [kw(m, 0, 1), m.XQNode(m.SPH_QUERY_PHRASE, [kw(m, 1, 2), kw(m, 0, 3)])], opt=3), ranker=m.SPH_RANK_PROXIMITY_BM25,
                  max_matches=50)]
    for inv in (64, 0):  # bitmap kernel and block kernel
        ctx.set("bitmap_inv", inv)
        seg = m.Segment(ctx, hi)
        try:
            got = batch.search(seg, qs)
            for q, g in zip(qs, got):
                assert g.status == 0
                assert g.total_found == n_docs
                assert list(g.rowid) == list(range(len(g.rowid))) and len(g.rowid) == q.max_matches  # ties: lowest rowids win
                assert len(set(int(w) for w in g.weight)) == 1
            for i in (0, 3):
                want = to_orc(orc, qs[i]).run(orc_index_of(orc, hi))
                assert (got[i].weight == want.weight).all() and (got[i].rowid == want.rowid).all()
        finally:
            seg.close()
    ctx.set("bitmap_inv", 64)


# ------------------------------------------------------------------ ExtQuorum_c proper ('"a b c d"/N', 1 < N < words)
def QUORUM(m, thr, *k, mask=0xFFFFFFFF):
    return m.XQNode(m.SPH_QUERY_QUORUM, list(k), None, mask, thr)


@pytest.mark.parametrize("block,fmt", [(128, 1), (32, 0)])
def test_quorum_node(orc, dev, block, fmt):
    """A real quorum node: docs holding at least N of the keywords, tfidf summed in the order ExtQuorum_c's child list
    has at that rowid (rare keywords' doclists end early and reorder it), hits sorted without the end flag."""
    _tree_only(dev)
    m, ctx, batch = dev
    rng = np.random.default_rng(606 + block + fmt)
    n_docs = 40000
    probs = [0.5, 0.3, 0.12, 0.05, 0.02, 0.004, 0.0008, 0.9]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=30, end_markers=True)
    # make a few keywords stop early, so that the child list is reordered well before the corpus ends
    keep = ~(((W == 3) & (R > 9000)) | ((W == 5) & (R > 21000)) | ((W == 2) & (R > 33000)))
    W, R, H = W[keep], R[keep], H[keep]
    nt = len(probs) + 1  # last keyword has no postings
    hi = m.index_from_hits(W, R, H, n_terms=nt, total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
    qs = []
    for i in range(90):
        k = int(rng.integers(3, 7))
        ts = [int(t) for t in rng.choice(nt, size=k, replace=False)]
        thr = int(rng.integers(max(2, k - 2), k))  # k - thr + 1 driver keywords (<= 4 per query on the device path)
        kws = [kw(m, t, j + 1) for j, t in enumerate(ts)]
        if rng.random() < 0.3:
            rng.shuffle(kws)  # children arrive in any order; the node sorts them by query position
        root = QUORUM(m, thr, *kws)
        rk = int(rng.choice([m.SPH_RANK_BM25, m.SPH_RANK_NONE]))
        if k <= 4 and i % 2:
            rk = int(rng.choice([m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_PROXIMITY, m.SPH_RANK_WORDCOUNT, m.SPH_RANK_SPH04]))
        if i % 5 == 0 and rk in (m.SPH_RANK_BM25, m.SPH_RANK_NONE):  # below other operators (weight-sum rankers only)
            extra = kw(m, [t for t in range(nt) if t not in ts][0], k + 1)
            root = [m.XQNode.AND(root, extra), OR(m, root, extra), ANDNOT(m, root, extra)][i % 3]
        qs.append(m.Query(root, ranker=rk, max_matches=int(rng.choice([10, 1000])),
                          field_weights=[int(x) for x in rng.integers(-2, 9, 3)] if rng.random() < 0.4 else None))
    check_batch(orc, dev, hi, qs)
    # declined shapes stay loud: a field-limited keyword, a repeated keyword
    seg = m.Segment(ctx, hi)
    r = batch.search(seg, [m.Query(QUORUM(m, 2, kw(m, 0, 1, 0b01), kw(m, 1, 2), kw(m, 2, 3)), ranker=m.SPH_RANK_BM25),
                           m.Query(QUORUM(m, 2, kw(m, 0, 1), kw(m, 0, 2), kw(m, 2, 3)), ranker=m.SPH_RANK_BM25)])
    assert r[0].status == -2 and r[1].status == -2
    seg.close()


# ------------------------------------------------------------------ real index files
def test_index_files_on_device(orc, dev, tmp_path):
    """Indexes opened from files -- the reference's own fixtures and a larger one written in the documented layout
    (tests/test_index_files.py) -- searched on the device with their .spm dead rows installed."""
    import os
    from test_index_files import IDX, synth, write_index
    m, ctx, batch = dev
    # 1. reference fixtures: one row each, every keyword and keyword pair under every ranker
    # (+ two RT RAM segments, mrk_rt_ram_open: decoded from the RT codecs and searched like any other segment)
    for name in ("t250_plain2", "t233_test", "t233_reload", "t406_index0", "rt:t406_index", "rt:t406_idx320"):
        hi = m.open_rt_ram(os.path.join(IDX, name[3:]))[0] if name.startswith("rt:") else m.open_index(os.path.join(IDX, name))
        qs = []
        ws = list(range(len(hi.words)))
        for rk in (m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_WORDCOUNT, m.SPH_RANK_SPH04, m.SPH_RANK_PROXIMITY):
            hit_rk = rk not in (m.SPH_RANK_BM25, m.SPH_RANK_NONE)
            if hit_rk and ctx_path(ctx) != 0:
                continue
            for a in ws:
                qs.append(m.Query(kw(m, a, 1), ranker=rk))
                for b in ws:
                    if a != b:
                        qs.append(m.Query(m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), ranker=rk))
                        if hit_rk:
                            qs.append(m.Query(PHRASE(m, kw(m, a, 1), kw(m, b, 2)), ranker=rk))
                        elif ctx_path(ctx) == 0:
                            qs.append(m.Query(OR(m, kw(m, a, 1), kw(m, b, 2)), ranker=rk))
        check_batch(orc, dev, hi, qs)
        if name == "t250_plain2" and ctx_path(ctx) == 0:  # filters by attribute name over the ingested .spa rows
            seg = m.Segment(ctx, hi)
            seg.set_attrs(hi.attr_rows)
            _, off, cnt = hi.attrs["id"]
            third, fourth = hi.find_word("third"), hi.find_word("fourth")
            root = OR(m, kw(m, third, 1), kw(m, fourth, 2))
            res = batch.search(seg, [m.Query(root, ranker=m.SPH_RANK_BM25, filters=[m.Filter(off, cnt, values=[4])]),
                                     m.Query(root, ranker=m.SPH_RANK_BM25, filters=[m.Filter(off, cnt, min=3, max=3)]),
                                     m.Query(root, ranker=m.SPH_RANK_BM25, filters=[m.Filter(hi.attrs["mode"][1], hi.attrs["mode"][2], values=[2])]),
                                     m.Query(root, ranker=m.SPH_RANK_BM25, filters=[m.Filter(hi.attrs["mode"][1], hi.attrs["mode"][2], values=[2], exclude=True)])])
            assert [r.status for r in res] == [0, 0, 0, 0]
            assert [r.rowid.tolist() for r in res] == [[1], [0], [0, 1], []] and [r.total_found for r in res] == [1, 1, 2, 0]
            seg.close()
    # 2. a written index: dictionary lookups by keyword, dead rows from the .spm file
    src = synth(m, n_terms=60, n_docs=20000, block=32)
    words = sorted("kw%04d" % (t * 37 % 1009) for t in range(len(src.dict)))
    dead = [int(r) for r in np.random.default_rng(4).choice(20000, 700, replace=False)]
    prefix = str(tmp_path / "idx")
    write_index(prefix, src, words, dead_rows=dead)
    hi = m.open_index(prefix)
    assert hi.info["n_dead"] == 700
    seg = m.Segment(ctx, hi)
    seg.set_dead_rows(hi.dead_bitmap)
    oi = orc_index_of(orc, hi)
    oi.dead_rows = hi.dead_bitmap
    rng = np.random.default_rng(5)
    qs = []
    for _ in range(40):
        a, b, c = (hi.find_word(words[int(t)]) for t in rng.choice(len(words), 3, replace=False))
        assert min(a, b, c) >= 0
        rk = int(rng.choice([m.SPH_RANK_BM25, m.SPH_RANK_NONE] + ([m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_SPH04] if ctx_path(ctx) == 0 else [])))
        root = m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)) if rng.random() < 0.6 or ctx_path(ctx) != 0 else m.XQNode.AND(OR(m, kw(m, a, 1), kw(m, b, 2)), kw(m, c, 3))
        qs.append(m.Query(root, ranker=rk, max_matches=200))
    try:
        got = batch.search(seg, qs)
        n_found = 0
        for q, g in zip(qs, got):
            want = to_orc(orc, q).run(oi)
            assert g.status == 0 and g.total_found == want.total_found
            assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
            assert not set(g.rowid.tolist()) & set(dead)
            n_found += g.total_found
        assert n_found > 1000
    finally:
        seg.close()


# ------------------------------------------------------------------ term position modifiers (ExtTermPos_T)
@pytest.mark.parametrize("block,fmt", [(128, 1), (32, 0)])
def test_term_position_modifiers(orc, dev, block, fmt):
    """'^word', 'word$', '^word$' and '@field[N] word' (searchnode.cpp:2259-2405): alone, in AND chains (the modified
    keyword sorts last, GetDocsCount() = INT_MAX), below OR / MAYBE / ANDNOT, with field limits, under every ranker.
    The reference's own vectors for this node (test_055, test_080) are in the golden fixture."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("position modifiers run on the packed path")
    rng = np.random.default_rng(8080 + block)
    n_docs = 30000
    probs = [0.4, 0.2, 0.1, 0.03, 0.008, 0.002, 0.0005]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=6, end_markers=True)
    nt = len(probs) + 1
    hi = m.index_from_hits(W, R, H, n_terms=nt, total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
    rankers = [m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_WORDCOUNT, m.SPH_RANK_SPH04,
               m.SPH_RANK_PROXIMITY, m.SPH_RANK_MATCHANY, m.SPH_RANK_FIELDMASK]

    def tkw(t, pos, mask=0xFFFFFFFF):
        kind = int(rng.integers(0, 5))
        return m.XQNode.keyword(t, pos, mask, field_start=kind in (1, 3), field_end=kind in (2, 3),
                                field_max_pos=int(rng.integers(1, 5)) if kind == 4 else 0)

    qs = []
    for i in range(160):
        k = int(rng.integers(1, 5))
        ts = [int(t) for t in rng.choice(nt, size=k, replace=False)]
        mask = lambda: int(rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 0b011, 0b100, 0b110]))
        leaves = [tkw(t, j + 1, mask()) if rng.random() < 0.6 else kw(m, t, j + 1, mask()) for j, t in enumerate(ts)]
        if not any(n.term_pos() for n in leaves):
            leaves[0] = m.XQNode.keyword(ts[0], 1, field_start=True)
        shape = int(rng.integers(0, 5)) if k > 1 else 0
        if k == 1:
            root = leaves[0]
        elif shape == 0:
            root = m.XQNode.AND(*leaves)
        elif shape == 1:
            root = OR(m, *leaves)
        elif shape == 2:
            root = m.XQNode.AND(OR(m, *leaves[:2]), *leaves[2:]) if k > 2 else OR(m, *leaves)
        elif shape == 3:
            root = ANDNOT(m, m.XQNode.AND(*leaves[:-1]) if k > 2 else leaves[0], leaves[-1])
        else:
            root = m.XQNode(m.SPH_QUERY_MAYBE, [leaves[0], leaves[1]]) if k == 2 else m.XQNode.AND(leaves[0], OR(m, *leaves[1:]))
        fw = [int(x) for x in rng.integers(-2, 9, 3)] if rng.random() < 0.3 else None
        qs.append(m.Query(root, ranker=int(rng.choice(rankers)), max_matches=int(rng.choice([20, 1000])), field_weights=fw))
    seg = m.Segment(ctx, hi)
    oi = orc_index_of(orc, hi)
    n_found = n_run = 0
    try:
        got = []
        for i in range(0, len(qs), batch.max_queries):
            got += batch.search(seg, qs[i:i + batch.max_queries])
        for q, g in zip(qs, got):
            if g.status == -2:
                continue
            want = to_orc(orc, q).run(oi)
            assert g.status == 0 and g.total_found == want.total_found
            assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
            n_found += g.total_found
            n_run += 1
        assert n_run >= 150 and n_found > 20000
    finally:
        seg.close()


# ------------------------------------------------------------------ BEFORE operator (ExtOrder_c)

@pytest.mark.parametrize("block,fmt", [(128, 1), (32, 0)])
def test_before_operator(orc, dev, block, fmt):
    """'a << b << c' over 2..4 plain keywords (repeated ones included, position modifiers too): at the root and below
    AND / OR / MAYBE / ANDNOT, every ranker.  The doc is the first operand's doc (its tfidf and fields alone); the hits of
    every complete in-order run reach the ranker.  Reference vectors: test_052 in the golden fixture."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("BEFORE runs on the packed path")
    rng = np.random.default_rng(5252 + block)
    n_docs = 30000
    probs = [0.5, 0.35, 0.2, 0.1, 0.04, 0.01, 0.002]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=12, end_markers=True)
    nt = len(probs) + 1
    hi = m.index_from_hits(W, R, H, n_terms=nt, total_docs=n_docs, skiplist_block_size=block, hit_format=fmt, n_fields=3)
    rankers = [m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_WORDCOUNT, m.SPH_RANK_SPH04,
               m.SPH_RANK_PROXIMITY, m.SPH_RANK_MATCHANY, m.SPH_RANK_FIELDMASK]
    qs = []
    for i in range(200):
        k = int(rng.integers(2, 5))
        ts = [int(t) for t in rng.choice(nt, size=k, replace=bool(i % 3 == 0))]
        mask = lambda: int(rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0b011, 0b110]))
        n_ord = k if i % 2 == 0 else int(rng.integers(2, k + 1))
        leaves = []
        for j, t in enumerate(ts):
            tp = int(rng.integers(0, 12))
            leaves.append(m.XQNode.keyword(t, j + 1, mask(), field_start=tp == 1, field_end=tp == 2, field_max_pos=6 if tp == 3 else 0))
        node = BEFORE(m, *leaves[:n_ord])
        rest = leaves[n_ord:]
        shape = int(rng.integers(0, 4))
        if not rest:
            root = node
        elif shape == 0:
            root = m.XQNode.AND(node, *rest)
        elif shape == 1:
            root = OR(m, node, *rest)
        elif shape == 2:
            root = ANDNOT(m, node, rest[0])
        else:
            root = m.XQNode(m.SPH_QUERY_MAYBE, [rest[0], node])
        fw = [int(x) for x in rng.integers(-2, 9, 3)] if rng.random() < 0.3 else None
        qs.append(m.Query(root, ranker=int(rng.choice(rankers)), max_matches=int(rng.choice([20, 1000])), field_weights=fw))
    seg = m.Segment(ctx, hi)
    oi = orc_index_of(orc, hi)
    n_found = n_run = 0
    try:
        got = []
        for i in range(0, len(qs), batch.max_queries):
            got += batch.search(seg, qs[i:i + batch.max_queries])
        for q, g in zip(qs, got):
            if g.status == -2:
                continue
            want = to_orc(orc, q).run(oi)
            assert g.status == 0 and g.total_found == want.total_found
            assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
            n_found += g.total_found
            n_run += 1
        assert n_run >= 180 and n_found > 20000, (n_run, n_found)
    finally:
        seg.close()


# ------------------------------------------------------------------ MVA filters over the blob pool
def test_mva_filters(orc, dev):
    """Filter_MVAValues_Any_c / _All_c / Filter_MVARange_Any_c / _All_c (sphinxfilter.cpp:340-383) over 32- and 64-bit multi-value
    attributes in the blob pool: device == oracle, with trees, hit rankers, a second plain filter, dead rows; a damaged pool is
    refused at mrk_segment_set_blobs."""
    from test_oracle_filters import mva_rows
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("filters run on the packed path")
    rng = np.random.default_rng(4242)
    n_docs = 30000
    probs = [0.5, 0.3, 0.1, 0.02]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=2, max_pos=12)
    hi = m.index_from_hits(W, R, H, n_terms=len(probs), total_docs=n_docs, n_fields=2)
    rows, pool, v32, v64 = mva_rows(rng, n_docs)
    rows = np.concatenate([rows, rng.integers(0, 9, (n_docs, 1)).astype(np.uint32)], axis=1)  # + a plain attribute at dword 4
    B = 1 << 33
    F = m.Filter

    def rand_mva():
        attr = int(rng.integers(0, 2))
        k = int(rng.integers(0, 4))
        common = dict(mva_bits=(32, 64)[attr], blob_attr_id=attr, n_blob_attrs=2, exclude=bool(rng.random() < 0.25))
        scale = 1 if attr == 0 else B
        lo = int(rng.integers(0, 35)) if attr == 0 else int(rng.integers(-20, 15))
        if k == 0:
            return F(0, 0, values=sorted(set(int(x) * scale for x in (rng.integers(0, 40, 4) if attr == 0 else rng.integers(-20, 20, 4)))), **common)
        if k == 1:
            return F(0, 0, values=sorted(set(int(x) * scale for x in (rng.integers(0, 40, 8) if attr == 0 else rng.integers(-20, 20, 8)))), mva_all=True, **common)
        return F(0, 0, min=lo * scale, max=(lo + int(rng.integers(0, 12))) * scale, mva_all=(k == 3), has_equal_min=bool(rng.random() < 0.6),
                 has_equal_max=bool(rng.random() < 0.6), **common)

    qs = []
    rankers = [m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_SPH04]
    for i in range(120):
        a, b = (int(t) for t in rng.choice(len(probs), 2, replace=False))
        root = [kw(m, a, 1), m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), OR(m, kw(m, a, 1), kw(m, b, 2)), PHRASE(m, kw(m, a, 1), kw(m, b, 2))][i % 4]
        fl = [rand_mva()] + ([F(128, 32, values=[1, 4, 7])] if rng.random() < 0.3 else [])
        qs.append(m.Query(root, ranker=int(rng.choice(rankers)), max_matches=int(rng.choice([20, 1000])), filters=fl))
    seg = m.Segment(ctx, hi)
    oi = orc_index_of(orc, hi)
    oi.attrs, oi.blobs = rows, pool
    try:
        seg.set_attrs(rows)
        assert batch.search(seg, qs[:1])[0].status == -2  # no blob pool yet: declined
        bad = pool.copy()
        bad[int(rows[77][2]) + 1] = 250  # a length that runs past the pool's end for a short row
        with pytest.raises(m.MrkError):
            seg.set_blobs(bad[: int(rows[-1][2])], 2, rows)  # (and the last rows' offsets lie past the truncated pool)
        seg.set_blobs(pool, 2, rows)
        n_found = 0
        for q, g in zip(qs, batch.search(seg, qs)):
            want = to_orc(orc, q).run(oi)
            assert g.status == 0 and g.total_found == want.total_found, (g.status, g.total_found, want.total_found, q.filters)
            assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
            n_found += g.total_found
        assert n_found > 20000, n_found
        with pytest.raises(m.MrkError, match="blob attribute"):  # a locator that disagrees with the pool the segment was given: a caller's error
            batch.search(seg, [m.Query(kw(m, 0, 1), filters=[F(0, 0, values=[1], mva_bits=32, blob_attr_id=1, n_blob_attrs=3)])])
    finally:
        seg.close()


# ------------------------------------------------------------------ attribute filters (EarlyReject)
def test_attribute_filters(orc, dev):
    """SPH_FILTER_VALUES / SPH_FILTER_RANGE over integer attributes of the row-wise storage (32-bit, 64-bit, bit fields) and
    SPH_FILTER_FLOATRANGE over a float attribute (negative values, infinities, a NaN row: fails every bound),
    include / exclude, open ranges, strict bounds, two filters at once; with trees, phrases, every ranker and dead rows.
    Rejected rows are neither ranked nor counted (sphinxsearch.cpp:1055-1064, sphinx.cpp:11903-11917)."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("filters run on the packed path")
    rng = np.random.default_rng(31337)
    n_docs = 40000
    probs = [0.45, 0.3, 0.1, 0.03, 0.006, 0.001]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=20, end_markers=True)
    nt = len(probs)
    hi = m.index_from_hits(W, R, H, n_terms=nt, total_docs=n_docs, n_fields=3)
    # rows: [id lo, id hi (64-bit id), gid (32-bit), flags: 3-bit field at bit 4 and 1-bit at bit 9 of dword 3, price (64-bit, signed use)]
    rows = np.zeros((n_docs, 7), np.uint32)  # (dword 6: a float attribute)
    ids = np.arange(n_docs, dtype=np.uint64) * np.uint64(7) + np.uint64(1 << 33)
    rows[:, 0], rows[:, 1] = (ids & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ids >> np.uint64(32)).astype(np.uint32)
    rows[:, 2] = rng.integers(0, 50, n_docs)
    rows[:, 3] = (rng.integers(0, 8, n_docs) << 4) | (rng.integers(0, 2, n_docs) << 9) | (rng.integers(0, 16, n_docs)) | (rng.integers(0, 1 << 20, n_docs) << 10 << 1)
    price = rng.integers(-1000, 100000, n_docs).astype(np.int64)
    rows[:, 4], rows[:, 5] = (price.view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32), (price.view(np.uint64) >> np.uint64(32)).astype(np.uint32)
    score = (rng.normal(0.0, 50.0, n_docs)).astype(np.float32)
    score[::997] = np.float32("inf")
    score[5::1999] = np.float32("nan")
    score[3::1013] = np.float32(-0.0)
    rows[:, 6] = score.view(np.uint32)
    F = m.Filter

    def rand_filter():
        k = int(rng.integers(0, 6))
        if k == 5:
            lo = float(np.float32(rng.normal(0.0, 40.0)))
            return F(192, 32, fmin=lo if rng.random() < 0.9 else float("-inf"), fmax=float(np.float32(lo + abs(rng.normal(0.0, 60.0)))) if rng.random() < 0.9 else float("inf"),
                     exclude=bool(rng.random() < 0.25), has_equal_min=bool(rng.random() < 0.7), has_equal_max=bool(rng.random() < 0.7))
        excl = bool(rng.random() < 0.25)
        if k == 0:
            return F(64, 32, values=sorted(set(int(v) for v in rng.integers(0, 50, int(rng.integers(1, 9))))), exclude=excl)
        if k == 1:
            lo = int(rng.integers(0, 40))
            return F(64, 32, min=lo, max=lo + int(rng.integers(0, 20)), exclude=excl, has_equal_min=bool(rng.random() < 0.7), has_equal_max=bool(rng.random() < 0.7))
        if k == 2:
            return F(96 + 4, 3, values=sorted(set(int(v) for v in rng.integers(0, 8, 3))), exclude=excl) if rng.random() < 0.5 else F(96 + 9, 1, values=[1], exclude=excl)
        if k == 3:
            lo = int(rng.integers(-1000, 90000))
            return F(128, 64, min=lo, max=lo + int(rng.integers(0, 30000)), exclude=excl, open_left=bool(rng.random() < 0.2), open_right=bool(rng.random() < 0.2))
        base = (1 << 33) + 7 * int(rng.integers(0, n_docs))
        return F(0, 64, min=base, max=base + 7 * 5000, exclude=excl)

    qs = []
    rankers = [m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_SPH04, m.SPH_RANK_WORDCOUNT]
    for i in range(150):
        a, b, c = (int(t) for t in rng.choice(nt, 3, replace=False))
        shape = i % 6
        root = [kw(m, a, 1), m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), OR(m, kw(m, a, 1), kw(m, b, 2)), m.XQNode.AND(OR(m, kw(m, a, 1), kw(m, b, 2)), kw(m, c, 3)),
                PHRASE(m, kw(m, a, 1), kw(m, b, 2)), ANDNOT(m, kw(m, a, 1), kw(m, b, 2))][shape]
        fl = [rand_filter() for _ in range(int(rng.integers(1, 3)))]
        wf = None
        if rng.random() < 0.35:  # a filter on the weight itself ('WHERE weight() ...'): m_pWeightFilter
            lo = int(rng.integers(0, 3000))
            wf = [F(0, 32, min=lo, max=lo + int(rng.integers(0, 4000)), exclude=bool(rng.random() < 0.3), has_equal_min=bool(rng.random() < 0.7))
                  if rng.random() < 0.8 else F(0, 32, values=[1, 1500, 1551, 1577, 2500])]
            if rng.random() < 0.3:
                fl = None
        qs.append(m.Query(root, ranker=int(rng.choice(rankers)), max_matches=int(rng.choice([30, 1000])), filters=fl, weight_filters=wf))
    seg = m.Segment(ctx, hi)
    oi = orc_index_of(orc, hi)
    oi.attrs = rows
    dead = np.zeros((n_docs + 31) // 32, np.uint32)
    for r in rng.choice(n_docs, 900, replace=False):
        dead[r >> 5] |= np.uint32(1 << (int(r) & 31))
    r0 = batch.search(seg, qs[:1])[0]
    assert r0.status == -2  # no attribute rows yet: declined, not answered
    seg.set_attrs(rows)
    seg.set_dead_rows(dead)
    oi.dead_rows = dead
    n_found = 0
    try:
        got = batch.search(seg, qs)
        for q, g in zip(qs, got):
            want = to_orc(orc, q).run(oi)
            assert g.status == 0 and g.total_found == want.total_found
            assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
            n_found += g.total_found
        assert n_found > 50000, n_found
        # more filters / values than the device path holds: declined
        assert batch.search(seg, [m.Query(kw(m, 0, 1), filters=[F(64, 32, values=list(range(9)))])])[0].status == -2
        assert batch.search(seg, [m.Query(kw(m, 0, 1), filters=[F(64, 32, min=1, max=2)] * 3)])[0].status == -2
    finally:
        seg.close()


# ------------------------------------------------------------------ cutoff (MatchExtended stops after N matches)
def test_cutoff(orc, dev):
    """CSphQuery::m_iCutoff (sphinx.cpp:12197-12199, 12261-12267): the first `cutoff` matches in rowid order that got past the filters
    and the dead-row map are all the sorter sees -- they are the result set (best first), and total_found counts them.  Every
    kind of path (single keyword, bitmap AND, trees, PHRASE, the generic evaluator), every ranker, with filters and dead rows;
    cutoffs below, at and above the number of matches and the top-K size."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("cutoff runs on the packed path")
    rng = np.random.default_rng(777)
    n_docs = 60000
    probs = [0.5, 0.4, 0.2, 0.05, 0.01, 0.002, 0.0003]
    W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=20, end_markers=True)
    nt = len(probs)
    hi = m.index_from_hits(W, R, H, n_terms=nt, total_docs=n_docs, n_fields=3)
    rows = np.zeros((n_docs, 3), np.uint32)
    rows[:, 0] = np.arange(n_docs)
    rows[:, 2] = rng.integers(0, 10, n_docs)
    F = m.Filter
    rankers = [m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_BM25, m.SPH_RANK_NONE, m.SPH_RANK_SPH04, m.SPH_RANK_WORDCOUNT]
    qs = []
    for i in range(160):
        a, b, c = (int(t) for t in rng.choice(nt, 3, replace=False))
        d, e = (int(t) for t in rng.choice(nt, 2))
        shape = i % 8
        root = [kw(m, a, 1), m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), OR(m, kw(m, a, 1), kw(m, b, 2)), m.XQNode.AND(OR(m, kw(m, a, 1), kw(m, b, 2)), kw(m, c, 3)),
                PHRASE(m, kw(m, a, 1), kw(m, b, 2)), ANDNOT(m, kw(m, a, 1), kw(m, b, 2)),
                m.XQNode.AND(kw(m, a, 1), kw(m, b, 2), kw(m, c, 3), kw(m, d, 4), kw(m, e, 5)),  # five streams: the generic evaluator
                m.XQNode(m.SPH_QUERY_NEAR, [kw(m, a, 1), m.XQNode(m.SPH_QUERY_PROXIMITY, [kw(m, b, 2), kw(m, c, 3)], opt=6)], opt=4)][shape]
        fl = [F(64, 32, values=sorted(set(int(v) for v in rng.integers(0, 10, 4))))] if rng.random() < 0.4 else None
        qs.append(m.Query(root, ranker=int(rng.choice(rankers)), max_matches=int(rng.choice([20, 1000])), filters=fl,
                          cutoff=int(rng.choice([1, 2, 7, 20, 21, 300, 1000, 1024]))))
    # dense x dense under BM25 / NONE: the shape the bitmap kernel takes when nothing bounds the rows
    qs.append(m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2)), ranker=m.SPH_RANK_BM25, max_matches=1000, cutoff=100))
    qs.append(m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2)), ranker=m.SPH_RANK_NONE, max_matches=10, cutoff=1000))
    seg = m.Segment(ctx, hi)
    seg.set_attrs(rows)
    oi = orc_index_of(orc, hi)
    oi.attrs = rows
    dead = np.zeros((n_docs + 31) // 32, np.uint32)
    for r in rng.choice(n_docs, 3000, replace=False):
        dead[r >> 5] |= np.uint32(1 << (int(r) & 31))
    try:
        for with_dead in (False, True):
            if with_dead:
                seg.set_dead_rows(dead)
                oi.dead_rows = dead
            n_cut = 0
            got = batch.search(seg, qs)
            for q, g in zip(qs, got):
                want = to_orc(orc, q).run(oi)
                assert g.status == 0, (g.status, q.cutoff)
                assert g.total_found == want.total_found, (g.total_found, want.total_found, q.cutoff)
                assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
                n_cut += int(g.total_found == q.cutoff)
            assert n_cut > 60, n_cut  # about half of them did stop at the cutoff
        # what stays on the host: a cutoff past the device's top-K size, a cutoff next to a weight filter
        assert batch.search(seg, [m.Query(kw(m, 0, 1), cutoff=1025)])[0].status == -2
        assert batch.search(seg, [m.Query(kw(m, 0, 1), cutoff=5, weight_filters=[F(0, 32, min=0, max=5000)])])[0].status == -2
    finally:
        seg.close()


# ------------------------------------------------------------------ malformed postings never reach a kernel
def test_corrupt_postings_are_rejected_at_load(dev):
    """Segment creation walks every doclist once (the load-time transcode): rowids beyond the row count, hitlist offsets
    past .spp, truncated or descending entries fail mrk_segment_create instead of becoming out-of-bounds device reads."""
    m, ctx, batch = dev
    if ctx_path(ctx) != 0:
        pytest.skip("the walk happens in the packed-format transcode")
    hi = m.synth_index(5000, [0.4, 0.1], seed=3, n_fields=2, max_pos=20)
    good = m.Segment(ctx, hi)
    good.close()

    def broken(total_docs=None, spd=None, spp_cut=None):
        h = m.HostIndex(hi.spd if spd is None else spd, hi.spp if spp_cut is None else hi.spp[:spp_cut], hi.spe, hi.dict.copy(),
                        hi.total_docs if total_docs is None else total_docs, hi.skiplist_block_size, hi.hit_format, hi.n_fields)
        with pytest.raises(m.MrkError):
            m.Segment(ctx, h)

    broken(total_docs=100)  # rowids >= the row count
    broken(spp_cut=16)      # hitlist offsets past the (cut) .spp
    spd = hi.spd.copy()
    off = int(hi.dict[0]["doclist_off"])
    spd[off + 40: off + 44] = 0x80  # a run of continuation bytes: deltas explode / entries no longer parse
    broken(spd=spd)
    d = hi.dict.copy()
    d[0]["docs"] += 5  # the dictionary promises more docs than the doclist holds
    with pytest.raises(m.MrkError):
        m.Segment(ctx, m.HostIndex(hi.spd, hi.spp, hi.spe, d, hi.total_docs, hi.skiplist_block_size, hi.hit_format, hi.n_fields))


def test_bad_postings_behind_a_packing_decline_are_rejected(dev):
    """The doclists the load-time transcode does not walk to the end (a field mask wider than 8 bits stops it; segments
    with more than 8 fields or ctx pack = 0 never start it) are walked by the validate-only pass: a rowid beyond the row
    count or a hitlist offset past .spp behind such an entry fails mrk_segment_create on every path -- nothing of it
    reaches scan_kernel's dead-row / attribute reads.  Same descriptors as tests/test_validate_cpu.py."""
    from test_validate_cpu import crafted
    m, ctx, batch = dev
    cases = crafted()
    for pack in (1, 0):
        ctx.set("pack", pack)
        try:
            for name, hi in cases.items():
                if name.startswith("good"):
                    m.Segment(ctx, hi).close()
                else:
                    with pytest.raises(m.MrkError):
                        m.Segment(ctx, hi)
        finally:
            ctx.set("pack", 1)

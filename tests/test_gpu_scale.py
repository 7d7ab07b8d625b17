"""GPU parity at bench scale (BASELINE.json configs 2 and 3 shapes): 10 M docs by default, MRK_SCALE_DOCS=100000000
for the full size.  The three device paths (bitmap AND kernel / packed block scan / VLB-direct) must agree with each
other bit for bit and with the oracle; results obey the sorter's order; a second run returns the same bytes."""
import os

import numpy as np
import pytest

from test_gpu_parity import ANDNOT, OR, kw, orc_index_of, to_orc

pytestmark = pytest.mark.gpu

N_DOCS = int(os.environ.get("MRK_SCALE_DOCS", 10_000_000))
SEED_OFF = int(os.environ.get("MRK_SCALE_SEED", 0))  # added to every query generator's seed: a soak draws other queries over the same corpora
# document probabilities: four common keywords, two in between, four selective ones
PROBS = [0.3, 0.12, 0.06, 0.031, 0.012, 0.004, 0.0011, 0.0004, 0.00013, 0.00005]


@pytest.fixture(scope="module")
def corpus():
    import manticoresearch_amd as m

    return m, m.synth_index(N_DOCS, PROBS, seed=20261004, n_fields=3, max_pos=64, end_markers=True)


def run_path(m, hi, qs, path, bitmap_inv):
    ctx = m.Context(0)
    ctx.set("path", path)
    ctx.set("bitmap_inv", bitmap_inv)
    batch = m.Batch(ctx, len(qs))
    seg = m.Segment(ctx, hi)
    try:
        r1 = batch.search(seg, qs)
        st = batch.stats()
        r2 = batch.search(seg, qs)
    finally:
        seg.close()
        batch.close()
        ctx.close()
    for a, b in zip(r1, r2):  # idempotence
        assert a.total_found == b.total_found and (a.rowid == b.rowid).all() and (a.weight == b.weight).all()
    return r1, st


def check_order(r):
    w, d = r.weight.astype(np.int64), r.rowid.astype(np.int64)
    assert ((w[:-1] > w[1:]) | ((w[:-1] == w[1:]) & (d[:-1] < d[1:]))).all()  # MatchRelevanceLt_fn
    assert r.total_found >= len(d) and len(np.unique(d)) == len(d)


def test_two_term_and_paths_agree(orc, corpus):
    m, hi = corpus
    rng = np.random.default_rng(5 + SEED_OFF)
    qs = []
    for a in range(len(PROBS)):
        for b in range(len(PROBS)):
            if a < b and rng.random() < 0.7:
                pair = (a, b) if rng.random() < 0.5 else (b, a)
                qs.append(m.Query(m.XQNode.AND(kw(m, pair[0], 1), kw(m, pair[1], 2)),
                                  ranker=m.SPH_RANK_BM25 if rng.random() < 0.8 else m.SPH_RANK_NONE, max_matches=1000,
                                  field_weights=[3, 1, 2] if rng.random() < 0.5 else None))
    bm, st_bm = run_path(m, hi, qs, 0, 64)
    pk, st_pk = run_path(m, hi, qs, 0, 0)
    vl, st_vl = run_path(m, hi, qs, 1, 0)
    assert st_bm["n_items_bm"] > 0 and st_pk["n_items_bm"] == 0 and st_pk["packed"] == 1 and st_vl["packed"] == 0
    oi = orc_index_of(orc, hi)
    for i, q in enumerate(qs):
        for other in (pk[i], vl[i]):
            assert bm[i].status == 0 and other.status == 0
            assert bm[i].total_found == other.total_found
            assert (bm[i].rowid == other.rowid).all() and (bm[i].weight == other.weight).all()
        check_order(bm[i])
        if i % 3 == 0 or N_DOCS <= 10_000_000:
            want = to_orc(orc, q).run(oi)
            assert bm[i].total_found == want.total_found
            assert (bm[i].rowid == want.rowid).all() and (bm[i].weight == want.weight).all()


def test_three_term_mixes_proximity_bm25(orc, corpus):
    """BASELINE config 3: a b c, (a|b) c, a (b|c), a b -c under SPH_RANK_PROXIMITY_BM25 (hitlist decode)."""
    m, hi = corpus
    rng = np.random.default_rng(6 + SEED_OFF)
    qs = []
    for _ in range(10):
        a, b, c = (int(x) for x in rng.choice(np.arange(2, len(PROBS)), 3, replace=False))
        ka, kb, kc = kw(m, a, 1), kw(m, b, 2), kw(m, c, 3)
        for root in (m.XQNode.AND(ka, kb, kc), m.XQNode.AND(OR(m, ka, kb), kc), m.XQNode.AND(ka, OR(m, kb, kc)),
                     ANDNOT(m, m.XQNode.AND(ka, kb), kc)):
            qs.append(m.Query(root, ranker=m.SPH_RANK_PROXIMITY_BM25, max_matches=1000))
    got, _ = run_path(m, hi, qs, 0, 64)
    plain, _ = run_path(m, hi, qs, 0, 0)
    oi = orc_index_of(orc, hi)
    n_ok = 0
    for q, g, p in zip(qs, got, plain):
        assert g.status == p.status
        if g.status != 0:
            continue
        assert g.total_found == p.total_found and (g.rowid == p.rowid).all() and (g.weight == p.weight).all()
        check_order(g)
        want = to_orc(orc, q).run(oi)
        assert g.total_found == want.total_found
        assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
        n_ok += 1
    assert n_ok >= 30


def test_proximity_pruning_under_ties(orc, corpus):
    """The weight bounds in front of the hit pass (prox_bounds; two-level lower-bound histogram, mrk_kprune.h) where they
    matter most: SPH_RANK_PROXIMITY has no BM25 part, so whole populations of matches share one weight and the sorter picks by
    rowid.  Pruned == unpruned bit for bit (common keywords, millions of matches), and == the oracle where the CPU gets through."""
    m, hi = corpus
    rng = np.random.default_rng(16 + SEED_OFF)
    qs, sizes = [], []
    for t in range(12):
        lo = 0 if t % 2 == 0 else 2  # even: the common keywords (prune on == off); odd: mid-range ones (oracle too)
        a, b, c = (int(x) for x in rng.choice(np.arange(lo, lo + 5), 3, replace=False))
        ka, kb, kc = kw(m, a, 1), kw(m, b, 2), kw(m, c, 3)
        root = (OR(m, ka, kb), m.XQNode.AND(OR(m, ka, kb), kc), m.XQNode.AND(ka, OR(m, kb, kc)), OR(m, ka, OR(m, kb, kc)))[t % 4]
        qs.append(m.Query(root, ranker=m.SPH_RANK_PROXIMITY if t % 3 else m.SPH_RANK_PROXIMITY_BM25, max_matches=(1000, 20, 300)[t % 3],
                          field_weights=(None, [3, 1, 2], [2, -1, 1], [1, 1, 1])[t % 4], index_weight=1 + (t % 5 == 0) * 2))
        sizes.append(lo)
    out = {}
    for prune in (1, 0):
        ctx = m.Context(0)
        ctx.set("prox_prune", prune)
        seg, batch = m.Segment(ctx, hi), m.Batch(ctx, len(qs))
        try:
            out[prune] = batch.search(seg, qs)
        finally:
            batch.close()
            seg.close()
            ctx.close()
    oi = orc_index_of(orc, hi)
    n_orc = 0
    for q, lo, g, p in zip(qs, sizes, out[1], out[0]):
        assert g.status == 0 and p.status == 0
        assert g.total_found == p.total_found and (g.rowid == p.rowid).all() and (g.weight == p.weight).all()
        check_order(g)
        if lo == 2 and g.total_found < 1_500_000:
            want = to_orc(orc, q).run(oi)
            assert g.total_found == want.total_found and (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
            n_orc += 1
    assert n_orc >= 3


@pytest.mark.parametrize("end_markers", [0, 2])
def test_proximity_bound_by_keywords_where_positions_own_the_end_flag(orc, end_markers):
    """ctx key prox_bound_keywords = 1: the tighter weight bound in front of the hit pass (a proximity run holds every keyword at most
    once).  It is sound where hits of different keywords at one position reach the ranker in query-position order -- a corpus WITHOUT
    field-end flags (0), or with the flag on the hit at the field's last POSITION whatever the word (2: what the reference's indexer
    writes).  Only 16 positions per field, so keywords share positions -- the last one included -- in most docs.  Device (pruned, tight
    bound) == device (unpruned) == oracle."""
    import manticoresearch_amd as m

    probs = [0.3, 0.12, 0.06, 0.031, 0.012, 0.004]
    hi = m.synth_index(3_000_000, probs, seed=20261005, n_fields=2, max_pos=16, end_markers=end_markers)
    rng = np.random.default_rng(26 + SEED_OFF)
    qs = []
    for t in range(16):
        a, b, c = (int(x) for x in rng.choice(np.arange(0 if t % 4 == 0 else 1, 5), 3, replace=False))
        ka, kb, kc = kw(m, a, 1), kw(m, b, 2), kw(m, c, 3)
        root = (OR(m, ka, kb), m.XQNode.AND(OR(m, ka, kb), kc), m.XQNode.AND(ka, OR(m, kb, kc)), OR(m, ka, OR(m, kb, kc)))[t % 4]
        qs.append(m.Query(root, ranker=m.SPH_RANK_PROXIMITY_BM25 if t % 3 else m.SPH_RANK_PROXIMITY, max_matches=(1000, 50, 300)[t % 3],
                          field_weights=(None, [3, 1], [2, -1], [1, 1])[t % 4]))
    out = {}
    for prune in (1, 0):
        ctx = m.Context(0)
        ctx.set("prox_prune", prune)
        ctx.set("prox_bound_keywords", 1)
        seg, batch = m.Segment(ctx, hi), m.Batch(ctx, len(qs))
        try:
            out[prune] = batch.search(seg, qs)
        finally:
            batch.close()
            seg.close()
            ctx.close()
    oi = orc_index_of(orc, hi)
    n_orc = 0
    for q, g, p in zip(qs, out[1], out[0]):
        assert g.status == 0 and p.status == 0
        assert g.total_found == p.total_found and (g.rowid == p.rowid).all() and (g.weight == p.weight).all()
        check_order(g)
        if g.total_found < 600_000:
            want = to_orc(orc, q).run(oi)
            assert g.total_found == want.total_found and (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
            n_orc += 1
    assert n_orc >= 6


def test_phrase_mix_with_field_weights(orc, corpus):
    """BASELINE config 5 shapes: PHRASE (alone, with another keyword, in an OR) + field weights, default ranker."""
    from test_gpu_parity import PHRASE
    m, hi = corpus
    rng = np.random.default_rng(7 + SEED_OFF)
    qs = []
    for _ in range(12):
        a, b, c, d = (int(x) for x in rng.choice(np.arange(0, 8), 4, replace=False))
        ph = PHRASE(m, kw(m, a, 1), kw(m, b, 2))
        fw = [int(x) for x in rng.integers(1, 12, 3)]
        for root in (ph, m.XQNode.AND(ph, kw(m, c, 3)), OR(m, ph, kw(m, d, 3))):
            qs.append(m.Query(root, ranker=m.SPH_RANK_PROXIMITY_BM25, max_matches=1000, field_weights=fw))
    got, _ = run_path(m, hi, qs, 0, 64)
    oi = orc_index_of(orc, hi)
    n_found = 0
    for q, g in zip(qs, got):
        assert g.status == 0
        check_order(g)
        want = to_orc(orc, q).run(oi)
        assert g.total_found == want.total_found
        assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
        n_found += g.total_found
    assert n_found > 1000  # the corpus is dense enough (positions 1..64) for phrases to occur


def test_generic_evaluator_shapes_at_scale(orc, corpus):
    """The generic per-doc evaluator at bench scale (BASELINE config 5: "PHRASE op + field weights" and what lies beyond the specialised
    passes): five common keywords under the hit rankers, a 5-word proximity, NEAR over phrases, BEFORE over groups, NOTNEAR, SENTENCE with
    a keyword standing in for the boundary word, NEAR over four operands at the root (probe launch) -- each against the oracle."""
    m, hi = corpus
    rng = np.random.default_rng(8 + SEED_OFF)
    X = m.XQNode
    qs = []
    for _ in range(4):
        a, b, c, d, e = (int(x) for x in rng.choice(np.arange(0, 7), 5, replace=False))
        s = int(rng.choice(np.arange(6, 10)))
        fw = [int(x) for x in rng.integers(1, 9, 3)]
        roots = [X.AND(kw(m, a, 1), kw(m, b, 2), kw(m, c, 3), kw(m, d, 4), kw(m, e, 5)),
                 X(m.SPH_QUERY_PROXIMITY, [kw(m, a, 1), kw(m, b, 2), kw(m, c, 3), kw(m, d, 4), kw(m, e, 5)], opt=12),
                 X(m.SPH_QUERY_NEAR, [X(m.SPH_QUERY_PROXIMITY, [kw(m, a, 1), kw(m, b, 2)], opt=4), X(m.SPH_QUERY_PHRASE, [kw(m, c, 3), kw(m, d, 4)])], opt=9),
                 X(m.SPH_QUERY_BEFORE, [OR(m, kw(m, a, 1), kw(m, s, 2)), kw(m, b, 3), kw(m, c, 4)]),
                 X(m.SPH_QUERY_NOTNEAR, [kw(m, a, 1), OR(m, kw(m, b, 2), kw(m, c, 3))], opt=2),
                 X(m.SPH_QUERY_SENTENCE, [kw(m, a, 1), kw(m, b, 2)], unit_term=c),
                 X(m.SPH_QUERY_NEAR, [kw(m, a, 1), kw(m, b, 2), kw(m, c, 3), kw(m, d, 4)], opt=6)]
        for i, root in enumerate(roots):
            qs.append(m.Query(root, ranker=[m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_SPH04, m.SPH_RANK_BM25][i % 3], max_matches=1000, field_weights=fw if i % 2 else None))
    got, _ = run_path(m, hi, qs, 0, 64)
    oi = orc_index_of(orc, hi)
    n_found = 0
    for q, g in zip(qs, got):
        assert g.status == 0, q.root
        check_order(g)
        want = to_orc(orc, q).run(oi)
        assert g.total_found == want.total_found, (g.total_found, want.total_found, q.root)
        assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
        n_found += g.total_found
    assert n_found > 100000


# ------------------------------------------------------------------ BASELINE.json's full size: 100 M docs
FULL_DOCS = int(os.environ.get("MRK_FULL_DOCS", 100_000_000))  # 0 skips the test


@pytest.mark.skipif(FULL_DOCS == 0, reason="MRK_FULL_DOCS=0")
def test_full_size_properties(orc):
    """The headline configuration (100 M docs, 2-term AND, BM25, top-1000) through size-independent properties: the
    three device paths agree bit for bit on one segment, results obey the sorter's order, a second run returns the
    same bytes, totals are consistent (|A and B| <= min(|A|, |B|), AND is symmetric in its result set), and a few
    selective queries -- cheap enough for the CPU -- equal the oracle."""
    import manticoresearch_amd as m

    probs = [0.25, 0.08, 0.03, 0.002, 0.0005, 0.0001]
    hi = m.synth_index(FULL_DOCS, probs, seed=20261005)
    docs = [int(hi.dict[t]["docs"]) for t in range(len(probs))]
    ctx = m.Context(0)
    seg = m.Segment(ctx, hi)
    pairs = [(a, b) for a in range(len(probs)) for b in range(len(probs)) if a != b]
    qs = [m.Query(m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), ranker=m.SPH_RANK_BM25, max_matches=1000) for a, b in pairs]
    batch = m.Batch(ctx, len(qs))
    try:
        res = {}
        for name, path, inv in (("bm", 0, 64), ("pk", 0, 0), ("vlb", 1, 0)):
            ctx.set("path", path)
            ctx.set("bitmap_inv", inv)  # read at submit: the same segment serves all three paths
            r1 = batch.search(seg, qs)
            st = batch.stats()
            r2 = batch.search(seg, qs)
            for a_, b_ in zip(r1, r2):
                assert a_.status == 0 and a_.total_found == b_.total_found and (a_.rowid == b_.rowid).all() and (a_.weight == b_.weight).all()
            res[name] = (r1, st)
        assert res["bm"][1]["n_items_bm"] > 0 and res["pk"][1]["n_items_bm"] == 0 and res["vlb"][1]["packed"] == 0
        by_pair = {}
        for i, (a, b) in enumerate(pairs):
            g = res["bm"][0][i]
            for other in (res["pk"][0][i], res["vlb"][0][i]):
                assert g.total_found == other.total_found and (g.rowid == other.rowid).all() and (g.weight == other.weight).all()
            check_order(g)
            assert g.total_found <= min(docs[a], docs[b])
            assert len(g.rowid) == min(1000, g.total_found)
            by_pair[(a, b)] = g
        for (a, b), g in by_pair.items():  # AND(a, b) and AND(b, a) find the same docs (weights may differ in tfidf sum order)
            h = by_pair[(b, a)]
            assert g.total_found == h.total_found
        oi = orc_index_of(orc, hi)
        for i, (a, b) in enumerate(pairs):
            if min(docs[a], docs[b]) < 100_000 and max(docs[a], docs[b]) < 3_000_000:
                want = to_orc(orc, qs[i]).run(oi)
                g = res["bm"][0][i]
                assert g.total_found == want.total_found and (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
        # ---- BASELINE config 3 at full size: 3-term mixes a b c | (a|b) c | a (b|c) | a b -c under SPH_RANK_PROXIMITY_BM25.
        # The window-driven tree kernel (scan_bt_kernel + rank_kernel) and the block path (scan_pk_kernel + rank_kernel) must
        # agree bit for bit; totals obey inclusion-exclusion; results obey the sorter's order; a rerun returns the same bytes;
        # the selective shapes equal the oracle.
        ctx.set("path", 0)
        ctx.set("bitmap_inv", 64)
        AND_, OR_, ANDNOT_ = m.XQNode.AND, (lambda *k: m.XQNode(m.SPH_QUERY_OR, list(k))), (lambda *k: m.XQNode(m.SPH_QUERY_ANDNOT, list(k)))
        triples = [(0, 1, 2), (1, 0, 2), (3, 0, 1), (2, 1, 0), (4, 1, 2), (0, 2, 3)]
        shapes = []
        for a, b, c in triples:
            ka, kb, kc = kw(m, a, 1), kw(m, b, 2), kw(m, c, 3)
            shapes += [AND_(ka, kb, kc), AND_(OR_(ka, kb), kc), AND_(ka, OR_(kb, kc)), ANDNOT_(AND_(ka, kb), kc)]
        q3 = [m.Query(r, ranker=m.SPH_RANK_PROXIMITY_BM25, max_matches=1000) for r in shapes]
        qn = [m.Query(r, ranker=m.SPH_RANK_NONE, max_matches=10) for r in
              [x for a, b, c in triples for x in (AND_(kw(m, a, 1), kw(m, c, 2)), AND_(kw(m, b, 1), kw(m, c, 2)), AND_(kw(m, a, 1), kw(m, b, 2)))]]
        b3 = m.Batch(ctx, len(q3))
        try:
            out = {}
            for name, inv in (("bt", 1024), ("blocks", 0)):
                ctx.set("bt_cover_inv", inv)
                r1 = b3.search(seg, q3)
                r2 = b3.search(seg, q3)
                for x, y in zip(r1, r2):
                    assert x.status == 0 and x.total_found == y.total_found and (x.rowid == y.rowid).all() and (x.weight == y.weight).all()
                out[name] = r1
            for x, y in zip(out["bt"], out["blocks"]):
                assert x.total_found == y.total_found and (x.rowid == y.rowid).all() and (x.weight == y.weight).all()
                check_order(x)
            pair_tot = [r.total_found for r in b3.search(seg, qn)]
            for t, (a, b, c) in enumerate(triples):
                abc, aorb_c, a_borc, ab_notc = (out["bt"][4 * t + i].total_found for i in range(4))
                ac, bc, ab = pair_tot[3 * t: 3 * t + 3]
                assert aorb_c == ac + bc - abc      # |(A u B) n C| = |A n C| + |B n C| - |A n B n C|
                assert a_borc == ab + ac - abc      # |A n (B u C)|
                assert ab_notc == ab - abc          # |A n B \ C|
                assert abc <= min(ab, ac, bc)
            for i, q in enumerate(q3):  # the shapes driven by a selective keyword are cheap enough for the CPU
                a = triples[i // 4][0]
                if docs[a] < 300_000 and i % 4 != 1:
                    want = to_orc(orc, q).run(oi)
                    g = out["bt"][i]
                    assert g.total_found == want.total_found and (g.rowid == want.rowid).all() and (g.weight == want.weight).all()
        finally:
            ctx.set("bt_cover_inv", 1024)
            b3.close()
    finally:
        ctx.set("path", 0)
        ctx.set("bitmap_inv", 64)
        batch.close()
        seg.close()
        ctx.close()


# ------------------------------------------------------------------ BASELINE config 5's mix in ONE 1024-query launch
def test_config5_mix_1024_queries_per_launch(orc):
    """BASELINE config 5 at parity-test size: 10 M docs, FOUR fields with weights (10, 5, 2, 1), the Zipf query mix -- 60 % 2-term AND,
    20 % 3-term AND / OR mixes, 20 % 2-3-word PHRASE -- under SPH_RANK_PROXIMITY_BM25, all 1024 queries in one launch (the shape
    bench.py's config5 leg times at 125 M docs).  Every query answers (none declined, none truncated), the launch with the pruning in
    front of the hit pass (prox_prune, default) equals the launch without it bit for bit -- a match the bounds kept out of the hit
    pass is still counted --, results obey the sorter's order, and every fourth query equals the oracle."""
    import manticoresearch_amd as m

    import bench

    n_docs = int(os.environ.get("MRK_C5_DOCS", 10_000_000))
    c = bench.zipf_c()
    ranks, strata = bench.make_queries(c, 128)
    probs = [min(0.5, c / r) for r in ranks]
    hi = m.synth_index(n_docs, probs, seed=bench.CORPUS_SEED + 5, n_fields=4, end_markers=True)
    gd = hi.dict["docs"].astype(np.int64)
    qs = bench.config5_queries(m, strata, 1024, 1000, n_docs, gd, (10, 5, 2, 1))
    assert len(qs) == 1024
    kinds = {"phrase": sum(q.root.op == m.SPH_QUERY_PHRASE for q in qs)}
    assert 150 < kinds["phrase"] < 260
    res = {}
    for prune in (1, 0):
        ctx = m.Context(0)
        ctx.set("prox_prune", prune)
        seg = m.Segment(ctx, hi)
        batch = m.Batch(ctx, len(qs))
        try:
            res[prune] = batch.search(seg, qs)
        finally:
            batch.close()
            seg.close()
            ctx.close()
    oi = orc_index_of(orc, hi)
    n_matches = 0
    for i, (q, g, h) in enumerate(zip(qs, res[1], res[0])):
        assert g.status == 0 and h.status == 0, (i, q.root)
        assert g.total_found == h.total_found and (g.rowid == h.rowid).all() and (g.weight == h.weight).all(), i
        check_order(g)
        n_matches += g.total_found
        if i % 4 == 0:
            want = to_orc(orc, q).run(oi)
            assert g.total_found == want.total_found, (i, g.total_found, want.total_found)
            assert (g.rowid == want.rowid).all() and (g.weight == want.weight).all(), i
    assert n_matches > 1_000_000

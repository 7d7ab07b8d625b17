"""GPU: object lifetimes behind the C-ABI.  Round 2 recorded a hang (gpurun_out/hang_lib.txt): a Batch finaliser posted its destroy
to the submission thread of a context that was already gone.  The library now counts a context's live segments / batches and
refuses to destroy it (MRK_E_INVAL, nothing destroyed); Context.close() takes its children down first."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ctx_destroy_refuses_with_children_alive_and_close_orders_them():
    import manticoresearch_amd as m
    from manticoresearch_amd import _lib

    hi = m.synth_index(50_000, [0.3, 0.1], seed=5)
    ctx = m.Context(0)
    seg = m.Segment(ctx, hi)
    batch = m.Batch(ctx, 4)
    q = m.Query(m.XQNode.AND(m.XQNode.keyword(0, 1), m.XQNode.keyword(1, 2)), ranker=m.SPH_RANK_BM25)
    want = batch.search(seg, [q])[0]
    # the C-ABI itself: children alive -> an error code, and the context still works afterwards
    assert _lib.lib().mrk_ctx_destroy(ctx._h) == _lib.MRK_E_INVAL
    assert b"still alive" in _lib.lib().mrk_last_error()
    again = batch.search(seg, [q])[0]
    assert again.total_found == want.total_found and np.array_equal(again.rowid, want.rowid)
    # Context.close() with a live Segment and Batch: children first, then the context; late close() calls are no-ops
    ctx.close()
    assert not ctx._h and not seg._h and not batch._h
    batch.close()
    seg.close()


def test_batcher_from_python_threads():
    import manticoresearch_amd as m

    hi = m.synth_index(300_000, [0.3, 0.2, 0.1, 0.05], seed=9)
    ctx = m.Context(0)
    seg = m.Segment(ctx, hi)
    kw = m.XQNode.keyword
    qs = [m.Query(m.XQNode.AND(kw(a, 1), kw(b, 2)), ranker=m.SPH_RANK_BM25, max_matches=200) for a in range(4) for b in range(4) if a != b]
    b = m.Batch(ctx, len(qs))
    want = b.search(seg, qs)
    bt = m.Batcher(ctx, max_batch=64)
    got = [None] * len(qs)

    def work(t):
        for i in range(t, len(qs), 4):
            got[i] = bt.search(seg, qs[i])

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for g, w in zip(got, want):
        assert g.status == 0 and g.total_found == w.total_found and np.array_equal(g.rowid, w.rowid) and np.array_equal(g.weight, w.weight)
    st = bt.stats()
    assert st["queries"] == len(qs) and st["launches"] <= len(qs)
    ctx.close()


def test_more_than_32_fields_is_declined_at_load():
    """The device path covers <= 32 fields (the doclist entry's field mask is one dword; beyond it the mask comes from the hits:
    ISphQword::CollectHitMask, restated in the oracle and pinned by test/test_183's rows -- tests/test_oracle_wide_fields.py).
    A 40-field index is refused when the segment is created, loudly, and the context stays usable."""
    import manticoresearch_amd as m
    from manticoresearch_amd import _lib

    W = np.array([1, 1, 2], np.uint64)
    R = np.array([0, 1, 1], np.uint32)
    H = np.array([(36 << 24) | 1, (3 << 24) | 1, (36 << 24) | 2], np.uint32)
    hi = m.index_from_hits(W, R, H, n_terms=2, total_docs=2, n_fields=40)
    ctx = m.Context(0)
    try:
        with pytest.raises(_lib.MrkError) as e:
            m.Segment(ctx, hi)
        assert e.value.code == _lib.MRK_E_UNSUPPORTED and "fields" in str(e.value)
        ok = m.Segment(ctx, m.synth_index(1000, [0.5], seed=3))
        ok.close()
    finally:
        ctx.close()

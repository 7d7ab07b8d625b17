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

import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import manticoresearch_amd as m
from oracle import oracle as orc
from helpers import synth_postings
from test_gpu_parity import to_orc, orc_index_of, kw
rng = np.random.default_rng(4321 + 128 + 1)
n_docs = 40000
probs = [0.5, 0.3, 0.12, 0.05, 0.02, 0.004, 0.9]
W, R, H = synth_postings(rng, n_docs, probs, n_fields=3, max_pos=40, end_markers=True)
hi = m.index_from_hits(W, R, H, n_terms=len(probs), total_docs=n_docs, skiplist_block_size=128, hit_format=1, n_fields=3)
qs = []
for _ in range(120):
    k = int(rng.integers(2, 5))
    ts = rng.choice(len(probs), size=k, replace=False)
    masks = [0xFFFFFFFF if rng.random() < 0.7 else int(rng.integers(1, 8)) for _ in ts]
    pos, ap = [], 0
    for _ in ts:
        ap += 1 if rng.random() < 0.85 else 2
        pos.append(ap)
    root = m.XQNode.AND(*[kw(m, int(t), p, mk) for t, p, mk in zip(ts, pos, masks)])
    qs.append((m.Query(root, ranker=int(rng.choice([m.SPH_RANK_PROXIMITY_BM25, m.SPH_RANK_PROXIMITY])),
                      max_matches=int(rng.choice([5, 100, 1000])),
                      field_weights=[int(x) for x in rng.integers(-3, 12, 3)] if rng.random() < 0.5 else None,
                      index_weight=int(rng.choice([1, 1, 2]))), list(zip(ts, pos, masks))))
ctx = m.Context(0); seg = m.Segment(ctx, hi); batch = m.Batch(ctx, 256)
oi = orc_index_of(orc, hi)
got = batch.search(seg, [q for q, _ in qs])
bad = 0
for (q, desc), g in zip(qs, got):
    q2 = m.Query(q.root, ranker=q.ranker, max_matches=1000, field_weights=q.field_weights, index_weight=q.index_weight)
    w = to_orc(orc, q2).run(oi)
    g2 = batch.search(seg, [q2])[0]
    wd = dict(zip(w.rowid.tolist(), w.weight.tolist())); gd = dict(zip(g2.rowid.tolist(), g2.weight.tolist()))
    diff = [(r, gd.get(r), wd.get(r)) for r in set(wd) | set(gd) if gd.get(r) != wd.get(r)]
    if diff:
        bad += 1
        print("MISMATCH", desc, "ranker", q.ranker, "fw", q.field_weights, "iw", q.index_weight, "total", g2.total_found, w.total_found, "ndiff", len(diff), diff[:5])
        if bad >= 4: break
print("bad", bad)

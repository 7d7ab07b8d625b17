#!/usr/bin/env python3
"""tools/c5_prune_check.py [--docs D] -- the config-5 mix on bench.py's config-5 corpus (field-end flags owned by positions) with the
tighter weight bound (prox_bound_keywords=1), with the bound by hits, and without pruning: the three runs must return the very same
(rowid, weight) lists and total_found."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import manticoresearch_amd as m  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=30_000_000)
ap.add_argument("--queries", type=int, default=1024)
args = ap.parse_args()
c = bench.zipf_c()
ranks, strata = bench.make_queries(c, 1024)
probs = [min(0.5, c / r) for r in ranks]
hi = m.synth_index(args.docs, probs, seed=bench.CORPUS_SEED + 5, n_fields=4, end_markers=2)
gd = hi.dict["docs"].astype(np.int64)
out = []
for prune, tight in ((1, 1), (1, 0), (0, 0)):
    ctx = m.Context(0)
    ctx.set("prox_prune", prune)
    ctx.set("prox_bound_keywords", tight)
    seg = m.Segment(ctx, hi)
    qs = bench.config5_queries(m, strata, args.queries, 1000, args.docs, gd, (10, 5, 2, 1))
    b = m.Batch(ctx, len(qs))
    b.submit(seg, qs)
    b.wait()
    res = b.results()
    print("prox_prune", prune, "prox_bound_keywords", tight, "scan_ms", round(b.stats()["scan_ms"], 3), flush=True)
    out.append([(r.status, r.total_found, r.rowid.tolist(), r.weight.tolist()) for r in res])
    del b, seg, ctx
bad = [i for i, (x, y, z) in enumerate(zip(*out)) if not (x == y == z)]
print("queries", len(out[0]), "differ", bad[:10])
sys.exit(1 if bad else 0)

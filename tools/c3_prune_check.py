#!/usr/bin/env python3
"""tools/c3_prune_check.py [--docs D] -- the config-3 queries with and without the weight-bound pruning in front of the hit pass
(ctx tunable prox_prune): the two runs must return the very same (rowid, weight) lists and total_found."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import manticoresearch_amd as m  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=64)
ap.add_argument("--shape", type=int, default=-1)
args = ap.parse_args()
c = bench.zipf_c()
ranks, strata = bench.make_queries(c, args.queries)
probs = [min(0.5, c / r) for r in ranks]
hi = m.synth_index(args.docs, probs, seed=bench.CORPUS_SEED)
gd = hi.dict["docs"].astype(np.int64)
out = []
for prune in (1, 0):
    ctx = m.Context(0)
    ctx.set("prox_bound_keywords", 1)  # the bench corpus carries no field-end flags: see bench.py
    ctx.set("prox_prune", prune)
    seg = m.Segment(ctx, hi)
    qs = bench.config3_queries(m, strata, args.queries, 1000, args.docs, gd)
    if args.shape >= 0:
        qs = [q for i, q in enumerate(qs) if i % 4 == args.shape]
    b = m.Batch(ctx, len(qs))
    b.submit(seg, qs)
    b.wait()
    res = b.results()
    st = b.stats()
    print("prox_prune", prune, "scan_ms", round(st["scan_ms"], 3), flush=True)
    del b, seg, ctx
    out.append([(r.status, r.total_found, r.rowid.tolist(), r.weight.tolist()) for r in res])
bad = [i for i, (x, y) in enumerate(zip(*out)) if x != y]
print("queries", len(out[0]), "differ", bad[:10])
for i in bad[:3]:
    x, y = out[0][i], out[1][i]
    print(i, "shape", i % 4, "total", x[1], y[1], "n", len(x[2]), len(y[2]), "first diff", next((j for j in range(min(len(x[2]), len(y[2]))) if x[2][j] != y[2][j] or x[3][j] != y[3][j]), None))
    print("  pruned  ", list(zip(x[2], x[3]))[:5], list(zip(x[2], x[3]))[-3:])
    print("  unpruned", list(zip(y[2], y[3]))[:5], list(zip(y[2], y[3]))[-3:])
sys.exit(1 if bad else 0)

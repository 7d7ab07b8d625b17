#!/usr/bin/env python3
"""tools/gen_time.py [--docs D] [--queries N] -- what the generic per-doc evaluator (csrc/mrk_keval.h) costs: the bench corpus,
query shapes only that path takes, next to the nearest shape a specialised path takes.  Scan time per launch from the library's
HIP events; candidates = docs that reached the evaluator (upper bound: the driver keywords' docs)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import manticoresearch_amd as m  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=64)
ap.add_argument("--reps", type=int, default=4)
args = ap.parse_args()
c = bench.zipf_c()
ranks, strata = bench.make_queries(c, 256)
probs = [min(0.5, c / r) for r in ranks]
hi = m.synth_index(args.docs, probs, seed=bench.CORPUS_SEED)
ctx = m.Context(0)
seg = m.Segment(ctx, hi)
kw = m.XQNode.keyword
N = args.queries
common = [strata["cc"][i] for i in range(256)]
sel = [strata["sc"][i][0] for i in range(256)]


def shapes(i):
    a, b = common[i]
    c_, d = common[(i + 1) % 256]
    e = common[(i + 2) % 256][0]
    s = sel[i]
    PH, PX, NEAR, BEF, NN, OR_ = m.SPH_QUERY_PHRASE, m.SPH_QUERY_PROXIMITY, m.SPH_QUERY_NEAR, m.SPH_QUERY_BEFORE, m.SPH_QUERY_NOTNEAR, m.SPH_QUERY_OR
    return {
        "and4 (specialised)": m.XQNode.AND(kw(s, 1), kw(a, 2), kw(b, 3), kw(c_, 4)),
        "and5": m.XQNode.AND(kw(s, 1), kw(a, 2), kw(b, 3), kw(c_, 4), kw(d, 5)),
        "and5 common": m.XQNode.AND(kw(a, 1), kw(b, 2), kw(c_, 3), kw(d, 4), kw(e, 5)),
        "proximity5 ~20": m.XQNode(PX, [kw(a, 1), kw(b, 2), kw(c_, 3), kw(d, 4), kw(e, 5)], opt=20),
        '"a b" NEAR/10 "c d"': m.XQNode(NEAR, [m.XQNode(PX, [kw(a, 1), kw(b, 2)], opt=5), m.XQNode(PX, [kw(c_, 3), kw(d, 4)], opt=5)], opt=10),
        "(a|b) << c << d": m.XQNode(BEF, [m.XQNode(OR_, [kw(a, 1), kw(b, 2)]), kw(c_, 3), kw(d, 4)]),
        'a NOTNEAR/3 (b | c)': m.XQNode(NN, [kw(a, 1), m.XQNode(OR_, [kw(b, 2), kw(c_, 3)])], opt=3),
    }


out = {}
batch = m.Batch(ctx, N)
for name in shapes(0):
    qs = [m.Query(shapes(i)[name], ranker=m.SPH_RANK_PROXIMITY_BM25, max_matches=1000) for i in range(N)]
    cq = m.prepare(qs)
    batch.submit_prepared(seg, cq, N)
    batch.wait()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        batch.submit_prepared(seg, cq, N)
        batch.wait()
    dt = (time.perf_counter() - t0) / args.reps
    st, res = batch.stats(), batch.results()
    out[name] = {"ms_per_launch": round(dt * 1e3, 2), "scan_ms": round(st["scan_ms"], 2), "ok": int(sum(r.status == 0 for r in res)),
                 "matches": int(sum(r.total_found for r in res)), "ref_MB": round(st["algo_bytes"] / 1e6, 1)}
    print(name, json.dumps(out[name]), flush=True)
batch.close()
seg.close()
ctx.close()
print(json.dumps({"docs": args.docs, "queries_per_launch": N, "shapes": out}))

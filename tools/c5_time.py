#!/usr/bin/env python3
"""tools/c5_time.py [--docs D] [--kind all|and2|mix3|phrase] [--queries N] [--set k=v] -- BASELINE config 5's launch on its own (bench.py's
config5_leg corpus and queries), optionally one query class only: per-launch times from the library's HIP events."""
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
ap.add_argument("--docs", type=int, default=125_000_000)
ap.add_argument("--kind", default="all")
ap.add_argument("--queries", type=int, default=1024)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--set", action="append", default=[])
args = ap.parse_args()
c = bench.zipf_c()
ranks, strata = bench.make_queries(c, 342)
probs = [min(0.5, c / r) for r in ranks]
t0 = time.time()
hi = m.synth_index(args.docs, probs, seed=bench.CORPUS_SEED + 5, n_fields=4, end_markers=2)  # as bench.py: field-end flags owned by positions
ctx = m.Context(0)
ctx.set("prox_bound_keywords", 1)  # (sound on this corpus: see bench.py)
for kv in args.set:
    ctx.set(kv.split("=")[0], int(kv.split("=")[1]))
seg = m.Segment(ctx, hi)
gd = hi.dict["docs"].astype(np.int64)
qs = bench.config5_queries(m, strata, 1024, 1000, args.docs, gd, (10, 5, 2, 1))


def kind_of(q):
    if q.root.op == m.SPH_QUERY_PHRASE:
        return "phrase"
    n = sum(1 for _ in [1])
    kids = q.root.children or []
    return "and2" if (q.root.op == m.SPH_QUERY_AND and len(kids) == 2 and all(k.word is not None for k in kids)) else "mix3"


if args.kind != "all":
    qs = [q for q in qs if kind_of(q) == args.kind]
qs = qs[: args.queries]
cq = m.prepare(qs)
b = m.Batch(ctx, len(qs))
for kinds in [None]:
    ts = []
    for _ in range(args.reps):
        t1 = time.perf_counter()
        b.submit_prepared(seg, cq, len(qs))
        b.wait()
        ts.append((time.perf_counter() - t1) * 1e3)
    st = b.stats()
    res = b.results()
    print(json.dumps({"kind": args.kind, "queries": len(qs), "wall_ms": [round(x, 2) for x in ts], "scan_ms": round(st["scan_ms"], 3), "select_ms": round(st["merge_ms"], 3),
                      "items": st["n_items"], "cands": st["n_cands"], "ok": int(sum(r.status == 0 for r in res)), "matches": int(sum(r.total_found for r in res)),
                      "setup_s": round(t1 - t0, 1)}), flush=True)

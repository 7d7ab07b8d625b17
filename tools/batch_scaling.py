#!/usr/bin/env python3
"""tools/batch_scaling.py [--docs D] -- scan / selection time of ONE launch as a function of the number of queries in it (dense x dense
2-keyword ANDs, BM25, top-1000): what a small launch of the batching front costs."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manticoresearch_amd as m  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=60_000_000)
ap.add_argument("--sizes", default="1,2,4,8,16,32,64,256")
ap.add_argument("--set", action="append", default=[])
args = ap.parse_args()
NT = 12
probs = [0.30 / (1.0 + 0.35 * i) for i in range(NT)]
hi = m.synth_index(args.docs, probs, seed=21)
ctx = m.Context(0)
for kv in args.set:
    ctx.set(kv.split("=")[0], int(kv.split("=")[1]))
seg = m.Segment(ctx, hi)
kw = m.XQNode.keyword
qs = []
for i in range(256):
    a = i % NT
    b = (a + 1 + (i // NT) % (NT - 1)) % NT
    qs.append(m.Query(m.XQNode.AND(kw(a, 1), kw(b, 2)), ranker=m.SPH_RANK_BM25, max_matches=1000))
for n in [int(x) for x in args.sizes.split(",")]:
    b = m.Batch(ctx, n)
    cq = m.prepare(qs[:n])
    lat, sc, se = [], [], []
    for _ in range(8):
        t = time.perf_counter()
        b.submit_prepared(seg, cq, n)
        b.wait()
        lat.append((time.perf_counter() - t) * 1e3)
        st = b.stats()
        sc.append(st["scan_ms"]), se.append(st["merge_ms"])
    print(json.dumps({"n": n, "wall_ms": round(min(lat[2:]), 4), "scan_ms": round(min(sc[2:]), 4), "select_ms": round(min(se[2:]), 4), "items": st["n_items"], "cands": st["n_cands"]}), flush=True)
    b.close()

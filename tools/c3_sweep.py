#!/usr/bin/env python3
"""tools/c3_sweep.py [--docs D] --cfg "k=v,k=v" [--cfg ...] -- the config-3 launch and the sc / cc launches of bench.py under several
context settings in ONE process (the corpus is generated once): scan / selection times per launch from the library's HIP events."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import manticoresearch_amd as m  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=100_000_000)
ap.add_argument("--queries", type=int, default=256)
ap.add_argument("--cfg", action="append", default=[], help='"key=value,key=value" (empty string = defaults)')
ap.add_argument("--what", default="c3,sc,cc")
args = ap.parse_args()
c = bench.zipf_c()
ranks, strata = bench.make_queries(c, args.queries)
probs = [min(0.5, c / r) for r in ranks]
hi = m.synth_index(args.docs, probs, seed=bench.CORPUS_SEED)
gd = hi.dict["docs"].astype(np.int64)
kw = m.XQNode.keyword
sets = {"c3": bench.config3_queries(m, strata, args.queries, 1000, args.docs, gd)}
for s in ("sc", "cc", "ss"):
    sets[s] = [m.Query(m.XQNode.AND(kw(a, 1), kw(b, 2)), ranker=m.SPH_RANK_BM25, max_matches=1000) for a, b in strata[s]]
for cfg in args.cfg or [""]:
    ctx = m.Context(0)
    for kv in [x for x in cfg.split(",") if x]:
        k, v = kv.split("=")
        ctx.set(k, int(v))
    seg = m.Segment(ctx, hi)
    out = {}
    for what in args.what.split(","):
        qs = sets[what]
        cq = m.prepare(qs)
        b = m.Batch(ctx, len(qs))
        ts, ms = [], []
        for _ in range(5):
            b.submit_prepared(seg, cq, len(qs))
            b.wait()
            ts.append(b.stats()["scan_ms"])
            ms.append(b.stats()["merge_ms"])
        out[what] = {"scan_ms": round(min(ts[1:]), 4), "sel_ms": round(min(ms[1:]), 4), "items": b.stats()["n_items"], "cands": b.stats()["n_cands"]}
        b.close()
    print(json.dumps({"cfg": cfg, **out}), flush=True)
    seg.close()
    ctx.close()

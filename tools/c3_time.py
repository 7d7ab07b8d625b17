#!/usr/bin/env python3
"""tools/c3_time.py [--docs D] [--reps R] [--shape S] -- the config-3 launch of bench.py on its own (same corpus, same queries):
scan / selection times per launch from the library's HIP events.  MRK_LIB_PATH picks a kernel-experiment library.
--shape 0..3 keeps only one of the four query shapes (a b c | (a|b) c | a (b|c) | a b -c)."""
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
ap.add_argument("--docs", type=int, default=100_000_000)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--queries", type=int, default=256)
ap.add_argument("--shape", type=int, default=-1)
ap.add_argument("--set", action="append", default=[], help="ctx tunable key=value")
args = ap.parse_args()
c = bench.zipf_c()
ranks, strata = bench.make_queries(c, args.queries)
probs = [min(0.5, c / r) for r in ranks]
t0 = time.time()
hi = m.synth_index(args.docs, probs, seed=bench.CORPUS_SEED)
ctx = m.Context(0)
ctx.set("prox_bound_keywords", 1)  # the bench corpus carries no field-end flags: see bench.py
for kv in args.set:
    k, v = kv.split("=")
    ctx.set(k, int(v))
seg = m.Segment(ctx, hi)
gd = hi.dict["docs"].astype(np.int64)
qs = bench.config3_queries(m, strata, args.queries, 1000, args.docs, gd)
if args.shape >= 0:
    qs = [q for i, q in enumerate(qs) if i % 4 == args.shape]
cq = m.prepare(qs)
bs = [m.Batch(ctx, len(qs)), m.Batch(ctx, len(qs))]
for b in bs:
    b.submit_prepared(seg, cq, len(qs))
    b.wait()
t1 = time.perf_counter()
for i in range(args.reps):
    bs[i % 2].wait()
    bs[i % 2].submit_prepared(seg, cq, len(qs))
for b in bs:
    b.wait()
dt = time.perf_counter() - t1
st = bs[0].stats()
res = bs[0].results()
print(json.dumps({"queries": len(qs), "queries_per_s": round(args.reps * len(qs) / dt, 1), "scan_ms": round(st["scan_ms"], 4), "merge_ms": round(st["merge_ms"], 4),
                  "algo_MB": round(st["algo_bytes"] / 1e6, 2), "dev_MB": round(st["dev_bytes"] / 1e6, 2), "items": st["n_items"], "ok": int(sum(r.status == 0 for r in res)),
                  "matches": int(sum(r.total_found for r in res)), "setup_s": round(t1 - t0 - 0, 1), "lib": os.environ.get("MRK_LIB_PATH", "libmrk.so")}))

#!/usr/bin/env python3
"""tools/bm_items_sweep.py [--docs D] -- the headline (common x common) launch of bench.py under different bm_target_items:
how many work items the bitmap kernel's window ranges are cut into (a wave's fixed costs want long runs, the chip wants enough
workgroups).  Scan time per launch from the library's HIP events."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import manticoresearch_amd as m  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=12_500_000)
ap.add_argument("--items", default="1024,2048,3072,4096,6144,8192,12288")
ap.add_argument("--min-windows", default="256", help="comma list: floor of windows per work item (ctx key bm_min_windows)")
args = ap.parse_args()
c = bench.zipf_c()
ranks, strata = bench.make_queries(c, 256)
probs = [min(0.5, c / r) for r in ranks]
hi = m.synth_index(args.docs, probs, seed=bench.CORPUS_SEED)
kw = m.XQNode.keyword
qs = [m.Query(m.XQNode.AND(kw(a, 1), kw(b, 2)), ranker=m.SPH_RANK_BM25, max_matches=1000) for a, b in strata["cc"]]
cq = m.prepare(qs)
out = {}
for items, minw in [(int(x), int(y)) for y in args.min_windows.split(",") for x in args.items.split(",")]:
    ctx = m.Context(0)
    ctx.set("bm_target_items", items)
    ctx.set("bm_min_windows", minw)
    seg = m.Segment(ctx, hi)
    b = m.Batch(ctx, len(qs))
    ts = []
    for _ in range(6):
        b.submit_prepared(seg, cq, len(qs))
        b.wait()
        ts.append(b.stats()["scan_ms"])
    out[f"{items}/{minw}"] = {"scan_ms": round(min(ts[1:]), 4), "n_items": b.stats()["n_items"], "cands": b.stats()["n_cands"]}
    print(items, minw, out[f"{items}/{minw}"], flush=True)
    b.close()
    seg.close()
    ctx.close()
print(json.dumps({"docs": args.docs, "sweep": out}))

#!/usr/bin/env python3
"""bench.py -- queries/sec of the match -> rank -> top-K path on MI355X.

Workload (BASELINE.json metric): 2-term AND, SPH_RANK_BM25, top-1000 over a synthetic Zipf
corpus (default 100 M docs), queries stratified into common x common / selective x common /
selective x selective thirds.  The query file holds 10 000 queries (seed 0x5EED0002),
cut into sets of 3 x 256; a "step" = one set: one batch (kernel launch) per stratum, the sets taken
in turn.  Index segments are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--docs D] [--queries Q]

N > 1, one rank per GPU: the corpus is split into N rowid-range shards (strong scaling: total docs
fixed; `--scaling weak` keeps --docs per GPU instead), every rank scans its shard, partial top-K
lists are all-gathered over RCCL and merged on device, totals are added up.  Launched either by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in the
environment) or plainly as `python bench.py --gpus N`: with no launcher environment the parent
process starts the N ranks itself -- as child processes, before anything touches the GPU -- and
prints rank 0's JSON line; it exits non-zero if any rank fails or fewer than N ranks joined.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
VOCAB = 1 << 20
AVG_TERMS_PER_DOC = 64.0
COMMON = (0.03, 0.3)       # document-probability band of "common" terms
SELECTIVE = (1e-4, 3e-3)   # ... of "selective" terms
QUERY_SEED = 0x5EED0002
CORPUS_SEED = 0x5EED0001
QUERY_FILE = 10_000        # queries per config; the steps cycle through them


def zipf_c() -> float:
    """C such that sum_r min(0.5, C/r) over r = 1..VOCAB equals AVG_TERMS_PER_DOC (s = 1)."""
    lo, hi = 0.1, 50.0
    r = np.arange(1, VOCAB + 1, dtype=np.float64)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if np.minimum(0.5, mid / r).sum() < AVG_TERMS_PER_DOC:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def rank_band(c: float, band) -> tuple:
    """Zipf ranks whose document probability C/r lies inside the band."""
    return int(math.ceil(c / band[1])), int(math.floor(c / band[0]))


def make_queries(c: float, q_per_stratum: int):
    """-> (ranks of the terms to materialise, {stratum: [(term_idx_a, term_idx_b), ...]})."""
    rng = np.random.default_rng(QUERY_SEED)
    cr, sr = rank_band(c, COMMON), rank_band(c, SELECTIVE)
    pairs = {"cc": [], "sc": [], "ss": []}
    for _ in range(q_per_stratum):
        a, b = rng.integers(cr[0], cr[1] + 1, 2)
        while a == b:
            b = rng.integers(cr[0], cr[1] + 1)
        pairs["cc"].append((int(a), int(b)))
        pairs["sc"].append((int(rng.integers(sr[0], sr[1] + 1)), int(rng.integers(cr[0], cr[1] + 1))))
        a, b = rng.integers(sr[0], sr[1] + 1, 2)
        while a == b:
            b = rng.integers(sr[0], sr[1] + 1)
        pairs["ss"].append((int(a), int(b)))
    ranks = sorted({r for v in pairs.values() for p in v for r in p})
    idx = {r: i for i, r in enumerate(ranks)}
    return ranks, {k: [(idx[a], idx[b]) for a, b in v] for k, v in pairs.items()}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=52)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--docs", type=int, default=100_000_000, help="documents in the whole corpus")
    ap.add_argument("--queries", type=int, default=256, help="queries per stratum per step")
    ap.add_argument("--item-bytes", type=int, default=0)
    ap.add_argument("--skiplist-block", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the oracle timing sample")
    ap.add_argument("--latency-samples", type=int, default=96)
    ap.add_argument("--strata", default="cc,sc,ss", help="subset of strata to run (profiling aid; the metric uses all three)")
    ap.add_argument("--path", type=int, default=0, help="0 = packed doclists (default), 1 = VLB-direct")
    ap.add_argument("--attr-nibbles", action="store_true", help="build the one-byte tf/field plane (ctx key attr_nibbles)")
    ap.add_argument("--ctx", action="append", default=[], metavar="KEY=VALUE", help="context tunable (mrk_ctx_set), e.g. attr_seq=0")
    ap.add_argument("--no-config3", action="store_true", help="skip the extra 3-term AND/OR mix leg (BASELINE config 3)")
    ap.add_argument("--no-config5", action="store_true", help="skip the extra Zipf-mix leg with PHRASE + field weights, 1024 queries per launch (BASELINE config 5, one GPU's share)")
    ap.add_argument("--config5-docs", type=int, default=125_000_000, help="docs of the config-5 leg's corpus (one eighth of 1 B)")
    ap.add_argument("--query-file", type=int, default=QUERY_FILE, help="queries in the query file the steps cycle through")
    ap.add_argument("--sets", type=int, default=0, help="steps kept in flight (sets of batches used in turn); 0 = 2 on one GPU, 4 sharded")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = --docs in total, split N ways (default); weak = --docs per GPU (BASELINE config 4: 8 x 100 M)")
    args = ap.parse_args()

    # (MRK_FORCE_DIST=1 with --gpus 1 rehearses the same parent -> ranks -> JSON relay with ONE rank: the only multi-rank
    # path a one-GPU box can run)
    if (args.gpus > 1 or os.environ.get("MRK_FORCE_DIST")) and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(max(1, args.gpus)))  # no torch, no HIP in this process
    if os.environ.get("MRK_DEBUG_HANG"):  # dump every thread's Python stack if the run is still going after that many seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["MRK_DEBUG_HANG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    # MRK_FORCE_DIST=1 runs the sharded code path (RCCL exchange + device merge) even with one rank
    force_dist = bool(os.environ.get("MRK_FORCE_DIST")) and "RANK" in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist

        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: local GPU {local_rank} of {torch.cuda.device_count()} visible")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", timeout=__import__("datetime").timedelta(seconds=900))
        if dist.get_world_size() != world:
            raise SystemExit(f"rank {rank}: {dist.get_world_size()} ranks joined, {world} expected")

    import manticoresearch_amd as m
    from manticoresearch_amd import dist as mdist

    t_setup = time.time()
    c = zipf_c()
    # the query file: QUERY_FILE queries in three strata, cut into sets of 3 x args.queries; step i runs set i mod n_qsets
    n_qsets = max(1, (args.query_file // 3) // args.queries)
    ranks, strata_all = make_queries(c, n_qsets * args.queries)
    strata = {k: v[: args.queries] for k, v in strata_all.items()}  # set 0 (the latency sample, config 3 and the CPU baseline draw from it)

    def qset(s_, k_):
        return strata_all[s_][k_ * args.queries:(k_ + 1) * args.queries]

    probs = [min(0.5, c / r) for r in ranks]
    # rowid-range shards of ONE corpus (postings are keyed on the global rowid): rank r holds rows [row0, row0 + shard_docs)
    corpus_docs = args.docs * world if args.scaling == "weak" else args.docs
    row0 = rank * corpus_docs // world
    shard_docs = (rank + 1) * corpus_docs // world - row0
    hi = m.synth_index(shard_docs, probs, seed=CORPUS_SEED, rowid_base=row0, skiplist_block_size=args.skiplist_block)
    t_gen = time.time() - t_setup

    ctx = m.Context(local_rank)
    # the main corpus is written WITHOUT field-end flags, config 5's with the flag on the hit at a field's last position whatever the
    # word (what the reference's indexer writes): in both, hits at one position reach the ranker in query-position order and the tighter of
    # the two weight bounds in front of the hit pass is sound (mrk_kprune.h, prox_bounds)
    ctx.set("prox_bound_keywords", 1)
    if args.item_bytes:
        ctx.set("item_bytes", args.item_bytes)
    ctx.set("path", args.path)
    if args.attr_nibbles:
        ctx.set("attr_nibbles", 1)
    for kv in args.ctx:
        ctx.set(kv.split("=")[0], int(kv.split("=")[1]))
    seg = m.Segment(ctx, hi, rowid_base=row0)
    batch = m.Batch(ctx, args.queries)
    sharded = world > 1 or force_dist
    state = {"i": 0, "pending": [], "rec": [False] * 8}

    # global DF / N so that every shard ranks with the same IDF (local_df, sphinxrt.cpp:6501-6521)
    local_docs = hi.dict["docs"].astype(np.int64)
    if world > 1 or force_dist:
        if not os.environ.get("MRK_TORCH_EXCHANGE"):  # default: the exchange and the DF sums through the library's own RCCL communicator
            mdist.lib_comm_init(ctx)
        global_docs, total_docs = mdist.global_df(local_docs, shard_docs, local_rank, ctx=ctx)
    else:
        global_docs, total_docs = local_docs, shard_docs

    kw = m.XQNode.keyword
    K = 1000

    def mkq(a, b):
        return m.Query(m.XQNode.AND(kw(a, 1), kw(b, 2)), ranker=m.SPH_RANK_BM25, max_matches=K, total_docs=int(total_docs),
                       local_docs={a: int(global_docs[a]), b: int(global_docs[b])})

    names = [x for x in ["cc", "sc", "ss"] if x in args.strata.split(",")]
    prepared = {s: [m.prepare([mkq(a, b) for a, b in qset(s, k_)]) for k_ in range(n_qsets)] for s in names}
    nq = args.queries
    # Work groups of a step.  One GPU: one batch per stratum (the cc launch is timed on its own for the roofline).
    # Sharded: the step's queries go down as ONE batch per rank, so a step costs one scan launch, one selection,
    # one RCCL all-gather of result rows and one merge.
    if sharded:
        prepared["all"] = [m.prepare([mkq(a, b) for s in names for a, b in qset(s, k_)]) for k_ in range(n_qsets)]
        groups = [("all", len(names) * nq)]
    else:
        groups = [(s, nq) for s in names]
    # two sets of batches used alternately: a step submits its batches and only then collects the previous step's
    # results, so host-side planning of one step overlaps the kernels of the other (what concurrent searchd workers
    # do); every step's results still land in host memory inside the timed region
    n_sets = args.sets if args.sets > 0 else (4 if sharded else 2)  # the sharded chain (scan, selection, exchange, merge) is longer: keep 4 steps in flight
    sets = [{g: (batch if (i == 0 and g == "cc") else m.Batch(ctx, n)) for g, n in groups} for i in range(n_sets)]
    merger = None
    if sharded:
        merger = mdist.ShardMerger(ctx, batch, len(names) * nq, K, world, local_rank, n_batches=1, n_sets=n_sets)
        for i in range(n_sets):
            merger.attach([sets[i]["all"]], set_index=i)

    per = {s: {"scan_ms": 0.0, "merge_ms": 0.0, "algo_bytes": 0, "n": 0} for s in ["cc", "sc", "ss", "all"]}

    host_ms = {"submit": 0.0, "wait": 0.0, "merge": 0.0, "n": 0}

    def record_stats(s, st) -> None:
        per[s]["scan_ms"] += st["scan_ms"]
        per[s]["merge_ms"] += st["merge_ms"]
        per[s].update(packed=st["packed"], n_items=st["n_items"], n_cands=st["n_cands"], n_items_bm=st["n_items_bm"])
        per[s]["algo_bytes"] = per[s].get("algo_bytes", 0) + st["algo_bytes"]  # (summed like scan_ms: the sets differ)
        per[s]["dev_bytes"] = per[s].get("dev_bytes", 0) + st["dev_bytes"]
        per[s]["plan_ms"] = per[s].get("plan_ms", 0.0) + st["plan_ms"]
        per[s]["submit_ms"] = per[s].get("submit_ms", 0.0) + st["submit_ms"]
        per[s]["n"] += 1

    def collect(idx: int, record: bool) -> None:
        cur = sets[idx]
        t_w = time.perf_counter()
        for g, _ in groups:
            cur[g].wait()
        host_ms["wait"] += (time.perf_counter() - t_w) * 1e3
        if record:
            for g, _ in groups:
                record_stats(g, cur[g].stats())

    used = [False] * 8

    def step(record: bool) -> None:
        idx = state["i"] % n_sets
        qk = state["i"] % n_qsets  # the query set of this step
        state["i"] += 1
        t_s = time.perf_counter()
        if merger is not None:
            # sharded: submit -> (event) -> all-gather -> merge -> host copy is ONE stream-ordered chain queued right
            # here; the host only waits when a set comes round again, n_sets steps later
            t_w = time.perf_counter()
            merger.finish(idx)  # merged rows in host memory; reruns / re-exchanges if a shard flagged a row
            if used[idx]:
                sets[idx]["all"].wait()
                if state["rec"][idx]:
                    record_stats("all", sets[idx]["all"].stats())
            host_ms["wait"] += (time.perf_counter() - t_w) * 1e3
            t_s = time.perf_counter()
            sets[idx]["all"].submit_prepared(seg, prepared["all"][qk], len(names) * nq)
            host_ms["submit"] += (time.perf_counter() - t_s) * 1e3
            t_m = time.perf_counter()
            merger.merge_attached(1, set_index=idx, to_host=True, after_submit=True)
            host_ms["merge"] += (time.perf_counter() - t_m) * 1e3
            host_ms["n"] += 1
            used[idx] = True
            state["rec"][idx] = record
            return
        for g, n in groups:
            sets[idx][g].submit_prepared(seg, prepared[g][qk], n)
        host_ms["submit"] += (time.perf_counter() - t_s) * 1e3
        host_ms["n"] += 1
        state["pending"].append((idx, record))
        while len(state["pending"]) >= n_sets:  # the set the next step reuses must be collected by then
            collect(*state["pending"].pop(0))

    def sync() -> None:
        while state["pending"]:
            collect(*state["pending"].pop(0))
        if merger is not None:
            for i in range(n_sets):
                merger.finish(i)
                if used[i]:
                    sets[i]["all"].wait()
                    if state["rec"][i]:
                        record_stats("all", sets[i]["all"].stats())
                    used[i] = False
        for st_ in sets:
            for bb in st_.values():
                bb.wait()
        if sharded:
            import torch

            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync()
    for k_ in host_ms:
        host_ms[k_] = 0
    if merger is not None and merger.timing is not None:
        merger.timing.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1 or force_dist:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sharded runs scan all strata in one launch: time the cc launch on its own (outside the timed region) for the
    # roofline fields
    if "cc" not in names:
        prepared["cc"] = [m.prepare([mkq(a, b) for a, b in qset("cc", k_)]) for k_ in range(n_qsets)]
        per.setdefault("cc", {"scan_ms": 0.0, "merge_ms": 0.0, "algo_bytes": 0, "n": 0})
    if sharded or "cc" not in names:
        for k_ in range(min(n_qsets, 4)):
            batch.submit_prepared(seg, prepared["cc"][k_], nq)
            batch.wait()
            record_stats("cc", batch.stats())
    # every cc set once more for its results: 8 B per returned match are output bytes, total_found checks the host-side count
    res_cc_sets = []
    for k_ in range(n_qsets):
        batch.submit_prepared(seg, prepared["cc"][k_], nq)
        batch.wait()
        res_cc_sets.append(batch.results())

    # single-query latency (batch of one), p50 over a stratified sample
    lat = []
    if rank == 0 and world == 1 and args.latency_samples > 0:
        sample = []
        per_s = max(1, args.latency_samples // 3)
        for s in names:
            sample += [mkq(a, b) for a, b in strata[s][:per_s]]
        singles = [m.prepare([q]) for q in sample]
        for cq in singles[:3]:
            batch.submit_prepared(seg, cq, 1)
            batch.wait()
        for cq in singles:
            t1 = time.perf_counter()
            batch.submit_prepared(seg, cq, 1)
            batch.wait()
            lat.append((time.perf_counter() - t1) * 1e3)

    total_queries = len(names) * nq * args.steps
    qps = total_queries / elapsed
    cc = per["cc"]
    scan_ms = cc["scan_ms"] / max(1, cc["n"])
    # which kernel carried the cc launch: the two-bitmap AND kernel (dense keywords) or the block scan
    bm_share = cc.get("n_items_bm", 0) / max(1, cc.get("n_items", 1))
    kernel_tag = "bm" if bm_share > 0.5 else ("pk" if cc.get("packed") else "vlb")
    # algorithmic bytes of the launch = SUM over its queries of min(B_ref(q), B_dev(q)):
    #   B_ref(q) = the two doclists in the reference's VLB format (CSphDictEntry::m_iDoclistLength) + 8 B per returned match
    #   B_dev(q) = what the device format makes the kernel read: both bitmaps (256 B per 2048-rowid window each) + the
    #              128-byte lines of tf / field words the MATCHED docs' slots touch in each keyword's packed array
    #              (counted on the host from the two doclists, mrk_host_index_pair_stats) + 8 B per returned match
    algo_detail = None
    out_bytes = sum(8 * len(r.rowid) for rs in res_cc_sets for r in rs) / max(1, len(res_cc_sets))
    if kernel_tag == "bm" and rank == 0:
        # per launch = the AVERAGE over the query file's cc sets (every one of them was timed in the loop, the same number of times
        # when --steps is a multiple of the set count)
        t_ps = time.time()
        nwin = (shard_docs + 2047) // 2048
        seq_plane = not args.attr_nibbles and not any(kv.startswith("attr_seq=0") for kv in args.ctx)
        tot = {"algo": 0, "ref": 0, "dev": 0, "ref_smaller": 0, "bitmap": 0, "lines": 0, "full": 0}
        for k_ in range(n_qsets):
            pairs_k, res_k = qset("cc", k_), res_cc_sets[k_]
            ps = m.pair_stats(hi, pairs_k)
            b_ref = [int(hi.dict[a]["doclist_len"]) + int(hi.dict[b]["doclist_len"]) + 8 * len(r.rowid) for (a, b), r in zip(pairs_k, res_k)]
            # the 128-byte lines the matched docs' slots touch in the plane the launched kernel gathers from: the slot-ordered
            # two-byte plane (default), the one-byte plane (--attr-nibbles) or the block decoder's interleaved words (attr_seq=0)
            line = [(p[5] + p[6]) if args.attr_nibbles else (p[7] + p[8]) if seq_plane else (p[3] + p[4]) for p in ps]
            b_dev = [2 * nwin * 256 + 128 * ln + 8 * len(r.rowid) for ln, r in zip(line, res_k)]
            if not os.environ.get("MRK_LIB_PATH"):  # (kernel-experiment libraries may skip the scoring step)
                assert all(p[0] == r.total_found for p, r in zip(ps, res_k)), "host intersection and device total_found disagree"
            tot["algo"] += sum(min(x, y) for x, y in zip(b_ref, b_dev))
            tot["ref"] += sum(b_ref)
            tot["dev"] += sum(b_dev)
            tot["ref_smaller"] += sum(x < y for x, y in zip(b_ref, b_dev))
            tot["bitmap"] += 2 * nwin * 256 * len(ps)
            tot["lines"] += 128 * sum(line)
            tot["full"] += sum(((p[1] + 127) // 128 + (p[2] + 127) // 128) * (128 if args.attr_nibbles else 256) for p in ps)
        algo = tot["algo"] / n_qsets
        algo_detail = {"sets_averaged": n_qsets, "sum_min_per_query": int(algo), "sum_B_ref": int(tot["ref"] / n_qsets), "sum_B_dev": int(tot["dev"] / n_qsets),
                       "queries_where_ref_is_smaller_per_set": round(tot["ref_smaller"] / n_qsets, 1),
                       "bitmap_bytes": int(tot["bitmap"] / n_qsets), "attr_line_bytes_touched": int(tot["lines"] / n_qsets),
                       "attr_plane": "one byte per posting" if args.attr_nibbles else "two bytes per posting, slot order" if seq_plane else "interleaved block words",
                       "attr_bytes_if_every_word_were_read": int(tot["full"] / n_qsets), "host_count_s": round(time.time() - t_ps, 1)}
    else:
        algo = min(cc["algo_bytes"], cc.get("dev_bytes", cc["algo_bytes"])) / max(1, cc["n"]) + out_bytes
    achieved = algo / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic, traffic_src, measured_strata = None, None, {}
    kernel_name = {"bm": "mrk::scan_bm_kernel", "pk": "mrk::scan_pk_kernel", "vlb": "mrk::scan_kernel"}[kernel_tag]
    device_format = {"bm": "doc-set bitmaps (2048-rowid windows) + packed tf/field bytes gathered by rank",
                     "pk": "packed 128-doc blocks (bit-packed rowid offsets + tf/field bytes)",
                     "vlb": "reference .spd VLB"}[kernel_tag]
    tr_path = os.path.join(ROOT, "profiles", "traffic.json")
    ksha = kernel_sources_sha()
    if os.path.exists(tr_path):
        try:
            tj = json.load(open(tr_path))
            if (tj.get("docs") == args.docs and tj.get("queries") == nq and tj.get("skiplist_block") == args.skiplist_block
                    and tj.get("kernel_tag") == kernel_tag and not args.attr_nibbles):
                if tj.get("kernel_sources_sha") == ksha:
                    traffic = tj.get("traffic_bytes_per_launch")
                    traffic_src = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this configuration and of THESE kernel sources "
                                   f"(sha {ksha}; tools/traffic.sh), calibrated on known-size kernels in this kernel's access patterns; "
                                   "a PMC pass cannot run inside this process")
                    measured_strata = tj.get("strata", {})
                else:  # measured on other kernel sources: not this run's kernel -- say so instead of reporting it
                    traffic_src = (f"profiles/traffic.json was measured at kernel sources {tj.get('kernel_sources_sha')}, this run has {ksha}: "
                                   "stale, not reported (re-run tools/traffic.sh)")
        except Exception:
            traffic = None

    out = {
        "metric": "queries/sec, 2-term AND BM25 top-1000 (p50 latency in p50_latency_ms)",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8 postings -> u32 rowids, f32 BM25, i32 weights",
        "data": "synthetic",
        "config": {
            "workload": f"{corpus_docs // 1_000_000}M docs (Zipf s=1, V=2^20, {AVG_TERMS_PER_DOC:.0f} terms/doc), 2-term AND, "
                        f"SPH_RANK_BM25, top-{K}, {3 * nq} queries/step in 3 strata (cc/sc/ss) out of a {n_qsets * 3 * nq}-query file, "
                        f"skiplist_block_size={args.skiplist_block}, inline hits",
            "docs": corpus_docs,
            "shards": world,
            "queries_per_step": 3 * nq,
            "k": K,
            "query_file": f"{n_qsets * 3 * nq} queries (seed {QUERY_SEED:#x}) in {n_qsets} sets of 3 x {nq}; step i runs set i mod {n_qsets}",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": f"{kernel_name} (common x common stratum launch)",
            "algo_bytes_per_launch": int(algo),
            "algo_bytes_detail": algo_detail,
            "ref_format_bytes_per_launch": int(cc["algo_bytes"] / max(1, cc["n"])),
            "device_format_bytes_per_launch": int(cc.get("dev_bytes", 0) / max(1, cc["n"])),
            "kernel_sources_sha": ksha,
            "device_format": device_format,
            "launch_ms": round(scan_ms, 4),
            "timed": ("cc launch timed after the loop: the sharded loop scans all strata in one launch per rank" if sharded
                      else "cc launches inside the timed region (HIP events on the scan stream)"),
        },
        # sc / ss: galloping over the block index reads less than the algorithmic bytes ("skip-assisted"): their
        # algorithmic rate may exceed the HBM peak and is never folded into the headline; measured bytes next to it
        "strata": {
            s: {"label": "headline (no skipping possible: algorithmic ~ read bytes)" if s == "cc" else "skip-assisted" if s in ("sc", "ss") else "all strata in one launch",
                "algo_GBps": round(min(per[s]["algo_bytes"], per[s].get("dev_bytes", per[s]["algo_bytes"])) / max(1e-9, per[s]["scan_ms"] * 1e-3) / 1e9, 1),
                "measured_MB": (round(measured_strata[s]["measured_bytes_per_launch"] / 1e6, 2) if s in measured_strata else None),
                "scan_ms": round(per[s]["scan_ms"] / max(1, per[s]["n"]), 4),
                "merge_ms": round(per[s]["merge_ms"] / max(1, per[s]["n"]), 4),
                "algo_MB": round(per[s]["algo_bytes"] / max(1, per[s]["n"]) / 1e6, 2), "dev_MB": round(per[s].get("dev_bytes", 0) / max(1, per[s]["n"]) / 1e6, 2), "items": per[s].get("n_items", 0), "cands": per[s].get("n_cands", 0),
                "host_plan_ms": round(per[s].get("plan_ms", 0.0) / max(1, per[s]["n"]), 4),
                "host_submit_ms": round(per[s].get("submit_ms", 0.0) / max(1, per[s]["n"]), 4)} for s in (names + ["all"]) if per[s]["n"]},
        "strata_run": names,
        "p50_latency_ms": round(float(np.percentile(lat, 50)), 4) if lat else None,
        "p95_latency_ms": round(float(np.percentile(lat, 95)), 4) if lat else None,
        "setup_s": {"generate": round(t_gen, 1), "index_MB": round((hi.spd.size + hi.spp.size + hi.spe.size) / 1e6, 1)},
    }

    out["host_ms_per_step"] = {k_: round(v / max(1, host_ms["n"]), 4) for k_, v in host_ms.items() if k_ != "n"}
    # BASELINE config 3 on the same corpus, outside the timed region: 3-term mixes a b c / (a|b) c / a (b|c) / a b -c under
    # SPH_RANK_PROXIMITY_BM25 (hitlist decode); a = selective keyword, b and c = common ones.  Extra information only.
    if rank == 0 and world == 1 and not sharded and not args.no_config3 and args.path == 0 and set(names) == {"cc", "sc", "ss"}:
        c3 = config3_queries(m, strata, nq, K, total_docs, global_docs)
        cq3 = m.prepare(c3)
        b3 = [m.Batch(ctx, nq), m.Batch(ctx, nq)]
        for bb in b3:
            bb.submit_prepared(seg, cq3, nq)
            bb.wait()
        t3 = time.perf_counter()
        reps = 6
        for i in range(reps):
            b3[i % 2].wait()
            b3[i % 2].submit_prepared(seg, cq3, nq)
        for bb in b3:
            bb.wait()
        dt3 = time.perf_counter() - t3
        st3 = b3[0].stats()
        res3 = b3[0].results()
        out["config3"] = {"workload": f"{nq} queries/launch: a b c | (a|b) c | a (b|c) | a b -c, SPH_RANK_PROXIMITY_BM25, top-{K}",
                          "queries_per_s": round(reps * nq / dt3, 1), "scan_ms": round(st3["scan_ms"], 4),
                          "merge_ms": round(st3["merge_ms"], 4), "algo_MB": round(st3["algo_bytes"] / 1e6, 2),
                          "ok": int(sum(r.status == 0 for r in res3)), "matches": int(sum(r.total_found for r in res3)),
                          "weight_bound": "by keywords (prox_bound_keywords=1: this corpus holds no field-end flags)"}
        for bb in b3:
            bb.close()
    # HBM footprint of what this run holds: the materialised query terms only (the corpus' other ~10^6 terms were never generated)
    if rank == 0:
        postings = int(hi.dict["docs"].astype(np.int64).sum())
        dev_b = int(seg.device_bytes)
        out["footprint"] = {
            "device_bytes": dev_b, "terms_materialised": int(len(hi.dict)), "postings_materialised": postings,
            "device_bytes_per_posting": round(dev_b / max(1, postings), 2),
            "reference_format_bytes": int(hi.spd.size + hi.spp.size + hi.spe.size),
            "holds": "reference .spd/.spp verbatim + 128-doc block index + packed blocks (rowid offsets, tf/field words, hit references) + "
                     "for keywords in >= 1/64 of the docs a doc-set bitmap, rank directory and the slot-ordered tf/field plane",
            "full_vocabulary_projection_bytes": int(dev_b / max(1, postings) * AVG_TERMS_PER_DOC * shard_docs),
            "projection_note": f"bytes per posting x {AVG_TERMS_PER_DOC:.0f} postings per doc x {shard_docs} docs: an upper estimate (the materialised terms "
                               "are the query terms: dense keywords, whose bitmaps and second tf/field plane cost most per posting, are over-represented)",
        }
    # BASELINE config 5 on one GPU's share (1 B docs / 8): Zipf query mix incl. PHRASE, field weights (10,5,2,1), SPH_RANK_PROXIMITY_BM25,
    # 1024 queries per launch.  Its own corpus (4 fields, field-end markers); outside the timed region.  Extra information only.
    if rank == 0 and world == 1 and not sharded and not args.no_config5 and args.path == 0 and set(names) == {"cc", "sc", "ss"}:
        try:
            out["config5"] = config5_leg(m, ctx, args, c, K)
        except Exception as e:  # (the headline line must not die with the extra leg)
            out["config5"] = {"error": f"{type(e).__name__}: {e}"}
    if merger is not None and merger.timing:
        out["dist_timing_ms_per_call"] = {k_: round(v / max(1, merger.timing["calls"]), 4) for k_, v in merger.timing.items() if k_ != "calls"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(hi, strata, total_docs, K, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None

    for st_ in sets:
        for bb in st_.values():
            bb.close()
    batch.close()
    seg.close()
    ctx.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def config5_queries(m, strata, n, K, total_docs, global_docs, fw):
    """BASELINE config 5's mix: 60 % 2-term AND, 20 % 3-term AND / OR mixes, 20 % 2-3-word PHRASE; field weights; PROXIMITY_BM25.
    Keywords from the same Zipf strata as the headline queries."""
    kw = m.XQNode.keyword
    OR_, ANDNOT_, PHR = m.SPH_QUERY_OR, m.SPH_QUERY_ANDNOT, m.SPH_QUERY_PHRASE
    rng = np.random.default_rng(QUERY_SEED + 5)
    pools = [strata["cc"], strata["sc"], strata["ss"]]
    out = []
    for i in range(n):
        a_, b_ = pools[i % 3][(i // 3) % len(pools[i % 3])]
        c_ = strata["cc"][(i * 7) % len(strata["cc"])][0]
        if c_ in (a_, b_):
            c_ = strata["cc"][(i * 7) % len(strata["cc"])][1]
        u = rng.random()
        if u < 0.6:
            root, used = m.XQNode.AND(kw(a_, 1), kw(b_, 2)), (a_, b_)
        elif u < 0.8:
            ka, kb, kc = kw(a_, 1), kw(b_, 2), kw(c_, 3)
            root = [m.XQNode.AND(ka, kb, kc), m.XQNode.AND(m.XQNode(OR_, [ka, kb]), kc), m.XQNode.AND(ka, m.XQNode(OR_, [kb, kc])),
                    m.XQNode(ANDNOT_, [m.XQNode.AND(ka, kb), kc])][i % 4]
            used = (a_, b_, c_)
        else:  # phrases of common words (a phrase of two rare words matches nothing in a synthetic corpus)
            p_, q_ = strata["cc"][i % len(strata["cc"])]
            words = [p_, q_] + ([c_] if (i % 2 and c_ not in (p_, q_)) else [])
            root, used = m.XQNode(PHR, [kw(w, j + 1) for j, w in enumerate(words)]), tuple(words)
        out.append(m.Query(root, ranker=m.SPH_RANK_PROXIMITY_BM25, max_matches=K, total_docs=int(total_docs), field_weights=list(fw),
                           local_docs={t: int(global_docs[t]) for t in used}))
    return out


def config5_leg(m, ctx, args, c, K):
    t0 = time.time()
    nq5, fw = 1024, (10, 5, 2, 1)
    ranks, strata = make_queries(c, 342)
    probs = [min(0.5, c / r) for r in ranks]
    hi5 = m.synth_index(args.config5_docs, probs, seed=CORPUS_SEED + 5, n_fields=4, end_markers=2, skiplist_block_size=args.skiplist_block)
    t_gen = time.time() - t0
    # (config 5's corpus carries field-end flags the way the reference's indexer writes them -- on the hit at a field's last position,
    # whatever the word: end_markers=2 -- so the statement made for the main corpus holds here too)
    seg5 = m.Segment(ctx, hi5)
    gd = hi5.dict["docs"].astype(np.int64)
    qs = config5_queries(m, strata, nq5, K, args.config5_docs, gd, fw)
    cq = m.prepare(qs)
    bs = [m.Batch(ctx, nq5), m.Batch(ctx, nq5)]
    for bb in bs:
        bb.submit_prepared(seg5, cq, nq5)
        bb.wait()
    t1 = time.perf_counter()
    reps = 6
    for i in range(reps):
        bs[i % 2].wait()
        bs[i % 2].submit_prepared(seg5, cq, nq5)
    for bb in bs:
        bb.wait()
    dt = time.perf_counter() - t1
    st, res = bs[0].stats(), bs[0].results()
    out = {"workload": f"{args.config5_docs // 1_000_000}M docs (one eighth of 1 B), 4 fields with weights {fw}, {nq5} queries/launch: 60 % 2-term AND, 20 % 3-term "
                       f"AND/OR mixes, 20 % 2-3-word PHRASE, SPH_RANK_PROXIMITY_BM25, top-{K}",
           "queries_per_s": round(reps * nq5 / dt, 1), "ms_per_launch": round(dt / reps * 1e3, 3), "scan_ms": round(st["scan_ms"], 4),
           "select_ms": round(st["merge_ms"], 4), "doclist_MB": round(st["algo_bytes"] / 1e6, 2), "ok": int(sum(r.status == 0 for r in res)),
           "declined": int(sum(r.status != 0 for r in res)), "matches": int(sum(r.total_found for r in res)),
           "device_bytes": int(seg5.device_bytes), "postings": int(gd.sum()), "setup_s": {"generate": round(t_gen, 1), "total": round(time.time() - t0, 1)},
           "weight_bound": "by keywords (prox_bound_keywords=1: this corpus flags the hit at a field's last position, as the reference's indexer does)"}
    for bb in bs:
        bb.close()
    seg5.close()
    return out


def kernel_sources_sha() -> str:
    """Short hash of the sources that shape the headline launch (kernel, its helpers, the work-item cut in the host code):
    profiles/traffic.json carries the one it was measured at."""
    import hashlib

    h = hashlib.sha256()
    for f in ("mrk_scan_bm.hip", "mrk_kprune.h", "mrk_kcommon.h", "mrk_dev.h", "mrk_host.cpp", "mrk_host_int.h", "mrk_pack.cpp"):
        with open(os.path.join(ROOT, "manticoresearch_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def config3_queries(m, strata, nq, K, total_docs, global_docs):
    """BASELINE config 3: 3-term mixes a b c / (a|b) c / a (b|c) / a b -c under SPH_RANK_PROXIMITY_BM25 (hitlist decode);
    a = the selective keyword of the i-th sc pair, b = its common one, c = a common keyword of the i-th cc pair."""
    kw = m.XQNode.keyword
    OR_, ANDNOT_ = m.SPH_QUERY_OR, m.SPH_QUERY_ANDNOT
    c3 = []
    for i in range(nq):
        a_, b_ = strata["sc"][i]
        c_ = strata["cc"][i][0] if strata["cc"][i][0] != b_ else strata["cc"][i][1]
        ka, kb, kc = kw(a_, 1), kw(b_, 2), kw(c_, 3)
        root = [m.XQNode.AND(ka, kb, kc), m.XQNode.AND(m.XQNode(OR_, [ka, kb]), kc), m.XQNode.AND(ka, m.XQNode(OR_, [kb, kc])),
                m.XQNode(ANDNOT_, [m.XQNode.AND(ka, kb), kc])][i % 4]
        c3.append(m.Query(root, ranker=m.SPH_RANK_PROXIMITY_BM25, max_matches=K, total_docs=int(total_docs),
                          local_docs={t: int(global_docs[t]) for t in (a_, b_, c_)}))
    return c3


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script as CHILD processes (env rendezvous
    on 127.0.0.1), relay rank 0's JSON line, return non-zero if a rank failed or the line does not report N ranks.
    Runs before torch or HIP are imported; nothing is exec'ed over a process that touched the GPU."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.2)
    if failed is None:
        failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:  # one rank down: the others would sit in the rendezvous / a collective until it times out
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        sys.stderr.write(f"bench.py: rank {failed} exited with {procs[failed].returncode}; {n}-rank run aborted\n")
        return 1
    reader.join(timeout=10)
    line = next((ln for ln in reversed(out0) if ln.lstrip().startswith("{")), None)
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    if json.loads(line).get("n_gpus") != n:
        sys.stderr.write(f"bench.py: the run reports {json.loads(line).get('n_gpus')} ranks, {n} expected\n")
        return 1
    sys.stdout.write(line if line.endswith("\n") else line + "\n")
    sys.stdout.flush()
    return 0


def cpu_baseline(hi, strata, total_docs, K, budget_s: float) -> dict:
    """The oracle (kind "port") on this box's host cores over a bounded, stratified sample of the
    SAME queries and index bytes: one independent query per thread, the reference's own
    parallelism model (searchd.cpp:5654).  Test infrastructure used as the reported baseline only."""
    from oracle import oracle as orc

    oi = orc.Index(hi.spd, hi.spp, hi.spe, hi.dict.view(orc.DICT_DTYPE), hi.total_docs, hi.skiplist_block_size,
                   hi.hit_format, hi.n_fields)
    cidx = oi.c_struct()

    def fq(a, b):
        return orc.FlatQuery(orc.op(orc.OP_AND, orc.term(a, 1), orc.term(b, 2)), ranker=orc.RANK_BM25, max_matches=K,
                             total_docs_override=int(total_docs))

    names = ["cc", "sc", "ss"]
    # single-thread pass: interleave strata so that the sample keeps the workload's mix
    sample, t_single = [], []
    t_begin = time.perf_counter()
    i = 0
    while time.perf_counter() - t_begin < budget_s * 0.35 and i < len(strata["cc"]):
        for s in names:
            q = fq(*strata[s][i])
            t1 = time.perf_counter()
            q.run(oi, cidx)
            t_single.append(time.perf_counter() - t1)
            sample.append(q)
        i += 1
    cores = orc.usable_cpus()
    # all-core pass (C worker pool): every query of the sample `repeat` times, one query per thread at a time
    one_pass = sum(t_single)
    repeat = max(1, int(round(budget_s * 0.65 * cores / max(one_pass, 1e-6))))
    repeat = min(repeat, 4 * cores)
    wall = orc.search_many(oi, sample, repeat, cores)
    return {
        "value": round(repeat * len(sample) / wall, 3),
        "unit": "queries/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{len(sample)} queries ({len(sample) // 3} per stratum, same index bytes) x {repeat} repeats on {cores} "
                  f"threads, one query per thread; single-thread {len(sample) / one_pass:.3f} q/s, "
                  f"p50 {np.percentile(t_single, 50) * 1e3:.2f} ms",
        "single_thread_qps": round(len(sample) / one_pass, 3),
    }


if __name__ == "__main__":
    main()

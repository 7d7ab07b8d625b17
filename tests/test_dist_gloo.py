"""CPU, world_size 2 over gloo: the one exchange step of the sharded path -- global DF all-reduce,
all-gather of partial top-K, all-reduce of totals -- with per-shard results produced by the oracle
and the merge checked against the oracle on the union corpus."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 100
N_DOCS = 30000
PROBS = [0.3, 0.1, 0.02]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import manticoresearch_amd as m
    from manticoresearch_amd import dist as mdist
    from oracle import oracle as orc

    hi = m.synth_index(N_DOCS, PROBS, seed=99, shard=rank, n_threads=1)
    gdocs, total = mdist.global_df(hi.dict["docs"].astype(np.int64), N_DOCS)
    oi = orc.Index(hi.spd, hi.spp, hi.spe, hi.dict.view(orc.DICT_DTYPE), N_DOCS, hi.skiplist_block_size, 1, 2)
    pairs = [(0, 1), (0, 2), (1, 2)]
    keys = torch.zeros((len(pairs), 1024), dtype=torch.int64)
    counts = torch.zeros((len(pairs),), dtype=torch.int32)
    totals = torch.zeros((len(pairs),), dtype=torch.int64)
    for qi, (a, b) in enumerate(pairs):
        r = orc.search(oi, orc.op(orc.OP_AND, orc.term(a, 1), orc.term(b, 2)), ranker=orc.RANK_BM25, max_matches=K,
                       total_docs_override=total, local_docs={a: int(gdocs[a]), b: int(gdocs[b])})
        k = ((r.weight.astype(np.int64).astype(np.uint64) ^ np.uint64(0x80000000)) & np.uint64(0xFFFFFFFF)) << np.uint64(32)
        k |= (~(r.rowid.astype(np.uint64) + np.uint64(rank * N_DOCS))) & np.uint64(0xFFFFFFFF)
        keys[qi, : len(k)] = torch.from_numpy(k.view(np.int64))
        counts[qi] = len(k)
        totals[qi] = r.total_found
    local_totals = totals.clone()
    keys_all, counts_all, totals = mdist.exchange_partial_topk(keys, counts, totals)
    # the one-collective form: a row per query = keys | count | total_found
    rows = torch.zeros((len(pairs), mdist.ROW_WORDS), dtype=torch.int64)
    rows[:, :1024] = keys
    rows[:, 1024] = counts.to(torch.int64)
    rows[:, 1025] = local_totals
    rows_all = mdist.exchange_rows(rows)
    assert rows_all.shape == (world, len(pairs), mdist.ROW_WORDS)
    assert torch.equal(rows_all[:, :, :1024], keys_all) and torch.equal(rows_all[:, :, 1024], counts_all.to(torch.int64))
    assert torch.equal(rows_all[:, :, 1025].sum(0), totals) and torch.equal(rows_all[rank], rows)
    # the exchange partitioned by query (round 3): each rank receives every shard's rows of ITS queries only -- three queries over two
    # ranks = slices of 2 and 1 -- and what it receives is the all-gather's rows of those queries
    recv, first, count = mdist.exchange_rows_partitioned(rows)
    per = (len(pairs) + world - 1) // world
    assert recv.shape == (world, per, mdist.ROW_WORDS) and (first, count) == ((0, 2), (2, 1))[rank]
    assert torch.equal(recv[:, :count], rows_all[:, first:first + count])
    # ... and the two flag words every rank must agree on travel as one all-reduce (max)
    flags = torch.tensor([1 if rank == 1 else 0, 0], dtype=torch.int32)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    assert flags.tolist() == [1, 0]
    if rank == 0:
        q.put((keys_all.numpy().view(np.uint64), counts_all.numpy(), totals.numpy(), gdocs, total))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_exchange_and_merge(orc):
    import torch.multiprocessing as mp
    import manticoresearch_amd as m

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    keys_all, counts_all, totals, gdocs, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert total == 2 * N_DOCS
    # union corpus = shard 0 rows followed by shard 1 rows; the oracle's answer on it is the expected merge
    his = [m.synth_index(N_DOCS, PROBS, seed=99, shard=s, n_threads=1) for s in range(2)]
    ois = [orc.Index(h.spd, h.spp, h.spe, h.dict.view(orc.DICT_DTYPE), N_DOCS, h.skiplist_block_size, 1, 2) for h in his]
    assert [int(x) for x in gdocs] == [int(his[0].dict[t]["docs"]) + int(his[1].dict[t]["docs"]) for t in range(3)]
    for qi, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
        want, tot = [], 0
        for s in range(2):
            r = orc.search(ois[s], orc.op(orc.OP_AND, orc.term(a, 1), orc.term(b, 2)), ranker=orc.RANK_BM25, max_matches=K,
                           total_docs_override=total, local_docs={a: int(gdocs[a]), b: int(gdocs[b])})
            tot += r.total_found
            want += [(-int(w), int(rid) + s * N_DOCS) for rid, w in zip(r.rowid, r.weight)]
        want.sort()
        assert int(totals[qi]) == tot
        merged = np.sort(np.concatenate([keys_all[s, qi, : counts_all[s, qi]] for s in range(2)]))[::-1][:K]
        got = [(-(int(k >> np.uint64(32)) ^ 0x80000000 if (int(k >> np.uint64(32)) ^ 0x80000000) < 2**31 else (int(k >> np.uint64(32)) ^ 0x80000000) - 2**32),
                (~int(k)) & 0xFFFFFFFF) for k in merged]
        assert got == want[:K]
        # ... and the merge of the shards' answers IS the answer of the unsharded corpus: the shards are slices of the one
        # corpus (generator keyed on the global rowid), ranked with its document frequencies
        if qi == 0:
            whole = m.synth_index(2 * N_DOCS, PROBS, seed=99, n_threads=1)
            ow = orc.Index(whole.spd, whole.spp, whole.spe, whole.dict.view(orc.DICT_DTYPE), 2 * N_DOCS, whole.skiplist_block_size, 1, 2)
            assert [int(x) for x in gdocs] == [int(whole.dict[t]["docs"]) for t in range(3)]
        r = orc.search(ow, orc.op(orc.OP_AND, orc.term(a, 1), orc.term(b, 2)), ranker=orc.RANK_BM25, max_matches=K)
        assert r.total_found == tot and [(-int(w), int(rid)) for rid, w in zip(r.rowid, r.weight)] == got

"""Worker of tests/test_gpu_dist.py (own process: torch must load its HIP runtime before libmrk.so does)."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def kw(m, t, pos, mask=0xFFFFFFFF, boost=1.0):
    return m.XQNode.keyword(t, pos, mask, boost)


def main(lib_comm: bool):
    import torch
    import torch.distributed as dist

    import manticoresearch_amd as m
    from manticoresearch_amd import dist as mdist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        probs = [0.3, 0.2, 0.05, 0.01, 0.002]
        n_docs, K = 300000, 1000
        hi = m.synth_index(n_docs, probs, seed=77)
        gdocs, total = mdist.global_df(hi.dict["docs"].astype(np.int64), n_docs, 0)
        assert total == n_docs and (gdocs == hi.dict["docs"]).all()
        qs = [m.Query(m.XQNode.AND(kw(m, a, 1), kw(m, b, 2)), ranker=m.SPH_RANK_BM25, max_matches=K, total_docs=int(total),
                      local_docs={a: int(gdocs[a]), b: int(gdocs[b])}) for a in range(5) for b in range(5) if a != b]
        nq = len(qs)
        ctx = m.Context(0)
        if lib_comm:  # the exchange through the library's own communicator (mrk_comm_init / mrk_shard_exchange)
            mdist.lib_comm_init(ctx)
            g2, t2 = mdist.global_df(hi.dict["docs"].astype(np.int64), n_docs, 0, ctx=ctx)
            assert t2 == n_docs and (g2 == hi.dict["docs"]).all()
        seg = m.Segment(ctx, hi, rowid_base=1000)
        n_sets = 3
        batches = [m.Batch(ctx, nq) for _ in range(n_sets)]
        merger = mdist.ShardMerger(ctx, batches[0], nq, K, 1, 0, n_batches=1, n_sets=n_sets)
        for i in range(n_sets):
            merger.attach([batches[i]], set_index=i)
        cq = m.prepare(qs)
        for rnd in range(3):  # sets reused: the chain must order itself without host waits in between
            for i in range(n_sets):
                merger.wait(i)
                batches[i].submit_prepared(seg, cq, nq)
                merger.merge_attached(1, set_index=i, to_host=True, after_submit=True)
        for i in range(n_sets):
            got = merger.results(i)
            batches[i].wait()
            want = batches[i].results()  # lazy host copy of the batch's own lists
            for q in range(nq):
                docid, weight, tot = got[q]
                assert tot == want[q].total_found and len(docid) == len(want[q].rowid)
                assert (docid == want[q].rowid + 1000).all() and (weight == want[q].weight).all()
        # the synchronous single-batch form
        b = m.Batch(ctx, nq)
        sm = mdist.ShardMerger(ctx, b, nq, K, 1, 0)
        b.submit_prepared(seg, cq, nq)
        b.wait()
        sm.merge()
        got = sm.results(0)
        want = b.results()
        for q in range(nq):
            assert (got[q][0] == want[q].rowid + 1000).all() and got[q][2] == want[q].total_found
        for bb in batches + [b]:
            bb.close()
        seg.close()
        # never a silently partial answer (ShardMerger.finish): 1.3 M docs that all match with one weight overflow the
        # candidate list, the row leaves flagged through the standing export and must come back repaired; a query the
        # shard declines (max_matches beyond the device top-K) must be reported, not merged as "no matches"
        n2 = 1_300_000
        rows = np.arange(n2, dtype=np.uint32)
        W = np.concatenate([np.full(n2, 1, np.uint64), np.full(n2, 2, np.uint64)])
        H = np.concatenate([np.full(n2, (1 << 24) | 1, np.uint32), np.full(n2, (1 << 24) | 2, np.uint32)])
        hi2 = m.index_from_hits(W, np.concatenate([rows, rows]), H, n_terms=2, total_docs=n2, n_fields=2)
        ctx.set("bitmap_inv", 0)
        seg2 = m.Segment(ctx, hi2, rowid_base=500)
        q2 = [m.Query(m.XQNode.AND(kw(m, 0, 1), kw(m, 1, 2)), ranker=m.SPH_RANK_BM25, max_matches=1000),
              m.Query(kw(m, 0, 1), ranker=m.SPH_RANK_BM25, max_matches=20),
              m.Query(kw(m, 1, 1), ranker=m.SPH_RANK_BM25, max_matches=5000)]
        b2 = m.Batch(ctx, len(q2))
        sm2 = mdist.ShardMerger(ctx, b2, len(q2), K, 1, 0)
        sm2.attach([b2])
        b2.submit_prepared(seg2, m.prepare(q2), len(q2))
        sm2.merge_attached(1, set_index=0, to_host=True, after_submit=True)
        sm2.wait(0)
        raw = sm2._merged(0)[:, 1025]
        assert int(raw[0]) >> 63 == 1 and int(raw[1]) >> 63 == 1 and (int(raw[2]) >> 62) & 1 == 1, [hex(int(x)) for x in raw]
        try:
            sm2.results(0)
            raise AssertionError("a declined query must raise")
        except m.MrkError:
            pass
        got2 = sm2.results(0, allow_declined=True)
        assert got2[2] is None
        for q in range(2):
            docid, weight, tot = got2[q]
            assert tot == n2 and list(docid) == [500 + i for i in range(q2[q].max_matches)] and len(set(weight.tolist())) == 1
        b2.close()
        seg2.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(lib_comm="--lib-comm" in sys.argv)
    print("dist chain ok")

"""Multi-GPU plumbing: one process per GPU, segments = contiguous rowid ranges (the reference's
disk-chunk model, sphinxrt.cpp:6018-6108), torch.distributed (RCCL on ROCm) for the one exchange
step the path has: partial top-K lists are all-gathered and merged, totals are all-reduced, and
document frequencies are summed once so that every shard ranks with the same IDF
(local_df: sphinxrt.cpp:6501-6521, sphinxsearch.cpp:4308-4315).

Merge order across shards: weight desc, then GLOBAL docid (rowid_base + rowid) asc -- the
reference compares (weight, local rowid) and leaves cross-chunk ties to arrival order
(sphinxsort.cpp:4541-4547); we fix the order so results are deterministic.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import MRK_MAX_K, check, lib


def global_df(local_docs: np.ndarray, shard_docs: int, device=None):
    """Sum per-term document counts and the document total over all ranks."""
    import torch
    import torch.distributed as dist

    dev = "cpu" if dist.get_backend() == "gloo" else f"cuda:{device}"
    t = torch.tensor(np.concatenate([local_docs.astype(np.int64), [shard_docs]]), dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu().numpy()
    return t[:-1].copy(), int(t[-1])


def exchange_partial_topk(keys, counts, totals):
    """all-gather [nq, MRK_MAX_K] keys and [nq] counts, all-reduce [nq] totals (in place).
    Works on CPU tensors (gloo) and device tensors (nccl = RCCL)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    keys_all = torch.empty((world,) + tuple(keys.shape), dtype=keys.dtype, device=keys.device)
    counts_all = torch.empty((world,) + tuple(counts.shape), dtype=counts.dtype, device=counts.device)
    # list-of-views form: identical on gloo and nccl (RCCL gathers straight into the slices)
    dist.all_gather([keys_all[i] for i in range(world)], keys.contiguous())
    dist.all_gather([counts_all[i] for i in range(world)], counts.contiguous())
    dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    return keys_all, counts_all, totals


class ShardMerger:
    """Device-side merge of the per-shard results of the batch's last submit."""

    def __init__(self, ctx, batch, n_queries: int, k: int, world: int, device: int):
        import torch

        self.torch = torch
        self.ctx, self.batch, self.nq, self.k, self.world = ctx, batch, n_queries, k, world
        dev = f"cuda:{device}"
        self.keys = torch.zeros((n_queries, MRK_MAX_K), dtype=torch.int64, device=dev)
        self.counts = torch.zeros((n_queries,), dtype=torch.int32, device=dev)
        self.totals = torch.zeros((n_queries,), dtype=torch.int64, device=dev)
        self.out_keys = torch.zeros((n_queries, MRK_MAX_K), dtype=torch.int64, device=dev)
        self.out_counts = torch.zeros((n_queries,), dtype=torch.int32, device=dev)

    def merge(self):
        torch = self.torch
        check(lib().mrk_batch_export_device(self.batch._h, self.keys.data_ptr(), self.counts.data_ptr(),
                                            self.totals.data_ptr()))
        keys_all, counts_all, _ = exchange_partial_topk(self.keys, self.counts, self.totals)
        torch.cuda.synchronize()
        check(lib().mrk_topk_merge(self.ctx._h, keys_all.data_ptr(), counts_all.data_ptr(), self.world, self.nq, self.k,
                                   self.out_keys.data_ptr(), self.out_counts.data_ptr()))
        return self.out_keys, self.out_counts, self.totals

    def results(self):
        """Decoded (global docid, weight) lists per query + total_found."""
        ok = self.out_keys.cpu().numpy().view(np.uint64)
        oc = self.out_counts.cpu().numpy()
        tot = self.totals.cpu().numpy()
        out = []
        for q in range(self.nq):
            k = ok[q, : oc[q]]
            weight = ((k >> np.uint64(32)).astype(np.uint32) ^ np.uint32(0x80000000)).view(np.int32)
            docid = ~k.astype(np.uint32)
            out.append((docid, weight, int(tot[q])))
        return out

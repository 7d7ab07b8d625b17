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


def lib_comm_init(ctx):
    """Give the context its own RCCL communicator (mrk_comm_init): rank 0's id travels over the torch.distributed group
    the launcher set up -- any other transport would do, the library itself only needs the 128 bytes.  From here on the
    exchange (ShardMerger) and the document-frequency sums (global_df) are C-ABI calls; torch.distributed only keeps
    bench.py's barrier and timing reduction."""
    import torch.distributed as dist

    box = [ctx.comm_unique_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(box[0], dist.get_world_size(), dist.get_rank())


def global_df(local_docs: np.ndarray, shard_docs: int, device=None, ctx=None):
    """Sum per-term document counts and the document total over all ranks (through the library's communicator when the
    context has one, else torch.distributed)."""
    if ctx is not None and ctx.has_comm:
        t = ctx.comm_allreduce(np.concatenate([local_docs.astype(np.int64), [shard_docs]]))
        return t[:-1].copy(), int(t[-1])
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


ROW_WORDS = MRK_MAX_K + 2  # MRK_ROW_WORDS: keys | count | total_found
ROW_RERUN, ROW_DECLINED = 1 << 63, 1 << 62  # MRK_ROW_RERUN / MRK_ROW_DECLINED: flag bits of the total_found word


def exchange_rows(rows):
    """ONE all-gather of [nq, ROW_WORDS] result rows -> [world, nq, ROW_WORDS] (gloo on CPU tensors, RCCL on device)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    rows_all = torch.empty((world,) + tuple(rows.shape), dtype=rows.dtype, device=rows.device)
    dist.all_gather([rows_all[i] for i in range(world)], rows.contiguous())
    return rows_all


def shard_slice(n_queries: int, world: int, rank: int):
    """(per, first, count) of the queries rank `rank` merges under the partitioned exchange: mrk_shard_slice's arithmetic
    (per = ceil(Q / N); the last ranks may get fewer, or none)."""
    per = (n_queries + world - 1) // world
    first = min(per * rank, n_queries)
    return per, first, min(per, n_queries - first)


def exchange_rows_partitioned(rows):
    """The exchange partitioned by QUERY over torch.distributed (gloo on CPU tensors, RCCL on device tensors): every rank sends each
    owner its rows of the owner's queries and receives every shard's rows of its own -> (recv [world, per, ROW_WORDS], first, count).
    The same all-to-all of row slices mrk_shard_exchange issues through the library's communicator (grouped ncclSend / ncclRecv)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    nq = rows.shape[0]
    per, first, count = shard_slice(nq, world, rank)
    recv = torch.zeros((world, per) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    ops = []
    for p in range(world):
        _, pf, pc = shard_slice(nq, world, p)
        if p == rank:
            recv[p, :count] = rows[first:first + count]
            continue
        if pc:
            ops.append(dist.P2POp(dist.isend, rows[pf:pf + pc].contiguous(), p))
        if count:
            ops.append(dist.P2POp(dist.irecv, recv[p, :count], p))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return recv, first, count


class ShardMerger:
    """Device-side merge of per-shard results.  One row per query carries the partial top-K keys, their count and
    total_found, so a step costs one collective and one merge launch.  attach() gives each batch a standing slice
    of an exchange buffer that its submits fill on their own stream; merge_attached() then queues all-gather ->
    merge -> host copy as a stream-ordered chain (RCCL stream -> event -> the library's merge stream) and returns
    without blocking; wait(set) blocks until that set's merged rows are in host memory.  merge() is the simple
    synchronous form for one batch."""

    def __init__(self, ctx, batch, n_queries: int, k: int, world: int, device: int, n_batches: int = 1, n_sets: int = 1):
        import torch

        assert n_sets <= 8  # MRK_MERGE_SLOTS
        self.torch = torch
        self.ctx, self.batch, self.k, self.world = ctx, batch, k, world
        self.per_batch = n_queries
        self.nq = n_queries * n_batches
        dev = f"cuda:{device}"

        def z(*shape):
            return torch.zeros(shape, dtype=torch.int64, device=dev)

        self.rows = [z(self.nq, ROW_WORDS) for _ in range(n_sets)]
        self.rows_all = [z(world, self.nq, ROW_WORDS) for _ in range(n_sets)]
        self.out_rows = [z(self.nq, ROW_WORDS) for _ in range(n_sets)]
        self.host_rows = [torch.zeros((self.nq, ROW_WORDS), dtype=torch.int64).pin_memory() for _ in range(n_sets)]
        self.gathered = [torch.cuda.Event() for _ in range(n_sets)]
        self.ready = [torch.cuda.Event() for _ in range(n_sets)]
        for ev in self.ready:  # created by their first record; the library re-records them on the batch streams
            ev.record()
        self.attached = [[] for _ in range(n_sets)]
        self.on_host = [False] * n_sets  # where the set's last merge wrote: pinned host rows or device rows
        self.rank = 0
        # the library partitions the merge by query: this rank's merged rows are [first, first + count) only
        self.partitioned = False
        self.first, self.count = 0, self.nq
        self.timing = {} if __import__("os").environ.get("MRK_DIST_TIMING") else None  # host ms per phase, summed

    def attach(self, batches, set_index: int = 0):
        """Every later submit of batches[i] writes its rows into slice i of exchange buffer `set_index`."""
        n = self.per_batch
        assert len(batches) * n <= self.nq
        for i, b in enumerate(batches):
            check(lib().mrk_batch_set_rows_dst(b._h, self.rows[set_index][i * n:].data_ptr()))
        self.attached[set_index] = list(batches)

    def merge(self):
        check(lib().mrk_batch_export_rows(self.batch._h, self.rows[0].data_ptr()))
        self.merge_attached(1, 0)
        self.wait(0)
        return self.out_rows[0][: self.per_batch]


    def merge_attached(self, n_batches: int, set_index: int = 0, to_host: bool = False, after_submit: bool = False):
        """Rows of the attached batches of one set -> all-gather over RCCL -> merge (-> pinned host rows), queued
        back to back; call wait(set_index) before reading or before reusing the set.  after_submit=True: the
        batches were only submitted, not waited for -- the collective is ordered behind their streams by events,
        so the host blocks nowhere (single attached batch per set)."""
        import time
        import torch.distributed as dist

        nq = n_batches * self.per_batch
        assert nq == self.nq, "merge_attached merges the whole attached set"
        t0 = time.perf_counter()
        if self.ctx.has_comm:
            # the library's own chain (mrk_shard_exchange): event on the batch's stream -> RCCL all-gather on the
            # communicator's stream -> merge kernel -> out rows; nothing of torch in it
            if after_submit:
                assert len(self.attached[set_index]) == 1
            bh = self.attached[set_index][0]._h if after_submit else None
            if lib().mrk_shard_partitioned(self.ctx._h):
                import ctypes as C
                import torch.distributed as tdist

                self.partitioned, self.rank = True, (tdist.get_rank() if tdist.is_initialized() else 0)
                f, c = C.c_uint32(), C.c_uint32()
                check(lib().mrk_shard_slice(nq, self.world, self.rank, C.byref(f), C.byref(c)))
                self.first, self.count = f.value, c.value
            check(lib().mrk_shard_exchange(self.ctx._h, bh, self.rows[set_index].data_ptr(), nq, self.k,
                                           (self.host_rows if to_host else self.out_rows)[set_index].data_ptr(), set_index))
            self.on_host[set_index] = to_host
            if self.timing is not None:
                self.timing["exchange"] = self.timing.get("exchange", 0.0) + (time.perf_counter() - t0) * 1e3
                self.timing["calls"] = self.timing.get("calls", 0) + 1
            return
        rows_all = self.rows_all[set_index]
        if after_submit:
            assert len(self.attached[set_index]) == 1
            ready = self.ready[set_index]
            check(lib().mrk_batch_record_event(self.attached[set_index][0]._h, ready.cuda_event))
            self.torch.cuda.current_stream().wait_event(ready)
        # list-of-views form: identical on gloo and nccl (RCCL gathers straight into the slices)
        dist.all_gather([rows_all[i] for i in range(self.world)], self.rows[set_index])
        ev = self.gathered[set_index]
        ev.record()  # on torch's current stream, behind the collective's completion
        t1 = time.perf_counter()
        check(lib().mrk_topk_merge_rows_async(self.ctx._h, rows_all.data_ptr(), self.world, nq, self.k,
                                              (self.host_rows if to_host else self.out_rows)[set_index].data_ptr(),
                                              ev.cuda_event, set_index))
        self.on_host[set_index] = to_host
        t2 = time.perf_counter()
        if self.timing is not None:
            for k_, v in zip(("exchange", "merge"), (t1 - t0, t2 - t1)):
                self.timing[k_] = self.timing.get(k_, 0.0) + v * 1e3
            self.timing["calls"] = self.timing.get("calls", 0) + 1

    def wait(self, set_index: int = 0):
        check(lib().mrk_merge_wait(self.ctx._h, set_index))

    def _merged(self, set_index: int):
        return (self.host_rows[set_index] if self.on_host[set_index] else self.out_rows[set_index].cpu()).numpy().view(np.uint64)

    def finish(self, set_index: int = 0):
        """wait() + never a silently partial answer: a merged row whose total_found word carries MRK_ROW_RERUN (some
        shard's candidate list overflowed and its row left before the rerun) makes EVERY rank -- all of them see the
        same merged rows -- run mrk_batch_wait on its attached batches (the rerun), export the repaired rows, and redo
        the exchange and the merge; a row with MRK_ROW_DECLINED (a shard declined the query) is left flagged for
        results() to report.  Returns the merged rows [nq, ROW_WORDS] as uint64."""
        self.wait(set_index)
        rows = self._merged(set_index)
        if self.partitioned:
            # only [first, first + count) of `rows` is this rank's; whether ANY rank saw a flagged row comes through the
            # all-reduced flag words: every rank reads the same values and takes the same path
            import ctypes as C

            rerun, decl = C.c_uint32(), C.c_uint32()
            check(lib().mrk_shard_flags(self.ctx._h, set_index, C.byref(rerun), C.byref(decl)))
            need_rerun = bool(rerun.value)
        else:
            need_rerun = bool((rows[:, MRK_MAX_K + 1] & np.uint64(ROW_RERUN)).any())
        if not self.attached[set_index] or not need_rerun:
            return rows
        n = self.per_batch
        for i, b in enumerate(self.attached[set_index]):
            check(lib().mrk_batch_wait(b._h))  # reruns this shard's overflowed queries, repairs the device-side lists
            check(lib().mrk_batch_export_rows(b._h, self.rows[set_index][i * n:].data_ptr()))
        self.merge_attached(len(self.attached[set_index]), set_index, to_host=self.on_host[set_index])
        self.wait(set_index)
        rows = self._merged(set_index)
        if self.partitioned:
            rows_mine = rows[self.first:self.first + self.count]
            bad = self.first + np.flatnonzero(rows_mine[:, MRK_MAX_K + 1] & np.uint64(ROW_RERUN))
        else:
            bad = np.flatnonzero(rows[:, MRK_MAX_K + 1] & np.uint64(ROW_RERUN))
        if bad.size:
            raise _lib.MrkError(_lib.MRK_E_UNSUPPORTED, f"queries {bad.tolist()[:8]}: a shard's candidate list overflowed again on the rerun")
        return rows

    def results_slice(self, set_index: int = 0, allow_declined: bool = False):
        """(first query, decoded results) of the queries THIS rank merged (the partitioned exchange; all of them with one rank)."""
        world, self.world = self.world, 1
        try:
            res = self.results(set_index, allow_declined=allow_declined)
        finally:
            self.world = world
        return self.first, res[self.first:self.first + self.count]

    def results(self, set_index: int = 0, nq=None, allow_declined: bool = False):
        """Decoded (global docid, weight) lists per query + total_found.  A query some shard declined raises MrkError
        (allow_declined=True: its entry is None instead)."""
        rows = self.finish(set_index)
        out = []
        if self.partitioned and self.world > 1:
            raise RuntimeError("partitioned exchange: this rank holds the queries of results_slice() only")
        for q in range(nq if nq is not None else self.nq):
            cnt, tot = int(rows[q, MRK_MAX_K]), int(rows[q, MRK_MAX_K + 1])
            if tot & ROW_DECLINED:
                if not allow_declined:
                    raise _lib.MrkError(_lib.MRK_E_UNSUPPORTED, f"query {q}: declined on a shard (MRK_E_UNSUPPORTED there); no merged answer")
                out.append(None)
                continue
            k = rows[q, :cnt]
            weight = ((k >> np.uint64(32)).astype(np.uint32) ^ np.uint32(0x80000000)).view(np.int32)
            docid = ~k.astype(np.uint32)
            out.append((docid, weight, tot))
        return out

"""GPU, one rank over RCCL (backend nccl, world_size 1): the stream-ordered shard-exchange chain of
manticoresearch_amd.dist.ShardMerger -- standing row export on the batch stream -> event -> all-gather -> event ->
merge kernel writing pinned host rows -- must hand back exactly what the batch itself reports.
(Two-list merges are checked in test_gpu_parity.py::test_row_export_and_merge, the collective's semantics with
two ranks over gloo in test_dist_gloo.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("mode", ["lib-comm", "torch"])
def test_stream_ordered_exchange_chain_one_rank(mode):
    """mode lib-comm: the library's own RCCL communicator (mrk_comm_init, mrk_comm_allreduce_i64, mrk_shard_exchange);
    mode torch: the all-gather issued by torch.distributed, the merge by the library."""
    # own process: torch has to load its HIP runtime before libmrk.so pulls in the system one
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(here, "dist_chain_worker.py")] + (["--lib-comm"] if mode == "lib-comm" else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "dist chain ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

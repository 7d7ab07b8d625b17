#!/bin/bash
# tools/trace_dist.sh DOCS -- kernel trace of the sharded bench path with ONE rank (no launcher: env rendezvous)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_dist
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MRK_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py --gpus 1 --docs $1 --steps 6 --warmup 3 --no-cpu-baseline --latency-samples 0 > $OUT/run.log 2>&1
tail -c 300 $OUT/run.log

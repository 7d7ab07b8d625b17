#!/bin/bash
# tools/prof_c3s.sh SHAPE -- kernel stats of one query shape of the config-3 launch (tools/c3_time.py --shape)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_c3s$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/tools/c3_time.py --reps 4 --shape $1 > $OUT/stats.log 2>&1
cut -c1-160 $OUT/stats/s_kernel_stats.csv | head -6

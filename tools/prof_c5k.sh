#!/bin/bash
# tools/prof_c5k.sh KIND -- kernel stats of one query kind (and2 | mix3 | phrase | all) of the config-5 launch (tools/c5_time.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_c5k_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $ROOT/tools/c5_time.py --reps 2 --kind $1 > $OUT/log.txt 2>&1
cut -c1-170 $OUT/s_kernel_stats.csv | head -7
grep "^{" $OUT/log.txt | cut -c1-200

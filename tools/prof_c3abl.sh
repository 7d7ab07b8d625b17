#!/bin/bash
# tools/prof_c3abl.sh SHAPE LIB... -- scan_bt_kernel / rank_kernel times of one config-3 query shape under kernel-experiment libraries
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  OUT=$ROOT/gpurun_out/prof_c3abl/$lib
  mkdir -p $OUT
  MRK_LIB_PATH=$ROOT/manticoresearch_amd/csrc/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $ROOT/tools/c3_time.py --reps 4 --shape $SHAPE > $OUT/log.txt 2>&1
  echo "$lib $(grep -E 'scan_bt|rank_kernel' $OUT/s_kernel_stats.csv | cut -d, -f1,4 | tr '\n' ' ')"
done

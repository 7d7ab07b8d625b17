#!/bin/bash
# tools/ablate.sh DOCS STRATA LIB... -- run the bench once per library variant (MRK_LIB_PATH), print the stratum timings
DOCS=$1; STRATA=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  MRK_LIB_PATH=$ROOT/manticoresearch_amd/csrc/$lib timeout -k 10 300 python3 $ROOT/bench.py --docs $DOCS --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --latency-samples 0 --strata $STRATA > /tmp/ab.log 2>&1 || { echo "$lib FAILED"; tail -3 /tmp/ab.log; exit 1; }
  echo "$lib $(grep -o '"strata": {.*}}, "strata_run"' /tmp/ab.log)"
done

#!/bin/bash
# tools/sweep_items.sh DOCS STRATA BYTES... -- bench once per work-item size
DOCS=$1; STRATA=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for ib in "$@"; do
  timeout -k 10 300 python3 $ROOT/bench.py --docs $DOCS --steps 3 --warmup 1 --no-cpu-baseline --latency-samples 0 --strata $STRATA --item-bytes $ib > /tmp/ab.log 2>&1 || { echo "$ib FAILED"; tail -3 /tmp/ab.log; exit 1; }
  echo "$ib $(grep -o '"value": [0-9.]*' /tmp/ab.log | head -1) $(grep -o '"strata": {.*}}, "strata_run"' /tmp/ab.log)"
done

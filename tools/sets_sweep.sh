#!/bin/bash
# tools/sets_sweep.sh DOCS -- the bench step at DOCS docs with 2 / 3 / 4 steps kept in flight (--sets)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for s in 2 3 4; do
  python $ROOT/bench.py --docs $1 --no-config3 --no-config5 --no-cpu-baseline --latency-samples 0 --sets $s 2>/dev/null > /tmp/sets_$s.json
  python3 - $s <<'PY'
import json, sys
d = json.loads(open("/tmp/sets_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("sets", sys.argv[1], d["value"], d["ms_per_step"], d["host_ms_per_step"])
PY
done

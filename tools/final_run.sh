#!/bin/bash
# tools/final_run.sh -- the end-of-round measurement set on the GPU box: GPU tests, the bench line, BASELINE config 2, and the bench under
# rocprofv3 --kernel-trace --stats (whose per-kernel averages must agree with the bench's HIP-event times).  Output under gpurun_out/.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
python -m pytest tests -x -q -m gpu > $OUT/fin_tests.log 2>&1
python bench.py > $OUT/fin_bench.json 2> $OUT/fin_bench.err
python bench.py --docs 10000000 --no-config3 --no-config5 > $OUT/fin_bench_10M.json 2> $OUT/fin_bench_10M.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fin_prof -o f -- python3 $ROOT/bench.py --no-cpu-baseline --no-config5 > $OUT/fin_prof_bench.json 2> $OUT/fin_prof.err
cd $ROOT
tail -2 $OUT/fin_tests.log
python - <<PY
import json
for f in ("fin_bench", "fin_bench_10M", "fin_prof_bench"):
    d = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms"], d["p50_latency_ms"], (d.get("config3") or {}).get("scan_ms"), d.get("cpu_baseline"))
PY
find $OUT/fin_prof -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-160

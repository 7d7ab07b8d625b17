#!/bin/bash
# tools/final_run2.sh -- the rest of the end-of-round set: one-eighth shard (12.5 M docs) bench lines, unsharded and as ONE rank of the
# sharded path; their kernel timelines; config 3 by kernel (stats + SQ counters + HBM traffic); config 5 by kernel.  Output: gpurun_out/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
python bench.py --docs 12500000 --no-config3 --no-config5 --no-cpu-baseline > $OUT/fin_b12.json 2> $OUT/fin_b12.err; echo "b12 done"
MRK_FORCE_DIST=1 python bench.py --gpus 1 --docs 12500000 --no-cpu-baseline > $OUT/fin_d12.json 2> $OUT/fin_d12.err; echo "d12 done"
bash tools/timeline.sh fin_12M5 12500000 > $OUT/fin_tl12.log 2>&1; echo "tl12 done"
DIST=1 bash tools/timeline.sh fin_12M5_dist 12500000 > $OUT/fin_tl12d.log 2>&1; echo "tl12d done"
bash tools/prof_c3k.sh > $OUT/fin_c3k.log 2>&1; echo "c3k done"
bash tools/traffic_c3.sh fin_traffic_c3 > $OUT/fin_c3_traffic.json 2> $OUT/fin_c3_traffic.err; echo "c3 traffic done"
bash tools/prof_c5k.sh all > $OUT/fin_c5k.log 2>&1; echo "c5k done"
python - <<PY
import json
for f in ("fin_b12", "fin_d12"):
    d = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms"], d["p50_latency_ms"], d.get("host_ms_per_step"))
PY

#!/bin/bash
# tools/prof_c3.sh DOCS -- SQ counters of the config-3 leg (3-term mixes, PROXIMITY_BM25) of bench.py
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_c3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --docs $1 --steps 1 --warmup 0 --no-cpu-baseline --latency-samples 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- python3 $ARGS > $OUT/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections
print(open(glob.glob("$OUT/stats/*kernel_stats.csv")[0]).read()[:1200])
for d in ("pmc1", "pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            if "mrk::" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES", "GRBM_GUI_ACTIVE"): cnt[k] += 1
    for k, v in acc.items():
        print(d, k, "dispatches", cnt[k], {a: round(b / max(cnt[k], 1)) for a, b in v.items()})
PY

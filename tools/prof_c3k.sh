#!/bin/bash
# tools/prof_c3k.sh -- the config-3 launch alone (tools/c3_time.py): kernel stats, then two SQ counter passes, then the four query shapes one by one
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_c3k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/c3_time.py --reps 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- python3 $ARGS > $OUT/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections
print(open(glob.glob("$OUT/stats/*kernel_stats.csv")[0]).read()[:1500])
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
cd $ROOT
for s in 0 1 2 3; do python3 tools/c3_time.py --reps 4 --shape $s 2>&1 | grep "^{" | cut -c1-260; done

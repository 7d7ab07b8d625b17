#!/bin/bash
# tools/timeline.sh TAG DOCS [bench args...] -- rocprofv3 --kernel-trace of one bench configuration (run on the GPU box):
# per-kernel statistics + the dispatch timeline of the last steps (start offset, duration, queue, kernel), so that overlap
# between the scan stream and the selection / exchange streams can be read off.  Output: gpurun_out/tl_TAG/{stats.csv,timeline.txt}
set -e
TAG=$1; DOCS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# DIST=1: the sharded path with ONE rank (env rendezvous: no launcher process between the profiler and the program)
if [ -n "$DIST" ]; then export MRK_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533; EXTRA_ARGS="--gpus 1"; fi
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/raw -o t -- python3 $ROOT/bench.py --docs $DOCS --steps 12 --warmup 3 --no-cpu-baseline --no-config3 --no-config5 --latency-samples 0 $EXTRA_ARGS "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob, collections
out = "$OUT"
for f in glob.glob(out + "/raw/**/*kernel_stats.csv", recursive=True):
    open(out + "/stats.csv", "w").write(open(f).read())
rows = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"]))
for f in glob.glob(out + "/raw/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "dma", r.get("Stream_Id", "?"), r["Direction"].replace("MEMORY_COPY_", "COPY ")))
rows.sort()
tail = rows[-150:]
t0 = tail[0][0] if tail else 0
with open(out + "/timeline.txt", "w") as fo:
    for s, e, q, st, k in tail:
        name = k.split("(")[0].replace("mrk::", "").replace("void ", "")[:48]
        fo.write(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  q{q:>3} s{st:>3}  {name}\n")
print(open(out + "/timeline.txt").read()[-6000:])
PY

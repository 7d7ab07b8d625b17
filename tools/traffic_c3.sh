#!/bin/bash
# tools/traffic_c3.sh [TAG] -- HBM bytes of the config-3 launch (3-term mixes, PROXIMITY_BM25, 256 queries, 100 M docs) by kernel:
# FETCH_SIZE and WRITE_SIZE in separate passes (no trace domains), corrected with the same calibration kernels as tools/traffic.sh
# (every access of these kernels is a 4-byte-per-lane load: c_gather4 for the scattered words, c_stream4 for the bitmap rows --
# the two differ by a few per cent, both are printed).  Next to them: the launch's algorithmic bytes (reference format) and the
# device format's bytes the planner counts for the same queries.
set -e
TAG=${1:-traffic_c3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/c3_time.py --reps 4"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/calib -o c -- $ROOT/tools/calib/fetch_calib 4 3 > $OUT/calib.json 2> $OUT/calib.log
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o w -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ARGS > $OUT/stats.log 2>&1
python3 - <<PY
import csv, json, glob, re, sys
sys.path.insert(0, "$ROOT")
import bench
def per_kernel(d, name):
    v = {}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = (re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0], r["Dispatch_Id"])
                v[k] = v.get(k, 0.0) + float(r["Counter_Value"]) * 1024
    out = {}
    for (n, d_), x in v.items(): out.setdefault(n, []).append(x)
    return out
def mean(x): return sum(x) / max(1, len(x))
known = json.loads([l for l in open("$OUT/calib.json") if l.startswith("{")][-1])
cal = per_kernel("calib", "FETCH_SIZE")
c_stream4 = mean(cal["calib_stream4"]) / known["calib_stream4"]["bytes"]
dense = []
for name, kn in known.items():
    m = re.search(r"calib_gather4<(\d+)>", name)
    if m and int(m.group(1)) <= 16:
        raw = [v for k, v in cal.items() if k.replace(" ", "") == name.replace(" ", "")]
        if raw: dense.append(mean(raw[0]) / (kn["lines128"] * 128))
c_gather = mean(dense)
fe, wr = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
times = {}
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        times[re.sub(r"^void ", "", r["Name"]).split("(")[0]] = float(r["AverageNs"]) / 1e6
line = json.loads([l for l in open("$OUT/stats.log") if l.startswith("{")][-1])
kernels = {}
tot = 0.0
for n in sorted(fe, key=lambda k: -mean(fe[k])):
    if not n.startswith("mrk::"): continue
    # (the first launches of the run are warm-up launches of the same batch: every dispatch of a kernel does the same work)
    f, w = mean(fe[n]), mean(wr.get(n, [0.0]))
    e = {"FETCH_SIZE_bytes_raw": int(f), "WRITE_SIZE_bytes_raw": int(w), "read_bytes_corrected": int(f / c_gather), "traffic_bytes": int(f / c_gather + w), "dispatches": len(fe[n])}
    if n in times:
        e["avg_ms"] = round(times[n], 4); e["traffic_GBps"] = round((f / c_gather + w) / times[n] / 1e6, 1)
    kernels[n] = e; tot += f / c_gather + w
out = {"docs": 100000000, "queries": 256, "workload": "config 3: a b c | (a|b) c | a (b|c) | a b -c, SPH_RANK_PROXIMITY_BM25, top-1000 (tools/c3_time.py)",
       "kernel_sources_sha": bench.kernel_sources_sha(), "c_stream4": c_stream4, "c_gather4_dense": c_gather, "kernels": kernels,
       "traffic_bytes_per_launch": int(tot), "algorithmic_bytes_reference_format": int(line["algo_MB"] * 1e6), "device_format_bytes_planned": int(line["dev_MB"] * 1e6),
       "traffic_over_algorithmic": round(tot / (line["algo_MB"] * 1e6), 3), "scan_ms_by_hip_events": line["scan_ms"],
       "correction": "read bytes = FETCH_SIZE x 1024 / c_gather4_dense (4 B/lane accesses; with c_stream4 instead the figure moves by the ratio of the two)",
       "command": "tools/traffic_c3.sh (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, --kernel-trace --stats: three separate runs of tools/c3_time.py --reps 4)"}
print(json.dumps(out, indent=1))
PY

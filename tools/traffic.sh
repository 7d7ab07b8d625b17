#!/bin/bash
# tools/traffic.sh DOCS -- HBM bytes of the cc-stratum scan launch from the PMC counters (run on the GPU box):
# FETCH_SIZE and WRITE_SIZE in separate passes, no trace domains; prints the JSON for profiles/traffic.json
set -e
DOCS=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --docs $DOCS --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --latency-samples 0 --strata cc"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o w -- python3 $ARGS > $OUT/write.log 2>&1
python3 - <<PY
import csv, json, glob
def launches(d, name):
    v = {}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "scan_" in r["Kernel_Name"]:
                k = (r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])
                v[k] = v.get(k, 0.0) + float(r["Counter_Value"])
    return v
fe, wr = launches("fetch", "FETCH_SIZE"), launches("write", "WRITE_SIZE")
names = sorted({k[0] for k in fe})
out = {"docs": $DOCS, "queries": 256, "skiplist_block": 128, "launches": {}}
for n in names:
    f = sorted(v for k, v in fe.items() if k[0] == n)
    w = sorted(v for k, v in wr.items() if k[0] == n)
    out["launches"][n] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w}
big = max(names, key=lambda n: max(v for k, v in fe.items() if k[0] == n))
f = [v for k, v in fe.items() if k[0] == big]; w = [v for k, v in wr.items() if k[0] == big]
fm, wm = sum(f) / len(f), (sum(w) / len(w) if w else 0.0)
out.update({"kernel": big, "FETCH_SIZE_KB_raw": fm, "WRITE_SIZE_KB_raw": wm,
            "traffic_bytes_per_launch": int(fm * 1024 * 2 + wm * 1024)})
print(json.dumps(out, indent=1))
PY

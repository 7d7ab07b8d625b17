#!/bin/bash
# tools/traffic.sh DOCS [TAG] -- HBM bytes of the cc-stratum scan launch from the PMC counters, CALIBRATED for the kernel's
# own access patterns (run on the GPU box).  Separate `--pmc` passes, no trace domains (MI355X_MICROARCH.md, HBM section).
#
#   1. tools/calib/fetch_calib under FETCH_SIZE: kernels with exactly known byte / line counts in scan_bm_kernel's two
#      access patterns -- 4 B/lane coalesced 256-B rows (the bitmap windows) and 4-B gathers at densities 1/2 .. 1/64
#      (the tf / field words gathered by rank) -- plus the 16 B/lane stream the guide calibrated.  Gives
#      c_stream4 = FETCH_SIZE / bytes read and, per density, FETCH_SIZE / (64-B lines touched x 64).
#   2. bench.py --strata cc under FETCH_SIZE and under WRITE_SIZE (the product library).
#   3. the same under FETCH_SIZE with libmrk_noscore.so (scan_bm_kernel without its scoring step = without the
#      gathers): FETCH_SIZE of the bitmap streams alone.
#   traffic = FETCH(streams) / c_stream4 + (FETCH(all) - FETCH(streams)) / c_gather + WRITE_SIZE, per launch.
# Prints the JSON for profiles/rNN_traffic.json (bench.py reads profiles/traffic.json).
set -e
DOCS=$1; TAG=${2:-traffic}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --docs $DOCS --steps 4 --warmup 1 --no-cpu-baseline --no-config3 --no-config5 --latency-samples 0 --strata cc"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/calib -o c -- $ROOT/tools/calib/fetch_calib 4 3 > $OUT/calib.json 2> $OUT/calib.log
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o w -- python3 $ARGS > $OUT/write.log 2>&1
export MRK_LIB_PATH=$ROOT/manticoresearch_amd/csrc/libmrk_noscore.so
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch_noscore -o f -- python3 $ARGS > $OUT/fetch_noscore.log 2>&1
unset MRK_LIB_PATH
# the selective strata (skip-assisted: galloping reads less than the algorithmic bytes) -- SURVEY 8(d) wants their measured bytes
for S in sc ss; do
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch_$S -o f -- python3 ${ARGS/--strata cc/--strata $S} > $OUT/fetch_$S.log 2>&1
done
python3 - <<PY
import csv, json, glob, re, sys
sys.path.insert(0, "$ROOT")
import bench
def launches(d, name, pat):
    v = {}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and pat in r["Kernel_Name"]:
                k = (re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0], r["Dispatch_Id"])
                v[k] = v.get(k, 0.0) + float(r["Counter_Value"])
    return v
def mean(x): return sum(x) / max(1, len(x))
known = json.loads([l for l in open("$OUT/calib.json") if l.startswith("{")][-1])
cal = launches("calib", "FETCH_SIZE", "calib_")
calib = {}
for name, kn in known.items():
    if not isinstance(kn, dict): continue
    raw = [v * 1024 for k, v in cal.items() if k[0].replace(" ", "") == name.replace(" ", "")]
    if not raw: continue
    e = {"FETCH_SIZE_bytes_raw": mean(raw), "launches": len(raw)}
    if "bytes" in kn:
        e["bytes_read"] = kn["bytes"]; e["ratio_to_bytes_read"] = mean(raw) / kn["bytes"]
    else:
        e.update(words_read=kn["words_read"], lines64=kn["lines64"], lines128=kn["lines128"],
                 ratio_to_64B_lines=mean(raw) / (kn["lines64"] * 64), ratio_to_128B_lines=mean(raw) / (kn["lines128"] * 128))
    calib[name] = e
c_stream4 = calib["calib_stream4"]["ratio_to_bytes_read"]
# the gathers of the cc launch touch (nearly) every 128-B line of the tf / field words: the dense calibration points apply
dense = [calib[k]["ratio_to_128B_lines"] for k in calib if k.startswith("calib_gather4") and int(re.search(r"<(\d+)>", k).group(1)) <= 16]
c_gather = mean(dense)
fe, wr, fs = launches("fetch", "FETCH_SIZE", "scan_"), launches("write", "WRITE_SIZE", "scan_"), launches("fetch_noscore", "FETCH_SIZE", "scan_")
names = sorted({k[0] for k in fe})
big = max(names, key=lambda n: max(v for k, v in fe.items() if k[0] == n))
f = [v * 1024 for k, v in fe.items() if k[0] == big]; w = [v * 1024 for k, v in wr.items() if k[0] == big]
s = [v * 1024 for k, v in fs.items() if k[0] == big]
fm, wm, sm = mean(f), mean(w), mean(s)
traffic = sm / c_stream4 + max(0.0, fm - sm) / c_gather + wm
strata = {}
for S in ("sc", "ss"):
    v = launches("fetch_" + S, "FETCH_SIZE", "scan_")
    per = {}
    for (n, d), x in v.items(): per.setdefault(n, []).append(x * 1024)
    # the stratum's launch is the kernel with the most bytes (the cc reference launch bench.py adds at the end is scan_bm)
    cand = {n: x for n, x in per.items() if "scan_bm" not in n} or per
    n = max(cand, key=lambda k: mean(cand[k]))
    strata[S] = {"kernel": n, "FETCH_SIZE_bytes_raw": mean(cand[n]), "launches": len(cand[n]),
                 "measured_bytes_per_launch": int(mean(cand[n]) / c_gather), "correction": "FETCH_SIZE / c_gather4_dense (block words and probes are 4 B/lane accesses)"}
out = {"docs": $DOCS, "queries": 256, "skiplist_block": 128, "kernel": big, "kernel_sources_sha": bench.kernel_sources_sha(), "kernel_tag": "bm" if "scan_bm" in big else "pk",
       "FETCH_SIZE_bytes_raw": fm, "WRITE_SIZE_bytes_raw": wm, "FETCH_SIZE_bytes_raw_streams_only": sm,
       "launches": {"fetch": len(f), "write": len(w), "fetch_streams_only": len(s)},
       "strata": strata, "calibration": calib, "c_stream4": c_stream4, "c_gather4_dense": c_gather,
       "traffic_bytes_per_launch": int(traffic),
       "traffic_formula": "FETCH(streams only) / c_stream4 + (FETCH(all) - FETCH(streams only)) / c_gather4_dense + WRITE_SIZE",
       "traffic_bytes_per_launch_uncalibrated_x2": int(fm * 2 + wm),
       "command": "tools/traffic.sh $DOCS (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, no trace domains; bench.py --strata cc --steps 3 --warmup 1; libmrk_noscore.so = -DMRK_BMEXP=2 for the streams-only pass)"}
print(json.dumps(out, indent=1))
PY

#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), no tracing options alongside.
# gfx950 correction: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2.
TAG=${1:-pmc}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --ramp-ms 0 --no-cpu-baseline > $OUT/$C.log 2>&1
  echo "$C pass exit $?"
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
vals = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_pow2" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals[c].append(float(r["Counter_Value"]))
fs = sum(vals["FETCH_SIZE"]) / max(len(vals["FETCH_SIZE"]), 1)
ws = sum(vals["WRITE_SIZE"]) / max(len(vals["WRITE_SIZE"]), 1)
traffic = int((2 * fs + ws) * 1024)
summary = {"kernel": "k_pow2<double,4096> (bench.py workload 4096x4096 c128)", "launches_sampled": len(vals["FETCH_SIZE"]),
           "FETCH_SIZE_KB_per_launch": fs, "WRITE_SIZE_KB_per_launch": ws,
           "read_bytes_corrected_x2": int(2 * fs * 1024), "write_bytes": int(ws * 1024),
           "hbm_traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 536870912,
           "traffic_over_algorithmic": round(traffic / 536870912, 4)}
json.dump(summary, open(f"{out}/pmc_summary.json", "w"), indent=1)
json.dump({"4096x4096": traffic}, open(f"{out}/pmc_traffic.json", "w"))
print(json.dumps(summary, indent=1))
PY

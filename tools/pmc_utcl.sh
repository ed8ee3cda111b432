#!/bin/bash
# Address-translation (UTCL1) counters of the cfg3-A stage kernels against the contiguous-row headline kernel: is the per-tile page spread
# (128 rows 2 MiB apart per stage-A tile) what the stages wait for?  Separate --pmc passes, nothing else traced.
# Usage (GPU box): bash tools/pmc_utcl.sh <tag>
TAG=${1:-pmc_utcl}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
         "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
         "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  for W in cfg3A_only cfg2; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/p${i}_$W -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --steps 3 --ramp-ms 0 --only $W > $OUT/p${i}_$W.log 2>&1
    echo "pass $i $W exit $?"
  done
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ndfft" not in k:
            continue
        name = re.sub(r"\(.*", "", k.replace("void ndfft::", ""))[:110] + " grid=" + str(r.get("Grid_Size"))
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in acc.items():
    d = {c: sum(x) / len(x) for c, x in v.items()}
    d["launches"] = max(len(x) for x in v.values())
    if d.get("TCP_UTCL1_REQUEST_sum"):
        d["utcl1_miss_per_request"] = round(d.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / d["TCP_UTCL1_REQUEST_sum"], 5)
    res[k] = d
json.dump(res, open(f"{out}/pmc_utcl_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY

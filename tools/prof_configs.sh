#!/bin/bash
# Per-kernel evidence for the non-headline configs: rocprofv3 kernel stats and (separate passes) HBM byte
# counters for cfg3 / cfg4 / cfg5.  Usage (on the GPU box): bash tools/prof_configs.sh <tag>
TAG=${1:-prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for W in cfg3 cfg4 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only $W --steps 30 --no-reread > $OUT/stats_$W.log 2>&1
  find $OUT/stats_$W -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/${W}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only $W --steps 3 --ramp-ms 0 --no-reread > $OUT/pmc_${W}_$C.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
res = {}
for w in ("cfg3", "cfg4", "cfg5"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/pmc_{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "ndfft" in r["Kernel_Name"]:
                    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ndfft::", ""))[:110]
                    vals[name][c].append(float(r["Counter_Value"]))
    res[w] = {k: {"launches": len(v["FETCH_SIZE"]),
                  "read_MB_per_launch_x2": round(2 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1) / 1024, 1),
                  "write_MB_per_launch": round(sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1) / 1024, 1)} for k, v in vals.items()}
json.dump(res, open(f"{out}/pmc_configs_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY

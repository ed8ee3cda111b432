#!/bin/bash
# rocprofv3 evidence for bench.py's three timed regions, one region per profiler run so that the per-kernel averages
# belong to it: kernel-trace stats, then (separate passes, as MI355X_MICROARCH.md prescribes) FETCH_SIZE / WRITE_SIZE.
# Usage (on the GPU box): bash tools/prof_bench.sh <tag> [pmc]
TAG=${1:-prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for PH in primary warm strong; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$PH -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --profile-phase $PH > $OUT/stats_$PH.log 2>&1
  echo "stats $PH exit $?"
  find $OUT/stats_$PH -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/${PH}_kernel_stats.csv
  head -4 $OUT/${PH}_kernel_stats.csv | cut -c1-260
  if [ "$2" == "pmc" ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${PH}_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --ramp-ms 0 --profile-phase $PH > $OUT/pmc_${PH}_$C.log 2>&1
      echo "pmc $PH $C exit $?"
    done
  fi
done
cd $GRAFT_REPO_ROOT
if [ "$2" == "pmc" ]; then
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
res = {}
for ph in ("primary", "warm", "strong"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/pmc_{ph}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "k_pow2" in r["Kernel_Name"]:
                    g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
                    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ndfft::", ""))[:100] + f" grid={g}"
                    vals[name][c].append(float(r["Counter_Value"]))
    res[ph] = {k: {"launches": len(v["FETCH_SIZE"]),
                   "read_bytes_x2_corrected": int(2 * 1024 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)),
                   "write_bytes": int(1024 * sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1))} for k, v in vals.items()}
    for k, v in res[ph].items():
        v["hbm_traffic_bytes_per_launch"] = v["read_bytes_x2_corrected"] + v["write_bytes"]
json.dump(res, open(f"{out}/pmc_bench_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:5000])
# the file bench.py quotes `roofline.traffic` from, stamped with the hash of the kernel text these counters belong to
import os, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", ".")
def pick(ph, grid_lanes):
    best = None
    for k, v in res.get(ph, {}).items():
        if "Pow2Kernel<double, 4096" in k and v["launches"]:
            best = v["hbm_traffic_bytes_per_launch"]
    return best
tag = os.path.basename(os.path.dirname(out)) if os.path.basename(out) == "prof_bench" else os.path.basename(out)
tj = {"4096x4096": pick("warm", 4096), "4096x4096_cold_rotating": pick("primary", 4096), "65536x4096": pick("strong", 65536),
      "kernel_source_sha": subprocess.check_output(["python3", os.path.join(root, "tools", "pmc_source_sha.py")], text=True).strip(),
      "source": f"profiles/r09/{tag}_pmc_bench_summary.json (tools/prof_bench.sh {tag} pmc: separate FETCH_SIZE / WRITE_SIZE --pmc passes of `bench.py --profile-phase ...`, "
                "x2 gfx950 read correction on FETCH_SIZE; kernel k_pow2<Pow2Kernel<double,4096,512,...>>)"}
json.dump(tj, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(tj, indent=1))
PY
fi

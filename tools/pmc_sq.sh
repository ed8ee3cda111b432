#!/bin/bash
# SQ instruction / LDS counters of the bench kernel (separate --pmc passes, nothing else traced).
# Usage (GPU box): bash tools/pmc_sq.sh <tag>
TAG=${1:-pmc_sq}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --ramp-ms 0 --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i ($C) exit $?"
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pow2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in acc.items()}
res["kernel"] = "k_pow2<double,4096,512x8,r8^4,half,nt-stores,TW_POWERS> (bench.py workload, per launch)"
if "SQ_LDS_BANK_CONFLICT" in res and res.get("SQ_LDS_IDX_ACTIVE"):
    res["lds_bank_conflict_fraction"] = round(res["SQ_LDS_BANK_CONFLICT"] / res["SQ_LDS_IDX_ACTIVE"], 4)
if "SQ_INSTS_VALU" in res and res.get("SQ_WAVES"):
    res["valu_insts_per_wave"] = round(res["SQ_INSTS_VALU"] / res["SQ_WAVES"], 1)
    res["lds_insts_per_wave"] = round(res.get("SQ_INSTS_LDS", 0) / res["SQ_WAVES"], 1)
json.dump(res, open(f"{out}/pmc_sq_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY

#!/bin/bash
# SQ counters of the short-lane kernels, with the wavefront (cross-lane shuffle) kernel and with the kernels it
# replaces (NDFFT_WAVE=0): LDS and VALU instructions per wave.  Usage (GPU box): bash tools/pmc_wave.sh <tag>
TAG=${1:-pmcwave}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for W in 1 0; do
  NDFFT_WAVE=$W timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
     --output-format csv -d $OUT/wave$W -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only smallsweep --steps 3 --ramp-ms 0 > $OUT/wave$W.log 2>&1
  echo "pmc NDFFT_WAVE=$W exit $?"
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
res = {}
for w in ("1", "0"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for f in glob.glob(f"{out}/wave{w}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ndfft" not in r["Kernel_Name"]: continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ndfft::", ""))[:90]
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": cnt[name] += 1
    res["NDFFT_WAVE=" + w] = {k: {"launches": cnt[k], "LDS_insts_per_wave": round(v["SQ_INSTS_LDS"] / max(v["SQ_WAVES"], 1), 2),
                                  "VALU_insts_per_wave": round(v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), 1),
                                  "SALU_insts_per_wave": round(v["SQ_INSTS_SALU"] / max(v["SQ_WAVES"], 1), 1),
                                  "VMEM_insts_per_wave": round((v["SQ_INSTS_VMEM_RD"] + v["SQ_INSTS_VMEM_WR"]) / max(v["SQ_WAVES"], 1), 1),
                                  "lds_bank_conflict_frac": round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), 3)} for k, v in acc.items()}
json.dump(res, open(f"{out}/pmc_wave_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY

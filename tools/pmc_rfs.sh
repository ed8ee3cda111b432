#!/bin/bash
# SQ counters of the real four-step's kernels (ndfft_r2c / nddct2 64 x 262144 f64 via tools/probes/rfs_prof.py), separate --pmc passes.  Usage (GPU box): bash tools/pmc_rfs.sh <tag>
TAG=${1:-pmc_rfs}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/probes/rfs_prof.py ndfft_r2c nddct2 > $OUT/p$i.log 2>&1
  echo "pass $i ($C) exit $?"
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ndfft::" in k and "double" in k:
            name = re.sub(r"\(.*", "", k.replace("void ndfft::", "").replace("ndfft::", ""))[:110]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in acc.items():
    d = {c: sum(x) / len(x) for c, x in v.items()}
    if d.get("SQ_LDS_IDX_ACTIVE"): d["lds_bank_conflict_fraction"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    if d.get("SQ_WAVES"):
        d["valu_insts_per_wave"] = round(d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"], 1); d["lds_insts_per_wave"] = round(d.get("SQ_INSTS_LDS", 0) / d["SQ_WAVES"], 1)
        d["vmem_insts_per_wave"] = round((d.get("SQ_INSTS_VMEM_RD", 0) + d.get("SQ_INSTS_VMEM_WR", 0)) / d["SQ_WAVES"], 1)
    if d.get("SQ_WAVE_CYCLES"):
        d["wait_any_fraction_of_wave_cycles"] = round(d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"], 3); d["wait_inst_any_fraction"] = round(d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"], 3)
        d["wait_inst_lds_fraction"] = round(d.get("SQ_WAIT_INST_LDS", 0) / d["SQ_WAVE_CYCLES"], 3); d["valu_active_fraction"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_WAVE_CYCLES"], 3)
    res[k] = d
json.dump(res, open(f"{out}/pmc_rfs_summary.json", "w"), indent=1)
for k, d in res.items(): print(k, {c: d[c] for c in d if c[0] != "S"})
PY

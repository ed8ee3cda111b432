#!/bin/bash
# SQ counters (instruction mix, busy / wait cycles, LDS conflicts) of EVERY kernel a tools/bench_configs.py selection launches,
# one --pmc pass per counter group, nothing else traced (gpurun refuses --pmc together with the trace domains).
# Usage (GPU box): bash tools/pmc_sq_row.sh <tag> "<bench_configs args>"        e.g.  pmc_sq_row.sh r09a_sq "--only cfg3 --steps 3 --no-reread"
TAG=${1:-pmc_sq_row}; ARGS=${2:---only cfg3 --steps 3 --no-reread}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py $ARGS --ramp-ms 0 > $OUT/p$i.log 2>&1
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
        if "ndfft" not in k and "k_jit" not in k:      # (hiprtc kernels are all called k_jit: grid / workgroup / VGPRs tell them apart)
            continue
        key = re.sub(r"\s+", "", k)[:230] + " grid=" + r["Grid_Size"] + " wg=" + r.get("Workgroup_Size", "?") + " vgpr=" + r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")) + " lds=" + r.get("LDS_Block_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, v in acc.items():
    d = {c: sum(x) / len(x) for c, x in v.items()}
    d["launches_seen"] = max(len(x) for x in v.values())
    w = d.get("SQ_WAVES")
    if w:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"):
            if c in d:
                d[c.replace("SQ_INSTS_", "").lower() + "_per_wave"] = round(d[c] / w, 1)
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
            if c in d:
                d[c.replace("SQ_", "").lower() + "_share_of_wave_cycles"] = round(d[c] / wc, 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    d["kernel"] = k
    rows.append(d)
json.dump(rows, open(f"{out}/pmc_sq_rows.json", "w"), indent=1)
for d in rows:
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items()}))
PY

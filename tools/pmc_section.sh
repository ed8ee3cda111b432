#!/bin/bash
# SQ counters of the kernels (name filter) that serve one section of tools/bench_configs.py (two --pmc passes: instruction mix, then where the wave cycles go).
# Usage (GPU box): bash tools/pmc_section.sh <tag> <section> <kernel name substring>
TAG=${1:-pmcsec}; SEC=${2:-radersweep}; FILT=${3:-k_jit}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
   --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only $SEC --steps 3 --ramp-ms 0 > $OUT/p1.log 2>&1
echo "pass 1 exit $?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
   --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only $SEC --steps 3 --ramp-ms 0 > $OUT/p2.log 2>&1
echo "pass 2 exit $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only $SEC --steps 10 --ramp-ms 0 > $OUT/kt.log 2>&1
echo "trace exit $?"
cd $GRAFT_REPO_ROOT
python3 - "$OUT" "$FILT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]; FILT = sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for ps in ("p1", "p2"):
    for f in glob.glob(f"{out}/{ps}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if FILT not in r["Kernel_Name"]: continue
            key = f"grid{r['Grid_Size']}_wg{r['Workgroup_Size']}_lds{r['LDS_Block_Size']}"
            acc[key][ps + ":" + r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][ps + ":" + r["Counter_Name"]] += 1
res = {}
for k, v in acc.items():
    g = lambda n: v.get(n, 0.0) / max(cnt[k].get(n, 1), 1)
    w1, w2 = max(g("p1:SQ_WAVES"), 1), max(g("p2:SQ_WAVES"), 1)
    res[k] = {"waves": w1, "VALU/wave": round(g("p1:SQ_INSTS_VALU") / w1, 1), "LDS/wave": round(g("p1:SQ_INSTS_LDS") / w1, 1), "SALU/wave": round(g("p1:SQ_INSTS_SALU") / w1, 1),
              "VMEM/wave": round((g("p1:SQ_INSTS_VMEM_RD") + g("p1:SQ_INSTS_VMEM_WR")) / w1, 1), "lds_conflict_frac": round(g("p1:SQ_LDS_BANK_CONFLICT") / max(g("p1:SQ_LDS_IDX_ACTIVE"), 1), 3),
              "wave_cycles/wave(quad)": round(g("p2:SQ_WAVE_CYCLES") / w2, 0), "wait_any": round(g("p2:SQ_WAIT_ANY") / max(g("p2:SQ_WAVE_CYCLES"), 1), 3),
              "wait_inst_any": round(g("p2:SQ_WAIT_INST_ANY") / max(g("p2:SQ_WAVE_CYCLES"), 1), 3), "active_inst_any": round(g("p2:SQ_ACTIVE_INST_ANY") / max(g("p2:SQ_WAVE_CYCLES"), 1), 3),
              "active_valu": round(g("p2:SQ_ACTIVE_INST_VALU") / max(g("p2:SQ_WAVE_CYCLES"), 1), 3), "active_lds": round(g("p2:SQ_ACTIVE_INST_LDS") / max(g("p2:SQ_WAVE_CYCLES"), 1), 3),
              "busy_cycles": g("p2:SQ_BUSY_CYCLES")}
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:8000])
PY
for f in $(find $OUT/kt -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; head -12 $f | cut -c1-200; done

#!/bin/bash
# per-kernel durations of one workload with two builds of the library: bash tools/probes/ab_prof.sh <only> <libA> <libB>
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/abprof; mkdir -p $OUT
cd /tmp
for L in $2 $3; do
  tag=$(basename $L .so)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/probes/ab_lib.py $GRAFT_REPO_ROOT/$L -- --only $1 --steps 30 > $OUT/$tag.log 2>&1
  echo "== $L"; grep -h "workload" $OUT/$tag.log | cut -c1-150
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ndfft" in r["Name"]:
        print(f'  calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][12:150]}')
PY
done

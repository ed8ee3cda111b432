#!/bin/bash
# Per-kernel durations of the column four-step (cfg3-A) at different chunk sizes: does stage B get faster when its
# input (stage A's output) is small enough to still be in the Infinity Cache?  Usage: bash tools/cs_chunk_prof.sh <tag>
TAG=${1:-cschunk}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for CH in 144 64 32 16; do
  NDFFT_CS_CHUNK_MB=$CH timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ch$CH -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only cfg3A_only --steps 20 > $OUT/ch$CH.log 2>&1
  f=$(find $OUT/ch$CH -name "*kernel_stats.csv" | head -1)
  echo "== chunk $CH MiB"; grep -h "cfg3A" $OUT/ch$CH.log | cut -c1-160
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "RealPow2Kernel" in r["Name"]:
        nm = r["Name"].split("RealPow2Kernel<")[1][:60]
        print(f'  {nm:62s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.1f} us total {float(r["TotalDurationNs"])/1e6:8.1f} ms')
PY
done

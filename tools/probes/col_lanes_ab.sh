#!/bin/bash
# f64 column tiles: 32 lanes (256-byte rows for f64, 512 for c128; up to 140 KiB = one workgroup per CU) vs 16 lanes, side build in tools/_ab
fmt='import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if "col" in r.get("path","") or "fft2" in r["workload"]: print(r["workload"][:70], r["us"], r.get("frac_of_8TBs"), r.get("path",""))'
for rep in 1 2; do
  for g in cfg4 fft2d generic; do
    echo "== default build ($g)"; python tools/bench_configs.py --only $g 2>/dev/null | python -c "$fmt"
    echo "== 16-lane f64 column tiles ($g)"; NDFFT_MI355X_LIB=$PWD/tools/_ab/libndfft_col16.so python tools/bench_configs.py --only $g 2>/dev/null | python -c "$fmt"
  done
done

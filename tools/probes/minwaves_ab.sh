#!/bin/bash
# cfg3 / cfg4 column kernels with a floor on waves per SIMD (= VGPR cap): variant libraries in tools/_ab
for rep in 1 2; do
  echo "== default build"; python tools/bench_configs.py --only cfg3 2>/dev/null | python -c "import sys,json; [print(r['workload'][:60], r['us'], r['frac_of_8TBs']) for r in map(json.loads, sys.stdin)]"
  for mw in 6 8; do echo "== min waves per SIMD $mw"; NDFFT_MI355X_LIB=$PWD/tools/_ab/libndfft_mw$mw.so python tools/bench_configs.py --only cfg3 2>/dev/null | python -c "import sys,json; [print(r['workload'][:60], r['us'], r['frac_of_8TBs']) for r in map(json.loads, sys.stdin)]"; done
done

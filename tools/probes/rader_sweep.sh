#!/bin/bash
# sweeps the developer knobs of the Rader kernel (jit.hip: NDFFT_RADER_EMAX, NDFFT_RADER_LPB / NDFFT_RADER_THREADS)
out=${1:-gpurun_out/rader_sweep.txt}
: > $out
for emax in 0 8 12 16 24; do
  for thr in 128 256 512; do
    echo "== EMAX=$emax THREADS=$thr" >> $out
    NDFFT_RADER_EMAX=$emax NDFFT_RADER_THREADS=$thr timeout 300 python tools/bench_configs.py --only radersweep --steps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(f\"{d['workload']:34s} {d['us']:8.1f} {d['frac_of_8TBs']:.3f} {d['path']}\")
" >> $out
  done
done

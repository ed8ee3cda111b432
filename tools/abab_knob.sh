#!/bin/bash
# A-B-A-B of ONE developer knob on the developer build (GPU box): tools/abab_knob.sh "<bench_configs args>" KNOB v1 v2 [...]   (three rounds, alternating, one process each)
ARGS=$1; KNOB=$2; shift 2
for round in 1 2 3; do
  for v in "$@"; do
    env NDFFT_MI355X_LIB=$PWD/ndrustfft_amd/csrc/libndfft_mi355x_dev.so $KNOB=$v python tools/bench_configs.py $ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-22s round $round  %8.2f us (re-read %8.2f)  %-14s %s' % ('$KNOB=$v', r['us'], r.get('us_reread', 0), r.get('path'), r['workload']))"
  done
done

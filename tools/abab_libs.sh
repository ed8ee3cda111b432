#!/bin/bash
# A-B-A-B of builds of the library on one box (GPU box): tools/abab_libs.sh "<bench_configs args>" name1=lib1.so name2=lib2.so ...
# (the product build is `product=ndrustfft_amd/csrc/libndfft_mi355x.so`); every build runs in its own process, three rounds, alternating.
ARGS=$1; shift
for round in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    NDFFT_MI355X_LIB=$PWD/$lib python tools/bench_configs.py $ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-10s round $round  %8.2f us  %-14s %s' % ('$name', r['us'], r.get('path'), r['workload'])); ('us_reread' in r and r.get('pairs', 1) > 1) and print('%-10s round $round  %8.2f us  %-14s %s [re-read]' % ('$name', r['us_reread'], r.get('path'), r['workload']))"
  done
done

#!/bin/bash
# ABAB check of a run-time knob against order / clock drift: abab.sh <ENVVAR> <A> <B> <bench section> [grep pattern]
var=$1; A=$2; B=$3; sec=$4; pat=${5:-.}
for v in $A $B $A $B $A $B; do
  echo "== $var=$v"
  env $var=$v python tools/bench_configs.py --only $sec --steps 30 > /tmp/abab.txt 2>&1
  python tools/probes/show.py /tmp/abab.txt | grep -E "$pat"
done

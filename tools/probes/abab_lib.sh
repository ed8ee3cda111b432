#!/bin/bash
# ABAB of the product library against a variant build: abab_lib.sh <variant.so> <bench section> [grep pattern]
lib=$1; sec=$2; pat=${3:-.}
for v in product variant product variant product variant; do
  echo "== $v"
  if [ $v = product ]; then python tools/bench_configs.py --only $sec --steps 30 > /tmp/abab.txt 2>&1; else python tools/probes/ab_lib.py $lib -- --only $sec --steps 30 > /tmp/abab.txt 2>&1; fi
  python tools/probes/show.py /tmp/abab.txt | grep -E "$pat"
done

#!/bin/bash
export TMPDIR=/tmp; cd /tmp
for w in r2c c128 c64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$w -- python3 $GRAFT_REPO_ROOT/tools/probes/cs_prof.py $w > /dev/null 2>&1
  echo "== $w"; find /tmp/p_$w -name "*kernel_stats.csv" | xargs cat | cut -c1-260 | awk -F'","' '{print $1" | calls "$2" | avg_ns "$4}' | head -6
done

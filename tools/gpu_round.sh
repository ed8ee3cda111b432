#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprof kernel trace.  Usage: tools/gpu_round.sh <tag> [pytest-args]
TAG=${1:-r01}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== device ==" > $OUT/summary.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 >> $OUT/summary.txt
echo "== smoke ==" >> $OUT/summary.txt
timeout 600 python __graft_entry__.py smoke >> $OUT/summary.txt 2>&1
echo "== pytest -m gpu ==" >> $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/summary.txt
tail -25 $OUT/pytest.log >> $OUT/summary.txt
echo "== bench ==" >> $OUT/summary.txt
timeout 900 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt; tail -5 $OUT/bench.err >> $OUT/summary.txt
echo "== rocprof ==" >> $OUT/summary.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err); echo "rocprof exit $?" >> $OUT/summary.txt
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -8 >> $OUT/summary.txt
cat $OUT/summary.txt

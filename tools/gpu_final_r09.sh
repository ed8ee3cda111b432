#!/bin/bash
# Round-6 closing visit: parity suite, the whole config table twice (HBM-sourced first) with the regression guard, bench.py, rocprofv3 kernel stats + PMC traffic
# of the bench regions and of cfg3 / cfg4 / cfg5, the reference bench matrix.   Usage (GPU box): bash tools/gpu_final_r09.sh <tag>
TAG=${1:-r09z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" > $OUT/summary.txt; tail -3 $OUT/pytest.log >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" >> $OUT/summary.txt
timeout 2400 python tools/bench_configs.py --steps 50 --preheat-s 10 --repeat 2 --compare profiles/r08/r08p_configs_final_repeat2.jsonl,profiles/r08/r08v_configs_final_repeat2.jsonl,profiles/r08/r08zz_configs_final_repeat2.jsonl > $OUT/configs_repeat2.jsonl 2> $OUT/configs.err; echo "table exit $?" >> $OUT/summary.txt
grep "^#" $OUT/configs.err > $OUT/configs_compare.txt
bash tools/prof_bench.sh $TAG/prof_bench pmc > $OUT/prof_bench.log 2>&1
bash tools/prof_configs.sh $TAG/prof_configs > $OUT/prof_configs.log 2>&1
timeout 900 python tools/ref_bench_table.py --md > $OUT/ref_bench_table.txt 2> $OUT/ref_bench_table.err
# keep what is merged back small: drop rocprof's raw traces, keep the summaries
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; cat $OUT/summary.txt; cat $OUT/configs_compare.txt | head -30

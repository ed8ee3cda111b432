#!/bin/bash
# Round-6 LAST closing visit (after the late changes: lane_split, staging tails): like tools/gpu_final_r09.sh without the parity suite (run by itself on the same
# build) and with the config table once.   Usage (GPU box): bash tools/gpu_final_r09b.sh <tag>
TAG=${1:-r09end}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" > $OUT/summary.txt
timeout 1500 python tools/bench_configs.py --steps 50 --preheat-s 10 --repeat 1 --compare profiles/r08/r08p_configs_final_repeat2.jsonl,profiles/r08/r08v_configs_final_repeat2.jsonl,profiles/r08/r08zz_configs_final_repeat2.jsonl,profiles/r09/r09final_configs_final_repeat2.jsonl > $OUT/configs.jsonl 2> $OUT/configs.err; echo "table exit $?" >> $OUT/summary.txt
grep "^#" $OUT/configs.err > $OUT/configs_compare.txt
bash tools/prof_bench.sh $TAG/prof_bench pmc > $OUT/prof_bench.log 2>&1
timeout 900 python tools/ref_bench_table.py --md > $OUT/ref_bench_table.txt 2> $OUT/ref_bench_table.err
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; cat $OUT/summary.txt; cat $OUT/configs_compare.txt | head -30

#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc3a; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --steps 3 --only cfg3 > $OUT/$N.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc3a'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        short='xcd_r2c' if ('RealPow2Kernel' in k and 'true, true' in k and ', 2,' in k) else 'xcd_c2r' if ('RealPow2Kernel' in k and 'true, true' in k) else None
        if short: agg[short+' grid='+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k,{c:round(sum(x)/len(x)) for c,x in v.items()})
PY

#!/bin/bash
# kbench + PMC counter passes for the product kernel. Usage: tools/gpu_kbench.sh <tag>
TAG=${1:-kb}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 tools/kbench 4096 15 > $OUT/kbench.txt 2>&1
cat $OUT/kbench.txt
if [ "$2" == "pmc" ]; then
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
P3="FETCH_SIZE"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P5="SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -- $GRAFT_REPO_ROOT/tools/kbench 4096 1 > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i exit $?"
  i=$((i+1))
done
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'+os.environ.get('TAGX','kb')
PY
fi

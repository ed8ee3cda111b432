#!/bin/bash
# A/B of the row workgroup size of the power-of-two real-op kernels (NDFFT_REAL_ROW_THREADS = 256 product, 128, 64) on one box
out=${1:-gpurun_out/rowthr_ab.txt}; : > $out
for v in product 128 64; do
  echo "== row threads $v" >> $out
  if [ $v = product ]; then python tools/bench_configs.py --only landscape --steps 30 > /tmp/o.txt 2>&1; else python tools/probes/ab_lib.py tools/probes/libndfft_rowthr$v.so -- --only landscape --steps 30 > /tmp/o.txt 2>&1; fi
  python tools/probes/show.py /tmp/o.txt | grep -E "nddct2|r2c" | grep -E "n=(128|256|512|1024|2048|4096) " >> $out
done

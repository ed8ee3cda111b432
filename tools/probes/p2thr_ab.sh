#!/bin/bash
# A/B of the workgroup size of the power-of-two C2C row kernels (NDFFT_POW2_ROW_THREADS = 256 product, 128, 64) on one box, warm and cold (6 rotating pairs)
out=${1:-gpurun_out/p2thr_ab.txt}; : > $out
for pairs in 1 6; do
for v in product 128 64 product; do
  echo "== threads $v pairs $pairs" >> $out
  if [ $v = product ]; then python tools/bench_configs.py --only pow2sweep --steps 40 --pairs $pairs > /tmp/o.txt 2>&1; else python tools/probes/ab_lib.py tools/probes/libndfft_p2thr$v.so -- --only pow2sweep --steps 40 --pairs $pairs > /tmp/o.txt 2>&1; fi
  python tools/probes/show.py /tmp/o.txt | grep -E "x(128|256|512|1024|2048) " >> $out
done
done

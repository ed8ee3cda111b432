#!/bin/bash
# real four-step A/B: long real-data lanes with the packed complex route (NDFFT_REAL_FOURSTEP=0) and the real four-step, and its knobs
for rep in 1 2; do
  echo "== packed complex four-step + PRE / POST"; NDFFT_REAL_FOURSTEP=0 python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"
  echo "== real four-step"; python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"
  echo "== real four-step, plain stores at the mirrored index"; NDFFT_RFS_MIRROR_PLAIN=1 python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"
done
echo "== real four-step, log2 N1 = 10"; NDFFT_RFS_LOGN1=10 python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"
echo "== real four-step, log2 N1 = 10, plain mirrored"; NDFFT_RFS_MIRROR_PLAIN=1 NDFFT_RFS_LOGN1=10 python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"

#!/bin/bash
# real four-step: the split n = N1 * N2 (log2 N1) per dtype, next to the packed complex route
export LONG_REAL_ONLY=nddct2,ndfft_r2c
echo "== packed complex four-step + PRE / POST"; NDFFT_REAL_FOURSTEP=0 python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"
for a in 8 9 10 11; do echo "== real four-step, log2 N1 = $a"; NDFFT_RFS_LOGN1=$a python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"; done
echo "== packed complex four-step + PRE / POST"; NDFFT_REAL_FOURSTEP=0 python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"
for a in 9 10; do echo "== real four-step, log2 N1 = $a"; NDFFT_RFS_LOGN1=$a python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"; done

#!/bin/bash
# real four-step, both directions, next to the packed complex route
export LONG_REAL_ONLY=nddct2,nddct3,ndfft_r2c,ndifft_r2c
for rep in 1 2; do
  echo "== packed complex four-step + PRE / POST"; NDFFT_REAL_FOURSTEP=0 python tools/probes/long_real.py 2>&1 | grep "64x"
  echo "== real four-step"; python tools/probes/long_real.py 2>&1 | grep "64x"
done

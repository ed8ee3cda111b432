#!/bin/bash
export LONG_REAL_ONLY=nddct4
for rep in 1 2; do
  echo "== packed complex four-step + PRE / POST"; NDFFT_REAL_FOURSTEP=0 python tools/probes/long_real.py 2>&1 | grep "64x"
  echo "== fused DCT-IV four-step"; python tools/probes/long_real.py 2>&1 | grep "64x"
done

#!/bin/bash
export LONG_REAL_ONLY=nddct3,ndifft_r2c
for rep in 1 2; do
  echo "== last pass through dispatch() (32-lane general column kernel)"; NDFFT_RFS_C2R_TILE=0 python tools/probes/long_real.py 2>&1 | grep "64x"
  echo "== last pass on 128-byte tiles"; python tools/probes/long_real.py 2>&1 | grep "64x"
done

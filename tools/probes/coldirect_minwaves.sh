#!/bin/bash
export LONG_REAL_ONLY=nddct2,nddct3,nddct4,ndfft_r2c,ndifft_r2c
for rep in 1 2 3; do
  echo "== before (no VGPR cap: 126-159 VGPRs)"; NDFFT_MI355X_LIB=$PWD/tools/_ab/libndfft_old.so python tools/probes/long_real.py 2>&1 | grep "64x"
  echo "== __launch_bounds__(THREADS, 4)"; python tools/probes/long_real.py 2>&1 | grep "64x"
done

#!/bin/bash
# real four-step, pass 2: store policy at lines shared by neighbouring tiles x runs of consecutive tiles per XCD
export LONG_REAL_ONLY=nddct2,ndfft_r2c
for rep in 1 2; do
for mp in 0 1 3; do for xc in 0 2 8 32; do
  echo "== plain-store mask $mp, xcd chunk $xc"; NDFFT_RFS_MIRROR_PLAIN=$mp NDFFT_RFS_XCD_CHUNK=$xc python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"
done; done; done

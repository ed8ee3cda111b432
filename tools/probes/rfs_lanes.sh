#!/bin/bash
export LONG_REAL_ONLY=nddct2,nddct3,ndfft_r2c,ndifft_r2c
for rep in 1 2; do for mp in 1 3; do for xc in 8 32; do
  echo "== plain-store mask $mp, xcd chunk $xc"; NDFFT_RFS_MIRROR_PLAIN=$mp NDFFT_RFS_XCD_CHUNK=$xc python tools/probes/long_real.py 2>&1 | grep "64x"
done; done; done

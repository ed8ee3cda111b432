#!/bin/bash
export LONG_REAL_ONLY=nddct3,ndifft_r2c
for rep in 1 2; do for xc in 0 4 8 32; do echo "== xcd chunk $xc"; NDFFT_RFS_XCD_CHUNK=$xc python tools/probes/long_real.py 2>&1 | grep "64x"; done; done

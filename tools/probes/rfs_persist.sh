#!/bin/bash
export LONG_REAL_ONLY=nddct2,nddct3,nddct4,ndfft_r2c,ndifft_r2c
for rep in 1 2; do for pc in 0 1 2 3 4; do echo "== persistent workgroups per CU: $pc"; NDFFT_RFS_PERSIST=$pc python tools/probes/long_real.py 2>&1 | grep "64x"; done; done

#!/bin/bash
export LONG_REAL_ONLY=ndfft_r2c NDFFT_RFS_MIRROR_PLAIN=1 NDFFT_RFS_XCD_CHUNK=8
for sg in 0 1 2 3 4 6 8; do echo "== stagger $sg"; NDFFT_RFS_STAGGER=$sg python tools/probes/long_real.py 2>&1 | grep "ndfft_r2c"; done

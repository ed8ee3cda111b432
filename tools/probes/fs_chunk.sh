#!/bin/bash
export LONG_REAL_ONLY=ndfft
for rep in 1 2; do for xc in 0 2 4 8 32; do echo "== four-step xcd chunk $xc"; NDFFT_FS_XCD_CHUNK=$xc python tools/probes/long_real.py 2>&1 | grep "ndfft "; done; done

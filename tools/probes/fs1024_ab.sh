#!/bin/bash
export LONG_REAL_ONLY=ndfft
for rep in 1 2; do
  echo "== default build"; python tools/probes/long_real.py 2>&1 | grep "ndfft "
  echo "== 4 c128 lanes per F = 1024 tile (64-byte rows, 70 KiB) with the XCD runs"; NDFFT_MI355X_LIB=$PWD/tools/_ab/libndfft_fs4.so python tools/probes/long_real.py 2>&1 | grep "ndfft "
done

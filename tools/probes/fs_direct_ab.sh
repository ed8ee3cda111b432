#!/bin/bash
# lane-fastest register kernels (col_direct.h) vs the staged column kernels in the four-step passes
for rep in 1 2; do
  echo "== staged column kernels (NDFFT_FS_DIRECT=0)"; NDFFT_FS_DIRECT=0 python tools/probes/long_real.py 2>&1 | grep -v amdgpu.ids
  echo "== lane-fastest register kernels"; python tools/probes/long_real.py 2>&1 | grep -v amdgpu.ids
done
echo "== lane-fastest, plain stores at the mirrored index"; NDFFT_RFS_MIRROR_PLAIN=1 python tools/probes/long_real.py 2>&1 | grep "nddct2\|ndfft_r2c"

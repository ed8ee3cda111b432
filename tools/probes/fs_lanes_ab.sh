#!/bin/bash
# A-B-A-B of the row four-step's tile width for 1024-point factors: default (8 lanes) vs libndfft_fs4.so (4 lanes: 64-byte rows, two workgroups per CU)
for rep in 1 2; do
  echo "== default"; python tools/bench_configs.py --only longlanes --steps 30 2>&1 | python tools/probes/show.py /dev/stdin
  echo "== 4 lanes for F = 1024"; python tools/probes/ab_lib.py ndrustfft_amd/csrc/libndfft_fs4.so -- --only longlanes --steps 30 2>&1 | python tools/probes/show.py /dev/stdin
done

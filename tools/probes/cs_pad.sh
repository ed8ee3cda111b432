#!/bin/bash
# Does padding the row pitch of col_split's intermediate (NDFFT_CS_PAD bytes) help?  A-B-A-B over the pads, cfg3-A, cfg3-A' and fft2.
for rep in 1 2; do
  for pad in 0 512 1024 256; do
    echo "== NDFFT_CS_PAD=$pad"
    for sec in cfg3A_only cfg3Ap_only; do
      env NDFFT_CS_PAD=$pad python tools/bench_configs.py --only $sec --steps 30 2>&1 | python tools/probes/show.py /dev/stdin
    done
  done
done

#!/bin/bash
for mb in 0 8 16 32 64 128; do
  echo "== NDFFT_CS_CHUNK_MB=$mb"
  NDFFT_CS_CHUNK_MB=$mb python tools/bench_configs.py --steps 60 --only cfg3 2>&1 | grep -E "cfg3A" | cut -c14-150
  NDFFT_CS_CHUNK_MB=$mb python tools/bench_configs.py --steps 60 --only fft2d 2>&1 | grep -E "fft2 (4096|8192)" | cut -c14-140
done

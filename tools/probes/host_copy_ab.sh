#!/bin/bash
# bounce-buffer host path (pageable caller arrays, 4096 x 4096 c128): streaming-store copies (default) vs memcpy (NDFFT_COPY_NT=0), A-B-A-B
for rep in 1 2 3; do
  for nt in 1 0; do
    echo -n "NDFFT_COPY_NT=$nt: "
    env NDFFT_COPY_NT=$nt python bench.py --steps 20 --warmup 2 --no-cpu-baseline --cold-pairs 0 --strong-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['host_api']['ms_per_call'], d['host_api']['registered']['steady_state_ms_per_call'])"
  done
done

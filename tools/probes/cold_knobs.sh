#!/bin/bash
# cold region of bench.py (6 rotating pairs) under run-time knobs, A-B-A-B
for rep in 1 2; do
  for kb in 512 128 2048 8192; do
    echo -n "NDFFT_XCD_CHUNK_KB=$kb: "
    env NDFFT_XCD_CHUNK_KB=$kb python bench.py --steps 100 --warmup 5 --profile-phase cold 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['frac_cold'], d['roofline']['cold']['avg_launch_us'])"
  done
done

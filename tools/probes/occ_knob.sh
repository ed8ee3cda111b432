#!/bin/bash
# resident workgroups per CU of the headline kernel (34 KiB of LDS each: 4 per CU) limited through extra dynamic LDS: 0 -> 4, 8 KiB -> 3, 20 -> 2
for rep in 1 2; do
  for kb in 0 8 20; do
    echo -n "NDFFT_POW2_LDS_PAD_KB=$kb: "
    env NDFFT_POW2_LDS_PAD_KB=$kb python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-host-api --strong-steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warm', d['roofline']['frac'], 'cold', d['roofline']['frac_cold'], 'strong', d['strong_cfg5']['per_gpu_frac'])"
  done
done

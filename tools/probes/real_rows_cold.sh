#!/bin/bash
# cfg4 (nddct1..4 rows, 256x256x512 f64) and f32 R2C rows: warm and cold (6 rotating pairs), load policy from the model (auto) vs forced plain loads
for rep in 1 2; do
  for sl in auto 0; do
    echo "== NDFFT_STREAM_LOADS=$sl"
    if [ $sl == auto ]; then unset NDFFT_STREAM_LOADS; else export NDFFT_STREAM_LOADS=$sl; fi
    python tools/bench_configs.py --only cfg4 --steps 40 2>&1 | python tools/probes/show.py /dev/stdin | grep -E "axis=[012] "
    python tools/bench_configs.py --only cfg4 --steps 40 --pairs 6 2>&1 | python tools/probes/show.py /dev/stdin | grep -E "axis=[012] "
  done
done

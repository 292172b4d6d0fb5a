#!/bin/bash
# A/B of two builds of the library inside one gpurun call: bash tools/ab_libs.sh old new [rounds]  (gif_amd/_variants/libgif_hip_<name>.so;
# 12 timed iterations without R1 and without per-launch events, arms alternating)
cd "$(dirname "$0")/.."
cp gif_amd/libgif_hip.so /tmp/libgif_keep.so
for r in $(seq 1 ${3:-3}); do
  for a in $1 $2; do
    cp gif_amd/_variants/libgif_hip_$a.so gif_amd/libgif_hip.so
    python bench.py --steps 12 --warmup 3 --no-prof --no-cpu-baseline --r1-every 100000 2> /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$a'.ljust(8), '%.1f ms  %.1f images/s' % (d['ms_per_step'], d['value']))"
  done
done
cp /tmp/libgif_keep.so gif_amd/libgif_hip.so

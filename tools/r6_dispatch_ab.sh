#!/bin/bash
# Run ON THE GPU BOX (gpurun) from the repo root: Winograd / direct dispatch A/B per channel count under f16x2 (review item 2 of round 5).
#   tools/r6_dispatch_ab.sh <out-dir under gpurun_out>
# ONE call, alternating arms (the boxes of the pool differ by +-1.5 %, and a box drifts over minutes: A B A B ...).
set -u
OUT=$1; mkdir -p $OUT
export GIF_EXPERIMENTAL=1
B="python bench.py --no-cpu-baseline --no-prof --steps 12 --warmup 3 --r1-every 100000"
run() {  # name, env assignments...
  local name=$1; shift
  ( env "$@" $B > $OUT/ab_$name.json 2> $OUT/ab_$name.err ) || echo "FAILED $name"
  python - "$OUT/ab_$name.json" "$name" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:34s} {d['ms_per_step']:8.2f} ms/step  {d['value']:7.2f} img/s")
except Exception as e:
    print(sys.argv[2], "no line:", e)
P
}
for rep in 1 2; do
  run default_$rep                X=0
  run fwd_minC256_$rep            GIF_WINOGRAD_MIN_C=256
  run wgrad_minC256_$rep          GIF_WINOGRAD_WGRAD_MIN_C=256
  run both_minC256_$rep           GIF_WINOGRAD_MIN_C=256 GIF_WINOGRAD_WGRAD_MIN_C=256
done
run fwd_minC512                   GIF_WINOGRAD_MIN_C=512
run both_minC512                  GIF_WINOGRAD_MIN_C=512 GIF_WINOGRAD_WGRAD_MIN_C=512
run no_winograd                   GIF_WINOGRAD=0
run fwd256_wgrad512               GIF_WINOGRAD_MIN_C=256 GIF_WINOGRAD_WGRAD_MIN_C=512

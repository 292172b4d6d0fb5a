#!/bin/bash
# A/B of the library variants built by tools/probes/sched_flag_variants.sh: 12 timed iterations without R1, per-family kernel time from the
# sampled HIP events of a second short run.  Usage (on the GPU box): bash tools/probes/sched_flag_ab.sh [variant ...]
cd "$(dirname "$0")/../.."
cp gif_amd/libgif_hip.so /tmp/libgif_base.so
arms=${@:-base maxilp memclause relaxocc trackers nopostmi nohighrp base}
for a in $arms; do
  if [ $a = base ]; then cp /tmp/libgif_base.so gif_amd/libgif_hip.so; else cp gif_amd/_variants/libgif_hip_$a.so gif_amd/libgif_hip.so; fi
  python bench.py --steps 12 --warmup 3 --no-prof --no-cpu-baseline --r1-every 100000 2> /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$a'.ljust(10), 'no-prof %.1f ms' % d['ms_per_step'], end='  ')"
  python bench.py --steps 8 --warmup 2 --prof-every 1 --no-cpu-baseline --r1-every 100000 2> /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        out=[]
        for k in ('roofline','roofline_conv_winograd_f16x2','roofline_wgrad_direct_f16x2','roofline_wgrad_winograd_f16x2','roofline_conv_direct_bf16x3_tapdense'):
            r=d.get(k)
            if r: out.append('%s %.1f' % (k.replace('roofline_','').replace('roofline','direct_h2'), r['achieved']))
        print(' | '.join(out))"
done
cp /tmp/libgif_base.so gif_amd/libgif_hip.so

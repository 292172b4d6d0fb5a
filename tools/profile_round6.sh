#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: round-6 kernel stats, PMC traffic, bench lines and probes.
#   tools/profile_round6.sh <out-dir under gpurun_out> <commit hash>
# Raw rocprofv3 output is summarised and deleted (the counter CSVs alone exceed gpurun's 64 MiB merge cap).
set -u
OUT=$1; COMMIT=${2:-unknown}
export TMPDIR=/tmp
mkdir -p $OUT
R=$(pwd)
B="python $R/bench.py --no-cpu-baseline"
# 1. per-kernel time (rocprofv3 --kernel-trace --stats), 5 timed + 2 warm-up training iterations
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/stats -- $B --steps 5 --warmup 2 --no-prof > $R/$OUT/stats_bench.json 2> $R/$OUT/stats.err )
DB=$(find $OUT/stats -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.md
rm -rf $OUT/stats
# 2. HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), one profiled iteration
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_$C -- $B --steps 1 --warmup 1 --no-prof > /dev/null 2> $R/$OUT/pmc_$C.err )
done
python tools/pmc_to_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --commit $COMMIT > $OUT/pmc_traffic.json 2> $OUT/pmc_json.err
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json   # the default bench line below relays THIS measurement
# 3. bench lines (the first one is the driver's command; the second SURVEY 8(d)'s >= 50 timed steps after 10 warm-up)
python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_default_50.json 2> /dev/null
$B --steps 12 --warmup 3 --no-prof --r1-every 100000 > $OUT/bench_default_12steps_no_r1_iteration.json 2> /dev/null
$B --steps 12 --warmup 3 --no-prof --r1-every 100000 --fp32-mfma bf16x3 > $OUT/bench_bf16x3_12steps_no_r1_iteration.json 2> /dev/null
$B --steps 12 --warmup 3 --no-prof --r1-every 100000 --fp32-mfma native > $OUT/bench_native_fp32_mfma.json 2> /dev/null
$B --steps 12 --warmup 3 --no-prof --r1-every 100000 --two-call-d > $OUT/bench_two_call_d_12steps_no_r1_iteration.json 2> /dev/null
$B --steps 12 --warmup 3 --batch 16 > $OUT/bench_config2_batch16.json 2> /dev/null
$B --steps 12 --warmup 3 --render-cond --gen-reg PATH_LEN_REG > $OUT/bench_config3_render_plreg.json 2> /dev/null
$B --steps 12 --warmup 3 --texture-interp > $OUT/bench_texture_interp.json 2> $OUT/bench_texture_interp.err
$B --steps 12 --warmup 3 --dtype f16 > $OUT/bench_f16_256.json 2> /dev/null
# config 5 WITH roofline and cpu_baseline
python bench.py --steps 8 --warmup 2 --dtype f16 --res 1024 --batch 8 --cpu-batch 1 --cpu-timeout 400 > $OUT/bench_f16_1024.json 2> $OUT/bench_f16_1024.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/stats1024 -- $B --steps 5 --warmup 2 --no-prof --dtype f16 --res 1024 --batch 8 > $R/$OUT/stats_bench_f16_1024.json 2> $R/$OUT/stats1024.err )
DB=$(find $OUT/stats1024 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats_f16_1024.md
rm -rf $OUT/stats1024
# run 29 with the texture-interpolation loss: per-kernel time of the added work
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/statstex -- $B --steps 4 --warmup 2 --no-prof --texture-interp > $R/$OUT/stats_bench_texture_interp.json 2> $R/$OUT/statstex.err )
DB=$(find $OUT/statstex -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats_texture_interp.md
rm -rf $OUT/statstex
GIF_PROF_DUMP=$OUT/shapes.csv $B --steps 8 --warmup 2 --prof-every 1 > $OUT/bench_shapes.json 2> /dev/null
# 4. probes
python tools/probes/h2_probe.py --time 2>&1 | grep -v amdgpu.ids > $OUT/h2_probe.txt
python tools/probes/h2_wino_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/h2_wino_probe.txt
python tools/probes/h2_fallback_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/h2_fallback_trace.txt
python tools/raster_bench.py --json $OUT/raster_bench.json > $OUT/raster_bench.txt 2>&1
for f in $OUT/bench_*.json; do echo "$f: $(head -c 200 $f | cut -c1-200)"; done
tail -3 $OUT/kernel_stats.md; head -c 300 $OUT/pmc_traffic.json

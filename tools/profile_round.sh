#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel stats, PMC traffic, bench lines and side measurements of one round.
#   tools/profile_round.sh <out-dir under gpurun_out> <commit hash>
# Raw rocprofv3 output is summarised and deleted (the counter CSVs alone exceed gpurun's 64 MiB merge cap).
set -u
OUT=$1; COMMIT=${2:-unknown}
export TMPDIR=/tmp
mkdir -p $OUT
R=$(pwd)
B="python $R/bench.py --no-cpu-baseline"
# 1. per-kernel time (rocprofv3 --kernel-trace --stats), 5 timed training iterations
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/stats -- $B --steps 5 --warmup 2 --no-prof > $R/$OUT/stats_bench.json 2> $R/$OUT/stats.err )
DB=$(find $OUT/stats -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.md
rm -rf $OUT/stats
# 1b. the same for the f16-activation step (row N1)
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/stats16 -- $B --steps 5 --warmup 2 --no-prof --dtype f16 > $R/$OUT/stats_bench_f16.json 2> $R/$OUT/stats16.err )
DB=$(find $OUT/stats16 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats_f16.md
rm -rf $OUT/stats16
# 2. HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), one profiled iteration
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_$C -- $B --steps 1 --warmup 1 --no-prof > /dev/null 2> $R/$OUT/pmc_$C.err )
done
python tools/pmc_to_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --commit $COMMIT > $OUT/pmc_traffic.json 2> $OUT/pmc_json.err
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json   # the default bench line below relays THIS measurement
# 3. bench lines
python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --steps 12 --warmup 3 --fp32-mfma native > $OUT/bench_native_fp32_mfma.json 2> /dev/null
GIF_FUSE_GRAD=0 $B --steps 12 --warmup 3 > $OUT/bench_no_gradient_epilogue_fusions.json 2> /dev/null
$B --steps 12 --warmup 3 > $OUT/bench_default_12steps_no_r1_iteration.json 2> /dev/null
$B --steps 12 --warmup 3 --reuse-generator-forward > $OUT/bench_one_generator_forward_NOT_headline.json 2> /dev/null
GIF_FUSE_D=0 $B --steps 12 --warmup 3 > $OUT/bench_two_discriminator_calls.json 2> /dev/null
$B --steps 12 --warmup 3 --batch 16 > $OUT/bench_config2_batch16.json 2> /dev/null
$B --steps 12 --warmup 3 --render-cond --gen-reg PATH_LEN_REG > $OUT/bench_config3_render_plreg.json 2> /dev/null
$B --steps 12 --warmup 3 --dtype f16 > $OUT/bench_f16_256.json 2> /dev/null
$B --steps 8 --warmup 2 --dtype f16 --res 1024 --batch 8 > $OUT/bench_f16_1024.json 2> /dev/null
# round 4: the f16 halo kernels / 256x256 weight-gradient tiles off (A/B in the same call)
GIF_F16_HALO=0 GIF_F16_HALO_WGRAD=0 $B --steps 8 --warmup 2 --dtype f16 --res 1024 --batch 8 --no-prof > $OUT/bench_f16_1024_no_halo_kernels.json 2> /dev/null
GIF_F16_WGRAD256=0 $B --steps 12 --warmup 3 --dtype f16 --no-prof > $OUT/bench_f16_256_wgrad_128_tiles.json 2> /dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/stats1024 -- $B --steps 5 --warmup 2 --no-prof --dtype f16 --res 1024 --batch 8 > $R/$OUT/stats_bench_f16_1024.json 2> $R/$OUT/stats1024.err )
DB=$(find $OUT/stats1024 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats_f16_1024.md
rm -rf $OUT/stats1024
GIF_PROF_DUMP=$OUT/shapes_f16_1024.csv $B --steps 4 --warmup 2 --prof-every 1 --dtype f16 --res 1024 --batch 8 > $OUT/bench_shapes_f16_1024.json 2> /dev/null
GIF_PROF_DUMP=$OUT/shapes.csv $B --steps 8 --warmup 2 --prof-every 1 > $OUT/bench_shapes.json 2> /dev/null
GIF_PROF_DUMP=$OUT/shapes_f16.csv $B --steps 8 --warmup 2 --prof-every 1 --dtype f16 > $OUT/bench_shapes_f16.json 2> /dev/null
# 4. side measurements
python tools/raster_bench.py --json $OUT/raster_bench.json > $OUT/raster_bench.txt 2>&1
python tools/probes/f16_error_by_layer.py > $OUT/f16_error_by_layer.txt 2>&1
python tools/probes/x3_power_trace.py > $OUT/x3_power_trace.txt 2>&1
python tools/probes/f16_halo_bench.py > $OUT/f16_halo_bench.txt 2>&1
python tools/probes/f16_halo_bench.py --wgrad >> $OUT/f16_halo_bench.txt 2>&1
GIF_F16_HALO_WGRAD=0 python tools/probes/f16_halo_bench.py --wgrad > $OUT/f16_wgrad_per_tap_kernel.txt 2>&1
python tools/probes/aten_crumbs.py > $OUT/aten_crumbs_f32_256.txt 2>&1
python tools/probes/aten_crumbs.py --dtype f16 --res 1024 --batch 8 > $OUT/aten_crumbs_f16_1024.txt 2>&1
for f in $OUT/bench_*.json; do echo "$f: $(head -c 260 $f | cut -c1-260)"; done
tail -3 $OUT/kernel_stats.md; head -c 400 $OUT/pmc_traffic.json; tail -5 $OUT/raster_bench.txt

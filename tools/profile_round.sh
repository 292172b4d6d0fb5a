#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel stats, PMC traffic and the bench lines of one round.
#   tools/profile_round.sh <out-dir under gpurun_out> <commit hash>
# Raw rocprofv3 output is summarised and deleted (the counter CSVs alone exceed gpurun's 64 MiB merge cap).
set -u
OUT=$1; COMMIT=${2:-unknown}
export TMPDIR=/tmp
mkdir -p $OUT
B="python bench.py --no-cpu-baseline"
# 1. per-kernel time (rocprofv3 --kernel-trace --stats), 5 timed training iterations
rocprofv3 --kernel-trace --stats -d $OUT/stats -- $B --steps 5 --warmup 2 --no-prof > $OUT/stats_bench.json 2> $OUT/stats.err
DB=$(find $OUT/stats -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.md
rm -rf $OUT/stats
# 2. HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), one profiled iteration
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- $B --steps 1 --warmup 1 --no-prof > /dev/null 2> $OUT/pmc_$C.err
done
python tools/pmc_to_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --commit $COMMIT > $OUT/pmc_traffic.json 2> $OUT/pmc_json.err
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
# 3. bench lines
python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --steps 12 --warmup 3 --fp32-mfma native > $OUT/bench_native_fp32_mfma.json 2> /dev/null
$B --steps 12 --warmup 3 --batch 16 > $OUT/bench_config2_batch16.json 2> /dev/null
$B --steps 12 --warmup 3 --render-cond --gen-reg PATH_LEN_REG > $OUT/bench_config3_render_plreg.json 2> /dev/null
$B --steps 12 --warmup 3 --dtype f16 > $OUT/bench_f16_256.json 2> /dev/null
$B --steps 8 --warmup 2 --dtype f16 --res 1024 --batch 8 > $OUT/bench_f16_1024.json 2> /dev/null
GIF_PROF_DUMP=$OUT/shapes.csv $B --steps 8 --warmup 2 > $OUT/bench_shapes.json 2> /dev/null
for f in $OUT/bench_*.json; do echo "$f: $(head -c 260 $f | cut -c1-260)"; done
tail -3 $OUT/kernel_stats.md; head -c 600 $OUT/pmc_traffic.json

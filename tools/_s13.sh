set -u
OUT=gpurun_out/s13; mkdir -p $OUT; R=$(pwd); export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1; tail -3 $OUT/pytest_all.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_5.json 2> $OUT/err1
B="python $R/bench.py --no-cpu-baseline --no-prof"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/st16 -- $B --steps 4 --warmup 2 --dtype f16 > $R/$OUT/f16_256.json 2> $R/$OUT/err2 )
DB=$(find $OUT/st16 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats_f16_256.md; rm -rf $OUT/st16
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/st1k -- $B --steps 4 --warmup 2 --dtype f16 --res 1024 --batch 8 > $R/$OUT/f16_1024.json 2> $R/$OUT/err3 )
DB=$(find $OUT/st1k -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats_f16_1024.md; rm -rf $OUT/st1k
GIF_PROF_DUMP=$OUT/shapes_f16_256.csv python bench.py --no-cpu-baseline --steps 4 --warmup 2 --dtype f16 > $OUT/f16_256_shapes.json 2>/dev/null
GIF_PROF_DUMP=$OUT/shapes_f16_1024.csv python bench.py --no-cpu-baseline --steps 4 --warmup 2 --dtype f16 --res 1024 --batch 8 > $OUT/f16_1024_shapes.json 2>/dev/null
for f in $OUT/*.json; do echo "$f $(python -c "import json,sys;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])")"; done

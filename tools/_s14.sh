set -u
OUT=gpurun_out/s14; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_f16.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-prof"
$B --steps 8 --warmup 2 --dtype f16 > $OUT/f16_256_fused.json 2> $OUT/err1
GIF_FUSE_GRAD=0 $B --steps 8 --warmup 2 --dtype f16 > $OUT/f16_256_unfused.json 2>> $OUT/err1
$B --steps 6 --warmup 2 --dtype f16 --res 1024 --batch 8 > $OUT/f16_1024_fused.json 2> $OUT/err2
GIF_FUSE_GRAD=0 $B --steps 6 --warmup 2 --dtype f16 --res 1024 --batch 8 > $OUT/f16_1024_unfused.json 2>> $OUT/err2
for f in $OUT/*.json; do echo "$f $(python -c "import json,sys;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])")"; done

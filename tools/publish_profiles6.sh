#!/bin/bash
# Copy what tools/profile_round6.sh produced (merged back under gpurun_out/<dir>) into profiles/ (tracked).
#   tools/publish_profiles5.sh gpurun_out/<dir> <commit>
set -eu
O=$1; C=$2; R=r6
cp $O/pmc_traffic.json profiles/pmc_traffic.json
for f in default default_50 default_12steps_no_r1_iteration bf16x3_12steps_no_r1_iteration native_fp32_mfma two_call_d_12steps_no_r1_iteration config2_batch16 config3_render_plreg texture_interp f16_256 f16_1024; do
  [ -s $O/bench_$f.json ] && cp $O/bench_$f.json profiles/${R}_bench_$f.json
done
for f in raster_bench.txt raster_bench.json h2_probe.txt h2_wino_probe.txt h2_fallback_trace.txt; do [ -s $O/$f ] && cp $O/$f profiles/${R}_$f; done
v=$(python -c "import json;d=json.load(open('$O/stats_bench.json'));print(f\"{d['value']:.1f} images/s, {d['ms_per_step']:.1f} ms/step\")")
{ echo "# rocprofv3 --kernel-trace --stats, round 6 (commit $C, f16x2 default): python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof"; echo
  echo "7 training iterations (5 timed + 2 warm-up) at 256x256, batch 32, fp32 tensors; summarised from the rocpd database by tools/rocpd_stats.py."
  echo "conv_gather_mfma_glds<.., X3, NST>: X3 = 2 are the f16x2 instantiations (NST = stages of the operand ring), X3 = 1 bf16x3; conv_wgrad_mfma<.., X3>; the bf16x3 ones are in this mode mostly the guarded twin launches"
  echo "(min ~4 us: they return at once) plus the tap-dense thin layers.  bench line of the profiled run: $v (profiler attached)"; echo; cat $O/kernel_stats.md; } > profiles/${R}_kernel_stats.md
s=$(python -c "import json;print(round(json.load(open('$O/bench_shapes.json'))['value'],1))")
{ echo "# Per-launch-shape timings of the profiled MFMA kernel families inside the bench region (round 6, commit $C, f16x2 default)"; echo
  echo "\`GIF_PROF_DUMP=file python bench.py --steps 8 --warmup 2 --no-cpu-baseline --prof-every 1\` (HIP events around every launch of the 8 timed steps; $s images/s)."
  echo "family 13 = direct conv fwd/dgrad on the f16x2 LDS-DMA kernel (tag = taps*10+stride, negative = transposed/dgrad; the time includes the guarded bf16x3 twin launch), 14 = wino_gemm_h2 (tag 2091), 15 / 16 = f16x2 weight gradient direct / Winograd plane GEMMs (tag +100 = modulated), 12 = bf16x3 direct kernel in the tap-dense K order (3x3 layers with 8..28 contraction channels), 5 = native register-staged kernel (1x1 / ToRGB-gradient layers with < 24 channels), 1 = native weight gradients (small-channel), 4 = Winograd transforms (last column TB/s)."
  echo "All rates are ALGORITHMIC fp32 TFLOP/s (direct-convolution count); the f16 pipe executes 3x (families 13, 15) or 3*16/36 = 1.33x (14, 16) of it.  Rows below 0.4 ms/step are folded into the totals."; echo
  python tools/shape_table.py $O/shapes.csv 8; } > profiles/${R}_conv_shapes.md
if [ -s $O/kernel_stats_f16_1024.md ]; then
v=$(python -c "import json;d=json.load(open('$O/stats_bench_f16_1024.json'));print(f\"{d['value']:.1f} images/s, {d['ms_per_step']:.1f} ms/step\")")
{ echo "# rocprofv3 --kernel-trace --stats, round 6 (commit $C), BASELINE configs[4]: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --dtype f16 --res 1024 --batch 8"; echo
  echo "7 training iterations at 1024x1024, batch 8, f16 activations; bench line of the profiled run: $v (profiler attached)"; echo; cat $O/kernel_stats_f16_1024.md; } > profiles/${R}_kernel_stats_f16_1024.md
fi
if [ -s $O/kernel_stats_texture_interp.md ]; then
v=$(python -c "import json;d=json.load(open('$O/stats_bench_texture_interp.json'));print(f\"{d['value']:.1f} images/s, {d['ms_per_step']:.1f} ms/step\")")
{ echo "# rocprofv3 --kernel-trace --stats, round 6 (commit $C): python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof --texture-interp"; echo
  echo "6 training iterations at 256x256, batch 32 WITH the texture-space interpolation loss of run 29 on every generator step (train.py:222-238); bench line of the profiled run: $v (profiler attached)."
  echo "Added per iteration: 2 rasteriser passes + vertex normals for the interpolated condition, one generator forward + backward on 31 images, texture_map_kernel (+ backward) and 31 tex_pair_loss launches (+ backward)."; echo; cat $O/kernel_stats_texture_interp.md; } > profiles/${R}_kernel_stats_texture_interp.md
fi
ls profiles | grep "^r6_"

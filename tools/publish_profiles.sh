#!/bin/bash
# Copy what tools/profile_round.sh produced (merged back under gpurun_out/<dir>) into profiles/ (tracked).
#   tools/publish_profiles.sh gpurun_out/<dir> <commit> <round tag, e.g. r3>
set -eu
O=$1; C=$2; R=$3
cp $O/pmc_traffic.json profiles/pmc_traffic.json
for f in default default_12steps_no_r1_iteration one_generator_forward_NOT_headline two_discriminator_calls native_fp32_mfma no_gradient_epilogue_fusions config2_batch16 config3_render_plreg f16_256 f16_1024 f16_1024_no_halo_kernels f16_256_wgrad_128_tiles; do
  [ -s $O/bench_$f.json ] && cp $O/bench_$f.json profiles/${R}_bench_$f.json
done
for f in raster_bench.txt raster_bench.json f16_error_by_layer.txt x3_power_trace.txt f16_halo_bench.txt f16_wgrad_per_tap_kernel.txt aten_crumbs_f32_256.txt aten_crumbs_f16_1024.txt; do [ -s $O/$f ] && cp $O/$f profiles/${R}_$f; done
v=$(python -c "import json;d=json.load(open('$O/stats_bench.json'));print(f\"{d['value']:.1f} images/s, {d['ms_per_step']:.1f} ms/step\")")
{ echo "# rocprofv3 --kernel-trace --stats, round ${R#r} (commit $C, bf16x3 default): python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof"; echo
  echo "7 training iterations (5 timed + 2 warm-up) at 256x256, batch 32, fp32 tensors; summarised from the rocpd database by tools/rocpd_stats.py."
  echo "bench line of the profiled run: $v (profiler attached)"; echo; cat $O/kernel_stats.md; } > profiles/${R}_kernel_stats.md
s=$(python -c "import json;print(round(json.load(open('$O/bench_shapes.json'))['value'],1))")
{ echo "# Per-launch-shape timings of the profiled MFMA kernel families inside the bench region (round ${R#r}, commit $C, bf16x3 default)"; echo
  echo "\`GIF_PROF_DUMP=file python bench.py --steps 8 --warmup 2 --no-cpu-baseline\` (HIP events around every launch of the 8 timed steps; $s images/s)."
  echo "family 8 = direct conv fwd/dgrad on the bf16x3 LDS-DMA kernel (tag = taps*10+stride, negative = transposed/dgrad), 12 = the same kernels in the tap-dense K order (3x3 layers with 8..28 contraction channels), 5 = native register-staged kernel (1x1 / ToRGB-gradient layers with < 24 channels), 9 / 1 = direct wgrad bf16x3 / native (tag +100 = modulated), 10 = wino_gemm_x3 (tag 2091), 11 = Winograd wgrad plane GEMMs on bf16x3, 4 = Winograd transforms (last column TB/s), 0 = native LDS-DMA kernel (none in this mode)."
  echo "All rates are ALGORITHMIC fp32 TFLOP/s (direct-convolution count); the bf16 pipe executes 6x (families 8, 9) or 6*16/36 = 2.67x (10, 11) of it.  Rows below 0.4 ms/step are folded into the totals."; echo
  python tools/shape_table.py $O/shapes.csv 8; } > profiles/${R}_conv_shapes.md
if [ -s $O/kernel_stats_f16.md ]; then
v=$(python -c "import json;d=json.load(open('$O/stats_bench_f16.json'));print(f\"{d['value']:.1f} images/s, {d['ms_per_step']:.1f} ms/step\")")
{ echo "# rocprofv3 --kernel-trace --stats, round ${R#r} (commit $C), f16 activations: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --dtype f16"; echo
  echo "7 training iterations at 256x256, batch 32; bench line of the profiled run: $v (profiler attached)"; echo; cat $O/kernel_stats_f16.md; } > profiles/${R}_kernel_stats_f16.md
fi
if [ -s $O/shapes_f16.csv ]; then
{ echo "# Per-launch-shape timings, f16 activations at 256x256, batch 32 (round ${R#r}, commit $C): family 6 = f16 conv fwd/dgrad, 7 = f16 wgrad; ALGORITHMIC TFLOP/s"; echo
  python tools/shape_table.py $O/shapes_f16.csv 8; } > profiles/${R}_conv_shapes_f16.md
fi
if [ -s $O/kernel_stats_f16_1024.md ]; then
v=$(python -c "import json;d=json.load(open('$O/stats_bench_f16_1024.json'));print(f\"{d['value']:.1f} images/s, {d['ms_per_step']:.1f} ms/step\")")
{ echo "# rocprofv3 --kernel-trace --stats, round ${R#r} (commit $C), BASELINE config 5: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --dtype f16 --res 1024 --batch 8"; echo
  echo "7 training iterations at 1024x1024, batch 8, f16 activations; bench line of the profiled run: $v (profiler attached)"; echo; cat $O/kernel_stats_f16_1024.md; } > profiles/${R}_kernel_stats_f16_1024.md
fi
if [ -s $O/shapes_f16_1024.csv ]; then
{ echo "# Per-launch-shape timings, f16 activations at 1024x1024, batch 8 (round ${R#r}, commit $C): family 6 = f16 conv fwd/dgrad (halo + gather kernels), 7 = f16 wgrad; ALGORITHMIC TFLOP/s"; echo
  python tools/shape_table.py $O/shapes_f16_1024.csv 4; } > profiles/${R}_conv_shapes_f16_1024.md
fi
echo published $O at $C as $R

#!/bin/bash
# Timing probe: how much of the bf16x3 kernels' time is the six MFMA products?  Builds a second library in which the three
# smallest products are left out (-DGIF_X3_FIRST_TERM=3: WRONG numerics, 16-bit products) and times the train step with both.
#   1. here (no GPU needed):   bash tools/probes/x3_three_products.sh build
#   2. on the GPU box:         bash tools/probes/x3_three_products.sh run <out-dir>
# Result of round 4: profiles/r4_x3_three_products_probe.txt
set -eu
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  for f in conv_igemm conv_wgrad conv_winograd; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DGIF_X3_FIRST_TERM=3 -c $f.hip -o _probe/$f.o &
  done; wait
  OBJS=$(ls _build/*.o | grep -v "conv_igemm.o\|conv_wgrad.o\|conv_winograd.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_probe3.so $OBJS _probe/conv_igemm.o _probe/conv_wgrad.o _probe/conv_winograd.o
else
  O=$2; mkdir -p $O
  B="python bench.py --no-cpu-baseline --no-prof --steps 12 --warmup 3"
  $B > $O/b_six.json; cp gif_amd/libgif_hip.so /tmp/keep.so; cp gif_amd/libgif_hip_probe3.so gif_amd/libgif_hip.so
  $B > $O/b_three.json || true
  cp /tmp/keep.so gif_amd/libgif_hip.so
fi

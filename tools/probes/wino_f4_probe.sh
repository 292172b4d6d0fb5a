#!/bin/bash
# F(4x4,3x3) GEMM timing probe (see wino_f4_probe.py).   build: here;   run: on the GPU box
set -eu
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DGIF_WINO_F4_PROBE -c conv_winograd.hip -o _probe/conv_winograd_f4.o
  OBJS=$(ls _build/*.o | grep -v "conv_winograd.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_f4probe.so $OBJS _probe/conv_winograd_f4.o
else
  echo "== today: F(2x2,3x3), fused output transform + epilogue"; python tools/probes/wino_f4_probe.py
  cp gif_amd/libgif_hip.so /tmp/keep.so; cp gif_amd/libgif_hip_f4probe.so gif_amd/libgif_hip.so
  echo "== probe: 36 position GEMMs over a quarter of the rows, M planes stored (timing only)"; python tools/probes/wino_f4_probe.py --f4
  cp /tmp/keep.so gif_amd/libgif_hip.so
fi

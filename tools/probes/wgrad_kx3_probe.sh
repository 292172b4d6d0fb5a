#!/bin/bash
# Timing probe (round 6): the f16x2 weight gradient with the three dx taps of a kernel row formed from ONE staged pair of tiles
# (-DGIF_WGRAD_KX3_PROBE: results are wrong, instruction and traffic counts are those of such a kernel up to the shifted conversion).
#   here:            bash tools/probes/wgrad_kx3_probe.sh build      -> gif_amd/libgif_hip_wgkx3.so
#   on the GPU box:  bash tools/probes/wgrad_kx3_probe.sh run
set -eu
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DGIF_WGRAD_KX3_PROBE -c conv_wgrad.hip -o _probe/conv_wgrad_kx3.o
  OBJS=$(ls _build/*.o | grep -v "conv_wgrad.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_wgkx3.so $OBJS _probe/conv_wgrad_kx3.o
else
  cp gif_amd/libgif_hip.so /tmp/keep.so
  echo "== normal library"; python tools/probes/wgrad_buf_probe.py child
  cp gif_amd/libgif_hip_wgkx3.so gif_amd/libgif_hip.so
  echo "== three taps per staged stage (timing only, wrong results)"; python tools/probes/wgrad_buf_probe.py child
  cp /tmp/keep.so gif_amd/libgif_hip.so
  echo "== normal library, again"; python tools/probes/wgrad_buf_probe.py child
fi

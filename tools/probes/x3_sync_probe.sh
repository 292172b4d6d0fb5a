#!/bin/bash
# Probe build of the library with cycle counters around the mid-stage sync of the bf16x3 direct kernel (-DGIF_X3_TIMING_PROBE).
#   here:            bash tools/probes/x3_sync_probe.sh build      -> gif_amd/libgif_hip_syncprobe.so
#   on the GPU box:  bash tools/probes/x3_sync_probe.sh run        (swaps the library in for the run, restores it)
set -eu
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DGIF_X3_TIMING_PROBE -c conv_igemm.hip -o _probe/conv_igemm_t.o
  OBJS=$(ls _build/*.o | grep -v "conv_igemm.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_syncprobe.so $OBJS _probe/conv_igemm_t.o
else
  cp gif_amd/libgif_hip.so /tmp/keep.so; cp gif_amd/libgif_hip_syncprobe.so gif_amd/libgif_hip.so
  python tools/probes/x3_sync_probe.py || true
  cp /tmp/keep.so gif_amd/libgif_hip.so
fi

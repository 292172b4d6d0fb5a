#!/bin/bash
# Timing probe (round 6): the direct f16x2 / bf16x3 kernels with the activation tile staged only for the FIRST kx tap of every kernel row
# (-DGIF_KXSHARE_PROBE: results are wrong) — the DMA issue / LDS-write saving of a "row + halo staged once, three shifted reads" K loop for
# stride-1 3x3 layers, before anybody builds it.
#   here:            bash tools/probes/kxshare_probe.sh build      -> gif_amd/libgif_hip_kxshare.so
#   on the GPU box:  bash tools/probes/kxshare_probe.sh run
set -eu
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DGIF_KXSHARE_PROBE -c conv_igemm.hip -o _probe/conv_igemm_kxshare.o
  OBJS=$(ls _build/*.o | grep -v "conv_igemm.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_kxshare.so $OBJS _probe/conv_igemm_kxshare.o
else
  echo "== normal library"; python tools/probes/kxshare_probe.py
  cp gif_amd/libgif_hip.so /tmp/keep.so; cp gif_amd/libgif_hip_kxshare.so gif_amd/libgif_hip.so
  echo "== activation tile staged once per kernel row (timing only, wrong results)"; python tools/probes/kxshare_probe.py
  cp /tmp/keep.so gif_amd/libgif_hip.so
  echo "== normal library, again"; python tools/probes/kxshare_probe.py
fi

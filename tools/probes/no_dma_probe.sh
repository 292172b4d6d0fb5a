#!/bin/bash
# Timing probe: the direct bf16x3 / f16x2 kernels WITHOUT the LDS-DMA issue in their K loop (-DGIF_NO_DMA_PROBE: only the first two stages
# are fetched, the loop then runs on stale LDS data — results are wrong).  Upper bound of what taking the DMA issue off the compute waves
# (loader waves) could buy.
#   here:            bash tools/probes/no_dma_probe.sh build      -> gif_amd/libgif_hip_nodma.so
#   on the GPU box:  bash tools/probes/no_dma_probe.sh run
set -eu
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DGIF_NO_DMA_PROBE -c conv_igemm.hip -o _probe/conv_igemm_nodma.o
  OBJS=$(ls _build/*.o | grep -v "conv_igemm.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_nodma.so $OBJS _probe/conv_igemm_nodma.o
else
  echo "== normal library"; python tools/probes/h2_probe.py --time 2>&1 | grep "^(32"
  cp gif_amd/libgif_hip.so /tmp/keep.so; cp gif_amd/libgif_hip_nodma.so gif_amd/libgif_hip.so
  echo "== K loop without DMA issue (timing only, wrong results)"; python tools/probes/h2_probe.py --time 2>&1 | grep "^(32"
  cp /tmp/keep.so gif_amd/libgif_hip.so
fi

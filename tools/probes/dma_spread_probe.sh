#!/bin/bash
# Timing + correctness probe (round 6): the bf16x3 / f16x2 direct kernels with the stage's LDS-DMA pieces issued one at a time BETWEEN the MFMAs
# of the step's first k-group (-DGIF_DMA_SPREAD) instead of back to back ahead of them.  Tap-grid launches only (the probe build does not handle
# the tap-dense order).
#   here:            bash tools/probes/dma_spread_probe.sh build      -> gif_amd/libgif_hip_spread.so
#   on the GPU box:  bash tools/probes/dma_spread_probe.sh run
set -eu
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DGIF_DMA_SPREAD -c conv_igemm.hip -o _probe/conv_igemm_spread.o
  OBJS=$(ls _build/*.o | grep -v "conv_igemm.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_spread.so $OBJS _probe/conv_igemm_spread.o
else
  echo "== normal library"; python tools/probes/kxshare_probe.py
  cp gif_amd/libgif_hip.so /tmp/keep.so; cp gif_amd/libgif_hip_spread.so gif_amd/libgif_hip.so
  echo "== DMA pieces spread between the MFMAs of the first k-group"; python tools/probes/kxshare_probe.py
  python -m pytest tests/test_gpu_f16x2.py -x -q -k "not_less_accurate or epilogue_and_determinism or adversarial_operands or rescale_path" 2>&1 | tail -3
  cp /tmp/keep.so gif_amd/libgif_hip.so
  echo "== normal library, again"; python tools/probes/kxshare_probe.py
fi

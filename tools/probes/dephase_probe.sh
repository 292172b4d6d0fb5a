#!/bin/bash
# Timing probe (round 6): do the chip-wide bursts of the direct f16x2 kernel cost time?  Every CU runs one 256 x 128 workgroup at a time and
# all of them start together, so ring fills (96 KB per CU) and tile stores (128 KB per CU) of the whole chip coincide while the K loops in
# between leave HBM idle.  -DGIF_DEPHASE_PROBE=N delays the first round of workgroups by k/N of a K loop (k = CU's index in its XCD mod N).
#   here:            bash tools/probes/dephase_probe.sh build      -> gif_amd/libgif_hip_dephase{2,4}.so
#   on the GPU box:  bash tools/probes/dephase_probe.sh run
set -eu
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  for n in 2 4; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DGIF_DEPHASE_PROBE=$n -c conv_igemm.hip -o _probe/conv_igemm_dephase$n.o &
  done; wait
  OBJS=$(ls _build/*.o | grep -v "conv_igemm.o")
  for n in 2 4; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_dephase$n.so $OBJS _probe/conv_igemm_dephase$n.o; done
else
  cp gif_amd/libgif_hip.so /tmp/keep.so
  echo "== normal library"; python tools/probes/kxshare_probe.py
  for n in 2 4; do
    cp gif_amd/libgif_hip_dephase$n.so gif_amd/libgif_hip.so
    echo "== first round of workgroups in $n phases"; python tools/probes/kxshare_probe.py
  done
  cp /tmp/keep.so gif_amd/libgif_hip.so
  echo "== normal library, again"; python tools/probes/kxshare_probe.py
fi

#!/bin/bash
# Timing probes (round 6) of the direct kernels' epilogue (results of the probe builds are WRONG):
#   nostore : the row loop without its global stores (-DGIF_NOSTORE_PROBE) — what a free output write would give
#   noepi   : no epilogue at all (-DGIF_EPI_PROBE=3): the bound for any rewrite
#   nofillwait : the K loop starts without waiting for the first ring stage (-DGIF_NOFILLWAIT_PROBE): the bound for a cross-tile prefetch
#   here:            bash tools/probes/epilogue_probe.sh build
#   on the GPU box:  bash tools/probes/epilogue_probe.sh run
set -eu
cd "$(dirname "$0")/../.."
VARIANTS="nostore:-DGIF_NOSTORE_PROBE noepi:-DGIF_EPI_PROBE=3 nofillwait:-DGIF_NOFILLWAIT_PROBE"
if [ "$1" = build ]; then
  make -s -j8 -C gif_amd/csrc ARCH=gfx950
  cd gif_amd/csrc; mkdir -p _probe
  for v in $VARIANTS; do
    n=${v%%:*}; f=$(echo ${v#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -Wno-unused-value $f -c conv_igemm.hip -o _probe/conv_igemm_$n.o &
  done; wait
  OBJS=$(ls _build/*.o | grep -v "conv_igemm.o")
  for v in $VARIANTS; do n=${v%%:*}; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgif_hip_$n.so $OBJS _probe/conv_igemm_$n.o; done
else
  cp gif_amd/libgif_hip.so /tmp/keep.so
  echo "== normal library"; python tools/probes/kxshare_probe.py
  cp gif_amd/libgif_hip_nostore.so gif_amd/libgif_hip.so
  echo "== row loop without stores"; python tools/probes/kxshare_probe.py
  cp gif_amd/libgif_hip_noepi.so gif_amd/libgif_hip.so
  echo "== no epilogue at all (one dword per lane)"; python tools/probes/kxshare_probe.py
  cp gif_amd/libgif_hip_nofillwait.so gif_amd/libgif_hip.so
  echo "== first ring stage read without waiting for it (three-stage kernels)"; python tools/probes/kxshare_probe.py
  cp /tmp/keep.so gif_amd/libgif_hip.so
  echo "== normal library, again"; python tools/probes/kxshare_probe.py
fi

#!/bin/bash
# Builds libgif_hip.so variants whose three MFMA sources (conv_igemm / conv_wgrad / conv_winograd) are compiled with one extra LLVM
# scheduling flag each, into gif_amd/_variants/ (git-ignored; they travel with gpurun).  tools/probes/sched_flag_ab.sh times them.
set -e
cd "$(dirname "$0")/../../gif_amd/csrc"
make -s -j8 ARCH=gfx950
mkdir -p ../_variants
declare -A V=(
  [maxilp]="-mllvm -amdgpu-sched-strategy=max-ilp"
  [memclause]="-mllvm -amdgpu-sched-strategy=max-memory-clause"
  [relaxocc]="-mllvm -amdgpu-schedule-relaxed-occupancy=true"
  [trackers]="-mllvm -amdgpu-use-amdgpu-trackers=1"
  [nopostmi]="-mllvm -enable-post-misched=0"
  [nohighrp]="-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1"
)
for n in "${!V[@]}"; do
  mkdir -p /tmp/variants/$n
  for f in conv_igemm conv_wgrad conv_winograd; do
    echo "/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. ${V[$n]} -c $f.hip -o /tmp/variants/$n/$f.o"
  done
done | xargs -P 8 -I{} bash -c "{}"
for n in "${!V[@]}"; do
  others=$(ls _build/*.o | grep -v "conv_igemm.o\|conv_wgrad.o\|conv_winograd.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_variants/libgif_hip_$n.so /tmp/variants/$n/*.o $others
  echo "built $n: ${V[$n]}"
done

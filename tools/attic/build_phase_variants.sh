#!/bin/bash
# tools/mlp_phase_<tag>.bin for the ring-depth / L2-warm-up variants of the 8-wave kernels (A/B on one box)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -sink-common-insts=false -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16"
for v in "r3w0 -DOSRL_RING_DEEP=3 -DOSRL_L2_WARM=0" "r4w0 -DOSRL_RING_DEEP=4 -DOSRL_L2_WARM=0" "r3w1 -DOSRL_RING_DEEP=3 -DOSRL_L2_WARM=1" "r4w1 -DOSRL_RING_DEEP=4 -DOSRL_L2_WARM=1" "r6w0 -DOSRL_RING_DEEP=6 -DOSRL_L2_WARM=0"; do
  set -- $v; tag=$1; shift
  /opt/rocm/bin/hipcc $F "$@" tools/mlp_phase.hip -o tools/mlp_phase_$tag.bin 2>&1 | grep -v "hip-link\|warning: \|^$" | head -5 &
done
wait; ls -la tools/mlp_phase_*.bin

#!/bin/bash
# round 5: phase stamps of the SHIPPED 8-wave 16-row chain kernels (tools/mlp_phase.hip = mlp.hip
# compiled with OSRL_PHASE_TIMING): stage-in / per layer: k-loop, barrier, epilogue, sync+save -- warm and cold weights
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5phase; rm -rf $O; mkdir -p $O
for shape in "2048 0 1 400 8" "2048 0 2 256 1" "2048 0 4 256 1" "2048 0 6 256 1"; do
  for cold in 0 1; do
    echo "== rows tile nets width out = $shape   cold weights = $cold"
    COLD=$cold timeout 60 tools/_lab/mlp_phase $shape 2>&1 | grep -E "mean cycles|kernel span|^rows" | cut -c1-420
  done
done > $O/r5_chain_phase.txt 2>&1
cat $O/r5_chain_phase.txt

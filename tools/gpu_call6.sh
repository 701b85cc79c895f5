#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
rm -f gpurun_out/c6_phase.txt
for cfg in "2048 0 1 400 8" "2048 0 2 256 1" "2048 0 4 256 1" "2048 0 1 256 4"; do
  echo "# tools/mlp_phase.bin $cfg" >> gpurun_out/c6_phase.txt
  timeout 60 tools/mlp_phase.bin $cfg >> gpurun_out/c6_phase.txt 2>&1
done
cat gpurun_out/c6_phase.txt

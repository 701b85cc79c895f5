#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, p.total_memory>>30)" > gpurun_out/c3_phase.txt 2>&1
for cfg in "20480 80 1 400 8" "20480 16 1 400 8" "20480 80 2 256 1" "20480 32 2 256 1"; do
  echo "# tools/mlp_phase.bin $cfg" >> gpurun_out/c3_phase.txt
  timeout 60 tools/mlp_phase.bin $cfg >> gpurun_out/c3_phase.txt 2>&1
done
cat gpurun_out/c3_phase.txt

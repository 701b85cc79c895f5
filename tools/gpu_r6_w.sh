#!/bin/bash
# round 6: bf16 MFMA rate probe (tools/mfma_bf16_probe.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6w; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mfma_bf16_probe.hip -o /tmp/mfma_bf16_probe 2>/dev/null
for i in 1 2; do timeout 120 /tmp/mfma_bf16_probe 2>&1 | tee -a $O/probe.txt; done

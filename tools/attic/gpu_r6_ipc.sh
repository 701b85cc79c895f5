#!/bin/bash
# round 6: the two-process slab exchange toy (tools/ipc_slab_lab.hip) on the lease's one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6ipc; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ipc_slab_lab.hip -o /tmp/ipc_slab_lab 2>/dev/null
for i in 1 2; do timeout 120 /tmp/ipc_slab_lab 2>&1 | tee -a $O/ipc_slab_lab.txt; echo "-- exit $?" | tee -a $O/ipc_slab_lab.txt; done

#!/bin/bash
# round 6: two processes on one GPU exchanging through IPC-mapped buffers (tests/test_gpu_ipc_dp.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ipc_dp.py -x -q --durations=5 > $O/pytest.log 2>&1; tail -40 $O/pytest.log

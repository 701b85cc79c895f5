#!/bin/bash
# the IPC process tests of the remaining engines (BEAR-L / COptiDICE / CDT), then the final evidence script
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6t; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ipc_dp.py -x -q -m gpu -k other_engines 2>&1 | tail -15 | tee $O/pytest.log
bash tools/gpu_r6_final.sh

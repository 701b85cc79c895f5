#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1  # lab switches (OSRL_*) are read only under this (engine/plan.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; rm -rf $O; mkdir -p $O
for v in 0 1; do
OSRL_DP_SIDE_COLL=$v timeout 600 python -m pytest "tests/test_gpu_dp_sim.py::test_captured_data_parallel_graph_equals_concatenated_batch[cpq-2]" -q -x -s > $O/t$v.log 2>&1; echo "side_coll=$v"; grep -v "^frame\|^  File \"/usr" $O/t$v.log | head -60
done

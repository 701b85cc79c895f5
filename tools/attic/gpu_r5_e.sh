#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; rm -rf $O; mkdir -p $O
for v in "cpq-2-False" "cpq_c4-2-False" "cpq_c4_w8-8-False" "cpq-2-True" "cpq_c4_w8-8-True"; do
timeout 600 python -m pytest "tests/test_gpu_dp_sim.py::test_captured_data_parallel_graph_equals_concatenated_batch[$v]" -q -x > $O/t_$v.log 2>&1; echo "$v rc=$?"; grep -E "passed|failed|Error|error:" $O/t_$v.log | head -5
done

#!/bin/bash
# round 5, call F: tests after the plan refactor (DP sim, bench path, BC, kernels touched this round)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_dp_sim.py tests/test_gpu_bench_path.py tests/test_gpu_bc_one_launch.py \
  "tests/test_gpu_kernels.py::test_forward2_with_a_kl_tail_writes_the_kl_rows" "tests/test_gpu_kernels.py::test_vae_ns_launches_equal_the_fused_launches" \
  -q > $O/t.log 2>&1; tail -15 $O/t.log

#!/bin/bash
# round 5, call D: data-parallel plan with the side-issued collectives (forced DP on one rank, DP sim tests), VAE Adam A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1  # lab switches (OSRL_*) are read only under this (engine/plan.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dp_sim.py tests/test_gpu_train_step.py -q -k "captured_data_parallel or world2_sharded and cpq or data_parallel or seeded_backward and c4 or bench_path" tests/test_gpu_bench_path.py "tests/test_gpu_kernels.py::test_forward2_with_a_kl_tail_writes_the_kl_rows" > $O/t.log 2>&1; tail -8 $O/t.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-70; }
run OSRL_FORCE_DP=1 OSRL_DP_SIDE_COLL=0
run OSRL_FORCE_DP=1 OSRL_DP_SIDE_COLL=1
run OSRL_FORCE_DP=1 OSRL_DP_SIDE_COLL=0 --config c4
run OSRL_FORCE_DP=1 OSRL_DP_SIDE_COLL=1 --config c4
run OSRL_VAE_ADAM_SIDE=0
run OSRL_VAE_ADAM_SIDE=1
run OSRL_VAE_ADAM_SIDE=1 --config c4
tail -3 $O/bench.err

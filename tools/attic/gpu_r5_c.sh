#!/bin/bash
# round 5, call C: the round's new GPU tests; default-plan bench lines (auto rule) for c2 / c4; forced-DP baselines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1  # lab switches (OSRL_*) are read only under this (engine/plan.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_path.py "tests/test_gpu_kernels.py::test_forward2_with_a_kl_tail_writes_the_kl_rows" \
  "tests/test_gpu_kernels.py::test_vae_ns_launches_equal_the_fused_launches" \
  "tests/test_gpu_data_eval.py::test_trained_cost_return_gap_vs_reference" -q > $O/t_new.log 2>&1; tail -8 $O/t_new.log
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_dp_sim.py -q -k "cpq_c4 or c4_w8" > $O/t_c4.log 2>&1; tail -4 $O/t_c4.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-70; }
run X=0 --config c4
run OSRL_VAE_NS=0 --config c4
run OSRL_FORCE_DP=1
run OSRL_FORCE_DP=1 --config c4
tail -3 $O/bench.err

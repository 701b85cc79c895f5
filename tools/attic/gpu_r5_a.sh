#!/bin/bash
# round 5, call A: new tests (bench path vs oracle, KL-tail pair, trained cost-return gap) + the all-CU VAE launches:
# full-size parity with OSRL_VAE_NS=1, then A/B of the step at c2 / c3 / c4
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1  # lab switches (OSRL_*) are read only under this (engine/plan.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_path.py "tests/test_gpu_kernels.py::test_forward2_with_a_kl_tail_writes_the_kl_rows" \
  "tests/test_gpu_data_eval.py::test_trained_cost_return_gap_vs_reference" -q > $O/t_new.log 2>&1; tail -6 $O/t_new.log
OSRL_VAE_NS=1 timeout 900 python -m pytest tests/test_gpu_train_step.py -q -k "full_size and (cpq or bcql) or parallel_branches and (c2_full or c4_full) or two_of_three" > $O/t_ns.log 2>&1; tail -6 $O/t_ns.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-70; }
for rep in 1 2; do
run OSRL_VAE_NS=0
run OSRL_VAE_NS=1
done
run OSRL_VAE_NS=0 --config c4
run OSRL_VAE_NS=1 --config c4
run OSRL_VAE_NS=0 --config c3 --steps 100
run OSRL_VAE_NS=1 --config c3 --steps 100
tail -3 $O/bench.err

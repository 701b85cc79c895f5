#!/bin/bash
# round 5, call Q: data-parallel CPQ with the VAE gradient riding in the [critic | cost-critic] all-reduce (3 collectives per
# step instead of 4): the DP tests, then forced data parallelism on one rank at C2 / C4, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dp_sim.py tests/test_gpu_train_step.py -m gpu -x -q -k "dp or world or data_parallel or parallel" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; v=$(env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | sed 's/.*"value": \([0-9.]*\).*/\1/'); echo "$v  $*" | tee -a $O/sweep.txt; }
for rep in 1 2; do
run X=0
run OSRL_FORCE_DP=1 OSRL_DP_MERGE_VAE=0
run OSRL_FORCE_DP=1 OSRL_DP_MERGE_VAE=1
run X=0 --config c4
run OSRL_FORCE_DP=1 OSRL_DP_MERGE_VAE=0 --config c4
run OSRL_FORCE_DP=1 OSRL_DP_MERGE_VAE=1 --config c4
done

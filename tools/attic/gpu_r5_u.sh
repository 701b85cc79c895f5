#!/bin/bash
# round 5, call U: C2 / C4 -- creation order of the two branches' first segments (same graph, other node order):
# M = main (all-CU VAE forward | backward | dW ..), S = side (actor forwards | N*B-row cost critics | critic phase)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5u; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_step.py -m gpu -x -q -k "cpq" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; v=$(env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | sed 's/.*"value": \([0-9.]*\).*/\1/'); echo "$v  $*" | tee -a $O/sweep.txt; }
for cfg in c2 c4; do
for o in MMM SMMM MSMM MSMSM MMSM MMSSM SSMMM MSSMM MMM; do
run OSRL_CREATE_ORDER=$o --config $cfg
done
done

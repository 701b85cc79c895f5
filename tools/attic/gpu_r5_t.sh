#!/bin/bash
# round 5, call T: the 400-wide N*B-row launches on 48-row tiles, two workgroups per CU (mlp_nb48.hip) -- parity of the big-row
# forward tests under it, then A/B at C2 / C4 / C3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5t; rm -rf $O; mkdir -p $O
OSRL_NB48=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py -m gpu -x -q -k "big_rows or nb or full_size or kl_tail or c2 or c3" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; v=$(env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | sed 's/.*"value": \([0-9.]*\).*/\1/'); echo "$v  $*" | tee -a $O/sweep.txt; }
for rep in 1 2; do
for cfg in c2 c4 c3; do
run X=0 --config $cfg
run OSRL_NB48=1 --config $cfg
run OSRL_NB48=1 OSRL_NB_WAVES=4 --config $cfg
done
done

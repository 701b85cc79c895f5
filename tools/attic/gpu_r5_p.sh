#!/bin/bash
# round 5, call P: C4 -- the action draws as forward tails (N * act_dim = 60 > the plan rule's 32), VAE Adam's branch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5p; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --config c4"
run() { E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; v=$(env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | sed 's/.*"value": \([0-9.]*\).*/\1/'); echo "$v  $*" | tee -a $O/sweep.txt; }
for rep in 1 2 3; do
run X=0
run OSRL_HEAD_TAILS=1
run OSRL_HEAD_TAILS=1 OSRL_VAE_ADAM_SIDE=1
run OSRL_HEAD_TAILS=1 OSRL_VAE_ADAM_SIDE=0
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4e; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-60; }
for rep in 1 2; do
run OSRL_HEAD_TAILS=0
run OSRL_HEAD_TAILS=1
run OSRL_HEAD_TAILS=0 OSRL_VAE_DW_SPLITS=2
run OSRL_HEAD_TAILS=0 OSRL_DW_T_CRITIC=0 OSRL_DW_T_COST=0
done
run OSRL_HEAD_TAILS=0 --config c4
run OSRL_HEAD_TAILS=1 --config c4

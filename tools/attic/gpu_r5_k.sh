#!/bin/bash
# round 5, call K2: C2 -- the all-CU VAE launches together with the VAE Adam on the side branch, three A/B pairs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5k2; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-70; }
for rep in 1 2 3; do
run X=0
run OSRL_VAE_NS=1 OSRL_VAE_ADAM_SIDE=1
done
run X=0 --steps 20 --warmup 5
run OSRL_VAE_NS=1 OSRL_VAE_ADAM_SIDE=1 --steps 20 --warmup 5

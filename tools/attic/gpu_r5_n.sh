#!/bin/bash
# round 5, call N: C3 (BCQ-Lag) -- row splits of the 4096-row dW plans (the online critics' forward beside the VAE phase
# was measured here first: 620 vs 640, gpurun_out/r5n)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5n2; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --config c3 --steps 100 --warmup 10"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-70; }
for rep in 1 2; do
run X=0
run OSRL_BCQ_DW_SPLITS=4
run OSRL_BCQ_DW_SPLITS=6
run OSRL_BCQ_DW_SPLITS=8
done

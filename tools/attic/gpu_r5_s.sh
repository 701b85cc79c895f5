#!/bin/bash
# round 5, call S: C2 / C4 -- the VAE group's dW on smaller tiles (the knock-out prices put its cost in the 210 CUs its
# 416-register workgroups take from the side branch, not in its own duration)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5s; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; v=$(env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | sed 's/.*"value": \([0-9.]*\).*/\1/'); echo "$v  $*" | tee -a $O/sweep.txt; }
for rep in 1 2; do
run X=0
run OSRL_VAE_DW_TILE=4 OSRL_VAE_DW_SPLITS=2
run OSRL_VAE_DW_TILE=4 OSRL_VAE_DW_SPLITS=3
run OSRL_VAE_DW_TILE=4 OSRL_VAE_DW_SPLITS=4
run OSRL_VAE_DW_TILE=3 OSRL_VAE_DW_SPLITS=2
run OSRL_VAE_DW_TILE=3 OSRL_VAE_DW_SPLITS=3
run OSRL_VAE_DW_TILE=2 OSRL_VAE_DW_SPLITS=1
run OSRL_VAE_DW_TILE=2 OSRL_VAE_DW_SPLITS=2
run X=0 --config c4
run OSRL_VAE_DW_TILE=4 OSRL_VAE_DW_SPLITS=3 --config c4
run OSRL_VAE_DW_TILE=3 OSRL_VAE_DW_SPLITS=2 --config c4
run OSRL_VAE_DW_TILE=2 OSRL_VAE_DW_SPLITS=2 --config c4
done

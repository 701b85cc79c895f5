#!/bin/bash
# round 5, call O: C2 -- one-factor sweep of the plan knobs around the shipped plan (all-CU VAE launches + VAE Adam on the
# side branch changed what runs beside what since round 4 tuned them)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5o; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; v=$(env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | sed 's/.*"value": \([0-9.]*\).*/\1/'); echo "$v  $*" | tee -a $O/sweep.txt; }
for rep in 1 2; do
run X=0
run OSRL_VAE_DW_SPLITS=2
run OSRL_VAE_DW_SPLITS=4
run OSRL_DW_S_CRITIC=1
run OSRL_DW_S_CRITIC=3
run OSRL_DW_S_COST=1
run OSRL_DW_S_COST=3
run OSRL_DW_T_CRITIC=4 OSRL_DW_S_CRITIC=2
run OSRL_DW_T_COST=4 OSRL_DW_S_COST=2
run OSRL_FUSE_DW_ADAM=0
run OSRL_FUSE_DW_ADAM=1
run OSRL_ENC_TILE=64
run X=0
done

#!/bin/bash
# round 5, call R (timing experiment): BCQ-Lag's 4096-row TRAINING forwards on the 80 / 64-row N*B kernels, saving nothing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5r; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --config c3 --steps 100 --warmup 10"
run() { E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; v=$(env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | sed 's/.*"value": \([0-9.]*\).*/\1/'); echo "$v  $*" | tee -a $O/sweep.txt; }
for rep in 1 2; do
run X=0
run OSRL_NB_IGNORE_SAVE=1 OSRL_BCQ_TRAIN_TILE=80
run OSRL_NB_IGNORE_SAVE=1 OSRL_BCQ_TRAIN_TILE=80 OSRL_NB64=1
run OSRL_NB_IGNORE_SAVE=1 OSRL_BCQ_TRAIN_TILE=80 OSRL_NB64=0
done
tail -3 $O/bench.err

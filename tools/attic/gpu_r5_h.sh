#!/bin/bash
# round 5, call H: VAE Adam on the side branch (3 A/B pairs at C2), BCQ-Lag with the merged clamp launch (tests + C3)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_data_eval.py -q -k "bcql" > $O/t.log 2>&1; tail -4 $O/t.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-70; }
for rep in 1 2 3; do
run OSRL_VAE_ADAM_SIDE=0
run OSRL_VAE_ADAM_SIDE=1
done
run X=1 --config c3 --steps 100
run X=1 --config c3 --steps 100

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4q; rm -rf $O; mkdir -p $O
OSRL_NB64=3 timeout 900 python -m pytest tests/test_gpu_train_step.py -q -x -k "bcql" > $O/t1.log 2>&1; grep -n "passed\|failed" $O/t1.log | tail -2
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --no-cold --steps 100 --warmup 10"
run() { echo -n "$* "; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-60; }
for rep in 1 2; do
for w in 2 3; do run OSRL_NB64=$w --config c3; done
done
for w in 0 2 3; do run OSRL_NB64=$w --config c2; done; for w in 0 2; do run OSRL_NB64=$w --config c4; done

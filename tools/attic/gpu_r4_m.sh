#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; rm -rf $O; mkdir -p $O
OSRL_OOD_SPLIT=5 timeout 600 python -m pytest tests/test_gpu_train_step.py -q -x -k "cpq and (golden or full_size)" > $O/t1.log 2>&1; tail -3 $O/t1.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --no-cold"
run() { echo -n "$* "; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-60; }
for rep in 1 2; do
for k in 10 8 6 5 3 0; do run OSRL_OOD_SPLIT=$k; done
done
for k in 10 5 0; do run OSRL_OOD_SPLIT=$k --config c4; done

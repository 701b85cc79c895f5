#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --no-cold"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-60; }
for rep in 1 2; do
run OSRL_ENC_AFTER=vae
run OSRL_ENC_AFTER=cost
done
run OSRL_ENC_AFTER=vae --config c4
run OSRL_ENC_AFTER=cost --config c4

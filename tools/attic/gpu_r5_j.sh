#!/bin/bash
# round 5, call J: "one block over" K-split of the 25th column block in the 8-wave 16-row kernels: phase stamps on / off,
# kernel + train-step parity, A/B of the step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; rm -rf $O; mkdir -p $O
for x in 0 1; do echo "== OSRL_XSPLIT=$x"; OSRL_XSPLIT=$x timeout 60 tools/_lab/mlp_phase 2048 0 1 400 8 2>&1 | grep -E "mean cycles|kernel span" | cut -c1-420; done | tee $O/phase.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py -q -x > $O/t.log 2>&1; tail -5 $O/t.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-70; }
for rep in 1 2; do
run OSRL_XSPLIT=0
run OSRL_XSPLIT=1
done
run OSRL_XSPLIT=0 --config c3 --steps 100
run OSRL_XSPLIT=1 --config c3 --steps 100

#!/bin/bash
# round 5, call M: the dual step by the OOD statistic's launch (osrl_cpq_ood_dual) -- kernel test, CPQ parity, A/B at C2 / C4
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py tests/test_gpu_bench_path.py -m gpu -x -q -k "ood or cpq or c2 or c4 or vae_ns or graph" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-70; }
for rep in 1 2 3; do
run OSRL_OOD_DUAL=0
run OSRL_OOD_DUAL=1
done
run OSRL_OOD_DUAL=0 --config c4
run OSRL_OOD_DUAL=1 --config c4

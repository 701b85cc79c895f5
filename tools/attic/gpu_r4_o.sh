#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4o; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_kernels.py -q -x > $O/t1.log 2>&1; grep -n "passed\|failed" $O/t1.log | tail -2
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --no-cold --steps 100 --warmup 10"
for c in c3 c2; do $B --config $c 2>>$O/bench.err | cut -c1-60; done

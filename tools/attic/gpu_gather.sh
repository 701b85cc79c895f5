#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/gather; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bc_one_launch.py tests/test_gpu_data_eval.py tests/test_gpu_train_step.py -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for cfg in c2 c1 c3 c2 c1; do
  timeout 300 python bench.py --config $cfg --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-roofline 2>>$O/err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['value'], d['ms_per_step'])"
done

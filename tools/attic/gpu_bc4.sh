#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bc1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bc_one_launch.py tests/test_gpu_train_step.py tests/test_gpu_data_eval.py -x -q -k "bc or BC or one_launch" > $O/pytest4.log 2>&1; grep -E "passed|failed|Error|assert" $O/pytest4.log | head -8
for rep in 1 2; do
for v in "OSRL_BC_DIRECT=1" "OSRL_BC_DIRECT=0" "OSRL_BC_ONE_LAUNCH=0" "HIP_FORCE_DEV_KERNARG=0 OSRL_BC_DIRECT=1" "HIP_FORCE_DEV_KERNARG=0 OSRL_BC_DIRECT=0"; do
  env $v timeout 300 python bench.py --config c1 --no-cpu-baseline --no-extras > $O/c1_v.json 2> $O/c1_v.err
  python -c "import json; d=json.load(open('$O/c1_v.json')); print('c1 $v', d['value'], d['ms_per_step'])" || tail -5 $O/c1_v.err
done; done
OSRL_BC_DIRECT=1 timeout 300 python bench.py --config c1 --no-cpu-baseline --no-extras > $O/c1_direct.json 2>/dev/null

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bc1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bc_one_launch.py tests/test_gpu_kernels.py tests/test_gpu_train_step.py -x -q > $O/pytest2.log 2>&1; grep -E "passed|failed|Error|Fatal|File \"/.*tests/" $O/pytest2.log | head -12
echo "=== stamps"
OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_stamps.so timeout 200 python tools/step_stamps.py 256 256 2>&1 | grep -v amdgpu.ids | tail -13 | tee $O/stamps.txt
for rep in 1 2; do for one in 1 0; do
  OSRL_BC_ONE_LAUNCH=$one timeout 300 python bench.py --config c1 --no-cpu-baseline --no-extras > $O/c1_one$one.json 2> $O/c1_one$one.err
  python -c "import json; d=json.load(open('$O/c1_one$one.json')); print('c1 one_launch=$one', d['value'], d['ms_per_step'])" || tail -5 $O/c1_one$one.err
done; done
for cfg in c2 c3; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras > $O/$cfg.json 2> $O/$cfg.err
  python -c "import json; d=json.load(open('$O/$cfg.json')); print('$cfg', d['value'], d['ms_per_step'], d.get('roofline',{}).get('in_step_sites_us'))" || tail -5 $O/$cfg.err
done

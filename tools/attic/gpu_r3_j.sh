#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py tests/test_gpu_dp_sim.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -5
timeout 300 python tools/dw_bench.py 2>&1 | grep -v amdgpu.ids | head -8
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'
for r in 1 2; do
  for v in "A=1" "OSRL_DW_FLAT=0" "OSRL_VAE_DW_T5=0 OSRL_DW_FLAT=0" "OSRL_VAE_DW_SPLITS=4" "OSRL_VAE_DW_SPLITS=2"; do
    echo -n "[$v] "; env $v python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P"
  done
done

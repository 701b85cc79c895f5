#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -5
for v in "OSRL_NB_FUSED_HEAD=1" "OSRL_NB_FUSED_HEAD=0"; do echo "== $v"; env $v timeout 300 python tools/kbench.py --glue 2>&1 | grep -E "rows=20480 tile=80|fwd enc|fwd q x2 rows=20480"; done
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'
for r in 1 2; do
  for v in "OSRL_NB_FUSED_HEAD=1" "OSRL_NB_FUSED_HEAD=0"; do
    echo -n "[c2 $v] "; env $v python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P"
    echo -n "[c3 $v] "; env $v python bench.py --no-extras --no-cpu-baseline --no-roofline --config c3 --steps 100 2>/dev/null | python -c "$P"
  done
done

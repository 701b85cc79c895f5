#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for b in 1 2; do for i in 1 2; do
  v=$(OSRL_BRANCHES=$b timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['last_stats']['loss/cost_critic_loss'], d['last_stats']['loss/alpha_value'])")
  echo "branches=$b steps/s=$v"
done; done
OSRL_BRANCHES=2 timeout 600 python -m pytest tests/test_gpu_train_step.py -m gpu -q --timeout=600 -k "graph" 2>&1 | tail -3
OSRL_BRANCHES=2 bash tools/gpu_prof_step.sh > gpurun_out/c26_prof.txt 2>&1; head -44 gpurun_out/c26_prof.txt | cut -c1-100

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for t in 0 80; do
  v=$(OSRL_BCQ_TILE=$t timeout 200 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['last_stats'])")
  echo "bcq tile=$t steps/s=$v"
done
OSRL_BCQ_TILE=80 timeout 600 python -m pytest tests/test_gpu_train_step.py -m gpu -q --timeout=600 -k "bcql" 2>&1 | tail -3

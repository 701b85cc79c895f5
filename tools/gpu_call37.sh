#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 120 python tools/act_bench.py 2>&1 | tail -7
echo "--- server off"
OSRL_ACT_SERVER=0 timeout 120 python tools/act_bench.py 2>&1 | tail -7 | head -3
timeout 300 python -m pytest tests/test_gpu_data_eval.py -m gpu -q --timeout=120 -k "fast_policy or evaluate or end_to_end" 2>&1 | tail -4

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cdt.py tests/test_gpu_dp_sim.py -m gpu -q --timeout=600 -k "cdt" > gpurun_out/c41_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/c41_pytest.log | head -5
timeout 200 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('c5 steps/s', d['value'], d['ms_per_step'], d['last_stats'])"

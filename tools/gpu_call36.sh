#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/c36_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c36_pytest.log
tail -5 gpurun_out/c36_pytest.log
timeout 200 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('c5 steps/s', d['value'], d['ms_per_step'])"
timeout 200 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('c2 steps/s', d['value'], d['ms_per_step'])"

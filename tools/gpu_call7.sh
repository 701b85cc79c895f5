#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/parity_margins.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c7_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c7_pytest.log
timeout 300 python tests/bench_eval.py > gpurun_out/c7_eval_bench.txt 2>&1
tail -60 gpurun_out/c7_pytest.log | cut -c1-300
tail -20 gpurun_out/c7_eval_bench.txt
cat gpurun_out/parity_margins.txt

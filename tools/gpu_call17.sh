#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_dp_sim.py -m gpu -q --timeout=600 -k "cpq or CPQ or data_parallel or checkpoint or rebuild" > gpurun_out/c17_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c17_pytest.log
tail -5 gpurun_out/c17_pytest.log
for i in 1 2; do timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('steps/s', d['value'], d['last_stats'])"; done
bash tools/gpu_prof_step.sh > gpurun_out/c17_prof.txt 2>&1; head -45 gpurun_out/c17_prof.txt | cut -c1-110

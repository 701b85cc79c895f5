#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/kbench.py --big > gpurun_out/c2_kbench.txt 2>&1
OSRL_OOD_TILE=80 timeout 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c2_bench_t80.json 2> gpurun_out/c2_bench_t80.err
OSRL_ENC_TILE=80 timeout 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c2_bench_enc80.json 2> gpurun_out/c2_bench_enc80.err
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py -m gpu -q --timeout=600 > gpurun_out/c2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c2_pytest.log
grep -v "^$" gpurun_out/c2_kbench.txt | cut -c1-100
tail -3 gpurun_out/c2_pytest.log

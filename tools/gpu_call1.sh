#!/bin/bash
# round-2 GPU call 1: full GPU test suite, kernel micro-benchmarks (incl. the 80-row tiles), bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity_margins.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c1_pytest.log
timeout 300 python tools/kbench.py > gpurun_out/c1_kbench.txt 2>&1
timeout 400 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
OSRL_OOD_TILE=80 timeout 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c1_bench_t80.json 2> gpurun_out/c1_bench_t80.err
OSRL_ENC_TILE=80 timeout 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c1_bench_enc80.json 2> gpurun_out/c1_bench_enc80.err
OSRL_FORCE_DP=1 timeout 200 python bench.py --gpus 1 --config c4 --no-cpu-baseline > gpurun_out/c1_bench_c4_dp.json 2> gpurun_out/c1_bench_c4_dp.err
tail -5 gpurun_out/c1_pytest.log
cat gpurun_out/c1_bench.json | cut -c1-1500

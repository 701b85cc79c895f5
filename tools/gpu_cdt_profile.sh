#!/bin/bash
# CDT (C5) step: GPU tests of the transformer path, throughput, rocprofv3 per-kernel statistics -> gpurun_out/cdt/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/cdt; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cdt.py -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python bench.py --config c5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err > $O/bench_c5.json; cut -c1-100 $O/bench_c5.json
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o cdt -- python $GRAFT_REPO_ROOT/tools/prof_one.py cdt 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
S=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $S $O/cdt_kernel_stats.csv; rm -rf $O/prof
head -12 $O/cdt_kernel_stats.csv | cut -c1-160

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/cdt; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cdt.py -q -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for r in 1 2; do timeout 300 python bench.py --config c5 --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c5', d['value'], d['ms_per_step'])"; done
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o cdt -- python $GRAFT_REPO_ROOT/tools/prof_one.py cdt 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
S=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $S $O/cdt_kernel_stats.csv; rm -rf $O/prof
grep -E "gelu|ln_fwd|ln_bwd" $O/cdt_kernel_stats.csv | cut -c1-140

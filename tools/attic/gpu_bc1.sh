#!/bin/bash
# the one-launch BC step: parity tests, then c1 with / without it (bench.py --config c1) and a kernel trace of both
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bc1; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bc_one_launch.py -x -q 2>&1 | tail -25 > $O/pytest.log; tail -12 $O/pytest.log
timeout 300 python -m pytest tests/test_gpu_train_step.py -x -q -k "bc" 2>&1 | tail -5 > $O/pytest_bc.log; tail -3 $O/pytest_bc.log
for one in 1 0; do
  OSRL_BC_ONE_LAUNCH=$one timeout 300 python bench.py --config c1 --no-cpu-baseline --no-extras > $O/c1_one$one.json 2> $O/c1_one$one.err
  python -c "import json; d=json.load(open('$O/c1_one$one.json')); print('one_launch=$one', d['value'], d['unit'], d['ms_per_step'])" || tail -5 $O/c1_one$one.err
done
(cd /tmp && OSRL_BC_ONE_LAUNCH=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o c1 -- python $GRAFT_REPO_ROOT/bench.py --config c1 --no-cpu-baseline --no-extras --steps 300 > $O/c1_prof.json 2> $O/prof.err)
K=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp $K $O/c1_kernel_stats.csv && head -6 $O/c1_kernel_stats.csv | cut -c1-200
rm -rf $O/prof
(cd /tmp && OSRL_BC_ONE_LAUNCH=0 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o c1 -- python $GRAFT_REPO_ROOT/bench.py --config c1 --no-cpu-baseline --no-extras --steps 300 > $O/c1_prof0.json 2> $O/prof0.err)
K=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp $K $O/c1_kernel_stats_plan.csv && head -8 $O/c1_kernel_stats_plan.csv | cut -c1-160
rm -rf $O/prof
OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_stamps.so timeout 200 python tools/step_stamps.py 256 256 > $O/stamps.txt 2>&1; cat $O/stamps.txt

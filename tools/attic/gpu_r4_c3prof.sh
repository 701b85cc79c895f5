#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4c3; rm -rf $O; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o c3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --no-cold --config c3 --steps 100 --warmup 10 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
python tools/timeline.py $T > $O/timeline.txt 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
head -12 $O/trace_summary.txt; cut -c1-80 $O/bench_profiled.json

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4cdt; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --config c5 --steps 20 --warmup 5"
for rep in 1 2; do $B 2>>$O/bench.err | cut -c1-120; done
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o cdt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --config c5 --steps 10 --warmup 3 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
S=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $S $O/cdt_kernel_stats.csv
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
rm -rf $O/prof
head -40 $O/cdt_kernel_stats.csv | cut -c1-150

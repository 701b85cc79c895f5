#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bc1; mkdir -p $O
for r4 in 1 0; do
(cd /tmp && OSRL_ROWS4=$r4 OSRL_BC_ONE_LAUNCH=0 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o c1 -- python $GRAFT_REPO_ROOT/bench.py --config c1 --no-cpu-baseline --no-extras --steps 300 > $O/c1_prof0.json 2> $O/prof0.err)
K=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp $K $O/c1_kernel_stats_plan_r4$r4.csv && echo "== ROWS4=$r4" && head -8 $O/c1_kernel_stats_plan_r4$r4.csv | cut -c1-150
rm -rf $O/prof
done

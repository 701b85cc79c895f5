#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_act
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_act -o act -- python $GRAFT_REPO_ROOT/tools/act_bench.py > $GRAFT_REPO_ROOT/gpurun_out/c9_act.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_act -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_act -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-220
tail -8 gpurun_out/c9_act.txt
find gpurun_out/prof_act -name "*kernel_trace.csv" -size +30M -delete

#!/bin/bash
# round 6: kernel durations of the Trainer-API path's graph, with / without a pipeline built on the engine before
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6s; rm -rf $O; mkdir -p $O
for m in none pipe; do
  (cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof_$m -o t -- python $GRAFT_REPO_ROOT/tools/r6_single_after_pipe.py $m > $O/run_$m.txt 2> $O/err_$m.txt)
  T=$(find $O/prof_$m -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py $T > $O/timeline_api_$m.txt 2>&1
  rm -rf $O/prof_$m
  grep api_path $O/run_$m.txt; head -45 $O/timeline_api_$m.txt
done

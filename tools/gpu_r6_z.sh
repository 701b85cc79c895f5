#!/bin/bash
# round 6: kernel-trace timeline of the C2 step with plan.ood_rows (one step per graph for readability)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6z; rm -rf $O; mkdir -p $O
for cfg in c2; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$cfg -o bench -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps-per-graph 1 --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 20 > $O/bench_profiled_$cfg.json 2> $O/prof_$cfg.err)
  T=$(find $O/prof_$cfg -name "*kernel_trace.csv" | head -1)
  python tools/timeline_graph.py $T 1 > $O/timeline_$cfg.txt 2>&1
  python tools/trace_summary.py $T > $O/trace_summary_$cfg.txt 2>&1
  rm -rf $O/prof_$cfg
  cut -c1-100 $O/bench_profiled_$cfg.json; cat $O/timeline_$cfg.txt | cut -c1-110
done

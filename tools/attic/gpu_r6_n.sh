#!/bin/bash
# round 6: forced data parallelism on one rank, RCCL vs IPC exchange: per-kernel durations inside the replayed graph
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6n; rm -rf $O; mkdir -p $O
for ex in rccl ipc; do
  (cd /tmp && OSRL_FORCE_DP=1 OSRL_DP_EXCHANGE=$ex rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$ex -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --no-cold --steps 200 --warmup 20 > $O/bench_$ex.json 2> $O/prof_$ex.err)
  T=$(find $O/prof_$ex -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py $T > $O/timeline_forced_dp_$ex.txt 2>&1
  python tools/trace_summary.py $T > $O/trace_summary_$ex.txt 2>&1
  rm -rf $O/prof_$ex
  cut -c1-100 $O/bench_$ex.json; head -40 $O/timeline_forced_dp_$ex.txt
done
